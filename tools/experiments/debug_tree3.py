#!/usr/bin/env python3
"""two ranks @ 3 ms weight-gradient delay against the undelayed single process, per unit, under the environment given"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import test_gpu_ddp as T
tmp = pathlib.Path(tempfile.mkdtemp())
levels = os.environ.get("LEVELS_LIT") or getattr(T, os.environ.get("LEVELS", "_LEVELS_RAGGED"))
_, one0 = T._run_tree(tmp, levels, "a", delay_us=0)
two3, _ = T._run_tree(tmp, levels, "b", delay_us=3000, single=False, hw_queues=os.environ.get("HWQ", ""))
keys, ranges = two3["keys"], two3["ranges"]
for step in range(3):
    st, ref = two3["hist"][step], one0["hist"][step]
    print(os.environ.get("TAG", ""), "step", step, [(k.replace("cross_fusion_encoders", "enc").replace("tokens_to_features", "K9").replace("patches_to_token", "K1"),
          f"{((st['grad'][lo:hi] - ref['grad'][lo:hi]).norm() / ref['grad'][lo:hi].norm().clamp_min(1e-30)).item():.3f}") for k, (lo, hi) in zip(keys, ranges)], flush=True)
