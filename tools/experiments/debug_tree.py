#!/usr/bin/env python3
"""Which reducer unit differs, and in which run?  The two-rank and the single-process wrapper runs of tests/test_gpu_ddp.py at weight-gradient
delays 0 and 3000 us, every run against the undelayed single process, per unit."""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import test_gpu_ddp as T

tmp = pathlib.Path(tempfile.mkdtemp())
levels = getattr(T, os.environ.get("LEVELS", "_LEVELS_RAGGED"))
two0, one0 = T._run_tree(tmp, levels, "a", delay_us=0)
two3, one3 = T._run_tree(tmp, levels, "b", delay_us=3000)
keys, ranges = two0["keys"], two0["ranges"]
def per_unit(st, ref):
    return [(k, f"{((st['grad'][lo:hi] - ref['grad'][lo:hi]).norm() / ref['grad'][lo:hi].norm().clamp_min(1e-30)).item():.4f}") for k, (lo, hi) in zip(keys, ranges)]
for name, run in (("two@0", two0), ("one@3000", one3), ("two@3000", two3)):
    for step in range(3):
        print(name, "step", step, per_unit(run["hist"][step], one0["hist"][step]), flush=True)
print("order", [keys[u] for u in two3["order"]], "grouped", two3["grouped"], one3["grouped"])
