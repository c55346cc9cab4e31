#!/usr/bin/env python3
"""cProfile of the HOST side of one bare-encoder training step (B from the environment, default 4): at the reference's per-GPU batch the
step is bound by how fast the host enqueues it (bench legs.b4.host_enqueue_ms).  usage: B=4 python tools/host_profile_encoder.py"""
import cProfile, importlib.util, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from transfusion_amd.runner.trainer import FusionTrainStep
dev = torch.device("cuda", 0)
enc = b.make_encoder(dev); enc.train()
tr = FusionTrainStep(enc, lr=1e-4, weight_decay=2e-4, grad_clip=1.0)
batches = [b.make_batch(int(os.environ.get("B", 4)), dev, 0, variant=v) for v in range(2)]
for i in range(6): tr.step([batches[i % 2]], b.loss_fn)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for i in range(n): tr.step([batches[i % 2]], b.loss_fn)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step; device-complete {1e3 * (t2 - t0) / n:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(20): tr.step([batches[i % 2]], b.loss_fn)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
