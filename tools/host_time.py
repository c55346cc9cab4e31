#!/usr/bin/env python3
"""How long does the HOST take to enqueue one training step (no device sync inside the loop)?  usage: python tools/host_time.py"""
import importlib.util, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from transfusion_amd.runner.trainer import FusionTrainStep
dev = torch.device("cuda", 0)
enc = b.make_encoder(dev); enc.train()
tr = FusionTrainStep(enc, lr=1e-4, weight_decay=2e-4, grad_clip=1.0)
batch = b.make_batch(int(os.environ.get("B", 32)), dev, 0)
for _ in range(5): tr.step([batch], b.loss_fn)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n): tr.step([batch], b.loss_fn)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step; device-complete {1e3 * (t2 - t0) / n:.2f} ms/step")
