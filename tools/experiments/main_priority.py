#!/usr/bin/env python3
"""Experiment: does running the whole training step on a HIGH-priority HIP stream (the side stream of the weight gradients stays at
the default, lowest priority) make the dispatcher favour the chain's workgroups?  usage: PRIO=-1|0 python tools/experiments/main_priority.py"""
import importlib.util, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from transfusion_amd.runner.trainer import FusionTrainStep
dev = torch.device("cuda", 0)
prio = int(os.environ.get("PRIO", 0))
enc = b.make_encoder(dev); enc.train()
tr = FusionTrainStep(enc, lr=1e-4, weight_decay=2e-4, grad_clip=1.0)
batches = [b.make_batch(32, dev, 0, variant=v) for v in range(4)]
st = torch.cuda.Stream(device=dev, priority=prio)
st.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(st):
    for i in range(5): tr.step([batches[i % 4]], b.loss_fn)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for i in range(n): tr.step([batches[i % 4]], b.loss_fn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print(f"priority {prio}: {dt * 1e3:.3f} ms/step, {32 / dt:.1f} samples/s")
