"""Soak run: N training steps of the benchmark encoder on a FIXED batch (so the loss must fall), checking the loss and the
parameters for non-finite values.  usage: python tools/soak.py [steps] [batch] [lr]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from transfusion_amd.runner.trainer import FusionTrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = bench.make_encoder(dev)
enc.train()
trainer = FusionTrainStep(enc, lr=lr, weight_decay=2e-4, grad_clip=1.0, overlap=True)
batch = bench.make_batch(B, dev, 0)
first = last = None
for i in range(steps):
    loss = trainer.step([batch], bench.loss_fn)
    if i % 25 == 0 or i == steps - 1:
        v = float(loss.item())
        if not math.isfinite(v):
            raise SystemExit(f"non-finite loss at step {i}")
        first = v if first is None else first
        last = v
        print(f"step {i:4d} loss {v:.6f}", flush=True)
torch.cuda.synchronize()
finite = bool(torch.isfinite(trainer.flat.flat).all().item())
ok = finite and last < first
print(f"parameters finite: {finite}; loss {first:.6f} -> {last:.6f}: {'OK' if ok else 'FAIL'}")
sys.exit(0 if ok else 1)
