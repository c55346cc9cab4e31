#!/usr/bin/env python3
"""Training-step time at BASELINE configs[3] (long-context stress: 28x28 visual grid + 1024 narration tokens, d=1024, 4 heads ->
head dim 256, 4 layers), B=8 samples per GPU (64 over 8 GPUs).  A data point for DESIGN.md, not a bench line."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
from transfusion_amd.runner.trainer import FusionTrainStep
B, NV, NL, D, H, L = 8, 784, 1024, 1024, 4, 4
dev = torch.device("cuda", 0)
torch.manual_seed(42)
enc = CrossTransformerModuleBox(no_patches=8192, pos_embedding_layer=PositionalEmbeddingLayer("sin1d", 8192, D), lang_pos_embedding=None,
                                num_layers=L, patch_dropout=0.1, num_heads=H, fforward_multiplier=2, token_dropout=0.15,
                                back_to_img_fn="regroup", activ_f="gelu", final_norm="ln", input_f_size=D).to(dev).train()
tr = FusionTrainStep(enc, lr=1e-4, weight_decay=2e-4, grad_clip=1.0)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, NV, D, generator=g).to(dev)
lang = torch.nn.functional.normalize(torch.randn(B, NL, D, generator=g), dim=-1).to(dev)
lens = torch.randint(NL // 4, NL + 1, (B,), generator=g)
pad = (torch.arange(NL).view(1, -1) >= lens.view(-1, 1)).to(dev)
valid = (~pad).unsqueeze(-1).float()

def loss_fn(m, batch):
    v, l_, _, _ = m(x, lang, pad, lang_valid_rows=int(lens.sum()) if os.environ.get("DENSE") != "1" else None)
    return v.float().square().mean() + (l_.float().square() * valid).sum() / (valid.sum() * D)

for _ in range(3):
    tr.step([None], loss_fn)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    loss = tr.step([None], loss_fn)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
S = NV + NL
fl = 3 * L * (16 * S * D * D + 4 * S * S * D) * B
print(f"stress shape: {ms:.2f} ms/step, {B / ms * 1e3:.1f} samples/s, {fl / ms / 1e9:.0f} TFLOP/s on the block ({fl / ms / 1e9 / 2500 * 100:.1f} % of peak), loss {float(loss):.4f}")

# in-situ kernel table of this shape (library launch tracer, 3 steps)
import ctypes
from transfusion_amd import _lib as Lb
lib = Lb.load()
Lb.check(lib.tf_trace_start(), "tf_trace_start")
for _ in range(3):
    tr.step([None], loss_fn)
cap = 1 << 13
recs = (Lb.TfTraceRecord * cap)()
nrec = lib.tf_trace_stop(ctypes.addressof(recs), cap)
agg = {}
for i in range(min(nrec, cap)):
    r = recs[i]
    a = agg.setdefault(r.name.decode(), [0.0, 0, 0.0])
    a[0] += r.us; a[1] += 1; a[2] += r.flops
for name, (us, cnt, fl2) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {name:36s} {us / cnt:9.1f} us x{cnt / 3:5.1f} = {us / 3:8.1f} us/step  {'' if not fl2 else f'{fl2 / us / 1e6:7.1f} TF/s'}")
