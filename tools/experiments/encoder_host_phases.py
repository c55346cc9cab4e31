#!/usr/bin/env python3
"""Where the HOST time of one encoder forward / backward goes (B = 4, the wrapper's host-bound regime): wall time of
_make_desc, the tf_encoder_fwd call, _bind_grads and the tf_encoder_bwd call, by wrapping them.  A data point, not a test."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transfusion_amd import _lib as L
from transfusion_amd.modeling.cross_fusion.ego_fusion import cross_f_box_layers as M
from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer

acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, g)

B, NV, NL, D = int(os.environ.get("B", 4)), 196, 512, 768
dev = torch.device("cuda", 0)
enc = M.CrossTransformerModuleBox(no_patches=2048, pos_embedding_layer=PositionalEmbeddingLayer("sin1d", 2048, D), lang_pos_embedding=None,
                                  num_layers=4, patch_dropout=0.1, num_heads=4, fforward_multiplier=2, token_dropout=0.15,
                                  back_to_img_fn="regroup", activ_f="gelu", final_norm="ln", input_f_size=D).to(dev).train()
wrap(enc, "_make_desc", "make_desc")
wrap(enc, "_bind_grads", "bind_grads")
orig_call = L.call
def call(fn, *a, **k):
    t = time.perf_counter()
    r = orig_call(fn, *a, **k)
    acc[fn] = acc.get(fn, 0.0) + time.perf_counter() - t
    return r
L.call = call
M.L.call = call
x = torch.randn(B, NV, D, device=dev, requires_grad=True)
lang = torch.randn(B, NL, D, device=dev)
pad = torch.zeros(B, NL, dtype=torch.bool, device=dev)
gv = torch.randn(B, NV, D, device=dev)
def step():
    enc.zero_grad(set_to_none=True)
    v, l_, _, _ = enc(x, lang, pad)
    v.backward(gv.to(v.dtype))
for _ in range(10): step()
torch.cuda.synchronize(); acc.clear()
n = 100
t0 = time.perf_counter()
for _ in range(n): step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"B={B}: host {host / n * 1e6:.0f} us/step, device-complete {tot / n * 1e6:.0f} us/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:20s} {v / n * 1e6:8.1f} us/step")
