#!/usr/bin/env python3
"""Experiment: does running the B=32 block as two B=16 halves on two HIP streams (kernels of different character co-resident,
HBM-bound phases of one half under the MFMA phases of the other) beat one B=32 pass?  Forward + backward only (no optimizer);
two independent encoders stand in for one encoder working on two halves.  A data point for DESIGN.md."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer

NV, NL, D, H, L = 196, 512, 768, 4, 4
dev = torch.device("cuda", 0)


def make(B, seed):
    torch.manual_seed(seed)
    enc = CrossTransformerModuleBox(no_patches=2048, pos_embedding_layer=PositionalEmbeddingLayer("sin1d", 2048, D), lang_pos_embedding=None,
                                    num_layers=L, patch_dropout=0.1, num_heads=H, fforward_multiplier=2, token_dropout=0.15,
                                    back_to_img_fn="regroup", activ_f="gelu", final_norm="ln", input_f_size=D).to(dev).train()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, NV, D, generator=g).to(dev).requires_grad_(True)
    lang = torch.nn.functional.normalize(torch.randn(B, NL, D, generator=g), dim=-1).to(dev).requires_grad_(True)
    lens = torch.randint(NL // 4, NL + 1, (B,), generator=g)
    pad = (torch.arange(NL).view(1, -1) >= lens.view(-1, 1)).to(dev)
    gv = torch.randn(B, NV, D, generator=g).to(dev) * 1e-2
    gl = torch.randn(B, NL, D, generator=g).to(dev) * 1e-2
    return enc, x, lang, pad, gv, gl


def fwd_bwd(pack):
    enc, x, lang, pad, gv, gl = pack
    out = enc(x, lang, pad)
    v, l_ = out[0], out[1]
    torch.autograd.backward([v, l_], [gv.to(v.dtype), gl.to(l_.dtype)])


def timed(fn, n=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = make(32, 1)
ms_full = timed(lambda: fwd_bwd(full))
print(f"one B=32 pass:                    {ms_full:.3f} ms", flush=True)
ha, hb = make(16, 2), make(16, 3)


def seq():
    fwd_bwd(ha)
    fwd_bwd(hb)


ms_seq = timed(seq)
print(f"two B=16 passes, one stream:      {ms_seq:.3f} ms", flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def conc(stagger_fwd=False):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    # interleave at fwd / bwd granularity: fwd A | fwd B | bwd A | bwd B, each on its own stream
    with torch.cuda.stream(s1):
        enc, x, lang, pad, gv, gl = ha
        oa = enc(x, lang, pad)
    with torch.cuda.stream(s2):
        enc, x, lang, pad, gv, gl = hb
        ob = enc(x, lang, pad)
    with torch.cuda.stream(s1):
        torch.autograd.backward([oa[0], oa[1]], [ha[4].to(oa[0].dtype), ha[5].to(oa[1].dtype)])
    with torch.cuda.stream(s2):
        torch.autograd.backward([ob[0], ob[1]], [hb[4].to(ob[0].dtype), hb[5].to(ob[1].dtype)])
    cur.wait_stream(s1)
    cur.wait_stream(s2)


ms_conc = timed(conc)
print(f"two B=16 passes, two streams:     {ms_conc:.3f} ms", flush=True)
