"""Seeded case definitions shared by make_golden.py (run once, with the reference
importable) and the tests (run anywhere).  Inputs and weights come from numpy's
MT19937 ``RandomState`` so they can be regenerated bit-identically without
storing them; the small cases store them in the fixture anyway.
"""
from __future__ import annotations

import numpy as np

# name -> config.  "mask_lens": valid language length per sample (None = no mask passed at all)
ENCODER_CASES = {
    "enc_small": dict(B=2, Nv=12, Nl=9, d=64, h=4, L=2, mask_lens=[6, 9], seed=101),
    "enc_hd18": dict(B=2, Nv=20, Nl=7, d=72, h=4, L=1, mask_lens=[7, 3], seed=102),
    "enc_nomask": dict(B=1, Nv=9, Nl=5, d=32, h=2, L=1, mask_lens=None, seed=103),
    "enc_local1": dict(B=2, Nv=20, Nl=6, d=64, h=4, L=1, mask_lens=[4, 6], seed=104, grid=(4, 5), local_k=1),
    # real width, weights regenerated from the seed (not stored), outputs stored sub-sampled
    "enc_d768": dict(B=1, Nv=196, Nl=64, d=768, h=4, L=1, mask_lens=[40], seed=105, big=True),
}

LEVEL_CASES = {
    # feature map [B,C,H,W], patch p -> Nv = (H//p)*(W//p); H % p != 0 exercises F.fold's zero border
    "level_p2": dict(B=2, C=8, H=9, W=10, p=2, Nl=5, d=32, h=2, L=1, mask_lens=[5, 2], seed=201),
    "level_p1": dict(B=1, C=16, H=3, W=4, p=1, Nl=4, d=32, h=4, L=2, mask_lens=[3], seed=202),
}


def encoder_param_shapes(d: int, L: int, ff_mult: int = 2):
    ff = int(d * ff_mult)
    shapes = {
        "image_kind_embedding": (1, 1, d),
        "lang_kind_embedding": (1, 1, d),
        "heatmap_token": (1, 1, d),
    }
    for j in range(L):
        p = f"t_encoder.layers.{j}."
        shapes.update({
            p + "self_attn.in_proj_weight": (3 * d, d),
            p + "self_attn.in_proj_bias": (3 * d,),
            p + "self_attn.out_proj.weight": (d, d),
            p + "self_attn.out_proj.bias": (d,),
            p + "linear1.weight": (ff, d),
            p + "linear1.bias": (ff,),
            p + "linear2.weight": (d, ff),
            p + "linear2.bias": (d,),
            p + "norm1.weight": (d,),
            p + "norm1.bias": (d,),
            p + "norm2.weight": (d,),
            p + "norm2.bias": (d,),
        })
    shapes["final_norm_layer.weight"] = (d,)
    shapes["final_norm_layer.bias"] = (d,)
    return shapes


def make_encoder_params(seed: int, d: int, L: int):
    """Non-trivial values for EVERY parameter (biases and LN affine included)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shp in encoder_param_shapes(d, L).items():
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "final_norm_layer.weight":
            out[name] = (1.0 + 0.1 * rs.randn(*shp)).astype(np.float32)
        elif name.endswith("weight") and len(shp) == 2:
            out[name] = (rs.randn(*shp) / np.sqrt(shp[1])).astype(np.float32)
        elif "kind_embedding" in name or name == "heatmap_token":
            out[name] = rs.randn(*shp).astype(np.float32)
        else:
            out[name] = (0.1 * rs.randn(*shp)).astype(np.float32)
    return out


def make_encoder_inputs(seed: int, B: int, Nv: int, Nl: int, d: int, mask_lens):
    rs = np.random.RandomState(seed + 7919)
    x = rs.randn(B, Nv, d).astype(np.float32)
    lang = rs.randn(B, Nl, d).astype(np.float32)
    gv = rs.randn(B, Nv, d).astype(np.float32)      # cotangents
    gl = rs.randn(B, Nl, d).astype(np.float32)
    mask = None
    if mask_lens is not None:
        mask = np.zeros((B, Nl), dtype=bool)        # True = pad / ignore (torch convention)
        for b, n in enumerate(mask_lens):
            mask[b, n:] = True
        gl = gl * (~mask)[..., None]                 # padded rows carry no gradient
    return x, lang, mask, gv, gl


def make_level_extras(seed: int, B, C, H, W, p, d):
    rs = np.random.RandomState(seed + 104729)
    feat = rs.randn(B, C, H, W).astype(np.float32)
    conv_w = (rs.randn(d, C, p, p) / np.sqrt(C * p * p)).astype(np.float32)
    reg_w = (rs.randn(p * p * C, d) / np.sqrt(d)).astype(np.float32)
    reg_b = (0.1 * rs.randn(p * p * C)).astype(np.float32)
    gout = rs.randn(B, C, H, W).astype(np.float32)
    return feat, conv_w, reg_w, reg_b, gout
