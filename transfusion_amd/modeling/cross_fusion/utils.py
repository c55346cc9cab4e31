"""Token plumbing of the fusion block -- host-side mirror of the reference's
``modeling/cross_fusion/utils.py`` (same public names, arguments and error behaviour); the
arithmetic runs in libtfusion_hip.so.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from transfusion_amd import ops

cache_masks = {}


def get_visual_token_mask(img_shape, mask_type):
    """reference utils.py:9-32.  "global" -> None; "local_k" -> [Nv,Nv] float, 1 = blocked."""
    if mask_type == "global":
        return None
    elif "local" in mask_type:
        key = str(tuple(img_shape)) + mask_type
        if key not in cache_masks:
            k = int(mask_type.split("_")[-1])
            h, w = int(img_shape[0]), int(img_shape[1])
            r = torch.arange(h).view(h, 1, 1, 1)
            c = torch.arange(w).view(1, w, 1, 1)
            rr = torch.arange(h).view(1, 1, h, 1)
            cc = torch.arange(w).view(1, 1, 1, w)
            # the reference clamps the window INTO the image, so windows near a border keep (2k+1)^2 extent
            # only in the sense of clamped coordinates: a cell is open iff it is a clamped window coordinate
            lo_r, hi_r = (r - k).clamp(0, h - 1), (r + k).clamp(0, h - 1)
            lo_c, hi_c = (c - k).clamp(0, w - 1), (c + k).clamp(0, w - 1)
            open_ = (rr >= lo_r) & (rr <= hi_r) & (cc >= lo_c) & (cc <= hi_c)
            cache_masks[key] = (~open_).reshape(h * w, h * w).to(torch.float32)
        return cache_masks[key]
    else:
        raise NotImplementedError()


def get_sin1d_embed(no_embeds, dim):
    """reference utils.py:267-273 -> [1, no_embeds, dim]."""
    position = torch.arange(no_embeds).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2) * (-math.log(10000.0) / dim))
    pe = torch.zeros(no_embeds, 1, dim)
    pe[:, 0, 0::2] = torch.sin(position * div_term)
    pe[:, 0, 1::2] = torch.cos(position * div_term)
    return pe.permute((1, 0, 2))


def patchify_image(image, patch_w, patch_h):
    """reference utils.py:35-39.  [B,C,H,W] -> [B, H'W', C*ph*pw] (device gather kernel)."""
    return ops.patchify(image, patch_h, patch_w)


def regroup_patches(patches, init_h, init_w, patch_h, patch_w):
    """reference utils.py:42-46 (transpose + F.fold, kernel == stride): [B,Nv,C*ph*pw] -> [B,C,H,W]."""
    return ops.regroup(patches, init_h, init_w, patch_h, patch_w)


class PositionalEmbeddingLayer(nn.Module):
    """reference utils.py:172-218.  The add itself is fused into the token-assemble kernel; this module
    owns the table (buffer/parameter name ``pos_embedding``, shape [1, num_patches, token_dim])."""

    def __init__(self, embedding_type, num_patches, token_dim, temporal_dim=0):
        super().__init__()
        self.embedding_type = embedding_type
        self.num_patches = num_patches
        self.token_dim = token_dim
        self.temporal_dim = temporal_dim
        if temporal_dim:
            raise NotImplementedError("temporal embeddings belong to the SpaceTime variant (out of scope, SURVEY.md 2 #1)")
        if self.embedding_type == "learned":
            self.pos_embedding = nn.Parameter(torch.randn(1, self.num_patches, self.token_dim))
        elif self.embedding_type == "zero":
            self.pos_embedding = nn.Parameter(torch.zeros(1, self.num_patches, self.token_dim))
        elif self.embedding_type == "sin1d":
            self.register_buffer("pos_embedding", get_sin1d_embed(self.num_patches, self.token_dim))
        else:
            raise ValueError(f"{self.embedding_type=} is not recognized for {temporal_dim}")

    def forward(self, x):
        _, np_, _ = x.shape
        return x + self.pos_embedding[:, :np_, :].to(x.dtype)


class RegroupPatchesLayerBox(nn.Module):
    """reference utils.py:84-119: Dropout -> Linear(d -> ph*pw*C) -> (identity act / norm) -> fold.

    Parameter names ``linear.weight`` / ``linear.bias`` as in the reference.  ``init_h`` / ``init_w`` are
    overwritten per batch by the wrapper (cross_f_box_wrapper.py:180-181).
    """

    def __init__(self, token_dim, init_h, init_w, patch_h, patch_w, out_channels, backproj_dropout=0.1, activ_f=None,
                 final_norm=False):
        super().__init__()
        if activ_f is not None:
            raise NotImplementedError(f"backproj_activ_f={activ_f!r}: only null is used by the shipped configs")
        if final_norm:
            raise NotImplementedError("RegroupPatchesLayerBox.final_norm is unused by the egonao path")
        self.init_h, self.init_w = init_h, init_w
        self.patch_h, self.patch_w = patch_h, patch_w
        self.out_channels = out_channels
        self.backproj_dropout = backproj_dropout
        self.linear = nn.Linear(token_dim, patch_h * patch_w * out_channels)   # parameter holder; forward never called
        self.final_norm = final_norm
        self.precision = "bf16"          # "fp32": run.precision 32 (CrossFusionBoxWrapper.set_precision)
        self.accumulate_linear_grad = False       # see PatchToToken

    def forward(self, x, cls_f=None):
        p_drop = self.backproj_dropout if self.training else 0.0
        if self.precision == "fp32":
            # run.precision 32: split (+ dropout) -> three-pass GEMM -> the fold adds the result's hi + lo planes into the fp32 map
            return ops.back_project_fp32(x, self.linear.weight, self.linear.bias, p_drop, self.init_h, self.init_w, self.patch_h, self.patch_w)
        y = ops.linear(x, self.linear.weight, self.linear.bias, p_drop_in=p_drop, accumulate=self.accumulate_linear_grad)
        return ops.regroup(y, self.init_h, self.init_w, self.patch_h, self.patch_w, out_dtype=torch.float32)
