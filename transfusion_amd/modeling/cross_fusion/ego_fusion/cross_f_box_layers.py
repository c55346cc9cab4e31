"""Fusion encoder -- host-side mirror of the reference's
``modeling/cross_fusion/ego_fusion/cross_f_box_layers.py:13-108`` (CrossTransformerModuleBox):
same constructor keywords, forward signature, return tuple, parameter / buffer names and error
behaviour, so it drops into ``get_cross_box_encoder`` / reference checkpoints unchanged.

The whole forward (pos-emb + kind-emb + patch dropout + concat, L post-norm encoder layers with fused
attention, final LayerNorm) is ONE call into the native encoder runtime ``tf_encoder_fwd`` of
libtfusion_hip.so, and the whole backward one call to ``tf_encoder_bwd``; PyTorch sees a single
autograd node.  Compute dtype is bf16 (fp32 statistics / accumulation); parameters stay fp32.
"""
from __future__ import annotations

import ctypes as C
import os
import math

import torch
from torch import nn

from transfusion_amd import _lib as L
from transfusion_amd import ops

# dropout site ids used by the runtime (tf_api.hip): patch = 0, layer l -> 16 + 8*l + {1 attn, 2 dropout1, 3 ffn, 4 dropout2}
SITE_PATCH = 0


def site_of(layer: int, which: int) -> int:
    return 16 + 8 * layer + which


class _OutProj(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d, d))
        self.bias = nn.Parameter(torch.zeros(d))


class _SelfAttnParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names and initialisation
    (torch18_adapters.py:190-223: xavier_uniform in_proj_weight, zero biases)."""

    def __init__(self, d, nhead):
        super().__init__()
        self.embed_dim, self.num_heads = d, nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _OutProj(d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.kaiming_uniform_(self.out_proj.weight, a=math.sqrt(5))


class _EncoderLayerParams(nn.Module):
    """Parameter holder with nn.TransformerEncoderLayer's names (torch18_adapters.py:71-85)."""

    def __init__(self, d, nhead, ff):
        super().__init__()
        self.self_attn = _SelfAttnParams(d, nhead)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.LayerNorm(d, eps=1e-5)
        self.norm2 = nn.LayerNorm(d, eps=1e-5)


class _EncoderStack(nn.Module):
    """``nn.TransformerEncoder(layer, L)`` deep-copies one layer L times (cross_f_box_layers.py:58), so all
    layers start from identical weights; reproduced here."""

    def __init__(self, d, nhead, ff, num_layers):
        super().__init__()
        first = _EncoderLayerParams(d, nhead, ff)
        layers = [first]
        for _ in range(num_layers - 1):
            nxt = _EncoderLayerParams(d, nhead, ff)
            nxt.load_state_dict(first.state_dict())
            layers.append(nxt)
        self.layers = nn.ModuleList(layers)
        self.num_layers = num_layers


_LAYER_FIELDS = (
    ("in_w", "self_attn.in_proj_weight"), ("in_b", "self_attn.in_proj_bias"), ("out_w", "self_attn.out_proj.weight"),
    ("out_b", "self_attn.out_proj.bias"), ("w1", "linear1.weight"), ("b1", "linear1.bias"), ("w2", "linear2.weight"),
    ("b2", "linear2.bias"), ("n1_w", "norm1.weight"), ("n1_b", "norm1.bias"), ("n2_w", "norm2.weight"), ("n2_b", "norm2.bias"),
)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, need_grad, x, lang, pad_mask, *params):
        # need_grad comes from the caller: inside Function.forward grad mode is always off
        desc, keep = mod._make_desc(x, lang, pad_mask)
        B, Nl, d = lang.shape
        out_dtype = x.dtype
        # (ragged groups: x and the visual outputs are the concatenation [sum_g B_g nv_g, d] of the groups' tokens)
        vis_out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        lang_out = torch.empty(B, Nl, d, dtype=out_dtype, device=x.device)
        desc.vis_out, desc.vis_out_is_f32 = L.ptr(vis_out), ops._is_f32(vis_out)
        desc.lang_out, desc.lang_out_is_f32 = L.ptr(lang_out), ops._is_f32(lang_out)
        st = ops._stream()
        if desc.packed_rows > 0:
            check_packed_row_errors(sync=False)          # an earlier call whose lang_valid_rows disagreed with its mask raises here
        desc.repack = 1 if mod._wpack_dirty() else 0     # bf16 weight shadows are refreshed inside the forward call
        watch = desc.packed_rows > 0 and not torch.cuda.is_current_stream_capturing()
        slot = _claim_packed_slot(desc) if watch else None
        L.call("tf_encoder_fwd", desc, st)
        desc.repack = 0
        desc.packed_error_host = None        # (the descriptor lives on for the backward: the word belongs to THIS forward's row map)
        if watch:
            _watch_packed_rows(desc, slot, x.device)
        ctx.mod, ctx.desc, ctx.keep, ctx.gen = mod, desc, keep, keep["gen"]
        ctx.group_mods = list(mod._group_mods) if mod._group_mods else None
        mod._last_desc = desc               # debug / test hooks (packed_row_error, peek)
        # per-call tensors the descriptor points at live on ctx (the workspace item may be recycled by a later forward)
        ctx.held = (keep["inputs"], keep.get("mask"), keep.get("block_bits"))
        ctx.io = (x.dtype, lang.dtype, x.requires_grad, lang.requires_grad)
        ctx.x_shape = tuple(x.shape)
        ctx.nparams = len(params)
        if not need_grad:
            mod._release(keep)           # no graph: the saved activations are not needed
        return vis_out, lang_out

    @staticmethod
    def backward(ctx, g_vis, g_lang):
        mod, desc, keep = ctx.mod, ctx.desc, ctx.keep
        if keep["gen"] != ctx.gen:
            raise L.TfError("the activations of this forward were recycled: its workspace was released (a backward already ran, "
                            "e.g. retain_graph=True) and reused by a later forward of the same shape")
        x_dtype, lang_dtype, x_rg, lang_rg = ctx.io
        B, Nv, Nl, d = desc.B, desc.Nv, desc.Nl, desc.d
        dev = keep["work"].device
        g_vis = None if g_vis is None else g_vis.contiguous()
        g_lang = None if g_lang is None else g_lang.contiguous()
        if g_vis is None and g_lang is None:
            mod._release(keep)
            return (None,) * (5 + ctx.nparams)
        desc.d_vis_out, desc.d_vis_out_is_f32 = L.ptr(g_vis), 0 if g_vis is None else ops._is_f32(g_vis)
        desc.d_lang_out, desc.d_lang_out_is_f32 = L.ptr(g_lang), 0 if g_lang is None else ops._is_f32(g_lang)
        d_vis = torch.empty(ctx.x_shape, dtype=x_dtype, device=dev) if x_rg else None
        d_lang = torch.empty(B, Nl, d, dtype=lang_dtype, device=dev) if lang_rg else None
        desc.d_vis, desc.d_vis_is_f32 = L.ptr(d_vis), 0 if d_vis is None else ops._is_f32(d_vis)
        desc.d_lang, desc.d_lang_is_f32 = L.ptr(d_lang), 0 if d_lang is None else ops._is_f32(d_lang)
        grads, direct = mod._bind_grads(desc, dev)
        hook = mod.layer_grad_hook
        if hook is None:
            L.call("tf_encoder_bwd", desc, ops._stream())
        else:
            # one runtime call per layer, top layer first; after each, the hook may start reducing that layer's gradients
            # (data-parallel all-reduce overlapped with the rest of the backward)
            # The side stream is NOT joined after each call (defer_join): the hook orders its collective after the
            # side stream's own events, and whoever consumes the gradients calls ops.join_overlap() first.
            st = ops._stream()
            desc.defer_join = 1
            members = ctx.group_mods or (mod,)
            for layer in range(desc.L - 1, -1, -1):
                desc.bwd_hi, desc.bwd_nlayers = layer, 1
                L.call("tf_encoder_bwd", desc, st)
                for m in members:                      # a grouped call finishes this layer of EVERY member at once
                    m.layer_grad_hook(m, layer)
            desc.bwd_nlayers, desc.defer_join = 0, 0
            if not getattr(getattr(hook, "__self__", None), "joins_overlap", False):
                ops.join_overlap(dev)
        mod._release(keep)
        if direct:
            ops.note_grad_writer(dev)                   # (ops.note_grad_writer: FusionTrainStep joins this stream before the optimiser)
            return (None, None, d_vis, d_lang, None) + (None,) * ctx.nparams
        return (None, None, d_vis, d_lang, None) + tuple(grads)


# ---- packed batches: the host's row count against the mask's, read back lazily ------------------------------------------------------
# The device finds a `lang_valid_rows` that disagrees with the mask (row_map_kernel: it then keeps every access inside the tensors the
# host sized, and the step computes garbage for the truncated samples).  The error word travels to pinned host memory behind the
# forward, on its stream, with an event -- no host synchronisation in the step -- and the NEXT packed forward whose copy has landed
# raises (``check_packed_row_errors(sync=True)`` waits for all of them: end of an epoch, tests).
_packed_flags = []          # (slot, packed_rows the host claimed, device) of forwards whose word has not been looked at yet
_PACKED_SLOTS = 256
_packed_pin = None          # pinned int32 ring: the row-map kernel of a forward stores its verdict into the forward's word
_packed_next = 0


def _claim_packed_slot(desc):
    """A word of pinned host memory for this forward's verdict, preset to -1 (not landed), handed to the runtime through the descriptor:
    the row-map kernel stores 0 or the mask's row count into it -- no device-to-host copy, no event on the forward's stream."""
    global _packed_pin, _packed_next
    if _packed_pin is None:
        _packed_pin = torch.full((_PACKED_SLOTS,), -1, dtype=torch.int32).pin_memory()
    if len(_packed_flags) >= _PACKED_SLOTS - 1:        # the ring is full of words nobody looked at: look (and WAIT for the oldest ones)
        check_packed_row_errors(sync=False)
        while len(_packed_flags) >= _PACKED_SLOTS - 1:
            _check_one_packed_flag(sync=True)
    slot = _packed_next
    _packed_next = (_packed_next + 1) % _PACKED_SLOTS
    _packed_pin[slot] = -1
    desc.packed_error_host = _packed_pin.data_ptr() + 4 * slot
    return slot


def _watch_packed_rows(desc, slot, device):
    _packed_flags.append((slot, int(desc.packed_rows), device))
    if len(_packed_flags) > 64:
        check_packed_row_errors(sync=False)


def _check_one_packed_flag(sync: bool) -> bool:
    """Looks at the oldest pending error word; False when its copy has not landed yet (and ``sync`` is off)."""
    slot, want, device = _packed_flags[0]
    got = int(_packed_pin[slot])
    if got < 0:                          # the row-map kernel of that forward has not run (or its store has not landed) yet
        if not sync:
            return False
        torch.cuda.synchronize(device)
        got = int(_packed_pin[slot])
        if got < 0:
            raise L.TfError("lang_valid_rows: a forward's row-map verdict never reached the host (packed_error_host)")
    _packed_flags.pop(0)
    if got != 0:
        _packed_flags.clear()
        if got == want:                  # the total agrees, the split over the groups does not (row_map_kernel's `bad`)
            raise L.TfError(f"lang_valid_rows: the {want} token rows of an earlier grouped forward are not shared equally by its groups "
                            "(with grouped levels every group must drop the same tokens).  That step's outputs and gradients are wrong.")
        raise L.TfError(f"lang_valid_rows: an earlier forward was told {want} token rows, its padding mask holds {got} "
                        "(the count must be B * Nv + the number of un-masked language tokens).  That step's outputs and gradients are wrong.")
    return True


def check_packed_row_errors(sync: bool = True):
    """Raises TfError for any earlier packed forward whose ``lang_valid_rows`` disagreed with its padding mask."""
    while _packed_flags and _check_one_packed_flag(sync):
        pass


_warned_dense_rows = False


def _warn_dense_rows_once():
    """A masked batch reached an encoder in training without ``lang_valid_rows``: the call is the reference's dense computation (every
    padded token is carried through every row-wise kernel).  Said once, because nothing else would: the results are the same."""
    global _warned_dense_rows
    if not _warned_dense_rows:
        _warned_dense_rows = True
        import warnings
        warnings.warn("transfusion_amd: a language padding mask was given without lang_valid_rows -- the encoder runs on dense rows "
                      "(padded tokens included).  Pass lang_valid_rows=<number of un-masked tokens> (the pooling layer's valid_tokens) "
                      "to drop them from the row-wise kernels.", RuntimeWarning, stacklevel=3)


class CrossTransformerModuleBox(nn.Module):
    def __init__(
        self,
        no_patches,
        patch_dropout,
        input_f_size,
        pos_embedding_layer,
        num_layers=2,
        num_heads=4,
        classif_token=False,
        fforward_multiplier=2,
        token_dropout=0.1,
        back_to_img_fn="token",
        activ_f="relu",
        patch_norm=False,
        final_norm=False,
        lang_pos_embedding=None,
    ):
        super().__init__()
        self.no_patches = no_patches
        self.classif_token = classif_token
        self.back_to_img_fn = back_to_img_fn
        self.patch_norm = patch_norm
        self.final_norm = final_norm
        self.token_dim = input_f_size
        self.pos_embedding_layer = pos_embedding_layer
        self.image_kind_embedding = nn.Parameter(torch.randn(1, 1, self.token_dim))
        self.lang_kind_embedding = nn.Parameter(torch.randn(1, 1, self.token_dim))
        self.lang_pos_embedding = lang_pos_embedding
        self.heatmap_token = nn.Parameter(torch.randn(1, 1, self.token_dim))     # never used by forward (as in the reference)
        if self.classif_token:
            self.class_token = nn.Parameter(torch.randn(1, 1, self.token_dim))
        self.patch_dropout = patch_dropout
        self.token_dropout = token_dropout
        self.num_heads = num_heads
        self.num_layers = num_layers
        self.dim_feedforward = int(self.token_dim * fforward_multiplier)
        self.register_buffer("padding_mask", torch.zeros(size=(1,), dtype=torch.bool))

        if activ_f not in ("gelu", "relu"):          # nn.TransformerEncoderLayer's _get_activation_fn accepts exactly these two
            raise RuntimeError("activation should be relu/gelu, not {}".format(activ_f))
        self.activ_f = activ_f
        if self.token_dim % num_heads or self.token_dim % 8:
            raise ValueError(f"input_f_size={self.token_dim} must be divisible by num_heads={num_heads} and by 8")
        if num_layers > L.TF_MAX_LAYERS:
            raise ValueError(f"num_layers={num_layers} exceeds TF_MAX_LAYERS={L.TF_MAX_LAYERS}")
        self.t_encoder = _EncoderStack(self.token_dim, num_heads, self.dim_feedforward, num_layers)
        if self.final_norm == "ln":
            self.final_norm_layer = nn.LayerNorm(self.token_dim)
        elif self.final_norm == "bn":
            raise ValueError("not implemented")
        elif self.final_norm is False:
            self.final_norm_layer = nn.Identity()
        else:
            raise ValueError("not implemented")

        # runtime state (not part of state_dict)
        self.accumulate_into_grad = False     # True: backward adds straight into p.grad (flat-buffer training loop)
        self.fp8_projections = os.environ.get("TF_FP8_PROJ") == "1"   # forward QKV / FFN GEMMs with fp8 (e4m3) operands (BASELINE configs[4])
        # "bf16": bf16 compute (run.precision 16 / bf16).  "fp32": the fp32-accuracy mode for run.precision: 32 (Ego4Dv2 YAML, BASELINE
        # configs[2]): hi + lo bf16 planes, three MFMA passes per contraction (include/tfusion.h, TfEncoderDesc.precision)
        self.precision = os.environ.get("TF_PRECISION", "bf16")
        self.layer_grad_hook = None           # callable(module, layer): called as soon as that layer's gradients are enqueued
        # Packed batches (TfEncoderDesc.packed_rows): when forward() is told how many language tokens the padding mask leaves
        # (``lang_valid_rows``, a HOST integer -- the tokeniser / the length list that built the mask knows it), the masked tokens are
        # dropped from every row-wise kernel instead of being carried through the GEMMs, LayerNorms and attention as dead rows.
        # Their fused outputs come back as zeros (the reference multiplies them by the mask wherever it uses them, lm_layers.py:59-61)
        # and cotangents on them are ignored; everything else equals the dense computation.  False: always dense.
        self.pack_tokens = os.environ.get("TF_PACK_TOKENS", "1") != "0"
        self._packed_rows = 0
        self._params_cache = None
        self._param_ptr_cache = None          # (addresses of the parameters, their TfLayerParams block): _make_desc
        self._grad_ptr_cache = None           # the same for the .grad tensors of the direct-accumulation path: _bind_grads
        self._group_mods = None               # set for the duration of a forward_grouped call: [self, the other encoders of the group]
        self._group_stride = None
        self._group_nv = None                 # ... and, for RAGGED groups, the visual tokens per sample of every group
        self._wpack = None
        self._wpack_versions = None
        self._work_pool = {}
        self._grad_layout = None
        self._last_seed = 0

    # ---- parameter plumbing ----------------------------------------------------------------------------
    def _param_list(self):
        """[kind_v, kind_l, 12 tensors per layer in _LAYER_FIELDS order, (final norm w, b)] -- the Parameter OBJECTS, cached: the module
        tree is fixed after construction (`.to()` / `load_state_dict` change `.data` in place), and walking `named_parameters()` of
        every layer on every call was a visible part of the host time per step at small batches."""
        ps = self._params_cache
        if ps is None:
            ps = [self.image_kind_embedding, self.lang_kind_embedding]
            for layer in self.t_encoder.layers:
                sd = dict(layer.named_parameters())
                ps += [sd[name] for _, name in _LAYER_FIELDS]
            if self.final_norm == "ln":
                ps += [self.final_norm_layer.weight, self.final_norm_layer.bias]
            self._params_cache = ps
        return ps

    def _wpack_dirty(self) -> bool:
        mods = self._group_mods or (self,)
        vers = tuple((p.data_ptr(), p._version) for m in mods for p in m._param_list()) + (bool(self.fp8_projections), self.precision, len(mods))
        if vers != self._wpack_versions:
            self._wpack_versions = vers
            return True
        return False

    def mark_weights_updated(self):
        self._wpack_versions = None

    MAX_WORK_SHAPES = 3       # workspaces are kept for this many distinct (B, Nv, Nl, precision) shapes, least recently used first out

    def _get_work(self, key, plan, device):
        # a loader that pads the language tokens to the longest sample of each batch hands over a new Nl almost every step: without a
        # bound the pool would keep one multi-GB workspace per length ever seen.  An evicted workspace that still awaits its backward stays
        # alive through that backward's ctx; it just is not reused afterwards.
        pool = self._work_pool.pop(key, None)
        if pool is None:
            pool = []
            while len(self._work_pool) >= self.MAX_WORK_SHAPES:
                self._work_pool.pop(next(iter(self._work_pool)))
        self._work_pool[key] = pool           # (re)inserted last: dict order is the recency order
        for item in pool:
            if not item["busy"]:
                item["gen"] += 1
                return item
        if len(pool) >= 4:      # many forwards awaiting their backward: the pool stops tracking the oldest (its graph keeps it alive)
            pool.pop(0)
        item = {"work": torch.zeros(plan.work_bytes, dtype=torch.uint8, device=device), "busy": False, "gen": 0}
        pool.append(item)
        return item

    def _release(self, keep):
        keep["busy"] = False

    def _make_desc(self, x, lang, pad_mask):
        ops._require_cuda(x, lang, pad_mask, self.image_kind_embedding)
        gnv = self._group_nv                 # ragged groups (forward_grouped): visual tokens per sample of every group, or None
        if gnv is not None:
            B, Nl, d = lang.shape
            Nv = max(gnv)
            if x.dim() != 2 or x.shape[1] != d or B % len(gnv) or x.shape[0] != (B // len(gnv)) * sum(gnv):
                raise RuntimeError(f"ragged groups {gnv}: visual tokens {tuple(x.shape)} are not the concatenation of {B // len(gnv)} samples per group")
        else:
            B, Nv, d = x.shape
            Nl = lang.shape[1]
        if d != self.token_dim or lang.shape[2] != d or lang.shape[0] != B:
            raise RuntimeError(f"token shapes {tuple(x.shape)} / {tuple(lang.shape)} do not match input_f_size={self.token_dim}")
        params = self._param_list()
        # the parameters' addresses as one key: the dtype / layout checks and the 50-odd ctypes field stores below are redone only when a
        # tensor has moved (host time counts: at the reference's per-GPU batch the step is bound by how fast the host enqueues it)
        # (the cheap dtype / contiguity check runs on EVERY call, outside the cache: an address can survive a `.data` reassignment or a
        # `.to(dtype)` round trip through the caching allocator, and the kernels read these pointers as contiguous fp32 unchecked)
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise L.TfError("parameters must be contiguous fp32 (bf16 shadows are derived inside the runtime)")
        pkey = tuple(p.data_ptr() for p in params)
        pc = self._param_ptr_cache
        if pc is None or pc[0] != pkey:
            blk = (L.TfLayerParams * L.TF_MAX_LAYERS)()
            for j in range(self.num_layers):
                for k, (field, _) in enumerate(_LAYER_FIELDS):
                    setattr(blk[j], field, pkey[2 + 12 * j + k])
            pc = self._param_ptr_cache = (pkey, blk)
        if self.precision not in ("bf16", "fp32"):
            raise ValueError(f"precision={self.precision!r}: 'bf16' or 'fp32'")
        if self.precision == "fp32" and self.fp8_projections:
            raise ValueError("fp8_projections and precision='fp32' exclude each other")
        lib = L.load()
        e = L.TfEncoderDesc()
        e.B, e.Nv, e.Nl, e.d, e.H, e.L, e.ff = B, Nv, Nl, d, self.num_heads, self.num_layers, self.dim_feedforward
        e.precision = 1 if self.precision == "fp32" else 0
        e.act = 1 if self.activ_f == "relu" else 0
        e.packed_rows = 0                   # the plan is the dense one (an upper bound for every packed layout of this shape)
        if self._group_mods:
            e.groups, e.param_gstride = len(self._group_mods), int(self._group_stride)
        if gnv is not None:
            if len(gnv) != e.groups or e.groups > L.CONSTS["TF_MAX_GROUPS"]:
                raise ValueError(f"ragged groups {gnv}: {e.groups} encoders in the call (at most {L.CONSTS['TF_MAX_GROUPS']})")
            for i, n in enumerate(gnv):
                e.group_nv[i] = int(n)
        plan = L.TfEncoderPlan()
        L.check(lib.tf_encoder_plan_ex(C.addressof(e), C.addressof(plan)), "tf_encoder_plan_ex")
        if self._wpack is None or self._wpack.numel() != plan.wpack_bytes or self._wpack.device != x.device:
            self._wpack = torch.zeros(plan.wpack_bytes, dtype=torch.uint8, device=x.device)
            self._wpack_versions = None
        keep = self._get_work((B, Nv, Nl, self.precision) + (() if gnv is None else (tuple(gnv),)), plan, x.device)
        keep["busy"] = True
        e.training = 1 if self.training else 0
        e.final_norm = 1 if self.final_norm == "ln" else 0
        e.fp8_proj = 1 if self.fp8_projections else 0
        e.p_token, e.p_patch = float(self.token_dropout), float(self.patch_dropout)
        self._last_seed = ops.next_seed() if self.training else 0
        e.seed = self._last_seed
        e.kind_v, e.kind_l = pkey[0], pkey[1]
        C.memmove(C.addressof(e.p), C.addressof(pc[1]), C.sizeof(pc[1]))
        if e.final_norm:
            e.fn_w, e.fn_b = pkey[-2], pkey[-1]
        # positional tables: the sin1d BUFFERS are read by the assemble kernel; learned / zero tables are Parameters (utils.py:181-184)
        # whose gradient autograd needs, so forward() has already added those with a torch op and they are not passed here
        pe = self.pos_embedding_layer.pos_embedding
        if gnv is not None:                  # the LONGEST table of the group (group_stride checked that the others are its prefixes)
            pe = self._group_mods[max(range(len(gnv)), key=lambda i: gnv[i])].pos_embedding_layer.pos_embedding
        if not isinstance(pe, nn.Parameter):
            e.pe = self._table(pe, Nv, d, "pos_embedding")
        lpe = None if not self.lang_pos_embedding else self.lang_pos_embedding.pos_embedding
        if lpe is not None and not isinstance(lpe, nn.Parameter):
            e.pe_lang = self._table(lpe, Nl, d, "lang_pos_embedding")
        e.wpack, e.work = self._wpack.data_ptr(), keep["work"].data_ptr()
        bb = getattr(self, "_block_bits", None)
        if bb is not None:
            keep["block_bits"] = bb
            e.attn_block_bits = bb.data_ptr()
        e.overlap = ops.wgrad_overlap(x.device)
        x = x.contiguous()
        lang = lang.contiguous()
        keep["inputs"] = (x, lang)
        e.vis, e.vis_is_f32 = x.data_ptr(), ops._is_f32(x)
        e.lang, e.lang_is_f32 = lang.data_ptr(), ops._is_f32(lang)
        if pad_mask is not None:
            if pad_mask.shape != (B, Nl):
                raise RuntimeError(f"language_tokens_att_maks must be [B, Nl] = {(B, Nl)}, got {tuple(pad_mask.shape)}")
            # the same mask TENSOR OBJECT step after step saves one conversion kernel; identity + version, and the cache holds
            # the tensor (a data_ptr key would match a freed temporary's successor)
            cache = getattr(self, "_pad_u8_cache", None)
            if pad_mask.dtype == torch.bool and pad_mask.is_contiguous():
                m8 = pad_mask.view(torch.uint8)      # a bool tensor IS bytes of 0 / 1: no conversion kernel (8.6 us + a launch per forward)
            elif pad_mask.dtype == torch.uint8 and pad_mask.is_contiguous():
                m8 = pad_mask
            elif cache is not None and cache[0] is pad_mask and cache[1] == pad_mask._version:
                m8 = cache[2]
            else:
                m8 = pad_mask.to(torch.uint8).contiguous()
                self._pad_u8_cache = (pad_mask, pad_mask._version, m8)
            keep["mask"] = m8
            e.lang_pad_mask = m8.data_ptr()
            pr = int(self._packed_rows)
            if pr:
                nvis = x.shape[0] if gnv is not None else B * Nv
                if not (nvis <= pr <= nvis + B * Nl):
                    raise ValueError(f"lang_valid_rows: {pr - nvis} un-masked language tokens do not fit a [{B}, {Nl}] mask")
                e.packed_rows = pr
        if gnv is not None and not e.packed_rows:
            raise L.TfError("ragged groups run on packed token rows: pass lang_valid_rows (and a padding mask) to forward_grouped")
        return e, keep

    @staticmethod
    def _table(pe, n, d, what):
        if pe.dim() != 3 or pe.shape[1] < n or pe.shape[2] != d:
            raise RuntimeError(f"{n} tokens exceed the {what} table {tuple(pe.shape)}")
        if pe.dtype != torch.float32 or not pe.is_contiguous():
            raise L.TfError(f"{what} must be contiguous fp32")
        return pe.data_ptr()

    def _bind_grads(self, desc, device):
        """Point the runtime's gradient slots either at fresh zero buffers (returned through autograd) or,
        with ``accumulate_into_grad``, straight at the preallocated ``p.grad`` tensors."""
        params = self._param_list()
        direct = self.accumulate_into_grad
        if desc.groups > 1 and not direct:
            raise L.TfError("a grouped encoder call accumulates straight into .grad (accumulate_into_grad, FusionTrainStep)")
        grads = []
        if direct:
            for p in params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                elif p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.numel() != p.numel():
                    raise L.TfError("direct accumulation needs contiguous fp32 .grad tensors of the parameters' sizes")
                grads.append(p.grad)
            gkey = tuple(g.data_ptr() for g in grads)
            gc = self._grad_ptr_cache
            if gc is None or gc[0] != gkey:
                blk = (L.TfLayerParams * L.TF_MAX_LAYERS)()
                for j in range(self.num_layers):
                    for k, (field, _) in enumerate(_LAYER_FIELDS):
                        setattr(blk[j], field, gkey[2 + 12 * j + k])
                gc = self._grad_ptr_cache = (gkey, blk)
            desc.g_kind_v, desc.g_kind_l = gkey[0], gkey[1]
            C.memmove(C.addressof(desc.g), C.addressof(gc[1]), C.sizeof(gc[1]))
            if desc.final_norm:
                desc.g_fn_w, desc.g_fn_b = gkey[-2], gkey[-1]
            self._grad_keepalive = None
            return grads, direct
        else:
            # ONE zero fill for all gradients of the call (a zeros_like per parameter is ~50 launches per backward); each
            # gradient is a 256-B aligned view, and no reference is kept here, so AccumulateGrad can adopt the view as p.grad.
            # The layout is computed once per parameter set: one split + one view per parameter (host time counts: the wrapper
            # path is host-bound at the reference's batch sizes).
            lay = self._grad_layout
            sig = tuple((p.requires_grad, p.dtype == torch.float32) for p in params)
            if lay is None or lay[0] != sig:
                sizes = [((p.numel() + 63) // 64) * 64 if (p.requires_grad and p.dtype == torch.float32) else 0 for p in params]
                exact = [p.numel() == n for p, n in zip(params, sizes)]
                lay = self._grad_layout = (sig, sizes, exact, [tuple(p.shape) for p in params], max(sum(sizes), 1))
            _, sizes, exact, shapes, total = lay
            flat = torch.zeros(total, dtype=torch.float32, device=device)
            pieces = flat.split([n for n in sizes if n]) if total > 1 or any(sizes) else ()
            k = 0
            for p, n, ex, shp in zip(params, sizes, exact, shapes):
                if not p.requires_grad:
                    grads.append(None)
                elif n == 0:
                    grads.append(torch.zeros_like(p, memory_format=torch.contiguous_format))
                else:
                    piece = pieces[k]
                    k += 1
                    grads.append(piece.view(shp) if ex else piece[:p.numel()].view(shp))
        scratch = None

        def gp(t, ref):
            nonlocal scratch
            if t is not None:
                return t.data_ptr()
            if scratch is None or scratch.numel() < ref.numel():    # frozen parameter: gradient goes to a throw-away buffer
                scratch = torch.zeros(max(ref.numel(), 1 << 16), dtype=torch.float32, device=device)
            return scratch.data_ptr()

        it = iter(zip(grads, params))
        g, p = next(it); desc.g_kind_v = gp(g, p)
        g, p = next(it); desc.g_kind_l = gp(g, p)
        for j in range(self.num_layers):
            for field, _ in _LAYER_FIELDS:
                g, p = next(it)
                setattr(desc.g[j], field, gp(g, p))
        if desc.final_norm:
            g, p = next(it); desc.g_fn_w = gp(g, p)
            g, p = next(it); desc.g_fn_b = gp(g, p)
        self._grad_keepalive = scratch             # (the returned gradients are owned by autograd; the kernels are stream-ordered)
        return grads, direct

    # ---- reference forward contract (cross_f_box_layers.py:69-108) --------------------------------------
    def forward(self, x, language_tokens, language_tokens_att_maks, vis_tokens_mask=None, lang_valid_rows=None):
        """Reference signature (cross_f_box_layers.py:69) plus one optional keyword: ``lang_valid_rows``, the number of language tokens
        ``language_tokens_att_maks`` leaves un-masked in the whole batch, as a host integer.  When given (and ``pack_tokens``), the masked
        tokens are dropped from the computation (see ``pack_tokens``); when absent the call is the reference's dense computation."""
        # learned / zero positional tables are Parameters: added here by torch so that autograd produces their gradient (the
        # kernel then adds nothing for that table); the default sin1d buffers go into the assemble kernel
        if isinstance(self.pos_embedding_layer.pos_embedding, nn.Parameter):
            x = self.pos_embedding_layer(x)
        if self.lang_pos_embedding and isinstance(self.lang_pos_embedding.pos_embedding, nn.Parameter):
            # reference order (:76-78): (lang + kind) + pe; here (lang + pe) + kind -- the same sum up to fp32 rounding
            language_tokens = self.lang_pos_embedding(language_tokens)
        self._packed_rows = 0
        if lang_valid_rows is not None and language_tokens_att_maks is not None and self.pack_tokens:
            nvis = x.shape[0] if self._group_nv is not None else x.shape[0] * x.shape[1]
            self._packed_rows = nvis + int(lang_valid_rows)
        elif language_tokens_att_maks is not None and self.pack_tokens and self.training:
            _warn_dense_rows_once()
        self._block_bits = None
        if vis_tokens_mask is not None:
            if self._group_nv is not None:
                raise L.TfError("a visual token mask and ragged groups exclude each other")
            self._block_bits = self._pack_block_bits(vis_tokens_mask, x.shape[1], language_tokens.shape[1], x.device)
        params = self._param_list()
        need_grad = torch.is_grad_enabled() and (x.requires_grad or language_tokens.requires_grad or any(p.requires_grad for p in params))
        vis_tokens, lang_tokens = _EncoderFn.apply(self, need_grad, x, language_tokens, language_tokens_att_maks, *params)
        return vis_tokens, lang_tokens, None, None

    # ---- the wrapper's FPN levels as ONE launch sequence (TfEncoderDesc.groups) ----------------------------------
    _GROUP_CFG = ("num_layers", "num_heads", "token_dim", "dim_feedforward", "token_dropout", "patch_dropout", "activ_f", "final_norm",
                  "precision", "fp8_projections", "training", "accumulate_into_grad", "pack_tokens")

    def group_stride(self, mods, ragged=False):
        """``ragged``: the encoders may differ in their number of visual tokens (TfEncoderDesc.group_nv): their positional tables then
        need not have one shape, but every table must be a PREFIX of the longest one (the kernel reads one table by row).
        Byte stride between the parameters of consecutive encoders of ``mods`` (``mods[0] is self``) if they can run as one grouped
        call, else None: identical configuration, gradients accumulated in place, per-layer gradient hooks on all members or on none, fixed positional tables
        with equal contents, and every parameter AND every gradient tensor of encoder g exactly g * stride bytes after encoder 0's --
        what FusionTrainStep's flat buffers give.  Cached on the tensors' addresses."""
        if len(mods) < 2 or mods[0] is not self:
            return None
        key = tuple((p.data_ptr(), 0 if p.grad is None else p.grad.data_ptr()) for m in mods for p in m._param_list()) + (bool(ragged),)
        cached = getattr(self, "_group_check", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        stride = None
        # (per-layer gradient hooks -- a data-parallel reducer -- are fine when EVERY member has one: the grouped backward then runs layer
        # by layer and calls each member's hook for that layer)
        ok = all(isinstance(m, CrossTransformerModuleBox) and m.accumulate_into_grad for m in mods)
        ok = ok and len({m.layer_grad_hook is None for m in mods}) == 1
        ok = ok and all(getattr(m, a) == getattr(self, a) for m in mods for a in self._GROUP_CFG)
        if ok:
            p0 = self._param_list()
            ok = all(p.grad is not None and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous() for p in p0)
        if ok:
            base = mods[1]._param_list()[0].data_ptr() - p0[0].data_ptr()
            ok = base > 0 and base % 16 == 0
            for gi, m in enumerate(mods):
                pm = m._param_list()
                ok = ok and len(pm) == len(p0) and all(
                    q.shape == p.shape and q.grad is not None and q.data_ptr() - p.data_ptr() == gi * base
                    and q.grad.data_ptr() - p.grad.data_ptr() == gi * base for p, q in zip(p0, pm))
            if ok:
                stride = base
        if stride is not None:
            pe0 = self.pos_embedding_layer.pos_embedding
            ok = not isinstance(pe0, nn.Parameter) and not self.lang_pos_embedding
            ok = ok and all(not m.lang_pos_embedding and not isinstance(m.pos_embedding_layer.pos_embedding, nn.Parameter) for m in mods[1:])
            if ok and ragged:
                tables = [m.pos_embedding_layer.pos_embedding for m in mods]
                longest = max(tables, key=lambda t: t.shape[1])
                ok = all(t.dim() == 3 and t.shape[0] == longest.shape[0] and t.shape[2] == longest.shape[2]
                         and (t is longest or torch.equal(t, longest[:, :t.shape[1]])) for t in tables)
            elif ok:
                ok = all(m.pos_embedding_layer.pos_embedding.shape == pe0.shape
                         and (m.pos_embedding_layer.pos_embedding is pe0 or torch.equal(m.pos_embedding_layer.pos_embedding, pe0)) for m in mods[1:])
            if not ok:
                stride = None
        self._group_check = (key, stride)
        return stride

    def forward_grouped(self, mods, x, language_tokens, language_tokens_att_maks, lang_valid_rows=None, group_nv=None):
        """``x`` [G * B, Nv, d]: the visual tokens of the G encoders ``mods`` (``mods[0] is self``), group-major; ``language_tokens`` /
        mask [G * B, Nl, ...] likewise (the wrapper repeats the shared narration tokens).  One runtime call for all G encoders, each
        row range against its own parameters (TfEncoderDesc.groups).  ``lang_valid_rows`` counts the un-masked tokens of ALL groups.
        Returns what ``forward`` returns, for the stacked batch.  Check ``group_stride(mods)`` first.
        ``group_nv`` (RAGGED groups, TfEncoderDesc.group_nv): the encoders differ in their visual token count -- ``group_nv[g]`` per sample
        of group g, the reference's real FPN geometry (28 x 28 tokens on level 0, 14 x 14 on levels 1 - 3).  ``x`` is then the
        CONCATENATION [sum_g B * group_nv[g], d] of the groups' tokens and so is the first return value; needs ``lang_valid_rows`` and a
        mask (ragged groups exist on packed rows); check ``group_stride(mods, ragged=True)`` first."""
        ragged = group_nv is not None and len(set(group_nv)) > 1
        stride = self.group_stride(mods, ragged=ragged)
        if stride is None:
            raise L.TfError("these encoders cannot run as one grouped call (CrossTransformerModuleBox.group_stride)")
        if ragged and (lang_valid_rows is None or language_tokens_att_maks is None or not self.pack_tokens):
            raise L.TfError("ragged groups run on packed token rows: lang_valid_rows, a padding mask and pack_tokens are required")
        if group_nv is not None and not ragged:                      # equal counts after all: the plain grouped call on [G B, Nv, d]
            x = x.reshape(language_tokens.shape[0], group_nv[0], x.shape[-1])
        self._group_mods, self._group_stride = list(mods), stride
        self._group_nv = [int(n) for n in group_nv] if ragged else None
        try:
            return self.forward(x, language_tokens, language_tokens_att_maks, None, lang_valid_rows)
        finally:
            self._group_mods = None
            self._group_nv = None

    def _pack_block_bits(self, vis_tokens_mask, Nv, Nl, device):
        """vis_tokens_mask [Nv,Nv] (nonzero = blocked, reference utils.py:14-30) -> the [S, ceil(S/64)] u64 block-bit matrix of
        the joint sequence: the reference pads it with zeros for the language rows / columns and casts to bool
        (cross_f_box_layers.py:87-95), i.e. only visual-visual pairs can be blocked.  Cached per mask tensor."""
        if tuple(vis_tokens_mask.shape) != (Nv, Nv):
            raise RuntimeError(f"vis_tokens_mask must be [Nv, Nv] = {(Nv, Nv)}, got {tuple(vis_tokens_mask.shape)}")
        key = (vis_tokens_mask._version, Nv, Nl, str(device))
        cache = getattr(self, "_block_bits_cache", None)
        if cache is not None and cache[0] is vis_tokens_mask and cache[1] == key:     # identity, not data_ptr (see the padding-mask cache)
            return cache[2]
        S = Nv + Nl
        SW = (S + 63) // 64
        full = torch.zeros(S, SW * 64, dtype=torch.bool)
        full[:Nv, :Nv] = vis_tokens_mask.detach().to("cpu") != 0
        words = (full.view(S, SW, 64).to(torch.int64) << torch.arange(64, dtype=torch.int64)).sum(-1)   # wraps mod 2^64: bit 63 is the sign
        bits = words.contiguous().to(device)
        self._block_bits_cache = (vis_tokens_mask, key, bits)
        return bits

    def packed_row_error(self, desc=None):
        """Test / debug hook: 0 when the ``lang_valid_rows`` of every forward run on this descriptor's workspace (default: the last
        forward's) agreed with its mask, otherwise the row total the mask gave (host sync)."""
        desc = self._last_desc if desc is None else desc
        out = (C.c_int * 1)(0)
        L.check(L.load().tf_encoder_packed_error(C.byref(desc), out, C.c_void_p(ops._stream())), "tf_encoder_packed_error")
        torch.cuda.current_stream().synchronize()
        return int(out[0])

    def peek(self, desc_keep, name):
        """Test hook: copy an internal activation of the last forward out of the workspace (fp32)."""
        desc, keep = desc_keep
        lib = L.load()
        plan = L.TfEncoderPlan()
        L.check(lib.tf_encoder_plan_ex(C.addressof(desc), C.addressof(plan)), "plan")
        cap = plan.M * max(plan.ldq, plan.ffp, plan.dp)
        buf = torch.empty(cap, dtype=torch.float32, device=keep["work"].device)
        n = lib.tf_encoder_peek(C.byref(desc), name.encode(), buf.data_ptr(), cap, ops._stream())
        if n < 0:
            L.check(int(n), "tf_encoder_peek")
        return buf[:n].view(plan.M, -1)
