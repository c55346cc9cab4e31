"""The per-level pieces either side of the wrapper's grouped encoder call -- K1 (patch embedding: im2col + GEMM) and K9 (back-projection:
dropout + GEMM + fold) of ALL feature levels -- each as ONE autograd node that issues its levels' kernels side by side on the level
streams (reference cross_f_box_wrapper.py:177-212 with :266-274 and utils.py:84-119).

Why: at the reference's per-GPU batch the wrapper's step is bound by the host, and these pieces were most of what was left of its host
time after the encoders became one call -- four autograd nodes, ~10 torch ops and ~8 stream / event calls per level and direction.
Here a level costs its two or three kernel launches.  Numerically these are the same kernels with the same arguments as
``ops.patchify`` / ``ops.linear`` / ``ops.regroup`` (bf16 path); ``tests/test_gpu_grouped.py`` compares the wrapper with and without.
"""
from __future__ import annotations

import os

import torch

from . import _lib as L
from . import ops


_K9_WGRAD_SIDE = os.environ.get("TF_K9_WGRAD_SIDE", "1") != "0"     # A/B switch of the host mirror (see _LevelsK9Fn.backward)


def _on(streams, main, i):
    """Context for level i: its stream (ordered behind everything on ``main`` so far), or main itself."""
    if streams is None:
        return torch.cuda.stream(main)
    st = streams[i]
    st.wait_stream(main)
    return torch.cuda.stream(st)


def _join(streams, main, n):
    if streams is not None:
        for st in streams[:n]:
            main.wait_stream(st)


def _up64(n):
    return (n + 63) // 64 * 64


def k1_supported(mods, feats, d):
    """K a multiple of 64, d of 8 (d = 712, the Ego4D v1 width, costs one padded copy of the token gradients in the backward), bf16
    arithmetic, 4-D feature maps that tile into patches."""
    if d % 8:
        return False
    for m, f in zip(mods, feats):
        if getattr(m, "precision", "bf16") != "bf16" or f.dim() != 4 or not f.is_cuda:
            return False
        if (f.shape[1] * m.patch_h * m.patch_w) % 64 or m.weight.shape[0] != d:
            return False
    return True


def k9_supported(mods, d):
    if d % 8:              # (d % 64 != 0: the tokens are copied into rows padded to 64 in front of the GEMM, as ops.linear does)
        return False
    for m in mods:
        n = m.linear.weight.shape[0]
        if getattr(m, "precision", "bf16") != "bf16" or n % 64 or m.linear.weight.shape[1] != d or m.linear.bias is None:
            return False
    return True


class _LevelsK1Fn(torch.autograd.Function):
    """feats[g] [B, C_g, H_g, W_g] -> tokens [G * B, Nv, d] bf16, group-major (what the grouped encoder call takes): per level an im2col
    gather and a GEMM that writes straight into its slice of the stacked output.  Levels of unequal token counts: [sum_g B nv_g, d]."""

    @staticmethod
    def forward(ctx, cfg, *tensors):
        mods, streams, accumulate = cfg
        G = len(mods)
        feats, weights = tensors[:G], tensors[G:]
        main = torch.cuda.current_stream(feats[0].device)
        B = feats[0].shape[0]
        d = weights[0].shape[0]
        # tokens per level; levels of unequal token counts (RAGGED groups, TfEncoderDesc.group_nv) give the concatenation [sum_g B nv_g, d]
        nvs = [(f.shape[2] // m.patch_h) * (f.shape[3] // m.patch_w) for m, f in zip(mods, feats)]
        offs = [0]
        for n in nvs:
            offs.append(offs[-1] + B * n)
        out = torch.empty(offs[-1], d, dtype=torch.bfloat16, device=feats[0].device)
        saved, meta = [], []
        for g, (m, f, w) in enumerate(zip(mods, feats, weights)):
            Bc, Cc, H, W = f.shape
            K = Cc * m.patch_h * m.patch_w
            wsh, wsh_t = ops._weight_shadows(w, d, K, _up64(d))
            with _on(streams, main, g):
                f = f.contiguous()
                rows = torch.empty(Bc * nvs[g], K, dtype=torch.bfloat16, device=f.device)
                L.call("tf_patchify_fwd", ops._patch_args(f, rows, Bc, Cc, H, W, m.patch_h, m.patch_w), ops._stream())
                ops.gemm(rows, wsh, out[offs[g]:offs[g + 1]], d, K, L.TF_EPI_NONE)
            saved += [rows, wsh_t]
            meta.append((tuple(f.shape), f.dtype, m.patch_h, m.patch_w, K))
        _join(streams, main, G)
        ctx.save_for_backward(*saved)
        ctx.cfg = (mods, streams, accumulate, meta, B, offs, d)
        ctx.into = [(w.grad if (accumulate and w.grad is not None and w.grad.is_contiguous() and w.grad.dtype == torch.float32) else None)
                    for w in weights]
        return out.view(G * B, nvs[0], d) if len(set(nvs)) == 1 else out

    @staticmethod
    def backward(ctx, gy):
        mods, streams, accumulate, meta, B, offs, d = ctx.cfg
        G = len(mods)
        saved = ctx.saved_tensors
        gy = gy.reshape(offs[-1], d)
        if gy.dtype != torch.bfloat16 or not gy.is_contiguous():
            gy = gy.contiguous().to(torch.bfloat16)
        main = torch.cuda.current_stream(gy.device)
        dfeats, dws = [], []
        # The levels' weight gradients as ONE launch (tf_gemm_wgrad_multi), on the main stream, FIRST: their operands -- the upstream
        # gradient and the im2col rows saved by the forward -- exist when this node starts, and the launch runs beside the levels' dgrad
        # GEMMs and folds (with level streams it is issued behind their fork, below).  (Round 5: one tf_gemm_wgrad per level -- 8 launches of 25 - 98 row steps per wrapper step for K1 + K9, each
        # ending in its own atomic flush: 556 us of kernel time at 133 TFLOP/s.  Together the levels' 432 output tiles fill the chip at
        # ONE row chunk each.)
        probs = []
        for g, m in enumerate(mods):
            rows = saved[2 * g]
            K = meta[g][4]
            gw = ctx.into[g]
            if gw is None:
                gw = torch.zeros(d, K, dtype=torch.float32, device=gy.device)
                dws.append(gw.view(m.weight.shape))
            else:
                dws.append(None)
            probs.append(ops.wgrad_args(gy[offs[g]:offs[g + 1]], d, rows, K, gw.view(d, K), None))
        if streams is None:
            ops.wgrad_multi(probs, 0)
        for g, m in enumerate(mods):
            rows, wsh_t = saved[2 * g], saved[2 * g + 1]
            shape, dtype, ph, pw, K = meta[g]
            gyg = gy[offs[g]:offs[g + 1]]
            with _on(streams, main, g):
                if ctx.needs_input_grad[1 + g]:
                    dx = torch.empty(offs[g + 1] - offs[g], K, dtype=torch.bfloat16, device=gy.device)
                    ops.gemm(ops.to_bf16_padded(gyg, _up64(d)), wsh_t, dx, K, _up64(d), L.TF_EPI_NONE)   # (a copy only when d % 64 != 0)
                    df = torch.empty(shape, dtype=dtype, device=gy.device)
                    L.call("tf_patchify_bwd", ops._patch_args(df, dx, shape[0], shape[1], shape[2], shape[3], ph, pw), ops._stream(), ops._is_f32(df))
                    if streams is not None:
                        df.record_stream(main)
                    dfeats.append(df)
                else:
                    dfeats.append(None)
        if streams is not None:
            ops.wgrad_multi(probs, 0)          # (after the level streams were forked: they wait for what main held THEN, not for this launch)
        _join(streams, main, G)
        return (None,) + tuple(dfeats) + tuple(dws)


class _LevelsK9Fn(torch.autograd.Function):
    """fused tokens [G * B, Nv, d] -> per level the feature map [B, C_g, H_g, W_g] fp32: (input dropout) -> GEMM + bias -> fold."""

    @staticmethod
    def forward(ctx, cfg, fused, *params):
        mods, streams, accumulate, training = cfg
        G = len(mods)
        weights, biases = params[:G], params[G:]
        dev = fused.device
        main = torch.cuda.current_stream(dev)
        d = fused.shape[-1]
        # tokens per level from the maps to rebuild; fused is [G * B, Nv, d] or (levels of unequal token counts) [sum_g B nv_g, d]
        nvs = [(m.init_h // m.patch_h) * (m.init_w // m.patch_w) for m in mods]
        x = fused.reshape(-1, d)
        if x.shape[0] % sum(nvs):
            raise RuntimeError(f"regroup_patches: {x.shape[0]} tokens do not tile the levels' maps ({nvs} tokens per sample)")
        B = x.shape[0] // sum(nvs)
        offs = [0]
        for n in nvs:
            offs.append(offs[-1] + B * n)
        if x.dtype != torch.bfloat16 or not x.is_contiguous():
            x = x.contiguous().to(torch.bfloat16)
        outs, saved, meta = [], [], []
        for g, (m, w, b) in enumerate(zip(mods, weights, biases)):
            N = w.shape[0]
            Cc = N // (m.patch_h * m.patch_w)
            H, W = m.init_h, m.init_w
            Nv = nvs[g]
            if fused.dim() == 3 and fused.shape[1] != Nv:
                raise RuntimeError(f"regroup_patches: {fused.shape[1]} tokens do not tile a {H}x{W} map with {m.patch_h}x{m.patch_w} patches")
            dp = _up64(d)
            wsh, wsh_t = ops._weight_shadows(w, N, dp, N)
            p = float(m.backproj_dropout) if training else 0.0
            drop = ops.drop_params(p, ops.next_seed() if p > 0 else 0, 7)
            xg = x[offs[g]:offs[g + 1]]
            with _on(streams, main, g):
                if drop[0]:
                    xd = torch.empty_like(xg)
                    L.check(L.load().tf_dropout_apply(L.ptr(xg), L.ptr(xd), xg.numel(), drop[1], drop[0], drop[2], ops._stream()), "tf_dropout_apply")
                else:
                    xd = xg
                xd = ops.to_bf16_padded(xd, dp)                    # (a copy only when d % 64 != 0; the dropout mask is indexed on the unpadded rows)
                y = torch.empty(B * Nv, N, dtype=torch.bfloat16, device=dev)
                ops.gemm(xd, wsh, y, N, dp, L.TF_EPI_BIAS, bias=b)
                o = torch.empty(B, Cc, H, W, dtype=torch.float32, device=dev)
                L.call("tf_regroup_fwd", ops._patch_args(o, y, B, Cc, H, W, m.patch_h, m.patch_w), ops._stream(), 1)
                if streams is not None:
                    o.record_stream(main)
                    xd.record_stream(main)
            outs.append(o)
            saved += [xd, wsh_t]
            meta.append((N, Cc, H, W, m.patch_h, m.patch_w, drop))
        _join(streams, main, G)
        ctx.save_for_backward(*saved)
        ctx.cfg = (mods, streams, meta, B, offs, d, fused.dtype, tuple(fused.shape))
        ok = lambda t: t is not None and t.grad is not None and t.grad.is_contiguous() and t.grad.dtype == torch.float32
        ctx.into = [((w.grad, b.grad) if (accumulate and ok(w) and ok(b)) else None) for w, b in zip(weights, biases)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        mods, streams, meta, B, offs, d, xdtype, xshape = ctx.cfg
        G = len(mods)
        saved = ctx.saved_tensors
        dev = saved[0].device
        main = torch.cuda.current_stream(dev)
        dx = torch.empty(offs[-1], d, dtype=torch.bfloat16, device=dev) if ctx.needs_input_grad[1] else None
        dws, dbs, all_drows = [], [], []
        for g, m in enumerate(mods):
            xd, wsh_t = saved[2 * g], saved[2 * g + 1]
            N, Cc, H, W, ph, pw, drop = meta[g]
            go = gouts[g]
            with _on(streams, main, g):
                if go is None:
                    go = torch.zeros(B, Cc, H, W, dtype=torch.float32, device=dev)
                go = go.contiguous()
                drows = torch.empty(offs[g + 1] - offs[g], N, dtype=torch.bfloat16, device=dev)
                L.call("tf_regroup_bwd", ops._patch_args(go, drows, B, Cc, H, W, ph, pw), ops._stream())
                if streams is not None:
                    drows.record_stream(main)                              # read by the merged weight-gradient launch on the main stream
                all_drows.append(drows)
                if dx is not None:
                    dxg = dx[offs[g]:offs[g + 1]]
                    ops.gemm(drows, wsh_t, dxg, d, N, L.TF_EPI_NONE)
                    if drop[0]:
                        L.check(L.load().tf_dropout_apply(L.ptr(dxg), L.ptr(dxg), dxg.numel(), drop[1], drop[0], drop[2], ops._stream()), "tf_dropout_apply")
        _join(streams, main, G)
        # the levels' weight and bias gradients as ONE launch on the main stream, behind the gathers that produce their dY operands
        # (see _LevelsK1Fn.backward)
        probs = []
        for g, m in enumerate(mods):
            xd = saved[2 * g]
            N = meta[g][0]
            into = ctx.into[g]
            if into is None:
                gw = torch.zeros(N, d, dtype=torch.float32, device=dev)
                gb = torch.zeros(N, dtype=torch.float32, device=dev)
                dws.append(gw)
                dbs.append(gb)
            else:
                gw, gb = into
                dws.append(None)
                dbs.append(None)
            probs.append(ops.wgrad_args(all_drows[g], N, xd, xd.shape[1], gw.view(N, d), gb))      # (xd is padded to 64 columns: k_src = d masks the pad)
        # Under direct accumulation (FusionTrainStep: nothing is handed back to autograd) the launch goes to the encoders' weight-gradient
        # side stream: this node is the FIRST of the backward, so on the main stream the launch (~100 us) would stand in front of the whole
        # encoder backward; on the side stream it runs beside its first kernels, and the optimiser waits for it (ops.note_grad_writer).
        side = ops.side_stream(dev) if (all(i is not None for i in ctx.into) and _K9_WGRAD_SIDE) else None
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.wgrad_multi(probs, -1)
                ops.note_grad_writer(dev)
            for g in range(G):
                all_drows[g].record_stream(side)                          # (freed when this node returns: the allocator must know the reader)
                saved[2 * g].record_stream(side)
        else:
            ops.wgrad_multi(probs, 0)
        gx = None
        if dx is not None:
            gx = dx.view(xshape)
            if xdtype != torch.bfloat16:
                gx = gx.to(xdtype)
        return (None, gx) + tuple(dws) + tuple(dbs)


def levels_patch_embed(mods, feats, streams=None):
    """``mods``: the PatchToToken modules of the levels.  -> [G * B, Nv, d] bf16."""
    acc = all(getattr(m, "accumulate_linear_grad", False) for m in mods) and torch.is_grad_enabled()
    return _LevelsK1Fn.apply((list(mods), streams, acc), *feats, *[m.weight for m in mods])


def levels_back_project(mods, fused, streams=None):
    """``mods``: the RegroupPatchesLayerBox modules (``init_h`` / ``init_w`` set).  -> list of [B, C_g, H_g, W_g] fp32."""
    acc = all(getattr(m, "accumulate_linear_grad", False) for m in mods) and torch.is_grad_enabled()
    training = bool(mods[0].training)
    return list(_LevelsK9Fn.apply((list(mods), streams, acc, training), fused, *[m.linear.weight for m in mods], *[m.linear.bias for m in mods]))
