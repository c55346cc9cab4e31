#!/usr/bin/env python3
"""Runs individual hot kernels at the benchmark shape (for rocprofv3 --pmc / --kernel-trace passes and A/B timing).
usage: python tools/kernel_bench.py [attn|wgrad|gemm|ln|all] [reps]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transfusion_amd import _lib as Lb, ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, NV, NL, D, H = 32, 196, 512, 768, 4
S, M, hd, ff = NV + NL, B * (NV + NL), D // H, 2 * D
dev = torch.device("cuda:0")
bf = torch.bfloat16
g = torch.Generator().manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g).to(device=dev, dtype=bf)
st = ops._stream()


def traced(fn, reps=5):
    """per-kernel mean duration of `fn`'s launches (the library's launch tracer: HIP event pairs on the launching stream)"""
    import collections
    import ctypes
    lib = Lb.load()
    fn()
    torch.cuda.synchronize()
    Lb.check(lib.tf_trace_start(), "tf_trace_start")
    for _ in range(reps):
        fn()
    cap = 4096
    recs = (Lb.TfTraceRecord * cap)()
    n = lib.tf_trace_stop(ctypes.addressof(recs), cap)
    acc, cnt = collections.OrderedDict(), collections.Counter()
    for r in recs[:min(int(n), cap)]:
        k = r.name.decode()
        acc[k] = acc.get(k, 0.0) + r.us
        cnt[k] += 1
    for k, v in acc.items():
        print(f"    {k:44s} {v / cnt[k]:8.1f} us x {cnt[k] / reps:.0f}", flush=True)


def timeit(name, fn, flops):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:28s} {us:9.1f} us  {flops / us / 1e6:8.1f} TF/s", flush=True)


if which in ("attn", "all"):
    QKV, Y = rnd(M, 3 * D), rnd(M, D)
    O = torch.empty(M, D, device=dev, dtype=bf)
    lse, delta = torch.empty(B * H * S, device=dev), torch.empty(B * H * S, device=dev)
    dQKV = torch.empty(M, 3 * D, device=dev, dtype=bf)
    km = torch.zeros(B, S, dtype=torch.uint8, device=dev)
    km[:, S - 100:] = 1
    dsw = torch.empty(Lb.load().tf_attn_ds_bytes(B, H, S), dtype=torch.uint8, device=dev)
    # the benchmark's ragged batch: 196 visual tokens + U{128..512} language tokens per sample, packed rows, longest sample first
    lens = sorted((NV + torch.randint(128, 513, (B,), generator=g)).tolist(), reverse=True)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    pairs = sum(l * l for l in lens)
    for p, packed in ((0.15, False), (0.15, True), (0.0, True)):
        drop = ops.drop_params(p, 1, 1)
        dbits = ops.attn_dropmask(B, H, S, p, 1, 1, dev) if p > 0 else None
        att = Lb.TfAttnArgs(qkv=Lb.ptr(QKV), ld_qkv=3 * D, out=Lb.ptr(O), ld_out=D, lse=Lb.ptr(lse), key_mask=0 if packed else Lb.ptr(km), B=B, S=S, H=H, HDP=hd,
                            scale=1 / math.sqrt(hd), drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2], drop_bits=Lb.ptr(dbits),
                            dout=Lb.ptr(Y), ld_dout=D, dqkv=Lb.ptr(dQKV), ld_dqkv=3 * D, delta=Lb.ptr(delta), ds_work=Lb.ptr(dsw),
                            cu_rows=Lb.ptr(cu) if packed else 0)
        fl = 4.0 * (pairs * H * hd if packed else B * S * S * D)
        tag = f"p={p} {'packed' if packed else 'dense'}"
        timeit(f"attn_fwd {tag}", lambda: Lb.call("tf_attn_fwd", att, st), fl)
        timeit(f"attn_bwd {tag}", lambda: Lb.call("tf_attn_bwd", att, st), 2.0 * fl)
        traced(lambda: Lb.call("tf_attn_bwd", att, st))
if which in ("wgrad", "all"):
    X, dYq, dYf = rnd(M, D), rnd(M, 3 * D), rnd(M, ff)
    dW, db = torch.zeros(3 * D, D, device=dev), torch.zeros(3 * D, device=dev)
    timeit("wgrad_qkv", lambda: ops.wgrad(dYq, 3 * D, X, D, dW, db), 2.0 * M * 3 * D * D)
    timeit("wgrad_ffn", lambda: ops.wgrad(dYf, ff, X, D, dW[:ff], db[:ff]), 2.0 * M * ff * D)
    timeit("wgrad_d_d", lambda: ops.wgrad(X, D, X, D, dW[:D], db[:D]), 2.0 * M * D * D)
if which in ("wgradm",):
    # a layer's four weight gradients: four launches of the per-problem kernel (stand-alone and overlapped sizing) against ONE merged
    # launch at several workgroup counts; packed row count of the benchmark
    Mp = int(os.environ.get("KB_ROWS", "16672"))
    x, o, x1, hh = rnd(Mp, D), rnd(Mp, D), rnd(Mp, D), rnd(Mp, ff)
    dqkv, dy1, dy2, du = rnd(Mp, 3 * D), rnd(Mp, D), rnd(Mp, D), rnd(Mp, ff)
    pairs = [(dqkv, x), (dy1, o), (du, x1), (dy2, hh)]
    dWs = [torch.zeros(a.shape[1], b.shape[1], device=dev) for a, b in pairs]
    dbs = [torch.zeros(a.shape[1], device=dev) for a, _ in pairs]
    fl = sum(2.0 * Mp * a.shape[1] * b.shape[1] for a, b in pairs)

    def four(chunk):
        for (a, b), w, v in zip(pairs, dWs, dbs):
            ops.wgrad(a, a.shape[1], b, b.shape[1], w, v, m_chunk=chunk(a.shape[1], b.shape[1]))
    timeit("4 launches, self-sized", lambda: four(lambda n, k: 0), fl)

    def ovl_chunk(n, k):          # the encoder runtime's overlapped sizing (tf_api.hip: wgrad())
        tiles = ((n + 255) // 256) * ((k + 127) // 128)
        steps = (Mp + 31) // 32
        splits = max(1, min(steps, (256 + tiles // 2) // tiles))
        return ((steps + splits - 1) // splits) * 32
    timeit("4 launches, overlap-sized", lambda: four(ovl_chunk), fl)
    probs = [ops.wgrad_args(a, a.shape[1], b, b.shape[1], w, v) for (a, b), w, v in zip(pairs, dWs, dbs)]
    for blocks in [int(t) for t in os.environ.get("KB_BLOCKS", "144,256,288,432,512,576").split(",")]:
        timeit(f"merged, blocks={blocks}", lambda: ops.wgrad_multi(probs, blocks), fl)
    half = [probs[:2], probs[2:]]
    fl2 = [sum(2.0 * Mp * a.shape[1] * b.shape[1] for a, b in pairs[:2]), sum(2.0 * Mp * a.shape[1] * b.shape[1] for a, b in pairs[2:])]
    for blocks in (144, 288, 432):
        timeit(f"merged {{in,out}} blocks={blocks}", lambda: ops.wgrad_multi(half[0], blocks), fl2[0])
        timeit(f"merged {{w1,w2}} blocks={blocks}", lambda: ops.wgrad_multi(half[1], blocks), fl2[1])
if which in ("wgradp",):
    # every product of a layer's weight-gradient launch ALONE, at the chunking the merged launch gives it (three row chunks), then the merged
    # launch: the dispatch order tools/wgrad_product_pmc.sh reads its per-product FETCH_SIZE / WRITE_SIZE from
    Mp = int(os.environ.get("KB_ROWS", "16672"))
    x, o, x1, hh = rnd(Mp, D), rnd(Mp, D), rnd(Mp, D), rnd(Mp, ff)
    dqkv, dy1, dy2, du = rnd(Mp, 3 * D), rnd(Mp, D), rnd(Mp, D), rnd(Mp, ff)
    pairs = [("in_proj  dW[2304, 768]", dqkv, x), ("out_proj dW[768, 768]", dy1, o), ("linear1  dW[1536, 768]", du, x1), ("linear2  dW[768, 1536]", dy2, hh)]
    for nm, a, b in pairs + [("merged layer", None, None)]:
        if a is not None:
            w, v = torch.zeros(a.shape[1], b.shape[1], device=dev), torch.zeros(a.shape[1], device=dev)
            probs = [ops.wgrad_args(a, a.shape[1], b, b.shape[1], w, v)]
            tiles = ((a.shape[1] + 255) // 256) * ((b.shape[1] + 127) // 128)
            fl, by = 2.0 * Mp * a.shape[1] * b.shape[1], (Mp * a.shape[1] + Mp * b.shape[1]) * 2.0 + a.shape[1] * b.shape[1] * 4.0
        else:
            ws = [torch.zeros(a.shape[1], b.shape[1], device=dev) for _, a, b in pairs]
            vs = [torch.zeros(a.shape[1], device=dev) for _, a, _ in pairs]
            probs = [ops.wgrad_args(a, a.shape[1], b, b.shape[1], w, v) for (_, a, b), w, v in zip(pairs, ws, vs)]
            tiles = 144
            fl = sum(2.0 * Mp * a.shape[1] * b.shape[1] for _, a, b in pairs)
            by = sum((Mp * a.shape[1] + Mp * b.shape[1]) * 2.0 + a.shape[1] * b.shape[1] * 4.0 for _, a, b in pairs)
        print(f"# {nm}: {tiles} tiles x 3 chunks, algorithmic {by / 1e6:.1f} MB", flush=True)
        timeit(nm, lambda: ops.wgrad_multi(probs, 3 * tiles), fl)
if which in ("gemm", "all"):
    X, Xf = rnd(M, D), rnd(M, ff)
    Wq, Wo, W1, W2 = rnd(3 * D, D) * 0.03, rnd(D, D) * 0.03, rnd(ff, D) * 0.03, rnd(D, ff) * 0.03
    QKV, O, U, Hh, Y = torch.empty(M, 3 * D, device=dev, dtype=bf), torch.empty(M, D, device=dev, dtype=bf), torch.empty(M, ff, device=dev, dtype=bf), torch.empty(M, ff, device=dev, dtype=bf), rnd(M, D)
    b3, b1, bff = torch.zeros(3 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(ff, device=dev)
    drop = ops.drop_params(0.15, 1, 2)
    timeit("gemm_qkv (bias)", lambda: ops.gemm(X, Wq, QKV, 3 * D, D, Lb.TF_EPI_BIAS, bias=b3), 2.0 * M * 3 * D * D)
    timeit("gemm_ffn_up (gelu+drop)", lambda: ops.gemm(X, W1, U, ff, D, Lb.TF_EPI_BIAS_GELU_DROP, bias=bff, C2=Hh, drop=drop), 2.0 * M * ff * D)
    timeit("gemm_ffn_down (drop+res)", lambda: ops.gemm(Xf, W2, O, D, ff, Lb.TF_EPI_BIAS_DROP_RES, bias=b1, R=Y, drop=drop), 2.0 * M * ff * D)
    timeit("gemm_outproj (drop+res)", lambda: ops.gemm(X, Wo, O, D, D, Lb.TF_EPI_BIAS_DROP_RES, bias=b1, R=Y, drop=drop), 2.0 * M * D * D)
    timeit("gemm_d_d (none)", lambda: ops.gemm(X, Wo, O, D, D, Lb.TF_EPI_NONE), 2.0 * M * D * D)
if which in ("epi",):
    X = rnd(M, D)
    W1 = rnd(ff, D) * 0.03
    U, Hh = torch.empty(M, ff, device=dev, dtype=bf), torch.empty(M, ff, device=dev, dtype=bf)
    bff = torch.zeros(ff, device=dev)
    Y = rnd(M, ff)
    d15, d0 = ops.drop_params(0.15, 1, 2), (0, 0, 1.0)
    fl = 2.0 * M * ff * D
    timeit("ffn_up BIAS only", lambda: ops.gemm(X, W1, U, ff, D, Lb.TF_EPI_BIAS, bias=bff), fl)
    timeit("ffn_up NONE", lambda: ops.gemm(X, W1, U, ff, D, Lb.TF_EPI_NONE), fl)
    timeit("ffn_up ADD (reads R)", lambda: ops.gemm(X, W1, U, ff, D, Lb.TF_EPI_ADD, R=Y), fl)
    timeit("ffn_up GELU p=0", lambda: ops.gemm(X, W1, U, ff, D, Lb.TF_EPI_BIAS_GELU_DROP, bias=bff, C2=Hh, drop=d0), fl)
    timeit("ffn_up GELU p=.15", lambda: ops.gemm(X, W1, U, ff, D, Lb.TF_EPI_BIAS_GELU_DROP, bias=bff, C2=Hh, drop=d15), fl)
    timeit("ffn_up DROP_RES p=0", lambda: ops.gemm(X, W1, U, ff, D, Lb.TF_EPI_BIAS_DROP_RES, bias=bff, R=Y, drop=d0), fl)
    timeit("ffn_up DROP_RES p=.15", lambda: ops.gemm(X, W1, U, ff, D, Lb.TF_EPI_BIAS_DROP_RES, bias=bff, R=Y, drop=d15), fl)
    G2 = torch.empty(M, ff, device=dev, dtype=bf)
    timeit("ffn_up GELU_G p=0", lambda: ops.gemm(X, W1, G2, ff, D, Lb.TF_EPI_BIAS_GELU_DROP_G, bias=bff, C2=Hh, drop=d0), fl)
    timeit("ffn_up GELU_G p=.15", lambda: ops.gemm(X, W1, G2, ff, D, Lb.TF_EPI_BIAS_GELU_DROP_G, bias=bff, C2=Hh, drop=d15), fl)
    timeit("ffn_up DGELU p=0", lambda: ops.gemm(X, W1, Hh, ff, D, Lb.TF_EPI_DGELU_DROP, R=U, drop=d0), fl)
    timeit("ffn_up DGELU p=.15", lambda: ops.gemm(X, W1, Hh, ff, D, Lb.TF_EPI_DGELU_DROP, R=U, drop=d15), fl)
if which in ("ln",):
    M = int(os.environ.get("KB_ROWS", M))
    X, DY = rnd(M, D), rnd(M, D)
    DX, DXD = torch.empty(M, D, device=dev, dtype=bf), torch.empty(M, D, device=dev, dtype=bf)
    mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
    gam, bet, dgam, dbet = torch.ones(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    drop = ops.drop_params(0.15, 1, 1)
    ln = Lb.TfLnArgs(x=Lb.ptr(X), ldx=D, y=Lb.ptr(DX), ldy=D, y_is_f32=0, gamma=Lb.ptr(gam), beta=Lb.ptr(bet), mean=Lb.ptr(mean), rstd=Lb.ptr(rstd),
                     rows=M, d=D, rows_per_group=M, x_group_stride=M, y_group_stride=M, eps=1e-5, dy=Lb.ptr(DY), lddy=D, dy_is_f32=0,
                     dx=Lb.ptr(DX), lddx=D, dx_drop=Lb.ptr(DXD), lddxd=D, drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2], drop_ld=D,
                     dgamma=Lb.ptr(dgam), dbeta=Lb.ptr(dbet))
    timeit("ln_fwd  (GB/s as TF col)", lambda: Lb.call("tf_layernorm_fwd", ln, st), 2.0 * 2 * M * D * 1e3)
    timeit("ln_bwd  (GB/s as TF col)", lambda: Lb.call("tf_layernorm_bwd", ln, st), 2.0 * 4 * M * D * 1e3)
    g32 = torch.randn(18912000, device=dev)
    acc = torch.zeros(1, device=dev)
    timeit("sumsq 75.6MB (GB/s)", lambda: Lb.check(Lb.load().tf_sumsq(Lb.ptr(g32), g32.numel(), Lb.ptr(acc), st)), 4.0 * g32.numel() * 1e3)
if which in ("ksweep",):
    # time(K) = fixed (launch + prologue + epilogue) + K * slope  ->  asymptotic MFMA rate of the K-loop and the per-tile fixed cost
    for N in (768, 2304):
        O = torch.empty(M, N, device=dev, dtype=bf)
        for K in (256, 512, 768, 1536, 3072, 6144):
            Xk, Wk = rnd(M, K), rnd(N, K) * 0.03
            timeit(f"gemm N={N} K={K} (none)", lambda: ops.gemm(Xk, Wk, O, N, K, Lb.TF_EPI_NONE), 2.0 * M * N * K)
            if K == 1536:
                b1 = torch.zeros(N, device=dev)
                drop = ops.drop_params(0.15, 1, 2)
                Y = rnd(M, N)
                timeit(f"gemm N={N} K={K} (bias)", lambda: ops.gemm(Xk, Wk, O, N, K, Lb.TF_EPI_BIAS, bias=b1), 2.0 * M * N * K)
                timeit(f"gemm N={N} K={K} (drop+res)", lambda: ops.gemm(Xk, Wk, O, N, K, Lb.TF_EPI_BIAS_DROP_RES, bias=b1, R=Y, drop=drop), 2.0 * M * N * K)
if which in ("small",):
    # the 128x128 kernel at the per-level sizes of the wrapper path / the reference's batch (M = B * 708 for B = 4, 8)
    for Ms in (2832, 5664):
        for N in (768, 1536, 2304):
            for K in (768, 1536, 2304):
                if N != 768 and K != 768:
                    continue
                Xk, Wk = rnd(Ms, K), rnd(N, K) * 0.03
                O = torch.empty(Ms, N, device=dev, dtype=bf)
                timeit(f"gemm M={Ms} N={N} K={K} (none)", lambda: ops.gemm(Xk, Wk, O, N, K, Lb.TF_EPI_NONE), 2.0 * Ms * N * K)
if which in ("pack",):
    # the per-step re-pack of a d = 768 encoder's fp32 weights into bf16 shadows (W and W^T): tf_encoder_pack = one launch per layer
    import ctypes
    e = Lb.TfEncoderDesc()
    e.B, e.Nv, e.Nl, e.d, e.H, e.L, e.ff = 4, NV, NL, D, H, 4, ff
    plan = Lb.TfEncoderPlan()
    Lb.check(Lb.load().tf_encoder_plan_ex(ctypes.addressof(e), ctypes.addressof(plan)), "plan")
    wpack = torch.zeros(plan.wpack_bytes, dtype=torch.uint8, device=dev)
    work = torch.zeros(plan.work_bytes, dtype=torch.uint8, device=dev)
    ps = []
    for j in range(4):
        t = dict(in_w=torch.randn(3 * D, D), in_b=torch.randn(3 * D), out_w=torch.randn(D, D), out_b=torch.randn(D), w1=torch.randn(ff, D), b1=torch.randn(ff),
                 w2=torch.randn(D, ff), b2=torch.randn(D), n1_w=torch.ones(D), n1_b=torch.zeros(D), n2_w=torch.ones(D), n2_b=torch.zeros(D))
        t = {k: v.to(dev) for k, v in t.items()}
        ps.append(t)
        for k, v in t.items():
            setattr(e.p[j], k, v.data_ptr())
    kv = torch.zeros(D, device=dev)
    e.kind_v = e.kind_l = kv.data_ptr()
    e.wpack, e.work = wpack.data_ptr(), work.data_ptr()
    nbytes = 4 * (8 * D * D) * (4 + 2 + 2)
    timeit("tf_encoder_pack, 4 layers (GB/s)", lambda: Lb.check(Lb.load().tf_encoder_pack(ctypes.byref(e), ctypes.c_void_p(st)), "pack"), nbytes * 1e3)
if which in ("attnblk",):
    # vis_mask_type local_k on a 28 x 28 token grid (+ 512 language tokens), head dim 192: forward / backward with and without the
    # block-sparse tile maps (TfAttnArgs.block_skip_q / block_skip_k)
    Bb, Hh, gh = 8, 4, 28
    Nv_, Sx = gh * gh, gh * gh + 512
    Mx = Bb * Sx
    QKV, Y = rnd(Mx, 3 * D), rnd(Mx, D)
    O = torch.empty(Mx, D, device=dev, dtype=bf)
    lse, delta = torch.empty(Bb * Hh * Sx, device=dev), torch.empty(Bb * Hh * Sx, device=dev)
    dQKV = torch.empty(Mx, 3 * D, device=dev, dtype=bf)
    dsw = torch.empty(Lb.load().tf_attn_ds_bytes(Bb, Hh, Sx), dtype=torch.uint8, device=dev)
    for k in (1, 2, 4, 8):
        r, c = torch.arange(Nv_) // gh, torch.arange(Nv_) % gh
        near = ((r[:, None] - r[None, :]).abs() <= k) & ((c[:, None] - c[None, :]).abs() <= k)
        blk = torch.zeros(Sx, Sx, dtype=torch.bool)
        blk[:Nv_, :Nv_] = ~near
        SW = (Sx + 63) // 64
        full = torch.zeros(Sx, SW * 64, dtype=torch.bool)
        full[:, :Sx] = blk
        bits = (full.view(Sx, SW, 64).to(torch.int64) << torch.arange(64, dtype=torch.int64)).sum(-1).contiguous().to(dev)
        nb = (Sx + 127) // 128
        skq, skk = torch.zeros(nb, dtype=torch.int64, device=dev), torch.zeros(nb, dtype=torch.int64, device=dev)
        Lb.check(Lb.load().tf_attn_block_skip(Lb.ptr(bits), Sx, Lb.ptr(skq), Lb.ptr(skk), st), "skip")
        torch.cuda.synchronize()
        nq = sum(bin(int(v) & (2 ** 64 - 1)).count("1") for v in skq.cpu().tolist())
        nk = sum(bin(int(v) & (2 ** 64 - 1)).count("1") for v in skk.cpu().tolist())
        print(f"local_{k}: {nq} of {nb * SW} (128 q x 64 k) tiles and {nk} of {nb * ((Sx + 31) // 32)} (32 q x 128 k) tiles are fully blocked")
        for use in (False, True):
            att = Lb.TfAttnArgs(qkv=Lb.ptr(QKV), ld_qkv=3 * D, out=Lb.ptr(O), ld_out=D, lse=Lb.ptr(lse), key_mask=0, B=Bb, S=Sx, H=Hh, HDP=hd,
                                scale=1 / math.sqrt(hd), drop_thr=0, drop_key=0, drop_scale=1.0, block_bits=Lb.ptr(bits),
                                dout=Lb.ptr(Y), ld_dout=D, dqkv=Lb.ptr(dQKV), ld_dqkv=3 * D, delta=Lb.ptr(delta), ds_work=Lb.ptr(dsw),
                                block_skip_q=Lb.ptr(skq) if use else 0, block_skip_k=Lb.ptr(skk) if use else 0)
            fl = 4.0 * Bb * Sx * Sx * D
            timeit(f"attn_fwd local_{k} skip={use}", lambda: Lb.call("tf_attn_fwd", att, st), fl)
            timeit(f"attn_bwd local_{k} skip={use}", lambda: Lb.call("tf_attn_bwd", att, st), 2.0 * fl)
