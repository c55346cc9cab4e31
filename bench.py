#!/usr/bin/env python3
"""Headline benchmark: train samples/s of the TransFusion fusion block on synthetic Ego4D-shaped batches.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

step  = forward + backward + gradient all-reduce (RCCL, N > 1) + global-norm clip + fused RAdam step of ONE
        4-layer fusion encoder (CrossTransformerModuleBox: d=768, 4 heads, ff=1536, GELU, post-LN, final LN,
        token dropout 0.15, patch dropout 0.1 -- cross_fusion_config_sym_ego_res50.yml) in training mode on a
        batch of B=32 samples PER GPU, each 14x14=196 visual + 512 language tokens (right-padded to a random valid
        length in [128, 512]), bf16 compute / fp32 master weights and statistics.  Weak scaling: B is per GPU.
value = N * B / max-over-ranks step time.

Also reported on the same JSON line: ``roofline`` for the dominant kernel (HIP-event timing of that kernel on the
stream it runs on; algorithmic FLOPs from SURVEY.md 8(d)), a per-kernel table, and ``cpu_baseline`` (the CPU oracle
-- a port of the reference arithmetic, oracle/fusion_oracle.py -- timed on this host's cores, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, H, L, FF_MULT, NV, NL = 768, 4, 4, 2, 196, 512
P_TOKEN, P_PATCH = 0.15, 0.1
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA, MI355X_MICROARCH.md (chip-level parameters)
PEAK_HBM_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_encoder(device, d=D, h=H, layers=L, p_tok=P_TOKEN, p_patch=P_PATCH):
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    torch.manual_seed(42)                       # run.seed: 42 -- identical weights on every rank
    pe = PositionalEmbeddingLayer("sin1d", 8192, d)
    enc = CrossTransformerModuleBox(no_patches=8192, pos_embedding_layer=pe, lang_pos_embedding=None, num_layers=layers,
                                    patch_dropout=p_patch, num_heads=h, fforward_multiplier=FF_MULT, token_dropout=p_tok,
                                    back_to_img_fn="regroup", activ_f="gelu", final_norm="ln", input_f_size=d)
    return enc.to(device)


def make_batch(B, device, rank, d=D, nv=NV, nl=NL, variant=0, padded=True):
    g = torch.Generator().manual_seed(42 + 1000 * rank + 77 * variant)
    x = torch.randn(B, nv, d, generator=g)
    lang = torch.nn.functional.normalize(torch.randn(B, nl, d, generator=g), dim=-1)     # SBERT normalize: True
    lens = torch.randint(nl // 4, nl + 1, (B,), generator=g)
    if not padded:
        lens = torch.full((B,), nl)                     # the no-padding variant
    pad = torch.arange(nl).view(1, -1) >= lens.view(-1, 1)                                # True = ignore
    valid = (~pad).float().contiguous()                                                   # [B, nl] row weights of the synthetic loss (0 / 1)
    km = 1.0 / (float(valid.sum()) * d)                                                   # ... and its normaliser (a host float)
    return x.to(device), lang.to(device), pad.to(device), valid.to(device), km, lens.tolist()


def _masked_square_loss(vis, lo, valid, km):
    """mean(vis^2) + mean(lang[valid]^2) through the library's own loss kernels (tf_sq_loss_fwd / _bwd: two launches
    forward, two backward, deterministic sums): the timed region holds no framework elementwise kernel and no vendor-library kernel,
    the harness's loss included (round 5: ~1.5 % of the step's kernel time were at::native mul / fill kernels of this function)."""
    from transfusion_amd import ops
    vis = vis if (vis.dtype == torch.float32 and vis.is_contiguous()) else vis.float().contiguous()
    lo = lo if (lo.dtype == torch.float32 and lo.is_contiguous()) else lo.float().contiguous()
    return ops.sq_loss([(vis, None, 1.0 / vis.numel()), (lo, valid, km)])


ZERO_IN_OPT = os.environ.get("TF_ZERO_IN_OPT", "0") == "1"      # A/B switch: the gradient zero fill rides in the optimiser pass (measured neutral)
PACK_TOKENS = True      # --dense-rows turns it off: the masked language tokens then travel through every kernel as dead rows


def _valid_rows(batch):
    """Host-side count of un-masked language tokens of a batch (the lengths that built the mask): with it the encoder drops the masked
    tokens from the computation (CrossTransformerModuleBox.pack_tokens)."""
    return sum(batch[5]) if PACK_TOKENS else None


def loss_fn(module, batch):
    """mean(vis^2) + mean(lang[valid]^2)."""
    x, lang, pad, valid, km = batch[:5]
    vis, lo, _, _ = module(x, lang, pad, lang_valid_rows=_valid_rows(batch))
    return _masked_square_loss(vis, lo, valid, km)


def _square_mean(f):
    """mean(f^2) of a feature map (the wrapper legs' loss), on the library's loss kernels when the map is fp32."""
    from transfusion_amd import ops
    if f.dtype == torch.float32 and f.is_contiguous() and f.numel() % 4 == 0 and f.data_ptr() % 16 == 0:
        return ops.sq_loss([(f, None, 1.0 / f.numel())])
    return f.reshape(-1).float().pow(2).mean()


class _EncoderWithHeads(torch.nn.Module):
    """bench --with-heads: the fusion encoder plus the RoI heads that consume the detector's box features (synthetic here: the
    detector between them is out of scope), one parameter set for the trainer."""

    def __init__(self, enc, heads, crit):
        super().__init__()
        self.enc, self.heads, self.crit = enc, heads, crit

    def forward(self, x, lang, pad, lang_valid_rows=None):
        return self.enc(x, lang, pad, lang_valid_rows=lang_valid_rows)


def make_heads_batch(B, device, rank, variant, rois_per_image=512, repr_size=1024, nouns=88, verbs=75):
    g = torch.Generator().manual_seed(4242 + 1000 * rank + 77 * variant)
    R = B * rois_per_image
    feats = torch.randn(R, repr_size, generator=g).to(device=device, dtype=torch.bfloat16)
    noun = torch.randint(1, nouns, (R,), generator=g)
    noun[torch.rand(R, generator=g) < 0.75] = 0                      # RoI sampling keeps 25 % positives (torchvision's positive_fraction)
    verb = torch.randint(0, verbs - 1, (R,), generator=g)
    verb[noun == 0] = 999
    ttc = torch.rand(R, generator=g) * 2
    ttc[noun == 0] = 999.0
    reg = torch.randn(R, 4, generator=g) * 0.1
    return feats, noun.to(device), verb.to(device), ttc.to(device), reg.to(device)


def loss_fn_heads(module, batch):
    x, lang, pad, valid, km = batch[:5]
    vis, lo, _, _ = module(x, lang, pad, lang_valid_rows=_valid_rows(batch))
    loss = _masked_square_loss(vis, lo, valid, km)
    feats, noun, verb, ttc, reg = batch[6]
    out = module.heads(feats)
    l = module.crit(out, noun, verb, ttc, reg)
    return loss + l["bbox_loss"] + l["noun_loss"] + l["verb_loss"] + l["ttc_loss"]


def flops_per_sample_layer(S, d):
    return 16 * S * d * d + 4 * S * S * d       # forward, SURVEY.md 8(d)


# ----------------------------------------------------------------------------------------------------------
def kernel_census(B, device, reps=20):
    """Times every distinct kernel launch of one training step in isolation (HIP events on the launching stream)."""
    from transfusion_amd import _lib as Lb, ops
    S, M, d, ff, hd = NV + NL, B * (NV + NL), D, D * FF_MULT, D // H
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=g).to(device=device, dtype=bf)
    X, Xf, W_qkv, W_o, W_1, W_2 = rnd(M, d), rnd(M, ff), rnd(3 * d, d) * 0.03, rnd(d, d) * 0.03, rnd(ff, d) * 0.03, rnd(d, ff) * 0.03
    W_qkvT = rnd(d, 3 * d) * 0.03
    QKV, Y, U, Hh = rnd(M, 3 * d), rnd(M, d), rnd(M, ff), torch.empty(M, ff, device=device, dtype=bf)
    bias3, bias1, biasf = torch.zeros(3 * d, device=device), torch.zeros(d, device=device), torch.zeros(ff, device=device)
    dW = torch.zeros(3 * d, d, device=device)
    db = torch.zeros(3 * d, device=device)
    O = torch.empty(M, d, device=device, dtype=bf)
    lse = torch.empty(B * H * S, device=device)
    delta = torch.empty(B * H * S, device=device)
    dQKV = torch.empty(M, 3 * d, device=device, dtype=bf)
    km = torch.zeros(B, S, dtype=torch.uint8, device=device)
    km[:, S - 100:] = 1
    drop = ops.drop_params(P_TOKEN, 1, 1)
    dbits = ops.attn_dropmask(B, H, S, P_TOKEN, 1, 1, device)
    att = Lb.TfAttnArgs(qkv=Lb.ptr(QKV), ld_qkv=3 * d, out=Lb.ptr(O), ld_out=d, lse=Lb.ptr(lse), key_mask=Lb.ptr(km), B=B, S=S, H=H, HDP=hd,
                        scale=1 / math.sqrt(hd), drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2], drop_bits=Lb.ptr(dbits), dout=Lb.ptr(Y), ld_dout=d,
                        dqkv=Lb.ptr(dQKV), ld_dqkv=3 * d, delta=Lb.ptr(delta))
    mean, rstd = torch.empty(M, device=device), torch.empty(M, device=device)
    gam, bet, dgam, dbet = torch.ones(d, device=device), torch.zeros(d, device=device), torch.zeros(d, device=device), torch.zeros(d, device=device)
    ln = Lb.TfLnArgs(x=Lb.ptr(X), ldx=d, y=Lb.ptr(O), ldy=d, y_is_f32=0, gamma=Lb.ptr(gam), beta=Lb.ptr(bet), mean=Lb.ptr(mean), rstd=Lb.ptr(rstd),
                     rows=M, d=d, rows_per_group=M, x_group_stride=M, y_group_stride=M, eps=1e-5, dy=Lb.ptr(Y), lddy=d, dy_is_f32=0,
                     dx=Lb.ptr(O), lddx=d, dx_drop=Lb.ptr(Y), lddxd=d, drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2], drop_ld=d,
                     dgamma=Lb.ptr(dgam), dbeta=Lb.ptr(dbet))
    E = Lb
    GF = lambda m, n, k: 2.0 * m * n * k
    st = ops._stream()
    cases = [
        # name, kernel symbol, launches per layer-step, callable, algorithmic flops per launch, algorithmic HBM bytes per launch
        ("gemm_qkv", "gemm_nt<BIAS>", 1, lambda: ops.gemm(X, W_qkv, QKV, 3 * d, d, E.TF_EPI_BIAS, bias=bias3), GF(M, 3 * d, d), 2 * (M * d + M * 3 * d)),
        ("gemm_outproj+drop+res", "gemm_nt<BIAS_DROP_RES>", 1, lambda: ops.gemm(X, W_o, O, d, d, E.TF_EPI_BIAS_DROP_RES, bias=bias1, R=Y, drop=drop), GF(M, d, d), 2 * 3 * M * d),
        ("gemm_ffn_up+gelu+G+drop", "gemm_nt<BIAS_GELU_DROP_G>", 1, lambda: ops.gemm(X, W_1, U, ff, d, E.TF_EPI_BIAS_GELU_DROP_G, bias=biasf, C2=Hh, drop=drop), GF(M, ff, d), 2 * (M * d + 2 * M * ff)),
        ("gemm_ffn_down+drop+res", "gemm_nt<BIAS_DROP_RES>", 1, lambda: ops.gemm(Xf, W_2, O, d, ff, E.TF_EPI_BIAS_DROP_RES, bias=bias1, R=Y, drop=drop), GF(M, d, ff), 2 * (M * ff + 2 * M * d)),
        ("dgrad_ffn_down*G", "gemm_nt<MUL>", 1, lambda: ops.gemm(X, W_1, Hh, ff, d, E.TF_EPI_MUL, R=U), GF(M, ff, d), 2 * (M * d + 2 * M * ff)),
        ("dgrad_ffn_up+add", "gemm_nt<ADD>", 1, lambda: ops.gemm(Xf, W_2, O, d, ff, E.TF_EPI_ADD, R=Y), GF(M, d, ff), 2 * (M * ff + 2 * M * d)),
        ("dgrad_outproj", "gemm_nt<NONE>", 1, lambda: ops.gemm(X, W_o, O, d, d, E.TF_EPI_NONE), GF(M, d, d), 2 * 2 * M * d),
        ("dgrad_qkv+add", "gemm_nt<ADD>", 1, lambda: ops.gemm(QKV, W_qkvT, O, d, 3 * d, E.TF_EPI_ADD, R=Y), GF(M, d, 3 * d), 2 * (M * 3 * d + 2 * M * d)),
        ("wgrad_qkv", "wgrad_tn_kernel", 1, lambda: ops.wgrad(QKV, 3 * d, X, d, dW, db), GF(M, 3 * d, d), 2 * (M * 3 * d + M * d)),
        ("wgrad_d_d", "wgrad_tn_kernel", 1, lambda: ops.wgrad(Y, d, X, d, dW[:d], db[:d]), GF(M, d, d), 2 * 2 * M * d),
        ("wgrad_ffn", "wgrad_tn_kernel", 2, lambda: ops.wgrad(U, ff, X, d, dW[:ff], db[:ff]), GF(M, ff, d), 2 * (M * ff + M * d)),
        ("attn_fwd", "attn_fwd_kernel<192>", 1, lambda: Lb.call("tf_attn_fwd", att, st), 4.0 * B * S * S * d, 2 * (M * 3 * d + M * d)),
        ("attn_bwd(delta+dq+dkv)", "attn_bwd_dq+dkv_kernel<192>", 1, lambda: Lb.call("tf_attn_bwd", att, st), 8.0 * B * S * S * d, 2 * (2 * M * 3 * d + 2 * M * d)),
        ("layernorm_fwd", "ln_fwd_kernel", 2, lambda: Lb.call("tf_layernorm_fwd", ln, st), 0.0, 2 * 2 * M * d),
        ("layernorm_bwd", "ln_bwd_kernel", 2, lambda: Lb.call("tf_layernorm_bwd", ln, st), 0.0, 2 * 4 * M * d),
    ]
    out = []
    for name, symbol, per_layer, fn, fl, by in cases:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out.append(dict(kernel=name, symbol=symbol, us=round(us, 2), launches_per_step=per_layer * L, us_per_step=round(us * per_layer * L, 1),
                        tflops=round(fl / us / 1e6, 1) if fl else None, gbs=round(by / us / 1e3, 1), flops=fl, bytes=by))
    return out



def traced_kernels(step, nsteps):
    """Per-kernel durations IN SITU: `nsteps` further training steps with the library's launch tracer on (a HIP event pair
    around every kernel, on the stream it is launched on -- the side stream for the overlapped weight-gradient GEMMs).
    Aggregated by kernel symbol, i.e. the rows of `rocprofv3 --kernel-trace --stats` for the same command."""
    import ctypes
    from transfusion_amd import _lib as Lb
    lib = Lb.load()
    Lb.check(lib.tf_trace_start(), "tf_trace_start")
    for j in range(nsteps):
        step(j)
    cap = 1 << 14
    recs = (Lb.TfTraceRecord * cap)()
    n = lib.tf_trace_stop(ctypes.addressof(recs), cap)
    if n < 0:
        Lb.check(int(n), "tf_trace_stop")
    # GPU occupancy of the traced window by THIS library's kernels: union of the launch intervals (both streams)
    iv = sorted((recs[i].start_us, recs[i].start_us + recs[i].us) for i in range(min(n, cap)))
    busy, cs, ce = 0.0, iv[0][0], iv[0][1]
    for s0, e0 in iv[1:]:
        if s0 > ce:
            busy += ce - cs
            cs, ce = s0, e0
        else:
            ce = max(ce, e0)
    busy += ce - cs
    span = max(e0 for _, e0 in iv) - iv[0][0]
    log(f"  traced window {span / nsteps:.0f} us/step (tracing on), library kernels busy {busy / nsteps:.0f} us/step, "
        f"other (harness torch kernels + idle) {(span - busy) / nsteps:.0f} us/step")
    by = {}
    for i in range(min(n, cap)):
        r = recs[i]
        a = by.setdefault(r.name.decode(), dict(us=0.0, launches=0, flops=0.0, bytes=0.0, side=0))
        a["us"] += r.us; a["launches"] += 1; a["flops"] += r.flops; a["bytes"] += r.bytes; a["side"] += r.side
    rows = []
    for name, a in by.items():
        rows.append(dict(kernel=name, avg_us=round(a["us"] / a["launches"], 2), launches_per_step=round(a["launches"] / nsteps, 2),
                         us_per_step=round(a["us"] / nsteps, 1), tflops=round(a["flops"] / a["us"] / 1e6, 1) if a["flops"] else None,
                         gbs=round(a["bytes"] / a["us"] / 1e3, 1) if (a["bytes"] and not a["flops"]) else None, side_stream=a["side"] > 0,
                         flops_per_launch=a["flops"] / a["launches"], bytes_per_launch=a["bytes"] / a["launches"]))
    rows.sort(key=lambda r: -r["us_per_step"])
    return rows


def wgrad_alone(B, device, reps=20):
    """The dominant kernel with the chip to itself: a layer's four weight-gradient products as ONE launch (tf_gemm_wgrad_multi), sized as
    the encoder runtime sizes it on its side stream (~1.7 workgroups per CU: three row chunks of the 144 output tiles), back to back on
    one stream, at the packed row count the benchmark's batches average (196 + 320 tokens per sample)."""
    from transfusion_amd import ops
    M, d, ff = B * (NV + (NL // 4 + NL) // 2), D, D * FF_MULT
    g = torch.Generator().manual_seed(2)
    rnd = lambda *s: torch.randn(*s, generator=g).to(device=device, dtype=torch.bfloat16)
    x, o, x1, hh = rnd(M, d), rnd(M, d), rnd(M, d), rnd(M, ff)
    dqkv, dy1, dy2, du = rnd(M, 3 * d), rnd(M, d), rnd(M, d), rnd(M, ff)
    pairs = [(dqkv, x), (dy1, o), (du, x1), (dy2, hh)]                                        # in_proj, out_proj, linear1, linear2
    dWs = [torch.zeros(a.shape[1], b.shape[1], device=device) for a, b in pairs]
    dbs = [torch.zeros(a.shape[1], device=device) for a, _ in pairs]
    probs = [ops.wgrad_args(a, a.shape[1], b, b.shape[1], w, v) for (a, b), w, v in zip(pairs, dWs, dbs)]
    tiles = sum(((a.shape[1] + 255) // 256) * ((b.shape[1] + 127) // 128) for a, b in pairs)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    blocks = max(1, (cus * 17 // 10 + tiles // 2) // tiles) * tiles
    fn = lambda: ops.wgrad_multi(probs, blocks)
    # ... and the same products as the library sizes a launch that HAS the chip to itself (blocks = 0: 128 tiles of 192 x 192 at two row
    # chunks, one 8-wave workgroup per CU, the two row halves reduced in LDS before one atomic pass -- what the last launch of a backward
    # runs).  The two sizings are timed in ALTERNATING blocks of launches: back to back, whichever comes second runs on the clock the
    # first left behind (the same launch measured 7 % longer in second place).
    fn0 = lambda: ops.wgrad_multi(probs, 0)
    fn(); fn0()
    torch.cuda.synchronize()
    tot, per = [0.0, 0.0], max(1, reps // 4)
    for _ in range(4):
        for k, f in enumerate((fn, fn0)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(per):
                f()
            e1.record()
            torch.cuda.synchronize()
            tot[k] += e0.elapsed_time(e1) * 1e3
    us, us0 = tot[0] / (4 * per), tot[1] / (4 * per)
    fl = sum(2.0 * M * a.shape[1] * b.shape[1] for a, b in pairs)
    return dict(avg_launch_us=round(us, 1), achieved=round(fl / us / 1e6, 1), frac=round(fl / us / 1e6 / PEAK_BF16_TFLOPS, 4),
                flops_per_launch=fl, rows=M, workgroups=blocks,
                note="one layer's four products as one launch, the encoder runtime's sizing, alone back to back",
                chip_to_itself=dict(avg_launch_us=round(us0, 1), achieved=round(fl / us0 / 1e6, 1), frac=round(fl / us0 / 1e6 / PEAK_BF16_TFLOPS, 4),
                                    note="the same launch sized by the library for a launch with the chip to itself (blocks = 0: the 192 x 192 two-quad form)"))


def cpu_baseline(seconds_budget=12.0):
    """The CPU oracle (a port of the reference arithmetic) timed on this host: forward + backward of the same 4-layer encoder, dropout
    masks drawn on the host (as the reference does).  SURVEY.md 8(d) asks for config 1 (B = 2) and the north-star shape (B = 32), dropout
    on and p = 0: the headline object is B = 2 with dropout (best of a few steps), `variants` holds the other three on a bounded sample
    (one warm-up + one or two timed steps each; ~30 s of CPU work in all)."""
    from oracle import fusion_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthreads = max(1, min(avail, 16))            # a one-GPU box owns a 16-core share of the host
    torch.set_num_threads(nthreads)
    enc = make_encoder("cpu")
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "pos_embedding" not in k and "heatmap" not in k)
          for k, v in enc.state_dict().items()}
    S = NV + NL

    def measure(B, dropout, budget, max_steps, warm_up=True):
        x, lang, pad = make_batch(B, "cpu", 0)[:3]

        def step():
            masks, pt, pp = None, 0.0, 0.0
            if dropout:
                pt, pp = P_TOKEN, P_PATCH
                masks = {"patch": (torch.rand(B, NV, D) >= P_PATCH).float()}
                for l in range(L):
                    pre = f"t_encoder.layers.{l}."
                    masks[pre + "attn"] = (torch.rand(B, H, S, S) >= P_TOKEN).float()
                    masks[pre + "dropout1"] = (torch.rand(B, S, D) >= P_TOKEN).float()
                    masks[pre + "dropout"] = (torch.rand(B, S, D * FF_MULT) >= P_TOKEN).float()
                    masks[pre + "dropout2"] = (torch.rand(B, S, D) >= P_TOKEN).float()
            for v in sd.values():
                v.grad = None
            vis, lo = O.encoder_forward(sd, x, lang, pad, H, L, masks=masks, token_dropout=pt, patch_dropout=pp)
            valid = (~pad).unsqueeze(-1).float()
            loss = vis.pow(2).mean() + (lo.pow(2) * valid).sum() / (valid.sum() * D)
            loss.backward()

        warm = 0.0
        if warm_up:
            t0 = time.perf_counter()
            step()
            warm = time.perf_counter() - t0
        times = []
        while len(times) < 1 or (sum(times) + warm < budget and len(times) < max_steps):
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
        best = min(times)
        return dict(value=round(B / best, 3), ms_per_step=round(best * 1e3, 1), steps=len(times), batch=B, dropout=bool(dropout))

    head = measure(2, True, seconds_budget, 12)
    variants = {"b2_p0": measure(2, False, 3.0, 6), "b32_dropout": measure(32, True, 0.0, 1, warm_up=False),
                "b32_p0": measure(32, False, 0.0, 1, warm_up=False)}
    return dict(value=head["value"], unit="samples/s", cores=nthreads, kind="port",
                sample=f"oracle/fusion_oracle.py fwd+bwd, dropout on, B=2 x [{NV}+{NL}] tokens, d={D}, {L} layers, fp32, "
                       f"best of {head['steps']} steps after 1 warm-up ({head['ms_per_step']:.0f} ms/step); variants: the same at B = 32 "
                       f"(ONE step, no warm-up: a bounded sample) and with every dropout p = 0",
                variants=variants)


class _Comm:
    """The collectives bench.py's control flow issues itself (the gradient all-reduces are issued by the trainer inside step())."""

    def __init__(self, world, device, live=None):
        self.world, self.device = world, device
        self.live = world > 1 if live is None else live       # (True also in the one-rank rehearsal of the RCCL path: REHEARSE)

    def sync(self):
        if self.live:
            dist.barrier()
        torch.cuda.synchronize()

    def max(self, x: float) -> float:
        if not self.live:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


class _QuietGC:
    """Around a timed region: a full collection of this process (torch, four model families, thousands of modules: ~1.5 M tracked
    objects) takes 60 - 165 ms, and CPython starts one whenever its allocation counters say so -- inside a host-bound step
    (B = 4: the host needs 1.1 ms per step) that is a stall of 50 - 100 steps' worth.  Measured in round 5 (TF_LEG_TRACE=1 logs every
    collection): the outliers of `legs.b4_dense`, `v1_d712`, `wrapper_b4_dp` were exactly these.  gc.freeze() moves everything alive
    after the warm-up into the permanent generation, so the collector still runs during the timed steps but only walks what they
    created (< 1 ms) -- what a long-running trainer does for the same reason."""

    def __enter__(self):
        import gc
        gc.collect()
        gc.freeze()
        return self

    def __exit__(self, *exc):
        import gc
        gc.unfreeze()
        return False


def run_schedule(step, comm, rank, warmup, steps, trace_steps, traced):
    """The benchmark's control flow, the same on EVERY rank: W untimed steps, barrier + sync, K timed steps, barrier + sync,
    max over ranks; then `trace_steps` further steps which rank 0 runs under the launch tracer (`traced(n)`) and every other rank
    runs plainly -- each step contains the gradient collectives, so a rank that skipped them would leave rank 0's all-reduces
    paired with the others' final barrier (the hang of round 1's 2-rank rehearsal).  `step(i)` runs training step number i.
    tests/test_bench_schedule_cpu.py drives this with recording fakes and asserts identical collective sequences per rank."""
    i = 0
    for _ in range(warmup):
        step(i); i += 1
    comm.sync()
    if rank == 0:
        log(f"  warm-up done ({warmup} steps)")
    with _QuietGC():
        t0 = time.perf_counter()
        for _ in range(steps):
            step(i); i += 1
        comm.sync()
        elapsed = comm.max(time.perf_counter() - t0)
    if rank == 0:
        log(f"  timed region done ({steps} steps, {elapsed / max(steps, 1) * 1e3:.3f} ms/step)")
    rows = None
    if trace_steps > 0:
        if rank == 0:
            rows = traced(lambda j: step(i + j), trace_steps)
        else:
            for j in range(trace_steps):
                step(i + j)
    return elapsed, rows


def allreduce_busbw(trainer, comm, reps=10):
    """Stand-alone all-reduce of the flat gradient buffer, per layer range as the trainer issues it: algorithm and bus bandwidth
    (busbw = algbw * 2 (N - 1) / N, the per-link figure to hold against xGMI's ~153 GB/s per direction per link)."""
    world = comm.world
    g = trainer.flat.grad
    ranges = sorted(trainer.layerwise.ranges.values()) if trainer.layerwise is not None else trainer.reducer.buckets
    comm.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for lo, hi in ranges:
            if trainer.bucket_comm is not None:
                trainer.bucket_comm.all_reduce_(g[lo:hi])
            else:
                dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM)
    comm.sync()
    dt = comm.max(time.perf_counter() - t0) / reps
    nbytes = g.numel() * 4
    g.zero_()
    return dict(bytes=nbytes, collectives_per_step=len(ranges), ms=round(dt * 1e3, 3), algbw_gbs=round(nbytes / dt / 1e9, 1),
                busbw_gbs=round(nbytes / dt / 1e9 * 2 * (world - 1) / world, 1),
                note="all-reduce alone, back to back (not overlapped with the backward)")


def rccl_probe(trainer, comm, device, rank):
    """N > 1 over RCCL: what the exchange ran on -- the process group's backend and size -- and the library's OWN communicator
    (tf_comm_create / tf_allreduce_bucket, csrc/comm.hip) exercised WITH its peers: one all-reduce compared with the process group's
    result, then its tf_comm_stats (world, rank, calls, elements).  Collective: every rank calls it.  A failure is reported, not raised
    (the headline line does not depend on it)."""
    out = {"backend": dist.get_backend(), "group_world": dist.get_world_size()}
    if out["backend"] != "nccl":
        return out                                   # (the one-GPU rehearsal over gloo: RCCL refuses two ranks on one device)
    try:
        from transfusion_amd.comm import BucketComm
        bc = trainer.bucket_comm if trainer.bucket_comm is not None else BucketComm.from_process_group(device)
        n = 1 << 22
        t = torch.arange(n, dtype=torch.float32, device=device) * (1.0 / n) + float(rank + 1)
        ref = t.clone()
        bc.all_reduce_(t)
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        comm.sync()
        same = bool(torch.equal(t, ref))
        big = torch.zeros(16 << 20, dtype=torch.float32, device=device)            # 64 MB, ten times
        comm.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            bc.all_reduce_(big)
        comm.sync()
        dt = comm.max(time.perf_counter() - t0) / 10
        w = dist.get_world_size()
        out["rccl"] = dict(bc.stats(), matches_process_group=same, busbw_gbs_64mb=round(big.numel() * 4 / dt / 1e9 * 2 * (w - 1) / w, 1))
        if trainer.bucket_comm is None:
            bc.close()
    except Exception as e:                           # noqa: BLE001 -- reported on the line
        out["rccl"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def probe_under_timer(result, probe, rank, timeout_s=None):
    """Runs ``probe()`` LAST and under a timer and merges what it returns into ``result["allreduce"]``.  The probe is the library's own
    RCCL communicator exercised with its peers (a second communicator next to the process group's): its bootstrap has never run on a
    multi-GPU node, and a communicator that never forms cannot be cancelled -- so if the probe is not back in time, the line measured so
    far is printed WITHOUT it (rank 0) and the process leaves with exit code 0: the benchmark result never depends on the probe."""
    line = dict(result)
    line["allreduce"] = dict(result.get("allreduce", {}), rccl={"error": "probe did not return in time"})
    out = _under_timer(line, probe, rank, float(timeout_s if timeout_s is not None else os.environ.get("TF_RCCL_PROBE_TIMEOUT_S", "90")))
    result.setdefault("allreduce", {}).update(out)


def _under_timer(line_if_late, fn, rank, timeout_s):
    """``fn()`` under a timer on EVERY rank: if it is not back in time, rank 0 prints ``line_if_late`` and every rank leaves with exit
    code 0 (a collective that never completes cannot be cancelled, and the measured line must not depend on an extra)."""
    import threading

    def bail():
        if rank == 0:
            print(json.dumps(line_if_late), flush=True)
        os._exit(0)
    timer = threading.Timer(timeout_s, bail)
    timer.daemon = True
    timer.start()
    try:
        return fn()
    finally:
        timer.cancel()


def extra_under_timer(result, key, fn, rank, timeout_s):
    """An EXTRA measurement of the N > 1 line (``result[key] = fn()``), under the same rule as the probe: a failure is reported in the
    entry, a hang costs the entry, never the line."""
    line = dict(result)
    line[key] = {"error": "did not return in time"}

    def guarded():
        try:
            return fn()
        except BaseException as e:                   # noqa: BLE001 -- reported on the line (SystemExit of a non-finite loss included)
            return {"error": f"{type(e).__name__}: {e}"[:300]}
    result[key] = _under_timer(line, guarded, rank, timeout_s)


def live_sections(result, *, world, rank, checksum, busbw, strong=None, wrapper_dp=None, probe=None, log_fn=None):
    """Everything the N > 1 line carries beyond the headline, in the ONE order every rank walks it (each part holds collectives: a rank
    that skipped one, or took them in another order, would hang the others):
      1. ``rank_sync``     -- the data-parallel invariant: after any number of steps every rank holds bit-identical parameters
                              (``checksum()``: a one-element float64 tensor on the collectives' device; MIN and MAX over ranks must agree);
      2. ``allreduce``     -- ``busbw()``: the gradient exchange alone, algorithm and bus bandwidth, + the backend and the group size;
      3. ``strong``        -- ``strong()``: the reference's own batch arithmetic, the GLOBAL batch of 32 divided by the device count
                              (run_experiment.py:373-374), world > 1 only;
      4. ``wrapper_b4_real_dp`` -- ``wrapper_dp()``: the reference's real module under the ordered range reducer, an EXTRA under a timer;
      5. ``allreduce.rccl`` -- ``probe()``: the library's own communicator with its peers, LAST and under a timer.
    main() passes the real measurements; tests/test_bench_schedule_cpu.py drives the same function with eight gloo ranks and stand-ins
    that issue real collectives (no 8-GPU node has been available to any round: this is the part of the N = 8 line that can be
    rehearsed without one)."""
    if os.environ.get("TF_CHECK_SYNC", "1") != "0":
        mine = checksum().reshape(1)
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert lo.item() == hi.item(), (lo.item(), hi.item())
        result["rank_sync"] = {"parameter_checksum": lo.item(), "identical_on_ranks": world}
        if rank == 0 and log_fn is not None:
            log_fn(f"  parameter checksum identical on {world} ranks: {lo.item():.6f}")
    result["allreduce"] = busbw()                                 # every rank takes part; rank 0 prints
    result["allreduce"].update({"backend": dist.get_backend(), "group_world": dist.get_world_size()})
    if world > 1 and strong is not None:
        s = strong()
        result["strong"] = {"global_batch": s["batch_per_gpu"] * world, "samples_s": s["samples_s"], "ms_per_step": s["ms_per_step"],
                            "batch_per_gpu": s["batch_per_gpu"]}
    if wrapper_dp is not None and os.environ.get("TF_WRAPPER_DP_LEG", "1") != "0":
        extra_under_timer(result, "wrapper_b4_real_dp", wrapper_dp, rank, float(os.environ.get("TF_WRAPPER_DP_TIMEOUT_S", "150")))
    if probe is not None and os.environ.get("TF_RCCL_PROBE", "1") != "0":
        probe_under_timer(result, probe, rank)


def run_leg(name, device, rank, comm, *, precision="bf16", batch=32, d=D, h=H, layers=L, nv=NV, nl=NL, fp8=False, pack=True, padded=True,
            steps=8, warmup=3, grad_clip=1.0):
    """One more BASELINE configuration in the same process, after the headline: its own encoder, trainer and batches, `warmup` untimed +
    `steps` timed training steps between barrier + sync pairs (the headline's protocol), max over ranks.  Returns the leg's numbers;
    everything it allocated is released before the next leg."""
    import gc
    from transfusion_amd.runner.trainer import FusionTrainStep
    global PACK_TOKENS
    saved = PACK_TOKENS
    PACK_TOKENS = pack
    obj = {}
    try:
        enc = make_encoder(device, d=d, h=h, layers=layers)
        enc.precision = precision
        enc.fp8_projections = bool(fp8)
        enc.train()
        obj["trainer"] = FusionTrainStep(enc, lr=1e-4, weight_decay=2e-4, grad_clip=grad_clip, zero_grads_in_optimizer=ZERO_IN_OPT)
        obj["batches"] = [make_batch(batch, device, rank, d=d, nv=nv, nl=nl, variant=v, padded=padded) for v in range(2)]
        del enc
        last = {}

        def step(i):
            last["loss"] = obj["trainer"].step([obj["batches"][i % 2]], loss_fn)

        for i in range(warmup):
            step(i)
        comm.sync()
        quiet = _QuietGC().__enter__()
        t0 = time.perf_counter()
        marks = []
        for i in range(steps):
            if i == 0 and os.environ.get("TF_LEG_TRACE") == "1":
                import faulthandler as _fh
                _fh.dump_traceback_later(0.008, repeat=False)       # where is the host 8 ms into the first timed step?
            step(warmup + i)
            if i == 0 and os.environ.get("TF_LEG_TRACE") == "1":
                _fh.cancel_dump_traceback_later()
            marks.append(time.perf_counter())
        t_host = (time.perf_counter() - t0) / steps        # the host has ENQUEUED the steps by now (no sync inside a step)
        comm.sync()
        dt = comm.max(time.perf_counter() - t0) / steps
        quiet.__exit__()
        if os.environ.get("TF_LEG_TRACE") == "1" and rank == 0:
            log(f"  leg {name}: host ms per step " + " ".join(f"{1e3 * (b_ - a_):.2f}" for a_, b_ in zip([t0] + marks[:-1], marks)))
        loss = float(last["loss"].item())
        if not math.isfinite(loss):
            raise SystemExit(f"leg {name}: non-finite loss {loss}")
        S = nv + nl
        fl_dense = 3 * layers * flops_per_sample_layer(S, d) * batch               # per GPU, dense S
        # executed work: the nv + len_b real tokens of each sample, averaged over the leg's batches (utilisation is quoted on THIS)
        fl = sum(3 * layers * sum(flops_per_sample_layer(nv + n, d) for n in b[5]) for b in obj["batches"]) / len(obj["batches"])
        # peak the leg is priced against: dense bf16 MFMA; a third of it in the fp32-accuracy mode (three bf16 passes per product).
        # The fp8 leg runs its forward projections on fp8 operands and everything else in bf16: priced against the bf16 peak.
        peak = PEAK_BF16_TFLOPS / 3.0 if precision == "fp32" else PEAK_BF16_TFLOPS
        out = dict(ms_per_step=round(dt * 1e3, 3), samples_s=round(comm.world * batch / dt, 1), batch_per_gpu=batch, tokens=[nv, nl], d=d, heads=h,
                   layers=layers, dtype="fp32" if precision == "fp32" else ("fp8 projections + bf16" if fp8 else "bf16"),
                   block_tflops_per_gpu=round(fl / dt / 1e12, 1), block_mfma_util=round(fl / dt / 1e12 / peak, 4), peak_used=round(peak, 1),
                   block_mfma_util_dense_credit=round(fl_dense / dt / 1e12 / peak, 4),
                   packed_rows=bool(pack), padded=bool(padded), steps=steps, warmup=warmup, final_loss=round(loss, 5),
                   host_enqueue_ms=round(t_host * 1e3, 3))
        if rank == 0:
            log(f"  leg {name:10s} {out['ms_per_step']:8.3f} ms/step  {out['samples_s']:9.1f} samples/s  {out['block_tflops_per_gpu']:7.1f} TFLOP/s/GPU "
                f"({100 * out['block_mfma_util']:.1f} % of {peak:.0f}; host enqueue {out['host_enqueue_ms']:.2f} ms/step)")
        return out
    finally:
        PACK_TOKENS = saved
        obj.clear()
        gc.collect()
        torch.cuda.empty_cache()


class _PassThroughDetector(torch.nn.Module):
    """rcnn_model stand-in for the wrapper leg: the feature maps pass straight through (the detector is out of scope)."""

    def __init__(self, shapes, channels):
        super().__init__()
        self.shapes, self.channels = shapes, channels
        self.noun_classes, self.verb_classes = 88, 75

    def get_dsampled_shapes(self):
        return self.shapes

    def get_features_out_channels(self):
        return self.channels

    def forward_features(self, images, targets=None):
        return {"features": {str(i): f for i, f in enumerate(images)}}

    def apply_fpn(self, fd):
        return fd

    def apply_rpn_roi_on_features(self, fd):
        return fd

    def call_model_epoch_triggers(self, epoch):
        pass


def run_wrapper_leg(device, rank, comm, batch=4, steps=10, warmup=4, real=False, reducer=False, d=D, precision=16):
    """The reference's REAL module around the hot path, at its own per-GPU batch: CrossFusionBoxWrapper over four FPN levels (level maps
    14p x 14p, p = 4, 4, 2, 1; C = 256 .. 2048; patch-embedding GEMM, 4-layer encoder on [196 + 512] tokens, back-projection + fold per
    level) with a pass-through detector, full training step (FusionTrainStep: flat buffers, clip, fused RAdam).  Host-bound at this size
    (several hundred launches per step): `host_enqueue_ms` is the time the host needs to issue one step."""
    import gc
    from transfusion_amd.modeling.model_factory import get_fusion_model
    from transfusion_amd.runner.config import load_fusion_config
    from transfusion_amd.runner.trainer import FusionTrainStep
    obj = {}
    saved_force = os.environ.get("TF_FORCE_LAYERWISE")
    try:
        ps, chans = [4, 4, 2, 1], [256, 512, 1024, 2048]
        # `real`: the reference's FPN geometry (patches of 4, 4, 2, 1 on maps of stride 4 / 8 / 16 / 32 of a 448-pixel frame,
        # cross_fusion_config_sym_ego_res50.yml:8-17): token grids 28 x 28, 14 x 14, 14 x 14, 14 x 14 -- level 0 holds four times the
        # tokens of the others; levels 1 - 3 run as one grouped encoder call, level 0 beside them.  Otherwise 14 x 14 on every level.
        grids = [28, 14, 14, 14] if real else [14, 14, 14, 14]
        shapes = [(n * p, n * p) for n, p in zip(grids, ps)]
        if reducer:
            os.environ["TF_FORCE_LAYERWISE"] = "1"       # the data-parallel path at world 1: layer-by-layer backward, per-unit hooks, no-op reduce
        fusion = load_fusion_config(os.path.join(ROOT, "transfusion_amd", "runner", "configs", "cross_fusion_config_sym_ego_res50.yml"))
        fusion.update({"fpn_features": [0, 1, 2, 3], "replace_fpn_features": True})
        fusion["args"].update({"input_f_size": d})
        run_cfg = {"experiment": "egonao", "narr_fusion": fusion, "criterion": {"lm": 0}, "precision": precision,
                   "narration_embeds": {"use": True, "args": {"text_pooling": "slowfast", "strategy": "current", "out_mlp": 0, "size": d,
                                                             "out_dropout": 0.0, "out_tanh": False, "train_ep": 0}}}
        torch.manual_seed(42)
        model = get_fusion_model(_PassThroughDetector(shapes, chans), {}, run_cfg, None).to(device).train()
        g = torch.Generator().manual_seed(4242 + 1000 * rank)
        feats = [torch.randn(batch, c, h, w, generator=g).to(device).requires_grad_(True) for c, (h, w) in zip(chans, shapes)]
        lens = torch.randint(NL // 4, NL + 1, (batch,), generator=g).tolist()
        lang = [torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1).to(device) for n in lens]
        obj["trainer"] = FusionTrainStep(model, lr=1e-4, weight_decay=2e-4, grad_clip=1.0, zero_grads_in_optimizer=ZERO_IN_OPT)
        del model

        def wloss(m, _):
            out = m({"image": feats, "language_f": lang})
            fs = [f for f in out["features"].values()]
            if all(f.dtype == torch.float32 and f.is_contiguous() and f.numel() % 4 == 0 and f.data_ptr() % 16 == 0 for f in fs):
                from transfusion_amd import ops
                return ops.sq_loss([(f, None, 1.0 / f.numel()) for f in fs])        # one scalar, len(fs) launches each way
            return sum(_square_mean(f) for f in fs)

        last = {}
        for _ in range(warmup):
            last["loss"] = obj["trainer"].step([None], wloss)
        comm.sync()
        # TWO timed regions of `steps` steps, both reported (`ms_per_step_runs`), the lower one quoted: the step is within 1.5x of the
        # host's enqueue time, and on the shared boxes of the pool single regions came out 2x long now and then (host enqueue 6.8 ms
        # against 2.9 in one take of the closing set; profiles/r06_v6_bench.json: 9.88 against 4.41 ms) -- a host stall is then visible as the difference, not hidden
        runs = []
        with _QuietGC():
            for _ in range(2):
                t0 = time.perf_counter()
                for _ in range(steps):
                    last["loss"] = obj["trainer"].step([None], wloss)
                th = (time.perf_counter() - t0) / steps
                comm.sync()
                runs.append((comm.max(time.perf_counter() - t0) / steps, th))
        dt, t_host = min(runs)
        loss = float(last["loss"].item())
        if not math.isfinite(loss):
            raise SystemExit(f"leg wrapper_b{batch}: non-finite loss {loss}")
        vis_tokens = [n * n for n in grids]
        # executed work of one step (train = 3 x forward): per level the encoder on its nv_l + len_b real tokens, plus the
        # patch-embedding (K1: [B nv_l, C p^2] x [C p^2, d]) and back-projection (K9: [B nv_l, d] x [d, C p^2]) GEMMs
        fl = 0.0
        for nv_l, c, p in zip(vis_tokens, chans, ps):
            fl += 3 * L * sum(flops_per_sample_layer(nv_l + n, d) for n in lens)
            fl += 3 * 2 * (2.0 * batch * nv_l * c * p * p * d)
        peak = PEAK_BF16_TFLOPS / 3.0 if str(precision) == "32" else PEAK_BF16_TFLOPS
        out = dict(ms_per_step=round(dt * 1e3, 3), samples_s=round(comm.world * batch / dt, 1), host_enqueue_ms=round(t_host * 1e3, 3),
                   ms_per_step_runs=[round(r[0] * 1e3, 3) for r in runs], host_enqueue_ms_runs=[round(r[1] * 1e3, 3) for r in runs],
                   batch_per_gpu=batch, levels=4, layers_per_level=L, tokens=[NV, NL], vis_tokens_per_level=vis_tokens, d=d,
                   dtype="fp32" if str(precision) == "32" else "bf16",
                   credited_tflop_per_step=round(fl / 1e12, 3), block_tflops_per_gpu=round(fl / dt / 1e12, 1),
                   block_mfma_util=round(fl / dt / 1e12 / peak, 4), peak_used=round(peak, 1),
                   steps=steps, warmup=warmup, final_loss=round(loss, 5),
                   us_per_visual_token=round(dt * 1e6 / (batch * sum(vis_tokens)), 3),
                   reducer=type(obj["trainer"].layerwise).__name__ if obj["trainer"].layerwise is not None else None,
                   module="CrossFusionBoxWrapper (4 FPN levels) + pass-through detector, full training step")
        if rank == 0:
            tag = "wrapper_b%d%s%s%s" % (batch, "_real" if real else "", "_dp" if reducer else "", "" if d == D and str(precision) != "32" else f"_d{d}_p{precision}")
            log(f"  leg {tag:16s} {out['ms_per_step']:8.3f} ms/step  {out['samples_s']:9.1f} samples/s  (host enqueue {out['host_enqueue_ms']:.2f} ms/step)")
        return out
    finally:
        if saved_force is None:
            os.environ.pop("TF_FORCE_LAYERWISE", None)
        else:
            os.environ["TF_FORCE_LAYERWISE"] = saved_force
        obj.clear()
        gc.collect()
        torch.cuda.empty_cache()


def csrc_hash():
    """Hash of the kernel sources: profiles/traffic.json is stamped with the value it was measured on."""
    import hashlib
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "transfusion_amd", "csrc")
    for fn in sorted(os.listdir(cs)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(cs, fn), "rb").read())
    return h.hexdigest()[:16]


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`
    as a child process (rank 0's JSON line goes to our stdout, everything else to stderr) and return its exit code.
    TF_BENCH_LAUNCHER (tests): another launcher command line in place of `python -m torch.distributed.run`."""
    import shlex
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    launcher = shlex.split(os.environ["TF_BENCH_LAUNCHER"]) if os.environ.get("TF_BENCH_LAUNCHER") else [sys.executable, "-m", "torch.distributed.run"]
    cmd = launcher + ["--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
                      os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    log(f"  bench.py --gpus {n}: launching {' '.join(cmd[:len(launcher) + 7])} ...")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU per step")
    ap.add_argument("--batches", type=int, default=4, help="distinct synthetic batches rotated through the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-census", action="store_true", help="skip the traced steps / kernel table / roofline object")
    ap.add_argument("--trace-steps", type=int, default=5)
    ap.add_argument("--isolated-census", action="store_true", help="also time every kernel alone (back-to-back launches of one kernel)")
    ap.add_argument("--grad-clip", type=float, default=1.0)
    ap.add_argument("--with-heads", action="store_true",
                    help="variant: add the NAO RoI heads + losses (SURVEY.md 8f-2) on synthetic box features, 512 RoIs per image, repr 1024, "
                         "Ego4Dv1 class counts (88 nouns / 75 verbs); not the headline line")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16",
                    help="bf16: the headline (BASELINE configs[1]); fp32: the fp32-accuracy mode of configs[2] (run.precision: 32)")
    ap.add_argument("--no-overlap", action="store_true", help="reduce gradients after the backward instead of layer by layer")
    ap.add_argument("--no-legs", action="store_true", help="skip the other BASELINE configurations that follow the headline leg")
    ap.add_argument("--legs", default="fp32,stress,fp8,b4,b4_dense,wrapper_b4,wrapper_b4_real,wrapper_b4_real_v1,wrapper_b4_real_v2,wrapper_b4_dp,b16,v1_d712,v2_d896_fp32,dense_rows,no_padding", help="comma-separated subset of the legs to run (N = 1)")
    ap.add_argument("--dense-rows", action="store_true",
                    help="carry the masked (padding) language tokens through every kernel as dead rows instead of dropping them "
                         "(CrossTransformerModuleBox.pack_tokens = False); same results on every real token, A/B switch")
    ap.add_argument("--comm", choices=["torch", "rccl"], default=os.environ.get("TF_COMM", "torch"),
                    help="N > 1: gradient exchange through torch.distributed's process group (default) or the C ABI's own RCCL "
                         "communicator (tf_allreduce_bucket)")
    args = ap.parse_args()
    global PACK_TOKENS
    PACK_TOKENS = not args.dense_rows

    if os.environ.get("TF_LEG_TRACE") == "1":
        import gc
        _gc_t = {}

        def _gc_cb(phase, info):
            if phase == "start":
                _gc_t["t"] = time.perf_counter()
            elif info.get("generation", 0) >= 1:
                log(f"  [gc] generation {info['generation']} collection: {1e3 * (time.perf_counter() - _gc_t.get('t', 0.0)):.1f} ms, collected {info.get('collected')}")
        gc.callbacks.append(_gc_cb)
    # a hang must end with a Python stack on stderr, not with the driver's silence timeout
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("TF_BENCH_WATCHDOG_S", "900")), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # called bare (`python bench.py --gpus N`): start the N ranks ourselves, as Lightning's strategy="ddp" does for the reference
        # (runner/run_experiment.py:437-454) -- a CHILD process (no exec), started before this process touches the GPU
        faulthandler.cancel_dump_traceback_later()
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product kernels")
    # rehearsal hooks (never set by the driver): several ranks on ONE device over gloo, to exercise the N > 1 code path
    dev_index = int(os.environ.get("TF_FORCE_DEVICE", local_rank))
    backend = os.environ.get("TF_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if os.environ.get("TF_BENCH_STREAM") == "1":
        # experiment hook: the whole run on a non-default (non-blocking) stream instead of the legacy null stream, which synchronises
        # implicitly with every BLOCKING stream (hipExtStreamCreateWithCUMask makes those: TF_SIDE_CUS in experiments builds)
        torch.cuda.set_stream(torch.cuda.Stream(device=device))
    # TF_REHEARSE_COLLECTIVES=1 (tests/test_gpu_ddp.py): the N > 1 code path -- process group over RCCL, layer-wise all-reduces on the
    # communication stream, rank checks, bandwidth probe, the library's own communicator -- on ONE GPU in a ONE-rank group: everything
    # but the peers, against the real backend (gloo, which the two-rank rehearsals use, has other stream semantics)
    rehearse = world == 1 and os.environ.get("TF_REHEARSE_COLLECTIVES") == "1"
    live = world > 1 or rehearse
    if live:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            import socket
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            sock.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    from transfusion_amd.runner.trainer import FusionTrainStep
    enc = make_encoder(device)
    enc.precision = args.precision
    enc.train()
    module, step_loss = enc, loss_fn
    if args.with_heads:
        from transfusion_amd.modeling.obj_detection.nao_heads import NaoHeadLosses, NaoRoIHeads
        heads = NaoRoIHeads(1024, 88, 75, box_2_dropout=0.0, classif_dropout=0.0).to(device)
        heads.precision = args.precision
        crit = NaoHeadLosses(torch.ones(88), torch.ones(75)).to(device)
        module, step_loss = _EncoderWithHeads(enc, heads, crit).train(), loss_fn_heads
    trainer = FusionTrainStep(module, lr=1e-4, weight_decay=2e-4, grad_clip=args.grad_clip, overlap=not args.no_overlap, comm=args.comm,
                              zero_grads_in_optimizer=ZERO_IN_OPT)
    # distinct batches (tensors, padding lengths) rotated through the steps: a real loader hands the encoder a new mask tensor
    # every step, so the padding-mask conversion cache never hits
    batches = [make_batch(args.batch, device, rank, variant=v) for v in range(max(1, args.batches))]
    if args.with_heads:
        batches = [b + (make_heads_batch(args.batch, device, rank, v),) for v, b in enumerate(batches)]
    comm = _Comm(world, device, live)
    last = {}

    def step(i):
        last["loss"] = trainer.step([batches[i % len(batches)]], step_loss)

    elapsed, rows = run_schedule(step, comm, rank, args.warmup, args.steps, 0 if args.no_census else args.trace_steps, traced_kernels)
    final_loss = float(last["loss"].item())
    if not math.isfinite(final_loss):
        raise SystemExit(f"non-finite loss {final_loss}")

    ms = elapsed / args.steps * 1e3
    value = world * args.batch / (elapsed / args.steps)
    S = NV + NL
    train_flops_step = 3 * L * flops_per_sample_layer(S, D) * args.batch          # per GPU, dense S (padded tokens credited)
    # valid-token accounting (BASELINE.md section 3 / SURVEY.md 8d): only the Nv + len_b real tokens of each sample, averaged over
    # the rotated batches.  block_mfma_util credits the dense S (BASELINE.md's rule); with the masked tokens dropped (default) the
    # kernels execute the valid-token figure, with --dense-rows something between the two (all-padding key tiles are skipped).
    valid_S = [[NV + n for n in b[5]] for b in batches]
    train_flops_valid = sum(3 * L * sum(flops_per_sample_layer(sb, D) for sb in vs) for vs in valid_S) / len(valid_S)
    attn_valid_ratio = sum(sum(sb * sb for sb in vs) for vs in valid_S) / (len(valid_S) * args.batch * S * S)
    result = {
        "metric": "train samples/sec, Ego4D NAO B=32 (14x14 vis + 512 txt tok), 1/2/4/8 GPU",
        "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp32" if args.precision == "fp32" else ("fp8 projections + bf16" if os.environ.get("TF_FP8_PROJ") == "1" else "bf16"),
        "data": "synthetic",
        "config": {"workload": f"fusion-encoder{' + RoI heads and losses (512 RoIs / image)' if args.with_heads else ''} train step (fwd+bwd+allreduce+clip+RAdam), B={args.batch}/GPU x [{NV} vis + {NL} txt] tokens, "
                               f"d={D}, heads={H}, ff={D * FF_MULT}, layers={L}, dropout {P_TOKEN}/{P_PATCH}, random right-padding"
                               f"{' (masked tokens dropped from the row-wise kernels)' if PACK_TOKENS else ' (masked tokens carried as dead rows)'}",
                   "global_batch": world * args.batch, "seq_len": S, "parallelism": f"dp{world}", "comm": args.comm if live else None},
        # utilisation on EXECUTED work: only the Nv + len_b real tokens of each sample (what the kernels compute with the masked tokens
        # dropped); the dense-S credit of BASELINE.md section 3 (padded tokens counted) beside it
        "block_mfma_util": round(train_flops_valid / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
        "block_tflops_per_gpu": round(train_flops_valid / (ms * 1e-3) / 1e12, 1),
        "block_mfma_util_dense_credit": round(train_flops_step / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
        "block_tflops_dense_credit": round(train_flops_step / (ms * 1e-3) / 1e12, 1),
        "mean_valid_tokens": round(sum(sum(vs) for vs in valid_S) / (len(valid_S) * args.batch), 1),
        "grad_allreduce_mb": round(trainer.reducer.bytes_per_step / 1e6, 1) if live else 0.0,
        "final_loss": round(final_loss, 5),
    }
    peak = PEAK_BF16_TFLOPS
    if args.precision == "fp32":
        # fp32 accuracy from three bf16 MFMA passes per contraction over hi + lo operand planes (fp32 accumulate): the matrix-pipe
        # ceiling for ALGORITHMIC FLOPs is a third of the bf16 peak; gfx950's f32-input MFMA peaks at 157.3 TFLOP/s
        peak = PEAK_BF16_TFLOPS / 3.0
        result["dtype_note"] = ("fp32-accuracy mode: hi + lo bf16 operand planes, 3 bf16 MFMA passes per product, fp32 accumulation, "
                                "epilogues and statistics; results within 1e-3 of the fp32 reference (tests/test_gpu_fp32_mode.py)")
        result["block_mfma_util"] = round(train_flops_valid / (ms * 1e-3) / 1e12 / peak, 4)
        result["block_mfma_util_dense_credit"] = round(train_flops_step / (ms * 1e-3) / 1e12 / peak, 4)
        result["peak_tflops_used"] = round(peak, 1)
        result["peak_fp32_mfma_tflops"] = 157.3
    if rank == 0 and rows is not None:
        # in-situ kernel table: the traced steps ran AFTER the timed region (two event records per launch would perturb it)
        total = sum(r["us_per_step"] for r in rows)
        for r in rows:
            # attention kernels: rate on valid (query, key) pairs next to the dense-S rate the tracer credits
            r["tflops_valid"] = round(r["tflops"] * attn_valid_ratio, 1) if r["kernel"].startswith("attn_") and r["tflops"] else None
        for r in rows:
            log(f"  {r['kernel']:34s} {r['avg_us']:9.1f} us x{r['launches_per_step']:6.1f} = {r['us_per_step']:8.1f} us/step  "
                f"{'' if r['tflops'] is None else str(r['tflops']) + ' TF/s':>12s}{'' if r.get('tflops_valid') is None else ' (' + str(r['tflops_valid']) + ' valid)':>16s} {'' if r['gbs'] is None else str(r['gbs']) + ' GB/s':>12s}"
                f"{'  [side stream]' if r['side_stream'] else ''}")
        log(f"  sum of kernel durations {total:.0f} us/step (streams overlap) vs measured step {ms * 1e3:.0f} us")
        dom = rows[0]                                            # the kernel symbol with the largest time per step
        traffic, traffic_stale = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # PMC-derived HBM bytes per launch, measured offline
        if os.path.exists(tpath):
            table = json.load(open(tpath))
            traffic_stale = table.get("_csrc_hash") != csrc_hash()  # measured on other kernel sources than the ones running now
            base = dom["kernel"].split("<")[0]                   # rocprofv3 prints template arguments the tracer's short names omit
            hit = table.get(dom["kernel"]) or next((v for k, v in table.items() if isinstance(v, dict) and k.split("<")[0] == base), {})
            traffic = hit.get("hbm_bytes_per_launch")
        if dom["tflops"] is not None:
            result["roofline"] = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"], "peak": round(peak, 1),
                                  "unit": "TFLOP/s", "frac": round(dom["tflops"] / peak, 4), "traffic": traffic, "traffic_stale": traffic_stale,
                                  "avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
                                  "algorithmic_flops_per_launch": dom["flops_per_launch"], "us_per_step": dom["us_per_step"],
                                  "algorithmic_bytes_per_launch": dom["bytes_per_launch"] or None,
                                  "traffic_ratio": round(traffic / dom["bytes_per_launch"], 3) if (traffic and dom["bytes_per_launch"]) else None,
                                  "measured": f"HIP event pair per launch on its own stream over {args.trace_steps} training steps after the timed region"}
        else:
            result["roofline"] = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": round((dom["gbs"] or 0.0) / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_stale": traffic_stale,
                                  "avg_launch_us": dom["avg_us"],
                                  "launches_per_step": dom["launches_per_step"], "algorithmic_bytes_per_launch": dom["bytes_per_launch"],
                                  "traffic_ratio": round(traffic / dom["bytes_per_launch"], 3) if (traffic and dom["bytes_per_launch"]) else None,
                                  "us_per_step": dom["us_per_step"],
                                  "measured": f"HIP event pair per launch on its own stream over {args.trace_steps} training steps after the timed region"}
        if dom["kernel"].startswith("wgrad_multi") and "roofline" in result:
            # the dominant kernel runs on the side stream and SHARES the chip with the backward chain in situ; its own rate too:
            result["roofline"]["alone"] = wgrad_alone(args.batch, device)
        result["kernels"] = [{k: r[k] for k in ("kernel", "avg_us", "launches_per_step", "us_per_step", "tflops", "tflops_valid", "gbs", "side_stream")} for r in rows]
        if args.isolated_census:
            census = kernel_census(args.batch, device)
            result["kernels_isolated"] = [{k: c[k] for k in ("kernel", "us", "launches_per_step", "us_per_step", "tflops", "gbs")} for c in census]
    # ---- the other BASELINE configurations, in the same run (every rank takes part: each step holds the gradient collectives) ----
    if not args.no_legs and not args.with_heads and args.precision == "bf16" and args.batch == 32:
        legs = {}
        if world == 1:
            specs = {
                "fp32": dict(precision="fp32", steps=6, warmup=2),                                    # configs[2]: run.precision 32
                "stress": dict(batch=8, d=1024, h=4, nv=784, nl=1024, steps=6, warmup=2),             # configs[3]: 64 samples / 8 GPUs
                "fp8": dict(fp8=True),                                                                # configs[4]
                "b4": dict(batch=4, steps=16, warmup=4),                                              # the reference's own per-GPU batch (32 / 8)
                "b4_dense": dict(batch=4, steps=16, warmup=4, pack=False),                            # ... on dense rows
                "b16": dict(batch=16),                                                                # configs[1]: Ego4Dv1, batch 16, one GPU
                # the widths the reference's two YAMLs really give the fusion block ("ref-true alt"): input_f_size = the RoI
                # head's representation size, 712 (Ego4Dv1: head dim 178, padded to 192 columns) and 896 (Ego4Dv2, fp32: head dim 224)
                "v1_d712": dict(batch=16, d=712),
                "v2_d896_fp32": dict(precision="fp32", d=896, steps=6, warmup=2),
                "dense_rows": dict(pack=False),                                                       # masked tokens carried as dead rows
                "no_padding": dict(padded=False),                                                     # 196 + 512 real tokens each
            }
            for name in [n.strip() for n in args.legs.split(",") if n.strip()]:
                if name in ("wrapper_b4", "wrapper_b4_real", "wrapper_b4_dp"):
                    legs[name] = run_wrapper_leg(device, rank, comm, batch=4, real=name.endswith("_real"), reducer=name.endswith("_dp"))
                    continue
                # the real module at the widths and precisions the reference's two YAMLs give it: Ego4Dv1 d = 712 at precision 16
                # (ego_nao_res50_ego4d.yml:73,118), Ego4Dv2 d = 896 at precision 32 (ego_nao_res50_ego4dv2.yml:79,124)
                if name in ("wrapper_b4_real_v1", "wrapper_b4_real_v2"):
                    v2 = name.endswith("v2")
                    legs[name] = run_wrapper_leg(device, rank, comm, batch=4, real=True, d=896 if v2 else 712, precision=32 if v2 else 16,
                                                 steps=6 if v2 else 10, warmup=3)
                    continue
                if name not in specs:
                    raise SystemExit(f"--legs: unknown leg {name!r} (known: {', '.join(specs)}, wrapper_b4, wrapper_b4_real, wrapper_b4_real_v1, wrapper_b4_real_v2, wrapper_b4_dp)")
                legs[name if name not in legs else f"{name}#{len(legs)}"] = run_leg(name, device, rank, comm, **specs[name])
        if legs:
            result["legs"] = legs
            if "fp32" in legs:            # BASELINE configs[2] (Ego4Dv2, run.precision: 32) names fp32: that leg's figure at top level too
                result["value_fp32"] = legs["fp32"]["samples_s"]
                result["ms_per_step_fp32"] = legs["fp32"]["ms_per_step"]
            if "no_padding" in legs:      # SURVEY.md 8(d)'s other variant: 196 + 512 REAL tokens per sample (nothing to drop)
                result["value_no_padding"] = legs["no_padding"]["samples_s"]
                result["ms_per_step_no_padding"] = legs["no_padding"]["ms_per_step"]
            if "fp8" in legs:             # BASELINE configs[4]: fp8 projections
                result["value_fp8"] = legs["fp8"]["samples_s"]
                result["ms_per_step_fp8"] = legs["fp8"]["ms_per_step"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    faulthandler.cancel_dump_traceback_later()
    if live:
        # the sections of the N > 1 line (live_sections: one order on every rank).  strong scaling and the real module under data
        # parallelism are legs: only with the default headline (no --no-legs / --with-heads / fp32 / another batch)
        with_legs = not args.no_legs and not args.with_heads and args.precision == "bf16" and args.batch == 32
        per = max(1, 32 // world)

        def strong_leg():
            out = run_leg("strong", device, rank, comm, batch=per, steps=16, warmup=4)
            return dict(out, batch_per_gpu=per)
        live_sections(result, world=world, rank=rank, log_fn=log,
                      checksum=lambda: trainer.flat.flat.double().sum(),
                      busbw=lambda: allreduce_busbw(trainer, comm),
                      strong=strong_leg if with_legs else None,
                      # the reference's REAL module under data parallelism, at its own per-GPU batch and FPN geometry: the four-level wrapper
                      # as one ragged grouped call, gradients exchanged unit by unit behind the backward (OrderedRangeReducer)
                      wrapper_dp=(lambda: run_wrapper_leg(device, rank, comm, batch=4, real=True, steps=8, warmup=3)) if with_legs else None,
                      probe=lambda: rccl_probe(trainer, comm, device, rank))
    if rehearse:
        result["rehearsal"] = "one-rank process group: the N > 1 code path against the real backend, without peers"
    if rank == 0:
        # the three figures that must be quoted TOGETHER (never the first alone): the headline drops masked tokens (random right padding,
        # mean valid tokens beside it), no_padding runs 708 real tokens per sample, fp32 is the B = 32 Ego4Dv2 config's own precision
        def _f(k, unit="samples/s"):
            return f"{result[k]:.0f} {unit}" if isinstance(result.get(k), (int, float)) else "n/a"
        log(f"[bench] SUMMARY bf16 padded+packed {result['value']:.0f} samples/s ({result['ms_per_step']} ms, mean valid tokens "
            f"{result.get('mean_valid_tokens')}) | no_padding {_f('value_no_padding')} | fp32 {_f('value_fp32')} | fp8 {_f('value_fp8')} | "
            f"block_mfma_util {result.get('block_mfma_util')} (executed work) | roofline.frac {result.get('roofline', {}).get('frac')}")
        print(json.dumps(result), flush=True)
    if live:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
