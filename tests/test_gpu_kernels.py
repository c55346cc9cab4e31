"""Per-kernel parity, HIP (through the C ABI) vs the CPU oracle / explicit fp32 math on the SAME bf16-rounded
inputs.  Tolerances are written next to each assert: bf16 outputs carry 2^-8 relative rounding."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from transfusion_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def bf(x):
    return x.to(torch.bfloat16)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def gelu_grad(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("M,N,K", [(300, 264, 128), (256, 256, 768), (1000, 2304, 768), (77, 8, 64),
                                   (5000, 776, 192), (4100, 2304, 64), (22656, 768, 768),    # these three take the large-tile kernel
                                   (9000, 2304, 128),    # 63 x 9 = 567 tiles of 144 x 256: the two-workgroups-per-CU form, ragged last row tile
                                   # row counts of packed batches: the large tile's height follows the grid (160 / 192 / 224-row tiles, ragged tails)
                                   (12010, 768, 192), (14003, 768, 128), (16519, 768, 192),
                                   # small row counts (the reference's per-GPU batch of 4): 64- and 96-row tiles of the 128-wide kernel
                                   (2070, 768, 128), (2833, 1536, 64), (3011, 768, 192),
                                   (2083, 768, 1536), (2083, 1536, 768), (1200, 2304, 768), (700, 768, 2304)])    # one workgroup per CU or fewer: the ring form of the 128-wide kernel
def test_gemm_epilogues(dev, guard, M, N, K):
    """(operands and outputs end at unmapped pages: ``guard``, tests/guard_alloc.py)"""
    from transfusion_amd import _lib as L, ops
    g = torch.Generator().manual_seed(M + N + K)
    A = guard(bf(torch.randn(M, K, generator=g)))
    W = guard(bf(torch.randn(N, K, generator=g) / math.sqrt(K)))
    bias = guard(torch.randn(N, generator=g))
    R = guard(bf(torch.randn(M, N, generator=g)))
    ref = A.float() @ W.float().t()
    # EPI_NONE / BIAS
    C = guard(torch.zeros(M, N, dtype=torch.bfloat16))
    ops.gemm(A, W, C, N, K, L.TF_EPI_NONE)
    assert rel(C, ref) < 4e-3            # bf16 output rounding only (fp32 accumulate)
    ops.gemm(A, W, C, N, K, L.TF_EPI_BIAS, bias=bias)
    assert rel(C, ref + bias) < 4e-3
    # ADD
    ops.gemm(A, W, C, N, K, L.TF_EPI_ADD, R=R)
    assert rel(C, bf(ref).float() + R.float()) < 4e-3
    # BIAS_GELU_DROP without and with dropout
    U = guard(torch.zeros(M, N, dtype=torch.bfloat16))
    H = guard(torch.zeros(M, N, dtype=torch.bfloat16))
    ops.gemm(A, W, U, N, K, L.TF_EPI_BIAS_GELU_DROP, bias=bias, C2=H)
    assert rel(U, ref + bias) < 4e-3
    assert rel(H, gelu(U.float())) < 4e-3
    p, seed, site = 0.15, 1234, 19
    drop = ops.drop_params(p, seed, site)
    ops.gemm(A, W, U, N, K, L.TF_EPI_BIAS_GELU_DROP, bias=bias, C2=H, drop=drop)
    keep = ops.dropout_mask(M * N, p, seed, site, dev).view(M, N).float()   # index = row * ldc2 + col, ldc2 == N here
    assert abs(keep.mean().item() - (1 - p)) < 0.02 or M * N < 5000
    assert rel(H, gelu(U.float()) * keep / (1 - p)) < 4e-3
    # BIAS_DROP_RES
    ops.gemm(A, W, C, N, K, L.TF_EPI_BIAS_DROP_RES, bias=bias, R=R, drop=drop)
    y = bf(ref + bias).float()
    assert rel(C, R.float() + y * keep / (1 - p)) < 4e-3
    # DGELU_DROP: C = acc * keep/(1-p) * gelu'(R)
    ops.gemm(A, W, C, N, K, L.TF_EPI_DGELU_DROP, R=R, drop=drop)
    assert rel(C, bf(ref).float() * keep / (1 - p) * gelu_grad(R.float())) < 5e-3


@pytest.mark.parametrize("M,N,K,grouped", [(200, 136, 72, False), (708 * 2, 768, 768, False), (1000, 384, 128, True),
                                           # one ragged 32-row step, tile tails in both directions, the benchmark's largest launch
                                           (33, 8, 8, False), (5000, 520, 264, False), (22656, 2304, 768, False)])
def test_wgrad(dev, guard, M, N, K, grouped):
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(7)
    dY = guard(bf(torch.randn(M, N, generator=g)))
    X = guard(bf(torch.randn(M, K, generator=g)))
    ref = dY.float().t() @ X.float()
    refb = dY.float().sum(0)
    if not grouped:
        dW = guard(torch.zeros(N, K))
        db = guard(torch.zeros(N))
        ops.wgrad(dY, N, X, K, dW, db)
        assert rel(dW, ref) < 1e-4          # fp32 accumulation of exact bf16 products, atomics order only
        assert rel(db, refb) < 1e-4
        # accumulation semantics: a second call adds
        ops.wgrad(dY, N, X, K, dW, db, m_chunk=128)
        assert rel(dW, 2 * ref) < 1e-4
    else:
        # padded row groups of 32 holding 18 valid rows each (head_dim 18 -> 32): N = 12 * 32
        rg, rgp = 18, 32
        n_src = (N // rgp) * rg
        dW = guard(torch.zeros(n_src, K))
        db = guard(torch.zeros(n_src))
        ops.wgrad(dY, N, X, K, dW, db, rg=rg, rgp=rgp, n_src=n_src)
        idx = torch.tensor([i for i in range(N) if i % rgp < rg])
        assert rel(dW, ref[idx]) < 1e-4
        assert rel(db, refb[idx]) < 1e-4
        # caller-sized M-splits select the 256x128-tile kernel (the one the encoder runtime's side stream uses)
        ops.wgrad(dY, N, X, K, dW, db, rg=rg, rgp=rgp, n_src=n_src, m_chunk=96)
        assert rel(dW, 2 * ref[idx]) < 1e-4
        assert rel(db, 2 * refb[idx]) < 1e-4


def _attn_ref(qkv, B, S, H, hd, hdp, key_mask, keep=None, p=0.0):
    """fp64 reference on the given (bf16-valued) packed qkv [B*S, 3*H*hdp]."""
    x = qkv.double().cpu().view(B, S, 3, H, hdp)[..., :hd]
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if key_mask is not None:
        s = s.masked_fill(key_mask.cpu().bool().view(B, 1, 1, S), float("-inf"))
    P = torch.softmax(s, dim=-1)
    Pd = P if keep is None else P * keep.double().cpu() / (1 - p)
    o = (Pd @ v).permute(0, 2, 1, 3)       # [B,S,H,hd]
    return o, P


@pytest.mark.parametrize("B,S,H,hd", [(2, 150, 2, 32), (1, 708, 4, 192), (2, 70, 4, 18), (1, 300, 2, 64), (1, 130, 1, 224),
                                      # tile edges and the other head widths: one key, exactly one / two tiles and one more, 8 heads of 8,
                                      # the padded 178 of Ego4Dv1, 160, 256 (one wave per SIMD forms), a batch of single-tile samples
                                      (3, 1, 2, 64), (2, 64, 1, 96), (1, 65, 2, 128), (2, 128, 3, 40), (1, 129, 8, 8), (1, 257, 4, 178),
                                      (2, 200, 1, 160), (1, 321, 2, 256), (5, 33, 4, 32)])
@pytest.mark.parametrize("ds", [False, True])
def test_attention_fwd_bwd(dev, guard, B, S, H, hd, ds):
    """``ds``: with TfAttnArgs.ds_work the backward is delta -> dK / dV (+ dS tiles) -> dQ = dS . K (S and dP computed once; head dims
    <= 192), without it the dQ kernel recomputes them.  Every operand sits flush against unmapped pages (``guard``, tests/guard_alloc.py):
    a kernel that reads or writes past a tensor's end faults here, in this test, every time -- [False-2-64-1-96] is the shape whose
    keep-bit image was over-read by attn_bwd_dkv16_kernel until round 6."""
    from transfusion_amd import _lib as L, ops
    hdp = (hd + 31) // 32 * 32
    g = torch.Generator().manual_seed(S + hd)
    ldq = (3 * H * hdp + 63) // 64 * 64
    qkv = torch.zeros(B * S, ldq)
    view = qkv[:, : 3 * H * hdp].view(B * S, 3, H, hdp)
    view[..., :hd] = torch.randn(B * S, 3, H, hd, generator=g)
    qkv = guard(bf(qkv))
    key_mask = torch.zeros(B, S, dtype=torch.uint8)
    key_mask[0, S - S // 3:] = 1
    key_mask = guard(key_mask)
    ldo = (H * hdp + 63) // 64 * 64
    out = guard(torch.zeros(B * S, ldo, dtype=torch.bfloat16))
    lse = guard(torch.zeros(B * H * S))
    for p in (0.0, 0.15):
        seed, site = 99, 17
        drop = ops.drop_params(p, seed, site)
        bits = guard(ops.attn_dropmask(B, H, S, p, seed, site, dev)) if p > 0 else None
        a = L.TfAttnArgs(qkv=L.ptr(qkv), ld_qkv=ldq, out=L.ptr(out), ld_out=ldo, lse=L.ptr(lse), key_mask=L.ptr(key_mask),
                         B=B, S=S, H=H, HDP=hdp, scale=1 / math.sqrt(hd), drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2],
                         drop_bits=L.ptr(bits))
        L.call("tf_attn_fwd", a, ops._stream())
        keep = None
        if p > 0:
            keep = ops.dropout_mask(B * H * S * S, p, seed, site, dev).view(B, H, S, S)
        p_eff = 1.0 - 1.0 / drop[2] if p > 0 else 0.0        # 16-bit threshold: the exact drop probability
        o_ref, P = _attn_ref(qkv[:, : 3 * H * hdp], B, S, H, hd, hdp, key_mask, keep, p_eff)
        o_hip = out[:, : H * hdp].float().cpu().view(B, S, H, hdp)
        assert rel(o_hip[..., :hd], o_ref) < 8e-3, f"fwd p={p}"          # bf16 P and bf16 output
        if hdp > hd:
            assert o_hip[..., hd:].abs().max().item() == 0.0            # pad columns stay exactly zero
        # LSE (log2 domain of the scaled scores)
        x = qkv[:, : 3 * H * hdp].double().cpu().view(B, S, 3, H, hdp)[..., :hd]
        s = (x[:, :, 0].permute(0, 2, 1, 3) @ x[:, :, 1].permute(0, 2, 3, 1)) / math.sqrt(hd)
        s = s.masked_fill(key_mask.cpu().bool().view(B, 1, 1, S), float("-inf"))
        lse_ref = torch.logsumexp(s, dim=-1) / math.log(2.0)
        assert (lse.cpu().view(B, H, S).double() - lse_ref).abs().max().item() < 2e-2

        # ---- backward ----
        do = torch.zeros(B * S, ldo)
        do[:, : H * hdp].view(B * S, H, hdp)[..., :hd] = torch.randn(B * S, H, hd, generator=g)
        do = guard(bf(do))
        dqkv = guard(torch.zeros(B * S, ldq, dtype=torch.bfloat16))
        delta = guard(torch.zeros(B * H * S))
        a.dout, a.ld_dout, a.dqkv, a.ld_dqkv, a.delta = L.ptr(do), ldo, L.ptr(dqkv), ldq, L.ptr(delta)
        dsw = guard(torch.full((L.load().tf_attn_ds_bytes(B, H, S) // 2,), float("nan"), dtype=torch.bfloat16)) if ds else None
        a.ds_work = L.ptr(dsw)                 # NaN-filled: tiles the dK / dV kernel does not write must not reach a stored dQ row
        L.call("tf_attn_bwd", a, ops._stream())
        xr = qkv[:, : 3 * H * hdp].double().cpu().view(B, S, 3, H, hdp)[..., :hd].clone().requires_grad_(True)
        q, k, v = xr[:, :, 0].permute(0, 2, 1, 3), xr[:, :, 1].permute(0, 2, 1, 3), xr[:, :, 2].permute(0, 2, 1, 3)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        s = s.masked_fill(key_mask.cpu().bool().view(B, 1, 1, S), float("-inf"))
        Pr = torch.softmax(s, -1)
        if keep is not None:
            Pr = Pr * keep.double().cpu() / (1 - p_eff)
        o = (Pr @ v).permute(0, 2, 1, 3)
        o.backward(do[:, : H * hdp].double().cpu().view(B, S, H, hdp)[..., :hd])
        g_hip = dqkv[:, : 3 * H * hdp].float().cpu().view(B, S, 3, H, hdp)
        for which, nm in enumerate("qkv"):
            got, want = g_hip[:, :, which, :, :hd].double(), xr.grad[:, :, which]
            # S = 1: softmax over one key has EXACTLY zero gradient w.r.t. q and k; the kernels form dS = P (dP - delta) with delta from
            # the bf16-rounded forward output, which leaves its rounding (2^-9 of |dO . v|) there -- an absolute floor for that case
            floor = (5e-3 if S == 1 else 1e-6) * want.numel() ** 0.5
            assert (got - want).norm().item() < 1.5e-2 * want.norm().item() + floor, f"d{nm} p={p}"
        if hdp > hd:
            assert g_hip[..., hd:].abs().max().item() == 0.0


@pytest.mark.parametrize("ds", [False, True])
@pytest.mark.parametrize("lens,H,hd", [([150, 1, 64, 129], 2, 64), ([708, 324, 500], 4, 192), ([33, 70], 4, 18), ([0, 200, 17], 1, 128),
                                       ([130, 260], 1, 224)])
def test_attention_packed_rows(dev, guard, lens, H, hd, ds):
    """TfAttnArgs.cu_rows: sample b owns rows cu[b] .. cu[b+1]-1 of the token-major tensors (ragged batch, no key mask); lse / delta /
    the dropout rows keep their dense [B, H, S] indexing.  Every sample against fp64 attention on its own rows, with dropout."""
    from transfusion_amd import _lib as L, ops
    B, S = len(lens), max(lens)
    hdp = (hd + 31) // 32 * 32
    M = sum(lens)
    g = torch.Generator().manual_seed(M + hd)
    ldq, ldo = (3 * H * hdp + 63) // 64 * 64, (H * hdp + 63) // 64 * 64
    qkv = torch.zeros(M + 5, ldq)                            # a few spare rows behind the last sample
    qkv[:M, : 3 * H * hdp].view(M, 3, H, hdp)[..., :hd] = torch.randn(M, 3, H, hd, generator=g)
    qkv = guard(bf(qkv))
    do = torch.zeros(M + 5, ldo)
    do[:M, : H * hdp].view(M, H, hdp)[..., :hd] = torch.randn(M, H, hd, generator=g)
    do = guard(bf(do))
    cu = guard(torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32))
    out = guard(torch.zeros(M + 5, ldo, dtype=torch.bfloat16))
    dqkv = guard(torch.zeros(M + 5, ldq, dtype=torch.bfloat16))
    lse, delta = guard(torch.zeros(B * H * S)), guard(torch.zeros(B * H * S))
    p, seed, site = 0.15, 7, 21
    drop = ops.drop_params(p, seed, site)
    bits = guard(ops.attn_dropmask(B, H, S, p, seed, site, dev))
    dsw = guard(torch.full((L.load().tf_attn_ds_bytes(B, H, S) // 2,), float("nan"), dtype=torch.bfloat16)) if ds else None
    a = L.TfAttnArgs(qkv=L.ptr(qkv), ld_qkv=ldq, out=L.ptr(out), ld_out=ldo, lse=L.ptr(lse), key_mask=0, B=B, S=S, H=H, HDP=hdp,
                     scale=1 / math.sqrt(hd), drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2], drop_bits=L.ptr(bits),
                     dout=L.ptr(do), ld_dout=ldo, dqkv=L.ptr(dqkv), ld_dqkv=ldq, delta=L.ptr(delta), cu_rows=L.ptr(cu), ds_work=L.ptr(dsw))
    L.call("tf_attn_fwd", a, ops._stream())
    L.call("tf_attn_bwd", a, ops._stream())
    torch.cuda.synchronize()
    keep = ops.dropout_mask(B * H * S * S, p, seed, site, dev).view(B, H, S, S).double().cpu()
    p_eff = 1.0 - 1.0 / drop[2]
    assert float(out[M:].float().abs().max()) == 0 and float(dqkv[M:].float().abs().max()) == 0      # nothing written past the last sample
    r0 = 0
    for b, n in enumerate(lens):
        if n == 0:
            continue
        x = qkv[r0:r0 + n, : 3 * H * hdp].double().cpu().view(n, 3, H, hdp)[..., :hd].clone().requires_grad_(True)
        q, k, v = x[:, 0].permute(1, 0, 2), x[:, 1].permute(1, 0, 2), x[:, 2].permute(1, 0, 2)          # [H, n, hd]
        sc = q @ k.transpose(-1, -2) / math.sqrt(hd)
        Pr = torch.softmax(sc, -1) * keep[b, :, :n, :n] / (1 - p_eff)
        o = (Pr @ v).permute(1, 0, 2)                                                                  # [n, H, hd]
        o.backward(do[r0:r0 + n, : H * hdp].double().cpu().view(n, H, hdp)[..., :hd])
        got_o = out[r0:r0 + n, : H * hdp].float().cpu().view(n, H, hdp)[..., :hd]
        assert rel(got_o, o.detach()) < 8e-3, (b, "fwd")
        lse_ref = torch.logsumexp(sc, -1).detach() / math.log(2.0)                                      # [H, n]
        assert (lse.cpu().view(B, H, S)[b, :, :n].double() - lse_ref).abs().max() < 2e-2
        got = dqkv[r0:r0 + n, : 3 * H * hdp].float().cpu().view(n, 3, H, hdp)[..., :hd].double()
        for which, nm in enumerate("qkv"):
            want = x.grad[:, which]
            floor = (5e-3 if n == 1 else 1e-6) * want.numel() ** 0.5
            assert (got[:, which] - want).norm().item() < 1.5e-2 * want.norm().item() + floor, (b, nm)
        r0 += n


def test_attention_online_softmax_rescale(dev, guard):
    """Forces the running-max update late in the key loop (a spike in the LAST key tile), which bounded
    random data never exercises."""
    from transfusion_amd import _lib as L, ops
    B, S, H, hd = 1, 200, 1, 64
    g = torch.Generator().manual_seed(5)
    q = torch.randn(S, hd, generator=g)
    k = torch.randn(S, hd, generator=g)
    v = torch.randn(S, hd, generator=g)
    k[S - 3] = 6.0 * q[10]          # a huge score for query 10 in the last tile
    qkv = guard(bf(torch.cat([q, k, v], dim=1)))
    out = guard(torch.zeros(S, hd, dtype=torch.bfloat16))
    lse = guard(torch.zeros(S))
    a = L.TfAttnArgs(qkv=L.ptr(qkv), ld_qkv=3 * hd, out=L.ptr(out), ld_out=hd, lse=L.ptr(lse), key_mask=0, B=B, S=S, H=H, HDP=hd,
                     scale=1 / math.sqrt(hd), drop_thr=0, drop_key=0, drop_scale=1.0)
    L.call("tf_attn_fwd", a, ops._stream())
    o_ref, _ = _attn_ref(qkv, 1, S, 1, hd, hd, None)
    assert rel(out.float().cpu().view(1, S, 1, hd), o_ref) < 8e-3
    assert (out.float().cpu()[10] - o_ref[0, 10, 0].float()).abs().max() < 5e-2


@pytest.mark.parametrize("rows,d,ld", [(37, 64, 64), (500, 768, 768), (64, 72, 128), (10, 712, 768),
                                       (1, 8, 8), (22656, 768, 768), (300, 1024, 1024), (129, 896, 896), (77, 1536, 1536)])
def test_layernorm_fwd_bwd(dev, guard, rows, d, ld):
    from transfusion_amd import _lib as L, ops
    from oracle import fusion_oracle as O
    g = torch.Generator().manual_seed(rows)
    x = torch.zeros(rows, ld)
    x[:, :d] = torch.randn(rows, d, generator=g) * 2 + 0.5
    xb = guard(bf(x))
    gamma = guard(1 + 0.1 * torch.randn(d, generator=g))
    beta = guard(0.1 * torch.randn(d, generator=g))
    y = guard(torch.full((rows, ld), 7.0, dtype=torch.bfloat16))
    mean = guard(torch.zeros(rows))
    rstd = guard(torch.zeros(rows))
    a = L.TfLnArgs(x=L.ptr(xb), ldx=ld, y=L.ptr(y), ldy=ld, y_is_f32=0, gamma=L.ptr(gamma), beta=L.ptr(beta), mean=L.ptr(mean),
                   rstd=L.ptr(rstd), rows=rows, d=d, rows_per_group=rows, x_group_stride=rows, y_group_stride=rows, eps=1e-5)
    L.call("tf_layernorm_fwd", a, ops._stream())
    xr = xb.float().cpu()[:, :d].clone().requires_grad_(True)
    gr, br = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
    yr = O.layer_norm(xr, gr, br)
    assert rel(y[:, :d], yr.detach()) < 4e-3
    if ld > d:
        assert y[:, d:].float().abs().max().item() == 0.0
    if rows >= 8192:
        # the launch above took the resident-grid form (>= 8 192 rows), a short launch takes the one-row-per-wave form: the same bits
        # (the encoder's batch-independence property rests on it: test_gpu_encoder.py::test_full_size_properties)
        y2, m2, r2 = guard(torch.zeros(1000, ld, dtype=torch.bfloat16)), guard(torch.zeros(1000)), guard(torch.zeros(1000))
        a2 = L.TfLnArgs(x=L.ptr(xb), ldx=ld, y=L.ptr(y2), ldy=ld, y_is_f32=0, gamma=L.ptr(gamma), beta=L.ptr(beta), mean=L.ptr(m2),
                        rstd=L.ptr(r2), rows=1000, d=d, rows_per_group=1000, x_group_stride=1000, y_group_stride=1000, eps=1e-5)
        L.call("tf_layernorm_fwd", a2, ops._stream())
        assert torch.equal(y2, y[:1000]) and torch.equal(m2, mean[:1000]) and torch.equal(r2, rstd[:1000])
    dy = guard(bf(torch.randn(rows, ld, generator=g)))
    dx = guard(torch.zeros(rows, ld, dtype=torch.bfloat16))
    dg = guard(torch.zeros(d))
    db = guard(torch.zeros(d))
    a.dy, a.lddy, a.dy_is_f32, a.dx, a.lddx, a.dgamma, a.dbeta = L.ptr(dy), ld, 0, L.ptr(dx), ld, L.ptr(dg), L.ptr(db)
    L.call("tf_layernorm_bwd", a, ops._stream())
    yr.backward(dy.float().cpu()[:, :d])
    assert rel(dx[:, :d], xr.grad) < 6e-3
    assert rel(dg, gr.grad) < 1e-3 and rel(db, br.grad) < 1e-3


@pytest.mark.parametrize("rows,d,ld,groups", [(2 * 4500, 768, 768, 2), (12000, 768, 768, (9000, 3000)), (9000, 712, 768, (8000, 600, 400)),
                                              (1450, 768, 768, (700, 300, 450))])
def test_layernorm_fwd_parameter_groups(dev, guard, rows, d, ld, groups):
    """LayerNorm forward with parameter groups (equal and ragged row ranges, gamma / beta p_gstride bytes apart) in both kernel forms
    (>= 8 192 rows: the resident-grid form): y, mean, rstd against fp64, pad columns zeroed; operands at guard pages."""
    from transfusion_amd import _lib as L, ops
    g = torch.Generator().manual_seed(rows + d)
    ranges = None
    if isinstance(groups, tuple):
        ranges, groups = groups, len(groups)
        assert sum(ranges) == rows
    bounds = [0]
    for k in range(groups):
        bounds.append(bounds[-1] + (ranges[k] if ranges else rows // groups))
    x = torch.zeros(rows, ld)
    x[:, :d] = torch.randn(rows, d, generator=g) * 1.7 + 0.4
    xb = guard(bf(x))
    gstride = ((d + 63) // 64) * 64
    gamma = guard((1 + 0.1 * torch.randn(groups, gstride, generator=g)).contiguous())
    beta = guard((0.1 * torch.randn(groups, gstride, generator=g)).contiguous())
    y = guard(torch.full((rows, ld), 5.0, dtype=torch.bfloat16))
    mean, rstd = guard(torch.zeros(rows)), guard(torch.zeros(rows))
    a = L.TfLnArgs(x=L.ptr(xb), ldx=ld, y=L.ptr(y), ldy=ld, y_is_f32=0, gamma=L.ptr(gamma), beta=L.ptr(beta), mean=L.ptr(mean), rstd=L.ptr(rstd),
                   rows=rows, d=d, rows_per_group=rows, x_group_stride=rows, y_group_stride=rows, eps=1e-5, pgroups=groups, p_gstride=gstride * 4)
    if ranges:
        for k, n in enumerate(ranges):
            a.group_rows[k] = n
    L.call("tf_layernorm_fwd", a, ops._stream())
    torch.cuda.synchronize()
    xd = xb.double().cpu()[:, :d]
    mu, var = xd.mean(1), xd.var(1, unbiased=False)
    gam = torch.cat([gamma.double().cpu()[k:k + 1, :d].expand(bounds[k + 1] - bounds[k], d) for k in range(groups)])
    bet = torch.cat([beta.double().cpu()[k:k + 1, :d].expand(bounds[k + 1] - bounds[k], d) for k in range(groups)])
    ref = (xd - mu[:, None]) / torch.sqrt(var[:, None] + 1e-5) * gam + bet
    assert rel(y[:, :d], ref) < 4e-3
    assert rel(mean, mu) < 1e-5 and rel(rstd, 1.0 / torch.sqrt(var + 1e-5)) < 1e-5
    if ld > d:
        assert y[:, d:].float().abs().max().item() == 0.0


@pytest.mark.parametrize("rows,d,ld,groups", [(2 * 4500, 768, 768, 2), (16640, 768, 768, 1), (3 * 333, 712, 768, 3), (130, 72, 128, 1),
                                              (2 * 257, 1024, 1024, 2), (1450, 768, 768, (700, 300, 450)), (12000, 768, 768, (9000, 3000))])
def test_layernorm_bwd_dropout_copy_and_parameter_groups(dev, guard, rows, d, ld, groups):
    """The per-layer form of the LayerNorm backward (ln_bwd2_kernel, both of its shipped configurations: >= 8192 rows per group and
    fewer): dx, the dropout-masked copy dx_drop = dx * keep / (1 - p) with the mask tf_dropout_mask replays, zeroed pad columns, and
    dgamma / dbeta per parameter group (equal row ranges, parameters p_gstride bytes apart) -- against fp64 torch, operands at guard pages."""
    from transfusion_amd import _lib as L, ops
    g = torch.Generator().manual_seed(rows + d)
    ranges = None
    if isinstance(groups, tuple):                           # ragged parameter groups (TfLnArgs.group_rows): the wrapper's unequal levels
        ranges, groups = groups, len(groups)
        assert sum(ranges) == rows
    bounds = [0]
    for k in range(groups):
        bounds.append(bounds[-1] + (ranges[k] if ranges else rows // groups))
    x = torch.zeros(rows, ld)
    x[:, :d] = torch.randn(rows, d, generator=g) * 1.5 - 0.3
    xb = guard(bf(x))
    dy = guard(bf(torch.randn(rows, ld, generator=g)))
    gstride = ((d + 63) // 64) * 64                         # floats between the groups' parameter vectors
    gamma = guard((1 + 0.1 * torch.randn(groups, gstride, generator=g)).contiguous())
    xd = xb.double().cpu()[:, :d]
    mean = xd.mean(1)
    rstd = 1.0 / torch.sqrt(xd.var(1, unbiased=False) + 1e-5)
    mean_d, rstd_d = guard(mean.float()), guard(rstd.float())
    dx = guard(torch.full((rows, ld), 3.0, dtype=torch.bfloat16))
    dxd = guard(torch.full((rows, ld), 3.0, dtype=torch.bfloat16))
    dg = guard(torch.zeros(groups, gstride))
    db = guard(torch.zeros(groups, gstride))
    p, seed, site = 0.15, 11, 5
    thr, key, scale = ops.drop_params(p, seed, site)
    a = L.TfLnArgs(x=L.ptr(xb), ldx=ld, gamma=L.ptr(gamma), mean=L.ptr(mean_d), rstd=L.ptr(rstd_d), rows=rows, d=d, rows_per_group=rows,
                   x_group_stride=rows, y_group_stride=rows, eps=1e-5, dy=L.ptr(dy), lddy=ld, dy_is_f32=0, dx=L.ptr(dx), lddx=ld,
                   dx_drop=L.ptr(dxd), lddxd=ld, drop_thr=thr, drop_key=key, drop_scale=scale, drop_ld=ld, dgamma=L.ptr(dg), dbeta=L.ptr(db),
                   pgroups=groups, p_gstride=gstride * 4)
    if ranges:
        for k, n in enumerate(ranges):
            a.group_rows[k] = n
    L.call("tf_layernorm_bwd", a, ops._stream())
    torch.cuda.synchronize()
    gam = torch.cat([gamma.double().cpu()[k:k + 1, :d].expand(bounds[k + 1] - bounds[k], d) for k in range(groups)])
    dyd = dy.double().cpu()[:, :d]
    xh = (xd - mean[:, None]) * rstd[:, None]
    gg = dyd * gam
    ref = rstd[:, None] * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True))
    assert rel(dx[:, :d], ref) < 6e-3
    keep = ops.dropout_mask(rows * ld, p, seed, site, dev).view(rows, ld)[:, :d].cpu()
    got = dxd.float().cpu()[:, :d]
    assert (got[~keep.bool()] == 0).all() and 0.1 < 1.0 - keep.float().mean().item() < 0.2
    assert rel(got, ref * keep.double() * scale) < 6e-3
    if ld > d:
        assert dx[:, d:].float().abs().max().item() == 0.0 and dxd[:, d:].float().abs().max().item() == 0.0
    for k in range(groups):
        sl = slice(bounds[k], bounds[k + 1])
        assert rel(dg[k, :d], (dyd[sl] * xh[sl]).sum(0)) < 1e-3 and rel(db[k, :d], dyd[sl].sum(0)) < 1e-3
        if gstride > d:
            assert dg[k, d:].abs().max().item() == 0.0 and db[k, d:].abs().max().item() == 0.0


def test_sumsq_set_overwrites_and_equals_the_accumulating_form(dev, guard):
    """tf_sumsq_set (ABI v10): out = sum(x^2) whatever out held, the same bits as tf_sumsq into a zeroed scalar; an odd length (scalar
    tail) and the flat gradient buffer's size; operand at a guard page."""
    from transfusion_amd import _lib as L, ops
    lib = L.load()
    for n in (18_912_000, 1_000_003, 5):
        x = guard(torch.randn(n, generator=torch.Generator().manual_seed(n)))
        a, b = torch.full((1,), 123.0, device=dev), torch.zeros(1, device=dev)
        L.check(lib.tf_sumsq_set(L.ptr(x), n, L.ptr(a), ops._stream()), "tf_sumsq_set")
        L.check(lib.tf_sumsq(L.ptr(x), n, L.ptr(b), ops._stream()), "tf_sumsq")
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        assert abs(a.item() / x.double().pow(2).sum().item() - 1.0) < 1e-5
        L.check(lib.tf_sumsq_set(L.ptr(x), n, L.ptr(a), ops._stream()), "tf_sumsq_set")      # again: overwritten, not doubled
        torch.cuda.synchronize()
        assert torch.equal(a, b)


def test_pack_and_patch_permutations(dev):
    from transfusion_amd import ops
    from oracle import fusion_oracle as O
    g = torch.Generator().manual_seed(3)
    w = torch.randn(72, 40, generator=g).to(dev)
    sh, sht = ops.pack_weight(w, 72, 64, 64, 128)
    assert torch.equal(sh[:, :40], bf(w)) and sh[:, 40:].abs().max() == 0
    assert torch.equal(sht[:40, :72], bf(w).t()) and sht[40:].abs().max() == 0 and sht[:, 72:].abs().max() == 0
    # im2col rows == the oracle's patch extraction; col2im == regroup fold with a zero border
    feat = torch.randn(2, 8, 9, 10, generator=g).to(dev)
    rows = ops.patchify(feat, 2, 2)
    B, C, H, W, p = 2, 8, 9, 10, 2
    x = feat.cpu()[:, :, : H // p * p, : W // p * p].reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, -1, C * p * p)
    assert torch.equal(rows.float().cpu(), bf(x).float())
    img = ops.regroup(rows, H, W, p, p, out_dtype=torch.float32)
    expect = torch.zeros(B, C, H, W)
    expect[:, :, : H // p * p, : W // p * p] = bf(feat.cpu()[:, :, : H // p * p, : W // p * p]).float()
    assert torch.equal(img.cpu(), expect)


@pytest.mark.parametrize("B,C,H,W,p", [(2, 256, 56, 56, 4), (3, 40, 15, 17, 2), (2, 2048, 14, 14, 1), (1, 12, 7, 300, 3), (2, 520, 29, 28, 2),
                                       # the reference's FPN levels at batch 4 (plane-tile kernels: 7 of 28 patch rows / whole planes per workgroup)
                                       (4, 256, 112, 112, 4), (4, 512, 56, 56, 4), (4, 1024, 28, 28, 2), (4, 2048, 14, 14, 1),
                                       # plane-tile plan edges: one patch row per workgroup, odd patch-row counts, 16 / 64 channels exactly
                                       (1, 8, 4, 2000, 4), (2, 16, 6, 10, 2), (1, 64, 3, 4, 1), (3, 128, 10, 6, 1), (2, 32, 18, 22, 2)])
def test_patch_permutations_tiled_kernels(dev, guard, B, C, H, W, p):
    """K1 / K9 permutations at the wrapper's level shapes and at ragged ones (channel tails, maps that are not a multiple of the patch,
    several channel chunks per patch row): im2col rows and the fold must be EXACT permutations of the bf16-rounded input.  (The feature
    map ends at unmapped pages: ``guard``.)"""
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + C + H + W + p)
    feat = guard(torch.randn(B, C, H, W, generator=g))
    Hp, Wp = H // p, W // p
    rows = ops.patchify(feat, p, p)
    x = feat.cpu()[:, :, : Hp * p, : Wp * p].reshape(B, C, Hp, p, Wp, p).permute(0, 2, 4, 1, 3, 5).reshape(B, -1, C * p * p)
    assert rows.shape[-1] >= C * p * p
    assert torch.equal(rows.float().cpu()[..., : C * p * p], bf(x).float())
    assert rows.float().cpu()[..., C * p * p:].abs().sum() == 0                 # row padding is zero
    for out_dtype in (torch.float32, torch.bfloat16):
        img = ops.regroup(rows, H, W, p, p, out_dtype=out_dtype)
        expect = torch.zeros(B, C, H, W)
        expect[:, :, : Hp * p, : Wp * p] = bf(feat.cpu()[:, :, : Hp * p, : Wp * p]).float()
        assert torch.equal(img.float().cpu(), expect)
    # bf16 feature map in (regroup's backward path: d(feat) -> d(token rows))
    rows16 = ops.patchify(feat.to(torch.bfloat16), p, p)
    assert torch.equal(rows16.float().cpu(), rows.float().cpu())


def test_linear_fn_fwd_bwd(dev):
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 50, 40, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(72, 40, generator=g) / 6).to(dev).requires_grad_(True)
    b = torch.randn(72, generator=g).to(dev).requires_grad_(True)
    y = ops.linear(x, w, b)
    gy = torch.randn(3, 50, 72, generator=g).to(dev)
    (y.float() * gy).sum().backward()
    xr = bf(x.detach()).float().cpu().requires_grad_(True)
    wr = bf(w.detach()).float().cpu().requires_grad_(True)
    br = b.detach().cpu().clone().requires_grad_(True)
    yr = xr @ wr.t() + br
    (yr * gy.cpu()).sum().backward()
    assert rel(y, yr.detach()) < 4e-3
    assert rel(x.grad, xr.grad) < 1e-2 and rel(w.grad, wr.grad) < 1e-2 and rel(b.grad, br.grad) < 1e-2


@pytest.mark.parametrize("name", ["radam_wd", "radam_groups", "radam_sgd"])
def test_radam_matches_reference_optimizer(dev, golden_dir, name):
    """FusedRAdam (tf_radam_step) against the parameters the reference's own RAdam class produced step by step
    (tests/golden/radam_*.npz: weight decay, two groups with different lr, degenerated_to_sgd), plus the optimiser protocol the
    reference relies on: per-group lr, state_dict round trip mid-run, parameter versions bumped by the raw-pointer update."""
    import os
    import numpy as np
    from cases import RADAM_CASES, make_radam_case
    from transfusion_amd.optim import FusedRAdam
    cfg = RADAM_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    params, grads = make_radam_case(cfg)

    def make():
        tp = [[torch.nn.Parameter(torch.from_numpy(p.copy()).to(dev)) for p in grp] for grp in params]
        groups = [dict({"params": ps}, **({"lr": gr["lr"]} if "lr" in gr else {})) for gr, ps in zip(cfg["groups"], tp)]
        return tp, FusedRAdam(groups, lr=cfg["lr"], betas=cfg["betas"], eps=cfg["eps"], weight_decay=cfg["weight_decay"],
                              degenerated_to_sgd=cfg["degenerated_to_sgd"])

    tp, opt = make()
    assert isinstance(opt, torch.optim.Optimizer)
    half = cfg["steps"] // 2
    for step in range(cfg["steps"]):
        if step == half:                       # checkpoint / resume: moments AND step counts travel in the state dict
            sd = opt.state_dict()
            old = tp
            tp, opt = make()
            for grp_new, grp_old in zip(tp, old):
                for pn, po in zip(grp_new, grp_old):
                    pn.data.copy_(po.data)
            opt.load_state_dict(sd)
        for gi, ps in enumerate(tp):
            for ti, p in enumerate(ps):
                p.grad = torch.from_numpy(grads[step][gi][ti].copy()).to(dev)
        v0 = tp[0][0]._version
        opt.step()
        assert tp[0][0]._version > v0
        for gi, ps in enumerate(tp):
            for ti, p in enumerate(ps):
                ref = torch.from_numpy(g[f"p/{step}/{gi}/{ti}"])
                assert (p.detach().cpu() - ref).abs().max().item() <= 2e-6 * (1 + ref.abs().max().item()), (step, gi, ti)
    for gi, ps in enumerate(tp):
        for ti, p in enumerate(ps):
            # moments: the device contracts v*b2 + (1-b2)*g*g into FMAs (one rounding fewer per step than the reference's separate ops)
            assert torch.allclose(opt.state[p]["exp_avg"].cpu(), torch.from_numpy(g[f"exp_avg/{gi}/{ti}"]), rtol=1e-4, atol=1e-7)
            assert torch.allclose(opt.state[p]["exp_avg_sq"].cpu(), torch.from_numpy(g[f"exp_avg_sq/{gi}/{ti}"]), rtol=1e-4, atol=1e-8)
    # a parameter without a gradient is skipped entirely (no moments, no weight decay), as in the reference
    tp, opt = make()
    before = tp[0][0].detach().clone()
    opt.step()
    assert torch.equal(tp[0][0].detach(), before) and len(opt.state) == 0


def _pack_bits(m):
    """bool [S, S] -> int64 [S, ceil(S/64)] little-endian bit words (the TfAttnArgs.block_bits layout)."""
    S = m.shape[0]
    SW = (S + 63) // 64
    full = torch.zeros(S, SW * 64, dtype=torch.bool)
    full[:, :S] = m
    return (full.view(S, SW, 64).to(torch.int64) << torch.arange(64, dtype=torch.int64)).sum(-1).contiguous()


@pytest.mark.parametrize("B,S,H,hd,p", [(2, 150, 2, 64, 0.0), (1, 333, 3, 18, 0.15)])
def test_attention_block_mask(dev, guard, B, S, H, hd, p):
    """Boolean attn_mask (TfAttnArgs.block_bits, the reference's local_k visual mask mechanism): random blocked pairs on the
    first two thirds of the sequence (every query keeps its own key), together with key padding and dropout."""
    from transfusion_amd import _lib as L, ops
    hdp = (hd + 31) // 32 * 32
    g = torch.Generator().manual_seed(3 * S + hd)
    ldq = (3 * H * hdp + 63) // 64 * 64
    qkv = torch.zeros(B * S, ldq)
    qkv[:, : 3 * H * hdp].view(B * S, 3, H, hdp)[..., :hd] = torch.randn(B * S, 3, H, hd, generator=g)
    qkv = guard(bf(qkv))
    nv = 2 * S // 3
    blk = torch.zeros(S, S, dtype=torch.bool)
    blk[:nv, :nv] = torch.rand(nv, nv, generator=g) < 0.6
    blk[torch.arange(S), torch.arange(S)] = False
    bits = guard(_pack_bits(blk))
    key_mask = torch.zeros(B, S, dtype=torch.uint8)
    key_mask[0, S - S // 5:] = 1
    key_mask = guard(key_mask)
    ldo = (H * hdp + 63) // 64 * 64
    out = guard(torch.zeros(B * S, ldo, dtype=torch.bfloat16))
    lse = guard(torch.zeros(B * H * S))
    seed, site = 5, 9
    drop = ops.drop_params(p, seed, site)
    dbits = guard(ops.attn_dropmask(B, H, S, p, seed, site, dev)) if p > 0 else None
    do = torch.zeros(B * S, ldo)
    do[:, : H * hdp].view(B * S, H, hdp)[..., :hd] = torch.randn(B * S, H, hd, generator=g)
    do = guard(bf(do))
    dqkv = guard(torch.zeros(B * S, ldq, dtype=torch.bfloat16))
    delta = guard(torch.zeros(B * H * S))
    a = L.TfAttnArgs(qkv=L.ptr(qkv), ld_qkv=ldq, out=L.ptr(out), ld_out=ldo, lse=L.ptr(lse), key_mask=L.ptr(key_mask), B=B, S=S, H=H,
                     HDP=hdp, scale=1 / math.sqrt(hd), drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2], drop_bits=L.ptr(dbits),
                     block_bits=L.ptr(bits), dout=L.ptr(do), ld_dout=ldo, dqkv=L.ptr(dqkv), ld_dqkv=ldq, delta=L.ptr(delta))
    L.call("tf_attn_fwd", a, ops._stream())
    L.call("tf_attn_bwd", a, ops._stream())
    keep = ops.dropout_mask(B * H * S * S, p, seed, site, dev).view(B, H, S, S).double().cpu() if p > 0 else None
    p_eff = 1.0 - 1.0 / drop[2] if p > 0 else 0.0
    xr = qkv[:, : 3 * H * hdp].double().cpu().view(B, S, 3, H, hdp)[..., :hd].clone().requires_grad_(True)
    q, k, v = xr[:, :, 0].permute(0, 2, 1, 3), xr[:, :, 1].permute(0, 2, 1, 3), xr[:, :, 2].permute(0, 2, 1, 3)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(key_mask.cpu().bool().view(B, 1, 1, S), float("-inf")).masked_fill(blk.view(1, 1, S, S), float("-inf"))
    Pr = torch.softmax(s, -1)
    if keep is not None:
        Pr = Pr * keep / (1 - p_eff)
    o = (Pr @ v).permute(0, 2, 1, 3)
    o.backward(do[:, : H * hdp].double().cpu().view(B, S, H, hdp)[..., :hd])
    assert rel(out[:, : H * hdp].float().cpu().view(B, S, H, hdp)[..., :hd], o.detach()) < 8e-3
    g_hip = dqkv[:, : 3 * H * hdp].float().cpu().view(B, S, 3, H, hdp)
    for which, nm in enumerate("qkv"):
        assert rel(g_hip[:, :, which, :, :hd], xr.grad[:, :, which]) < 1.5e-2, f"d{nm}"


def _local_block(gh, gw, k, Nl):
    """The reference's local_k mask (utils.py:14-30) on a gh x gw token grid, padded with Nl language tokens (nobody is blocked from
    them, they block nobody): bool [S, S], True = blocked."""
    Nv = gh * gw
    r, c = torch.arange(Nv) // gw, torch.arange(Nv) % gw
    near = ((r[:, None] - r[None, :]).abs() <= k) & ((c[:, None] - c[None, :]).abs() <= k)
    blk = torch.zeros(Nv + Nl, Nv + Nl, dtype=torch.bool)
    blk[:Nv, :Nv] = ~near
    return blk


@pytest.mark.parametrize("gh,gw,k,Nl,hd,packed", [(28, 28, 2, 100, 64, False), (28, 28, 4, 512, 192, True), (14, 14, 1, 70, 32, False), (40, 45, 1, 30, 64, False)])
def test_attention_block_sparse_tiles(dev, guard, gh, gw, k, Nl, hd, packed):
    """TfAttnArgs.block_skip_q / block_skip_k (the local_k mask as block-sparse tiles): the maps tf_attn_block_skip derives
    from the bit matrix are exactly the tiles a torch reduction of the mask finds fully blocked, a local mask on a 28 x 28 grid leaves a
    visual query block with less than half of its visual key tiles, and forward and backward with the maps give the SAME BITS as
    without them (a skipped tile holds only exact zeros) -- output, LSE, dQ, dK, dV; the last case has more than 64 query tiles of 32
    (no dK / dV map: zero words)."""
    from transfusion_amd import _lib as L, ops
    lib = L.load()
    B, H = 2, 2
    Nv, S = gh * gw, gh * gw + Nl
    blk = _local_block(gh, gw, k, Nl)
    bits = guard(_pack_bits(blk))
    nb, SW, nqt = (S + 127) // 128, (S + 63) // 64, (S + 31) // 32
    skq = guard(torch.full((nb,), -1, dtype=torch.int64))
    skk = guard(torch.full((nb,), -1, dtype=torch.int64))
    L.check(lib.tf_attn_block_skip(L.ptr(bits), S, L.ptr(skq), L.ptr(skk), ops._stream()), "tf_attn_block_skip")
    # the maps against a torch reduction of the mask (keys / rows past S count as blocked for the tile test only when the WHOLE word is:
    # the bit matrix pads with zeros, so a ragged last tile is never skippable)
    full = torch.zeros(nb * 128, SW * 64, dtype=torch.bool)
    full[:S, :S] = blk
    rows_exist = torch.arange(nb * 128) < S
    want_q = []
    for qb in range(nb):
        rsel = rows_exist[qb * 128:(qb + 1) * 128]
        t_all = full[qb * 128:(qb + 1) * 128][rsel].view(-1, SW, 64).all(-1).all(0)            # [SW]
        want_q.append(sum(1 << t for t in range(SW) if bool(t_all[t])) if SW <= 64 else 0)
    got_q = [int(v) & (2 ** 64 - 1) for v in skq.cpu().tolist()]
    assert got_q == want_q
    want_k = []
    for kb in range(nb):
        cols = full[:, kb * 128: min((kb + 1) * 128, SW * 64)]
        word = 0
        for j in range(nqt):
            rsel = slice(j * 32, min((j + 1) * 32, S))
            if bool(cols[rsel].all()):
                word |= 1 << j
        want_k.append(word if nqt <= 64 else 0)
    got_k = [int(v) & (2 ** 64 - 1) for v in skk.cpu().tolist()]
    assert got_k == want_k
    if gh == 28:
        vis_q = [bin(w).count("1") for w in want_q[: Nv // 128]]
        assert min(vis_q) >= (Nv // 64) // 2 - 1, vis_q            # most visual key tiles of a visual query block are gone
    # ---- same bits with and without the maps ----
    hdp = (hd + 31) // 32 * 32
    g = torch.Generator().manual_seed(S + hd)
    ldq = (3 * H * hdp + 63) // 64 * 64
    ldo = (H * hdp + 63) // 64 * 64
    lens = [S, S - 37] if packed else [S, S]
    rows = sum(lens) if packed else B * S
    qkv = torch.zeros(rows, ldq)
    qkv[:, : 3 * H * hdp].view(rows, 3, H, hdp)[..., :hd] = torch.randn(rows, 3, H, hd, generator=g)
    qkv = guard(bf(qkv))
    do = torch.zeros(rows, ldo)
    do[:, : H * hdp].view(rows, H, hdp)[..., :hd] = torch.randn(rows, H, hd, generator=g)
    do = guard(bf(do))
    cu = guard(torch.tensor([0, lens[0], lens[0] + lens[1]], dtype=torch.int32)) if packed else None
    dsw = guard(torch.zeros(lib.tf_attn_ds_bytes(B, H, S), dtype=torch.uint8))
    res = {}
    for use in (False, True):
        out = guard(torch.zeros(rows, ldo, dtype=torch.bfloat16))
        lse = guard(torch.zeros(B * H * S))
        dqkv = guard(torch.zeros(rows, ldq, dtype=torch.bfloat16))
        delta = guard(torch.zeros(B * H * S))
        dsw.fill_(0x7F)                                         # bf16 NaN patterns: a dS tile that is read without having been written shows
        a = L.TfAttnArgs(qkv=L.ptr(qkv), ld_qkv=ldq, out=L.ptr(out), ld_out=ldo, lse=L.ptr(lse), key_mask=0, B=B, S=S, H=H, HDP=hdp,
                         scale=1 / math.sqrt(hd), drop_thr=0, drop_key=0, drop_scale=1.0, block_bits=L.ptr(bits), dout=L.ptr(do), ld_dout=ldo,
                         dqkv=L.ptr(dqkv), ld_dqkv=ldq, delta=L.ptr(delta), ds_work=L.ptr(dsw), cu_rows=L.ptr(cu),
                         block_skip_q=L.ptr(skq) if use else 0, block_skip_k=L.ptr(skk) if use else 0)
        L.call("tf_attn_fwd", a, ops._stream())
        L.call("tf_attn_bwd", a, ops._stream())
        torch.cuda.synchronize()
        res[use] = (out.clone(), lse.clone(), dqkv.clone())
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][2], res[False][2])
    valid = torch.zeros(B, H, S, dtype=torch.bool)
    for b_, n_ in enumerate(lens):
        valid[b_, :, :n_] = True
    assert torch.equal(res[True][1].cpu().view(B, H, S)[valid], res[False][1].cpu().view(B, H, S)[valid])
    assert torch.isfinite(res[True][2].float()).all() and float(res[True][2].float().abs().max()) > 0


@pytest.mark.parametrize("M,N,K", [(300, 264, 128), (2500, 768, 768), (4096, 1536, 768), (700, 768, 1536),
                                   (900, 520, 192), (520, 264, 64), (1100, 776, 832)])     # odd counts of 64-value K-steps: the 128-deep MFMA's missing half is zeros
def test_gemm_fp8_operands(dev, M, N, K):
    """BASELINE configs[4]: fp8 (OCP e4m3) MFMA with fp32 accumulation for the projections.  Exactness: against the fp64
    product of the DEQUANTISED operands (only the bf16 rounding of the result remains); accuracy: against the product of the
    original bf16 operands (two 3-bit-mantissa operands: relative L2 error of a K-term dot product ~ 3-4 %)."""
    from transfusion_amd import _lib as L, ops
    g = torch.Generator().manual_seed(M + N)
    x = bf(torch.randn(M, K, generator=g)).to(dev)
    w = bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    xq, sx = ops.quant_rows_fp8(x)
    wq, sw = ops.quant_rows_fp8(w)
    xd = xq.view(torch.float8_e4m3fn).double().cpu()[:, :K] * sx.double().cpu()[:, None]
    wd = wq.view(torch.float8_e4m3fn).double().cpu()[:, :K] * sw.double().cpu()[:, None]
    assert (xd - x.double().cpu()).abs().max() <= 0.0625 * x.double().abs().max().cpu() + 1e-6    # half an e4m3 ulp at the top binade
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(xq, wq, out, N, xq.shape[1], L.TF_EPI_BIAS, bias=bias, scale_a=sx, scale_w=sw)
    ref_q = xd @ wd.t() + bias.double().cpu()
    ref = x.double().cpu() @ w.double().cpu().t() + bias.double().cpu()
    assert rel(out, ref_q) < 6e-3
    assert rel(out, ref) < 6e-2


def test_sumsq_is_a_pure_function_of_its_input(dev):
    """tf_sumsq (global-norm clipping): the block partials are summed in index order by the last block to arrive -- no float atomics --
    so repeated runs give the same BITS (data-parallel ranks holding the same reduced gradient then compute the same clip coefficient
    and stay bit-identical), launches accumulate into `out`, and the value is the fp64 sum to fp32 accuracy."""
    from transfusion_amd import _lib as L, ops
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    for n in (1, 1000, 18_912_000 + 3):
        x = (torch.randn(n + 4, generator=g) * 3).to(dev)[4:] if n > 1000 else (torch.randn(n, generator=g) * 3).to(dev)
        if x.data_ptr() % 16:
            x = x.clone()
        outs = []
        for rep in range(6):
            out = torch.zeros(1, device=dev)
            L.check(lib.tf_sumsq(L.ptr(x), n, L.ptr(out), ops._stream()), "tf_sumsq")
            outs.append(out)
        torch.cuda.synchronize()
        assert all(torch.equal(o, outs[0]) for o in outs), [float(o) for o in outs]
        ref = x.double().pow(2).sum().item()
        assert abs(outs[0].item() - ref) <= 2e-6 * ref + 1e-30
        acc = torch.zeros(1, device=dev)
        for _ in range(3):
            L.check(lib.tf_sumsq(L.ptr(x), n, L.ptr(acc), ops._stream()), "tf_sumsq")
        assert abs(acc.item() - 3 * ref) <= 1e-5 * ref + 1e-30


def test_sumsq_never_reads_a_stale_partial(dev):
    """The hand-off inside tf_sumsq (block partials by sc1 stores, a ticket, the last arriver sums behind an agent acquire; no release
    fence -- ADVICE r4) under the conditions that expose a stale read: 512 blocks, launches back to back with no host
    synchronisation, an unrelated streaming kernel on another stream (uneven load), and every scratch row (16, round-robin) holding
    the partials of a launch over DIFFERENT data of 1e6 times the magnitude.  One stale partial would change the small sums by many
    orders of magnitude; every result must equal its own input's bits."""
    from transfusion_amd import _lib as L, ops
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    n = 18_912_000
    big = (torch.randn(n, generator=g) * 1e3).to(dev)
    small = (torch.randn(n, generator=g) * 1e-3).to(dev)
    refs = {}
    for name, x in (("big", big), ("small", small)):
        out = torch.zeros(1, device=dev)
        L.check(lib.tf_sumsq(L.ptr(x), n, L.ptr(out), ops._stream()), "tf_sumsq")
        torch.cuda.synchronize()
        refs[name] = out.clone()
        assert abs(out.item() - x.double().pow(2).sum().item()) <= 2e-6 * x.double().pow(2).sum().item()
    noise_a, noise_b = torch.empty(64 << 20, device=dev), torch.ones(64 << 20, device=dev)
    side = torch.cuda.Stream()
    outs = [torch.zeros(1, device=dev) for _ in range(96)]
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(12):
            noise_a.copy_(noise_b)
    for i, out in enumerate(outs):                       # 16 launches over `big`, 16 over `small`, ...: launch i reuses the row of launch i - 16
        x = big if (i // 16) % 2 == 0 else small
        L.check(lib.tf_sumsq(L.ptr(x), n, L.ptr(out), ops._stream()), "tf_sumsq")
    torch.cuda.synchronize()
    for i, out in enumerate(outs):
        want = refs["big"] if (i // 16) % 2 == 0 else refs["small"]
        assert torch.equal(out, want), (i, out.item(), want.item())


@pytest.mark.parametrize("G,Mg,N,K", [(4, 2083, 768, 768), (4, 2083, 2304, 768), (4, 2832, 768, 1536), (3, 300, 264, 128), (4, 5000, 1536, 768),
                                      (2, 8300, 768, 768)])     # 128-wide (ring and two-slot), large-tile and two-per-CU forms
def test_gemm_grouped_rows(dev, guard, G, Mg, N, K):
    """TfGemmArgs.groups: G equal row ranges, each against its own weight / bias (w_gstride apart) -- the wrapper's FPN levels as one
    launch.  Must equal G separate launches bit for bit (same tiles, same arithmetic), ragged last row tile of every group included."""
    from transfusion_amd import _lib as L, ops
    g = torch.Generator().manual_seed(G * Mg + N)
    x = guard(bf(torch.randn(G * Mg, K, generator=g)))
    r = guard(bf(torch.randn(G * Mg, N, generator=g)))
    # one buffer holds [W_g | bias_g] blocks at a constant byte stride, as the encoder runtime's per-level shadow blocks do
    blk = (N * K * 2 + N * 4 + 255) // 256 * 256
    store = guard(torch.zeros((G - 1) * blk + N * K * 2 + N * 4, dtype=torch.uint8))    # ends with the last group's bias: no slack behind it
    Ws, bs = [], []
    for k in range(G):
        w = store[k * blk: k * blk + N * K * 2].view(torch.bfloat16).view(N, K)
        b = store[k * blk + N * K * 2: k * blk + N * K * 2 + N * 4].view(torch.float32)
        w.copy_(bf(torch.randn(N, K, generator=g) * 0.05))
        b.copy_(torch.randn(N, generator=g))
        Ws.append(w)
        bs.append(b)
    drop = ops.drop_params(0.15, 5, 2)
    for epi, kw in ((L.TF_EPI_BIAS, {}), (L.TF_EPI_BIAS_DROP_RES, {"R": r, "drop": drop}), (L.TF_EPI_ADD, {"R": r})):
        out_g = guard(torch.zeros(G * Mg, N, dtype=torch.bfloat16))
        has_b = epi != L.TF_EPI_ADD
        ops.gemm(x, Ws[0], out_g, N, K, epi, bias=bs[0] if has_b else None, groups=G, w_gstride=blk, **kw)
        out_s = torch.zeros_like(out_g)
        full = torch.zeros_like(out_g)
        for k in range(G):
            kw_k = dict(kw)
            if "R" in kw_k:
                kw_k["R"] = r                                   # full-height views: rows (and dropout indices) stay global
            # a single-group launch over ALL rows with group k's weights; its rows of range k are what the grouped launch must produce
            ops.gemm(x, Ws[k], full, N, K, epi, bias=bs[k] if has_b else None, **kw_k)
            out_s[k * Mg:(k + 1) * Mg] = full[k * Mg:(k + 1) * Mg]
        ref = torch.cat([x[k * Mg:(k + 1) * Mg].double().cpu() @ Ws[k].double().cpu().t() + (bs[k].double().cpu() if epi == L.TF_EPI_BIAS else 0)
                         for k in range(G)]) if epi == L.TF_EPI_BIAS else None
        assert (out_g.float() - out_s.float()).abs().max().item() <= 2e-2 * out_s.float().abs().max().item(), epi       # (tile shapes may differ: bf16 ulp)
        if ref is not None:
            assert rel(out_g, ref) < 6e-3


@pytest.mark.parametrize("G,Mg,N,K,chunk", [(4, 2083, 768, 768, 0), (4, 2832, 2304, 768, 704), (3, 300, 264, 136, 0), (2, 8300, 768, 1536, 2080)])
def test_wgrad_grouped_rows(dev, guard, G, Mg, N, K, chunk):
    """TfWgradArgs.groups: range g of the rows accumulates dY_g^T X_g into ITS dW / db (dw_gstride apart); both kernels (caller-sized
    256x128 and self-sized 128x128), ragged group heights."""
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(G * Mg + K)
    dy = guard(bf(torch.randn(G * Mg, N, generator=g) * 0.1))
    x = guard(bf(torch.randn(G * Mg, K, generator=g)))
    blk = N * K + N + 13                                     # floats per group: [dW | db | slack], an odd stride on purpose
    blk = (blk + 3) // 4 * 4
    store = guard(torch.zeros(G * blk, dtype=torch.float32))
    dW0 = store[: N * K].view(N, K)
    db0 = store[N * K: N * K + N]
    ops.wgrad(dy, N, x, K, dW0, db0, m_chunk=chunk, groups=G, dw_gstride=blk * 4)
    torch.cuda.synchronize()
    for k in range(G):
        dW = store[k * blk: k * blk + N * K].view(N, K).double().cpu()
        db = store[k * blk + N * K: k * blk + N * K + N].double().cpu()
        yk, xk = dy[k * Mg:(k + 1) * Mg].double().cpu(), x[k * Mg:(k + 1) * Mg].double().cpu()
        assert rel(dW, yk.t() @ xk) < 2e-5, k
        assert rel(db, yk.sum(0)) < 2e-5, k
    assert float(store.view(G, blk)[:, N * K + N:].abs().max()) == 0.0       # nothing written between the groups' tensors


def test_backward_through_repacked_weight_shadows_raises(dev):
    """ops.linear keeps bf16 shadows of a weight and re-packs them IN PLACE after an optimiser step.  forward, step, forward again and
    then the FIRST graph's backward would form dgrad with the new weights: the re-pack bumps the shadows' autograd version, so that
    backward raises torch's in-place-modification error."""
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(9)
    w = torch.nn.Parameter(torch.randn(64, 64, generator=g).to(dev))
    x = torch.randn(16, 64, generator=g).to(dev).requires_grad_(True)
    y1 = ops.linear(x, w)
    with torch.no_grad():
        w.add_(0.5)                                    # an optimiser step (bumps w's version -> the shadows are re-packed in place)
    y2 = ops.linear(x, w)
    y2.float().sum().backward()                        # the current graph is fine
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y1.float().sum().backward()


@pytest.mark.gpu
@pytest.mark.parametrize("protected", [True, False])
def test_linear_input_from_another_stream_survives_until_its_weight_gradient_ran(dev, protected, monkeypatch):
    """The root cause of round 4's one-off 0.37 two-rank gradient mismatch, as a unit test.  ``ops.linear`` SAVES its input for the weight
    gradient, and when that input is already bf16 / contiguous / unpadded it saves the caller's own tensor -- which may have been
    allocated on another stream (the grouped encoder's output on the main stream, handed to a level's back-projection on a level
    stream).  Autograd replays the node on the level stream and drops the saved tensor as soon as the backward is ENQUEUED; the caching
    allocator then recycles the block for the next same-sized allocation on its home stream, whose kernel may run before the weight
    gradient has read it.  Here the weight gradient sits behind a 3-ms spin (ops.debug_delay_wgrad) and a main-stream node of the same
    graph, which runs right after the linear node's backward, asks the allocator for blocks of the input's size until it is handed the
    input's own block, and scribbles over it.  With ``record_stream`` (ops._LinearFn.forward) the allocator never hands that block out
    while the level stream still owes the weight gradient; ``protected=False`` is the negative control (record_stream disabled): the
    block IS handed out and the weight gradient comes out wrong."""
    from transfusion_amd import ops
    M, K, N = 4096, 512, 256
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(M, K, generator=g).to(dev)
    w0 = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    cot = torch.randn(M, N, generator=g).to(dev)
    ref = cot.to(torch.bfloat16).float().t() @ x0.to(torch.bfloat16).float()          # dW = dY^T X on the bf16-rounded operands
    target = {"ptr": 0, "hit": False}

    class Scribbler(torch.autograd.Function):              # identity; its backward runs on the main stream, after the linear node's
        @staticmethod
        def forward(ctx, t):
            return t * 1.0

        @staticmethod
        def backward(ctx, grad):
            held = []
            for _ in range(512):                           # the allocator serves cached free blocks before it asks the driver for memory
                j = torch.empty(M, K, dtype=torch.bfloat16, device=grad.device)
                held.append(j)
                if j.data_ptr() == target["ptr"]:
                    j.fill_(7.0)
                    target["hit"] = True
                    break
            return grad

    def scenario(st):
        target["ptr"], target["hit"] = 0, False
        w = torch.nn.Parameter(w0.clone())
        main = torch.cuda.current_stream(dev)
        xb = x0.to(torch.bfloat16).requires_grad_(True)
        mid = xb * 1.0                                         # an intermediate allocated on main: the tensor ops.linear will save
        target["ptr"] = mid.data_ptr()
        probe = Scribbler.apply(mid)                           # created BEFORE the linear node: its backward runs after it
        st.wait_stream(main)
        with torch.cuda.stream(st):
            y = ops.linear(mid, w)
        main.wait_stream(st)
        loss = (y.float() * cot).sum() + probe.float().sum()
        del mid, probe, y                                      # only the graph holds the intermediate now
        prev = ops.debug_delay_wgrad(3000)
        try:
            loss.backward()
        finally:
            ops.debug_delay_wgrad(prev)
        torch.cuda.synchronize()
        return target["hit"], ((w.grad.float() - ref).norm() / ref.norm()).item()

    if not protected:
        monkeypatch.setattr(torch.Tensor, "record_stream", lambda self, s: None)
    # Several level streams: two streams that the runtime maps onto ONE hardware queue run in submission order, and the scribble then
    # waits behind the spin by accident (the mapping depends on how many streams the process has created).  With the
    # fix every stream must be right; the control needs one stream on a queue of its own.
    seen = [scenario(torch.cuda.Stream(device=dev)) for _ in range(8)]
    monkeypatch.undo()
    if protected:
        # the property: the weight gradient is computed from the intact input.  (Round 6: "the block is never handed out" was asserted
        # too, and failed on a box with a slow host -- the scribbler's 512 allocations took longer than the 3-ms spin, the recorded stream's
        # work was DONE, and the allocator rightly recycled the block: hits with an error of 6e-7.  A hit is harmless exactly when the
        # gradient is right, which is what is asserted.)
        assert all(err < 1e-2 for _, err in seen), seen
    elif not any(hit and err > 5e-2 for hit, err in seen):
        # (none of the eight streams got a hardware queue of its own: the control is void on this box, the product is not at fault)
        pytest.skip(f"the control did not reproduce the hazard on this box: {seen}")


def test_sq_loss_kernels(dev, guard):
    """tf_sq_loss_fwd / tf_sq_loss_bwd (ABI v9: the benchmark's synthetic loss, SURVEY.md 8d, as library kernels): value and gradient
    against fp64 torch, with and without row weights, accumulation across terms, the same bits on a second call (deterministic sum), an
    upstream gradient that is not 1."""
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(11)
    vis = torch.randn(7, 196, 72, generator=g)
    lo = torch.randn(7, 130, 72, generator=g)
    valid = (torch.rand(7, 130, generator=g) < 0.6).float()
    v, l, w = guard(vis).requires_grad_(True), guard(lo).requires_grad_(True), guard(valid)
    kv, km = 1.0 / vis.numel(), 1.0 / (float(valid.sum()) * 72)
    loss = ops.sq_loss([(v, None, kv), (l, w, km)])
    (loss * 3.0).backward()
    vr, lr = vis.double().requires_grad_(True), lo.double().requires_grad_(True)
    ref = (vr ** 2).sum() * kv + ((lr * valid.double().unsqueeze(-1)) ** 2).sum() * km
    (ref * 3.0).backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    assert rel(v.grad, vr.grad) < 1e-6 and rel(l.grad, lr.grad) < 1e-6
    assert float(l.grad.cpu()[valid == 0].abs().max()) == 0.0
    again = ops.sq_loss([(v.detach(), None, kv), (l.detach(), w, km)])
    assert float(again) == float(loss)                                   # no float atomics: the same bits
    fmap = guard(torch.randn(3, 5, 14, 14, generator=g)).requires_grad_(True)      # a feature map whose last dimension is not a multiple of 4
    lf = ops.sq_loss([(fmap, None, 0.5)])
    lf.backward()
    assert abs(float(lf) - 0.5 * float((fmap.detach().double() ** 2).sum())) < 1e-5 * float(lf) and rel(fmap.grad, fmap.detach()) < 1e-6
    big = guard(torch.randn(22656, 768, generator=g))                     # the benchmark's size: many blocks, the 8-deep body and its tail
    assert abs(float(ops.sq_loss([(big, None, 1.0)])) - float((big.double() ** 2).sum())) < 1e-5 * float((big.double() ** 2).sum())
