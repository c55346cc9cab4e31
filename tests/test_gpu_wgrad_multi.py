"""tf_gemm_wgrad_multi: several weight gradients dW_p += dY_p^T X_p (+ bias gradients) in ONE launch, against fp64 products of the same
bf16 values -- heterogeneous extents (M included), ragged row chunks and tile tails, padded row / column groups, the fp32-accuracy mode's
plane pairs, `groups` expansion, accumulation semantics, every chunk count the launcher can choose."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def bf(t):
    return t.to(torch.bfloat16)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _problem(g, M, N, K, dev, scale=1.0, guard=None):
    """guard: the conftest fixture -- operands flush against unmapped pages (tests/guard_alloc.py)"""
    put = guard if guard is not None else (lambda t: t.to(dev))
    dY = put(bf(torch.randn(M, N, generator=g) * scale))
    X = put(bf(torch.randn(M, K, generator=g)))
    return dY, X, dY.double().cpu().t() @ X.double().cpu(), dY.double().cpu().sum(0)


@pytest.mark.parametrize("blocks", [0, -1, 1, 40, 97, 512, -2])          # -2: the 192 x 192 two-quad form
def test_layer_shaped_problems(dev, guard, blocks):
    """four problems with a layer's operand relations (shared rows, different N / K, one without a bias) at a small width"""
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(11 + abs(blocks))
    M, d = 1000 + 7 * (abs(blocks) % 5), 136                                   # ragged last 32-row step
    shapes = [(3 * d, d, True), (d, d, True), (2 * d, d, False), (d, 2 * d, True)]
    probs, outs, keep = [], [], []
    for N, K, bias in shapes:
        dY, X, ref, refb = _problem(g, M, N, K, dev, 0.1, guard)
        dW = guard(torch.zeros(N, K))
        db = guard(torch.zeros(N)) if bias else None
        probs.append(ops.wgrad_args(dY, N, X, K, dW, db))
        outs.append((dW, db, ref, refb))
        keep += [dY, X]
    ops.wgrad_multi(probs, blocks)
    for dW, db, ref, refb in outs:
        assert rel(dW, ref) < 1e-4
        if db is not None:
            assert rel(db, refb) < 1e-4
    ops.wgrad_multi(probs, blocks)                                         # accumulation: a second launch adds
    for dW, db, ref, refb in outs:
        assert rel(dW, 2 * ref) < 1e-4
        if db is not None:
            assert rel(db, 2 * refb) < 1e-4


def test_benchmark_layer(dev):
    """the benchmark's four shapes at a packed row count (d = 768, ff = 1536, M not a multiple of 32)"""
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(5)
    M, d, ff = 16661, 768, 1536
    x, o, x1 = (bf(torch.randn(M, d, generator=g)).to(dev) for _ in range(3))
    hh = bf(torch.randn(M, ff, generator=g)).to(dev)
    dqkv, dy1, dy2 = bf(torch.randn(M, 3 * d, generator=g) * 0.05).to(dev), bf(torch.randn(M, d, generator=g) * 0.05).to(dev), bf(torch.randn(M, d, generator=g) * 0.05).to(dev)
    du = bf(torch.randn(M, ff, generator=g) * 0.05).to(dev)
    pairs = [(dqkv, x), (dy1, o), (du, x1), (dy2, hh)]
    dWs = [torch.zeros(a.shape[1], b.shape[1], device=dev) for a, b in pairs]
    dbs = [torch.zeros(a.shape[1], device=dev) for a, _ in pairs]
    refs = [(a.float().t() @ b.float(), a.float().sum(0)) for a, b in pairs]
    for blocks in (256, 0, -1):                  # an explicit count; the launcher's own sizing alone / beside a chain
        for w, v in zip(dWs, dbs):
            w.zero_(); v.zero_()
        ops.wgrad_multi([ops.wgrad_args(a, a.shape[1], b, b.shape[1], w, v) for (a, b), w, v in zip(pairs, dWs, dbs)], blocks)
        for (rw, rv), w, v in zip(refs, dWs, dbs):
            assert rel(w, rw) < 2e-5, blocks
            assert rel(v, rv) < 2e-5, blocks


def test_different_row_counts_and_tile_tails(dev):
    """problems of different M (the wrapper's levels with unequal token grids), N / K that are not tile multiples, a one-step problem"""
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(23)
    shapes = [(33, 8, 8), (5000, 520, 264), (1200, 264, 136), (31, 256, 128)]
    probs, outs, keep = [], [], []
    for M, N, K in shapes:
        dY, X, ref, refb = _problem(g, M, N, K, dev)
        dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        probs.append(ops.wgrad_args(dY, N, X, K, dW, db))
        outs.append((dW, db, ref, refb))
        keep += [dY, X]
    for blocks in (0, 64, -1, -2):
        for dW, db, _, _ in outs:
            dW.zero_(); db.zero_()
        ops.wgrad_multi(probs, blocks)
        for dW, db, ref, refb in outs:
            assert rel(dW, ref) < 1e-4
            assert rel(db, refb) < 1e-4


def test_padded_groups_and_leading_dimensions(dev):
    """padded row groups (head dim 18 -> 32) and column groups, operands with leading dimensions wider than their extents"""
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(31)
    M, N, K, rg, rgp = 1000, 384, 128, 18, 32
    dYw = bf(torch.randn(M, N + 64, generator=g)).to(dev)
    Xw = bf(torch.randn(M, K + 64, generator=g)).to(dev)
    dY, X = dYw[:, :N], Xw[:, :K]
    ref, refb = dY.double().cpu().t() @ X.double().cpu(), dY.double().cpu().sum(0)
    n_src = (N // rgp) * rg
    idx = torch.tensor([i for i in range(N) if i % rgp < rg])
    dW, db = torch.zeros(n_src, K, device=dev), torch.zeros(n_src, device=dev)
    # the same tensors transposed in role: column groups on the K side
    dW2 = torch.zeros(K, n_src, device=dev)
    probs = [ops.wgrad_args(dY, N, X, K, dW, db, rg=rg, rgp=rgp, n_src=n_src),
             ops.wgrad_args(X, K, dY, N, dW2, None, cg=rg, cgp=rgp, k_src=n_src)]
    for blocks in (48, -1, -2):
        dW.zero_(); db.zero_(); dW2.zero_()
        ops.wgrad_multi(probs, blocks)
        assert rel(dW, ref[idx]) < 1e-4
        assert rel(db, refb[idx]) < 1e-4
        assert rel(dW2, ref[idx].t()) < 1e-4


def test_layer_at_the_benchmark_size_takes_the_two_quad_form(dev):
    """a d = 768 layer's four products at a packed row count (ragged last step, odd step count): 128 tiles of 192 x 192 at two row chunks
    -- the launcher's own rule (blocks = -1) picks the two-quad form here; against the 256 x 128 form (blocks = 432) and fp64"""
    from transfusion_amd import ops
    g = torch.Generator().manual_seed(77)
    M, d = 16637, 768
    probs, outs, keep = [], [], []
    for N, K in [(3 * d, d), (d, d), (2 * d, d), (d, 2 * d)]:
        dY = bf(torch.randn(M, N, generator=g) * 0.1).to(dev)
        X = bf(torch.randn(M, K, generator=g)).to(dev)
        dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        dW2, db2 = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        probs.append((ops.wgrad_args(dY, N, X, K, dW, db), ops.wgrad_args(dY, N, X, K, dW2, db2)))
        outs.append((dW, db, dW2, db2, dY, X))
    ops.wgrad_multi([p[0] for p in probs], -1)
    ops.wgrad_multi([p[1] for p in probs], 432)
    for dW, db, dW2, db2, dY, X in outs:
        ref = dY.double().t() @ X.double()
        assert rel(dW, ref) < 1e-4 and rel(dW2, ref) < 1e-4
        assert rel(db, dY.double().sum(0)) < 1e-4 and rel(db2, dY.double().sum(0)) < 1e-4
        assert rel(dW, dW2) < 2e-6                                         # same products, different summation orders


def test_groups_expand(dev):
    """TfWgradArgs.groups: range g of the rows accumulates into ITS dW / db, dw_gstride bytes apart"""
    from transfusion_amd import ops
    G, Mg, N, K = 3, 333, 264, 136
    g = torch.Generator().manual_seed(41)
    dy = bf(torch.randn(G * Mg, N, generator=g) * 0.1).to(dev)
    x = bf(torch.randn(G * Mg, K, generator=g)).to(dev)
    blk = (N * K + N + 13 + 3) // 4 * 4
    store = torch.zeros(G * blk, dtype=torch.float32, device=dev)
    p = ops.wgrad_args(dy, N, x, K, store[: N * K].view(N, K), store[N * K: N * K + N], groups=G, dw_gstride=blk * 4)
    ops.wgrad_multi([p], 30)
    ops.wgrad_multi([p], -1)                                               # the launcher's own sizing adds the same again
    ops.wgrad_multi([p], -2)                                               # ... and the 192 x 192 form a third time
    ops.wgrad_multi([p], 0)
    store *= 0.25
    for k in range(G):
        dW = store[k * blk: k * blk + N * K].view(N, K)
        db = store[k * blk + N * K: k * blk + N * K + N]
        yk, xk = dy[k * Mg:(k + 1) * Mg].double().cpu(), x[k * Mg:(k + 1) * Mg].double().cpu()
        assert rel(dW, yk.t() @ xk) < 2e-5, k
        assert rel(db, yk.sum(0)) < 2e-5, k
    assert float(store.view(G, blk)[:, N * K + N:].abs().max()) == 0.0


def test_split_mode(dev):
    """hi + lo plane pairs (fp32-accuracy mode): dW = dY_hi^T X_hi + dY_lo^T X_hi + dY_hi^T X_lo to ~1e-5"""
    from transfusion_amd import ops
    from test_gpu_fp32_mode import planes, joined, KTOL
    g = torch.Generator().manual_seed(3)
    probs, outs, keep = [], [], []
    for M, N, K, ldy, ldx in [(1000, 200, 136, 256, 192), (1000, 136, 200, 192, 256)]:
        dY, X = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g)
        Yh, Yl = planes(dY, dev, ldy)
        Xh, Xl = planes(X, dev, ldx)
        dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        probs.append(ops.wgrad_args(Yh, N, Xh, K, dW, db, dY_lo=Yl, X_lo=Xl))
        Y64, X64 = joined(Yh, Yl).cpu()[:, :N], joined(Xh, Xl).cpu()[:, :K]
        outs.append((dW, db, Y64.t() @ X64, Y64.sum(0)))
        keep += [Yh, Yl, Xh, Xl]
    for blocks in (24, -1, -2):
        for dW, db, _, _ in outs:
            dW.zero_(); db.zero_()
        ops.wgrad_multi(probs, blocks)
        for dW, db, ref, refb in outs:
            assert rel(dW, ref) < KTOL
            assert rel(db, refb) < KTOL


def test_argument_errors(dev):
    from transfusion_amd import ops
    from transfusion_amd._lib import TfError
    g = torch.Generator().manual_seed(1)
    dY, X, _, _ = _problem(g, 64, 16, 16, dev)
    dW = torch.zeros(16, 16, device=dev)
    with pytest.raises(TfError):
        ops.wgrad_multi([ops.wgrad_args(dY, 16, X, 16, dW)] * 17)            # more than TF_WGRAD_MULTI_MAX problems
    with pytest.raises(TfError):
        ops.wgrad_multi([ops.wgrad_args(dY, 12, X, 16, dW)])                 # N not a multiple of 8
