"""torch.autograd bindings over the C ABI (include/tfusion.h).  PyTorch only provides device memory,
streams and the autograd graph here; every FLOP runs in libtfusion_hip.so.  All entries raise when
given CPU tensors -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from transfusion_amd import _lib as L

BIG = 1 << 28


def _up(x: int, a: int) -> int:
    return (x + a - 1) // a * a


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.TfError("transfusion_amd kernels need device tensors on an MI355X (gfx950); there is no CPU fallback")


def _stream() -> int:
    """The current HIP stream of the current device as a raw handle.  (torch.cuda.current_stream() builds a Stream object through
    _get_device_index / _lazy_init / is_available -- ~10 us a call, ~100 calls per wrapper step across the forward and the autograd
    thread: ~1 ms of the 3.5 ms the host needs to enqueue that step.  The two C entry points below return the same handle in < 1 us.)"""
    return _raw_stream(_cur_device())


try:
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:               # (a torch build without the private entry points: the public, slower spelling)
    def _raw_stream(_idx):
        return torch.cuda.current_stream().cuda_stream

    def _cur_device():
        return 0


def _is_f32(t) -> int:
    if t.dtype == torch.float32:
        return 1
    if t.dtype == torch.bfloat16:
        return 0
    raise L.TfError(f"unsupported dtype {t.dtype}: fp32 or bf16 expected")


def drop_params(p: float, seed: int, site: int):
    lib = L.load()
    if p <= 0.0:
        return 0, 0, 1.0
    return lib.tf_drop_threshold(p), lib.tf_drop_key(seed, site), lib.tf_drop_scale(p)


_seed_counter = [0]


def next_seed() -> int:
    """A fresh dropout stream id per forward, derived from torch's seed so runs are reproducible."""
    _seed_counter[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _seed_counter[0]) & 0xFFFFFFFFFFFFFFFF


# ------------------------------------------------------------------------------------------------------
# small helpers: padded bf16 copies
# ------------------------------------------------------------------------------------------------------
def to_bf16_padded(x2d: torch.Tensor, ld: int) -> torch.Tensor:
    """[M, K] fp32/bf16 -> bf16 [M, ld] with zero pad columns (tf_copy_rows)."""
    M, K = x2d.shape
    if x2d.dtype == torch.bfloat16 and K == ld and x2d.is_contiguous():
        return x2d
    if K % 8:
        raise L.TfError(f"feature width {K} must be a multiple of 8")
    x2d = x2d.contiguous()
    out = torch.empty(M, ld, dtype=torch.bfloat16, device=x2d.device)
    a = L.TfCopyRowsArgs(src=L.ptr(x2d), src_is_f32=_is_f32(x2d), ld_src=K, src_rpg=max(M, 1), src_gstride=max(M, 1),
                         dst=L.ptr(out), dst_is_f32=0, ld_dst=ld, dst_rpg=max(M, 1), dst_gstride=max(M, 1), rows=M, cols=K)
    L.call("tf_copy_rows", a, _stream())
    return out


def from_padded(x2d: torch.Tensor, cols: int, dtype) -> torch.Tensor:
    M, ld = x2d.shape
    if cols == ld and dtype == x2d.dtype:
        return x2d
    out = torch.empty(M, cols, dtype=dtype, device=x2d.device)
    a = L.TfCopyRowsArgs(src=L.ptr(x2d), src_is_f32=_is_f32(x2d), ld_src=ld, src_rpg=max(M, 1), src_gstride=max(M, 1),
                         dst=L.ptr(out), dst_is_f32=_is_f32(out), ld_dst=cols, dst_rpg=max(M, 1), dst_gstride=max(M, 1), rows=M, cols=cols)
    L.call("tf_copy_rows", a, _stream())
    return out


def pack_weight(w: torch.Tensor, rows_p: int, cols_p: int, ld: int, ld_t: int, want_t: bool = True, rg=BIG, rgp=BIG, cg=BIG, cgp=BIG, into=None,
                residual: bool = False):
    """fp32 [N,K] parameter -> bf16 shadow [rows_p, ld] and transpose [cols_p, ld_t]; rows / columns may be regrouped (source index
    (p // gp) * g + p % gp, valid iff p % gp < g): heads of width hd padded to hdp.  ``into``: (dst, dst_t) of an earlier call with
    the same geometry, re-packed in place (their pad columns are still the zeros they were created with).  ``residual``: the lo plane of
    the fp32-accuracy mode, bf16(w - bf16(w)), formed by the kernel (TfPackArgs.residual)."""
    N, K = w.shape
    w = w.detach().contiguous().float()
    if into is not None and into[0].shape == (rows_p, ld) and (not want_t or (into[1] is not None and into[1].shape == (cols_p, ld_t))):
        dst, dst_t = into[0], into[1] if want_t else None
        # the kernel rewrites them through raw pointers: tell autograd, so that a backward whose forward saved the OLD contents (forward,
        # optimiser step, forward again, then the first graph's backward) raises torch's in-place-modification error instead of
        # silently forming dgrad with the new weights
        torch._C._increment_version([t for t in (dst, dst_t) if t is not None])
    else:
        dst = torch.zeros(rows_p, ld, dtype=torch.bfloat16, device=w.device)
        dst_t = torch.zeros(cols_p, ld_t, dtype=torch.bfloat16, device=w.device) if want_t else None
    a = L.TfPackArgs(src=L.ptr(w), rows=N, cols=K, dst=L.ptr(dst), ld_dst=ld, dst_t=L.ptr(dst_t), ld_dst_t=ld_t,
                     rows_p=rows_p, cols_p=cols_p, rg=rg, rgp=rgp, cg=cg, cgp=cgp, dst_is_f32=0, residual=1 if residual else 0)
    L.call("tf_pack_weight", a, _stream())
    return dst, dst_t


def pack_bias(b: torch.Tensor, cols_p: int, cg=BIG, cgp=BIG):
    """fp32 [N] bias -> fp32 [cols_p] with the same column regrouping as its weight's rows."""
    b = b.detach().contiguous().float()
    dst = torch.zeros(cols_p, dtype=torch.float32, device=b.device)
    a = L.TfPackArgs(src=L.ptr(b), rows=1, cols=b.numel(), dst=L.ptr(dst), ld_dst=cols_p, dst_t=0, ld_dst_t=0, rows_p=1, cols_p=cols_p,
                     rg=BIG, rgp=BIG, cg=cg, cgp=cgp, dst_is_f32=1)
    L.call("tf_pack_weight", a, _stream())
    return dst


def layernorm_fwd(x, y, gamma, beta, mean, rstd, rows, d, eps=1e-5, x_lo=None, y_lo=None):
    """y = LN(x[:, :d]) per row; x bf16 [rows, ldx]; y bf16 [rows, ldy] (pad zeroed) or fp32 [rows, d]; mean / rstd fp32 [rows].
    ``*_lo``: the lo planes of the fp32-accuracy mode (value = hi + lo)."""
    a = L.TfLnArgs(x=L.ptr(x), ldx=x.stride(0), y=L.ptr(y), ldy=y.stride(0), y_is_f32=_is_f32(y), gamma=L.ptr(gamma), beta=L.ptr(beta),
                   mean=L.ptr(mean), rstd=L.ptr(rstd), rows=rows, d=d, rows_per_group=rows, x_group_stride=rows, y_group_stride=rows, eps=eps,
                   x_lo=L.ptr(x_lo), y_lo=L.ptr(y_lo))
    L.call("tf_layernorm_fwd", a, _stream())


def layernorm_bwd(x, gamma, mean, rstd, dy, dx, dgamma, dbeta, rows, d, dx_drop=None, drop=(0, 0, 1.0), eps=1e-5, x_lo=None, dy_lo=None,
                  dx_lo=None, dx_drop_lo=None):
    """dx (bf16 [rows, lddx]) and optionally dx_drop = dx * keep / (1 - p); dgamma / dbeta accumulated (fp32 atomics)."""
    a = L.TfLnArgs(x=L.ptr(x), ldx=x.stride(0), gamma=L.ptr(gamma), mean=L.ptr(mean), rstd=L.ptr(rstd), rows=rows, d=d, rows_per_group=rows,
                   x_group_stride=rows, y_group_stride=rows, eps=eps, dy=L.ptr(dy), lddy=dy.stride(0), dy_is_f32=_is_f32(dy), dx=L.ptr(dx),
                   lddx=dx.stride(0), dx_drop=L.ptr(dx_drop), lddxd=0 if dx_drop is None else dx_drop.stride(0), drop_thr=drop[0], drop_key=drop[1],
                   drop_scale=drop[2], drop_ld=dx.stride(0), dgamma=L.ptr(dgamma), dbeta=L.ptr(dbeta),
                   x_lo=L.ptr(x_lo), dy_lo=L.ptr(dy_lo), dx_lo=L.ptr(dx_lo), dx_drop_lo=L.ptr(dx_drop_lo))
    L.call("tf_layernorm_bwd", a, _stream())


def split_planes(x: torch.Tensor):
    """fp32 tensor -> (hi, lo) bf16 planes with hi + lo == x to 16 significant bits: the operand format of the fp32-accuracy
    mode (include/tfusion.h, TfGemmArgs.A_lo).  Plain torch ops: a host-side helper for tests and stand-alone callers; the encoder
    runtime produces its planes inside its own kernels."""
    hi = x.to(torch.bfloat16)
    lo = (x.float() - hi.float()).to(torch.bfloat16)
    return hi, lo


def gemm(A, W, C_out, N, K, epilogue=None, bias=None, R=None, C2=None, drop=(0, 0, 1.0), scale_a=None, scale_w=None, act=0,
         A_lo=None, W_lo=None, C_lo=None, R_lo=None, C2_lo=None, groups=1, w_gstride=0):
    """C = epilogue(A @ W^T).  A / W are bf16, or -- when ``scale_a`` / ``scale_w`` are given -- uint8 tensors of OCP e4m3
    values from ``quant_rows_fp8`` (fp8 MFMA, fp32 accumulate, result scaled per row and per output channel), or -- with the
    ``*_lo`` planes -- hi + lo bf16 pairs (fp32-accuracy mode: three MFMA passes, fp32 epilogue, hi + lo outputs)."""
    fp8 = scale_a is not None or scale_w is not None
    g = L.TfGemmArgs(A=L.ptr(A), lda=A.stride(0), W=L.ptr(W), ldw=W.stride(0), C=L.ptr(C_out), ldc=C_out.stride(0),
                     bias=L.ptr(bias), R=L.ptr(R), ldr=0 if R is None else R.stride(0), C2=L.ptr(C2),
                     ldc2=0 if C2 is None else C2.stride(0), M=A.shape[0], N=N, K=K,
                     epilogue=L.TF_EPI_NONE if epilogue is None else epilogue,
                     drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2],
                     fp8=1 if fp8 else 0, scale_a=L.ptr(scale_a), scale_w=L.ptr(scale_w), act=act,
                     A_lo=L.ptr(A_lo), W_lo=L.ptr(W_lo), C_lo=L.ptr(C_lo), R_lo=L.ptr(R_lo), C2_lo=L.ptr(C2_lo),
                     groups=int(groups), w_gstride=int(w_gstride),
                     c_is_f32=1 if (A_lo is not None and C_out.dtype == torch.float32) else 0)   # fp32-accuracy mode: an fp32 result, no planes
    L.call("tf_gemm_fwd", g, _stream())


def planes_of(x2d: torch.Tensor, ld: int, drop=(0, 0, 1.0), drop_ld: int = None):
    """fp32 [M, K] (any row stride) -> the (hi, lo) bf16 operand planes [M, ld] of the fp32-accuracy mode, zero padded, with the
    optional input dropout applied BEFORE the split (tf_split_planes: one kernel; torch spelt it as three elementwise passes and two
    pads).  The keep mask is the function of (row * drop_ld + col) that tf_dropout_mask replays."""
    _require_cuda(x2d)
    M, K = x2d.shape
    if x2d.dtype != torch.float32 or x2d.stride(1) != 1:
        x2d = x2d.float().contiguous()
    hi = torch.empty(M, ld, dtype=torch.bfloat16, device=x2d.device)
    lo = torch.empty(M, ld, dtype=torch.bfloat16, device=x2d.device)
    a = L.TfPlanesArgs(src=L.ptr(x2d), ld_src=x2d.stride(0), hi=L.ptr(hi), lo=L.ptr(lo), ld_dst=ld, dst_f32=0, ld_f32=0, rows=M, cols=K,
                       drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2], drop_ld=ld if drop_ld is None else drop_ld)
    L.call("tf_split_planes", a, _stream())
    return hi, lo


def dropout_f32_(x2d: torch.Tensor, cols: int, drop, drop_ld: int):
    """In place on the first ``cols`` columns of an fp32 [M, ld] tensor: x *= keep / (1 - p) with the mask of ``planes_of``."""
    a = L.TfPlanesArgs(src=L.ptr(x2d), ld_src=x2d.stride(0), hi=0, lo=0, ld_dst=0, dst_f32=L.ptr(x2d), ld_f32=x2d.stride(0), rows=x2d.shape[0],
                       cols=cols, drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2], drop_ld=drop_ld)
    L.call("tf_split_planes", a, _stream())


def quant_rows_fp8(x: torch.Tensor, ld_out: int = None):
    """bf16 [rows, cols] -> (uint8 [rows, ld_out] of OCP e4m3 values, fp32 [rows] scales): x ~= q * scale[:, None]."""
    _require_cuda(x)
    rows, cols = x.shape
    ld_out = (cols + 63) // 64 * 64 if ld_out is None else ld_out
    q = torch.empty(rows, ld_out, dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    L.check(L.load().tf_quant_rows_fp8(x.data_ptr(), x.stride(0), q.data_ptr(), ld_out, sc.data_ptr(), rows, cols, _stream()), "tf_quant_rows_fp8")
    return q, sc


# ---- the step clock (tf_clock_ptr): what makes a step captured in a HIP graph draw fresh dropout masks on every replay ----
_clock_host = [0]          # host mirror of the device word (graph replays advance the device side; a capturing caller keeps this in step: tests/graph_step.py)


def clock_ptr() -> int:
    p = L.load().tf_clock_ptr()
    if not p:
        L.check(-9, "tf_clock_ptr")
    return int(p)


def clock_advance(by: int = 1):
    """*clock += by on the current stream (the first node of a captured training step)."""
    L.check(L.load().tf_clock_advance(int(by), _stream()), "tf_clock_advance")
    _clock_host[0] += int(by)


def clock_set(value: int):
    L.check(L.load().tf_clock_set(int(value), _stream()), "tf_clock_set")
    _clock_host[0] = int(value)


def clock_value() -> int:
    """The clock as the host last left it (plus the replays a capturing caller has counted)."""
    return _clock_host[0]


def sumsq(x: torch.Tensor, out: torch.Tensor):
    """out[0] += sum(x^2) (tf_sumsq: deterministic, no float atomics); x contiguous fp32, out a 1-element fp32 tensor."""
    _require_cuda(x, out)
    if x.dtype != torch.float32 or not x.is_contiguous() or out.dtype != torch.float32:
        raise L.TfError("tf_sumsq needs contiguous fp32 tensors")
    L.check(L.load().tf_sumsq(L.ptr(x), x.numel(), L.ptr(out), _stream()), "tf_sumsq")


def _sq_rows(x, w):
    """(rows, d) of a term of tf_sq_loss: with row weights the tensor's own [..., d] rows (d % 4 == 0); without, the row structure means
    nothing to the kernels and any tensor whose element count is a multiple of 4 goes as [numel / 4, 4] (a 14 x 14 feature map)."""
    if w is None:
        if x.numel() % 4:
            raise L.TfError("tf_sq_loss: the element count must be a multiple of 4")
        return x.numel() // 4, 4
    d = x.shape[-1]
    rows = x.numel() // d
    if d % 4 or w.numel() != rows:
        raise L.TfError(f"tf_sq_loss: {w.numel()} row weights for {rows} rows of {d} (d % 4 must be 0)")
    return rows, d


class _SqLossFn(torch.autograd.Function):
    """sum_t scale_t * sum_r row_w_t[r]^2 |x_t[r, :]|^2 over the given terms (tf_sq_loss_fwd / tf_sq_loss_bwd): the benchmark's synthetic loss
     without a framework elementwise kernel in the step.  Every term: a contiguous fp32 CUDA tensor [..., d] (d % 4 == 0), an
    optional fp32 row-weight tensor with one entry per row, a float weight."""

    @staticmethod
    def forward(ctx, n_terms, *flat):
        xs, ws, scales = flat[:n_terms], flat[n_terms:2 * n_terms], flat[2 * n_terms:]
        out = torch.empty(1, dtype=torch.float32, device=xs[0].device)
        lib = L.load()
        args = []
        for i, (x, w, sc) in enumerate(zip(xs, ws, scales)):
            _require_cuda(x)
            if x.dtype != torch.float32 or not x.is_contiguous() or (w is not None and (w.dtype != torch.float32 or not w.is_contiguous())):
                raise L.TfError("tf_sq_loss needs contiguous fp32 tensors")
            rows, d = _sq_rows(x, w)
            a = L.TfSqLossArgs(x=L.ptr(x), rows=rows, d=d, row_w=L.ptr(w), scale=float(sc), out=L.ptr(out), accumulate=1 if i else 0)
            L.check(lib.tf_sq_loss_fwd(C.byref(a), C.c_void_p(_stream())), "tf_sq_loss_fwd")
            args.append(a)
        ctx.save_for_backward(*xs, *[w for w in ws if w is not None])
        ctx.has_w = [w is not None for w in ws]
        ctx.scales = [float(sc) for sc in scales]
        ctx.n_terms = n_terms
        return out[0]

    @staticmethod
    def backward(ctx, g):
        saved = list(ctx.saved_tensors)
        xs, rest = saved[:ctx.n_terms], saved[ctx.n_terms:]
        lib = L.load()
        g = g.contiguous() if g.dtype == torch.float32 else g.float().contiguous()
        grads = []
        for x, has_w, sc in zip(xs, ctx.has_w, ctx.scales):
            w = rest.pop(0) if has_w else None
            dx = torch.empty_like(x)
            rows, d = _sq_rows(x, w)
            a = L.TfSqLossArgs(x=L.ptr(x), rows=rows, d=d, row_w=L.ptr(w), scale=sc, g=L.ptr(g), dx=L.ptr(dx))
            L.check(lib.tf_sq_loss_bwd(C.byref(a), C.c_void_p(_stream())), "tf_sq_loss_bwd")
            grads.append(dx)
        return (None, *grads, *([None] * (2 * ctx.n_terms)))


def sq_loss(terms):
    """terms: [(x, row_w or None, scale), ...] -> the scalar sum of scale * sum_r row_w[r]^2 |x[r]|^2 (differentiable w.r.t. every x)."""
    xs = [t[0] for t in terms]
    ws = [t[1] for t in terms]
    scales = [float(t[2]) for t in terms]
    return _SqLossFn.apply(len(terms), *xs, *ws, *scales)


def set_gemm_concurrency(n: int):
    """Planning hint for the GEMM tile choice: ``n`` launch sequences share the chip (tf_set_gemm_concurrency)."""
    L.load().tf_set_gemm_concurrency(int(n))


_overlap_cache = {}      # (device index, stream handle) -> TfOverlap: every stream that runs encoders has its own side stream + events


def _overlap_key(device):
    idx = device.index if device.index is not None else _cur_device()
    return idx, _raw_stream(idx)


def overlap_handle(device):
    """The ``TfOverlap`` ctypes object of (device, current stream) (None when disabled); see ``wgrad_overlap``."""
    addr = wgrad_overlap(device)
    if addr is None:
        return None
    return _overlap_cache[_overlap_key(device)]


def wgrad_overlap_enabled() -> bool:
    """Whether encoder calls fork a side stream for their weight gradients (``TF_WGRAD_OVERLAP=0`` keeps one stream)."""
    return os.environ.get("TF_WGRAD_OVERLAP", "1") != "0"


def wgrad_overlap(device):
    """Address of the ``TfOverlap`` (side stream + events on which ``tf_encoder_bwd`` issues its weight-gradient GEMMs) that belongs to
    the CURRENT stream of ``device``, created on first use: encoders that run concurrently on different streams (the wrapper's feature
    levels) must not share events.  ``TF_WGRAD_OVERLAP=0`` keeps everything on the caller's stream (A/B switch)."""
    if os.environ.get("TF_WGRAD_OVERLAP", "1") == "0":
        return None
    key = _overlap_key(device)
    o = _overlap_cache.get(key)
    if o is None:
        o = L.TfOverlap()
        with torch.cuda.device(key[0]):
            L.check(L.load().tf_overlap_create(C.byref(o)), "tf_overlap_create")
        _overlap_cache[key] = o
    return C.addressof(o)


def join_overlap(device):
    """Make the current stream wait for every weight-gradient GEMM its side stream still owes (tf_overlap_join)."""
    o = overlap_handle(device)
    if o is not None:
        L.check(L.load().tf_overlap_join(C.byref(o), C.c_void_p(_stream())), "tf_overlap_join")


_side_streams = {}


def side_stream(device):
    """The side stream of ``overlap_handle(device)`` as a torch stream object (for event ordering from Python), or None."""
    o = overlap_handle(device)
    if o is None:
        return None
    key = _overlap_key(device)
    s = _side_streams.get(key)
    if s is None:
        s = torch.cuda.ExternalStream(o.stream, device=torch.device("cuda", key[0]))
        _side_streams[key] = s
    return s


# ---- streams that wrote into .grad DIRECTLY ----------------------------------------------------------------------------------------------
# A backward that adds its parameter gradients straight into ``.grad`` (the encoders under FusionTrainStep, ops.linear(accumulate=True))
# hands autograd ``None`` for them.  When nothing downstream of it needs a gradient either (a wrapper level on its own stream whose
# feature map comes from a frozen backbone) no tensor ever flows from that stream back to the one the optimiser runs on.  Under the
# delay probe the optimiser is nevertheless ordered behind such a stream (attributed to the engine joining the "leaf stream" of the
# parameter's AccumulateGrad node, which still runs as a no-op) -- but that edge cannot be removed for a negative control, and
# hardware-queue aliasing can hide a missing edge.  So the product does not rely on it: such writers note their
# stream here and FusionTrainStep.step() joins them explicitly before the exchange and the optimiser (a few event waits per step).
_grad_writer_streams = {}


def note_grad_writer(device=None):
    st = torch.cuda.current_stream(device)
    _grad_writer_streams[(st.device.index, st.cuda_stream)] = st


def join_grad_writers(device):
    """The current stream of ``device`` waits for every stream noted by ``note_grad_writer`` since the last call."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(idx)
    for key in [k for k in _grad_writer_streams if k[0] == idx]:
        st = _grad_writer_streams.pop(key)
        if st.cuda_stream != cur.cuda_stream:
            cur.wait_stream(st)


_zeros_cache = {}


def _zeros256(device):
    z = _zeros_cache.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.uint8, device=device)
        _zeros_cache[device] = z
    return z


# ---- stream-ordering probe (tests/test_gpu_ddp.py): every weight-gradient launch -- the encoder runtime's on its side streams, K1 / K9 /
# ops.linear's on the stream they are issued on -- behind a spin of `us` microseconds, so that a consumer (collective, optimiser,
# AccumulateGrad) that lacks an event edge to its producer reads too early EVERY time instead of once in many runs
_debug_wgrad_delay_us = 0


def debug_delay_wgrad(us: int) -> int:
    global _debug_wgrad_delay_us
    prev = _debug_wgrad_delay_us
    _debug_wgrad_delay_us = int(us)
    L.load().tf_debug_delay_wgrad(int(us))
    return prev


def debug_spin(us: int, stream=None):
    L.check(L.load().tf_debug_spin(int(us), C.c_void_p(_stream() if stream is None else stream)), "tf_debug_spin")


def wgrad(dY, N, X, K, dW, db=None, rg=BIG, rgp=BIG, n_src=None, cg=BIG, cgp=BIG, k_src=None, m_chunk=0, dY_lo=None, X_lo=None, groups=1,
          dw_gstride=0):
    if _debug_wgrad_delay_us:
        debug_spin(_debug_wgrad_delay_us)
    w = L.TfWgradArgs(dY=L.ptr(dY), ldy=dY.stride(0), X=L.ptr(X), ldx=X.stride(0), dW=L.ptr(dW), lddw=dW.stride(0), db=L.ptr(db),
                      dY_lo=L.ptr(dY_lo), X_lo=L.ptr(X_lo),
                      zeros=L.ptr(_zeros256(dY.device)), M=dY.shape[0], N=N, K=K, rg=rg, rgp=rgp,
                      n_src=dW.shape[0] if n_src is None else n_src, cg=cg, cgp=cgp,
                      k_src=dW.shape[1] if k_src is None else k_src, m_chunk=m_chunk, groups=int(groups), dw_gstride=int(dw_gstride))
    L.call("tf_gemm_wgrad", w, _stream())


def wgrad_args(dY, N, X, K, dW, db=None, rg=BIG, rgp=BIG, n_src=None, cg=BIG, cgp=BIG, k_src=None, dY_lo=None, X_lo=None, groups=1,
               dw_gstride=0):
    """One problem of ``wgrad_multi`` (a ``TfWgradArgs``; keeps no reference to the tensors)."""
    return L.TfWgradArgs(dY=L.ptr(dY), ldy=dY.stride(0), X=L.ptr(X), ldx=X.stride(0), dW=L.ptr(dW), lddw=dW.stride(0), db=L.ptr(db),
                         dY_lo=L.ptr(dY_lo), X_lo=L.ptr(X_lo), zeros=0, M=dY.shape[0], N=N, K=K, rg=rg, rgp=rgp,
                         n_src=dW.shape[0] if n_src is None else n_src, cg=cg, cgp=cgp,
                         k_src=dW.shape[1] if k_src is None else k_src, m_chunk=0, groups=int(groups), dw_gstride=int(dw_gstride))


def wgrad_multi(problems, blocks=0):
    """Several weight gradients (``wgrad_args`` results) as ONE launch on the current stream (tf_gemm_wgrad_multi)."""
    if _debug_wgrad_delay_us:
        debug_spin(_debug_wgrad_delay_us)
    arr = (L.TfWgradArgs * len(problems))(*problems)
    L.check(L.load().tf_gemm_wgrad_multi(arr, C.c_int(len(problems)), C.c_int(int(blocks)), C.c_void_p(_stream())), "tf_gemm_wgrad_multi")


# ------------------------------------------------------------------------------------------------------
# K1 / K9 permutations
# ------------------------------------------------------------------------------------------------------
def _patch_args(feat, cols, B, Cc, H, W, ph, pw):
    return L.TfPatchArgs(feat=L.ptr(feat), feat_is_f32=_is_f32(feat), cols=L.ptr(cols), ld_cols=cols.stride(0), B=B, C=Cc, H=H, W=W,
                         ph=ph, pw=pw)


class _PatchifyFn(torch.autograd.Function):
    """feat [B,C,H,W] -> rows [B*Hp*Wp, ld] bf16 (column = (c*ph+i)*pw+j, zero padded to ld)."""

    @staticmethod
    def forward(ctx, feat, ph, pw, ld):
        _require_cuda(feat)
        feat = feat.contiguous()
        B, Cc, H, W = feat.shape
        Hp, Wp = H // ph, W // pw
        cols = torch.empty(B * Hp * Wp, ld, dtype=torch.bfloat16, device=feat.device)
        L.call("tf_patchify_fwd", _patch_args(feat, cols, B, Cc, H, W, ph, pw), _stream())
        ctx.meta = (feat.shape, feat.dtype, ph, pw)
        return cols

    @staticmethod
    def backward(ctx, g):
        shape, dtype, ph, pw = ctx.meta
        B, Cc, H, W = shape
        g = g.contiguous()
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        dfeat = torch.empty(shape, dtype=dtype, device=g.device)
        L.call("tf_patchify_bwd", _patch_args(dfeat, g, B, Cc, H, W, ph, pw), _stream(), _is_f32(dfeat))
        return dfeat, None, None, None


def patchify(image, patch_h, patch_w, ld=None):
    B, Cc, H, W = image.shape
    K = Cc * patch_h * patch_w
    rows = _PatchifyFn.apply(image, patch_h, patch_w, K if ld is None else ld)
    if ld is None:
        return rows.view(B, (H // patch_h) * (W // patch_w), K)
    return rows


class _RegroupFn(torch.autograd.Function):
    """rows [B*Nv, >= C*ph*pw] bf16 -> [B,C,H,W] (F.fold with kernel == stride; uncovered border zero)."""

    @staticmethod
    def forward(ctx, rows, B, Cc, H, W, ph, pw, out_dtype):
        _require_cuda(rows)
        if rows.dtype != torch.bfloat16:
            rows = rows.to(torch.bfloat16)
        rows = rows.contiguous()
        out = torch.empty(B, Cc, H, W, dtype=out_dtype, device=rows.device)
        L.call("tf_regroup_fwd", _patch_args(out, rows, B, Cc, H, W, ph, pw), _stream(), _is_f32(out))
        ctx.meta = (rows.shape, B, Cc, H, W, ph, pw)
        return out

    @staticmethod
    def backward(ctx, g):
        shape, B, Cc, H, W, ph, pw = ctx.meta
        g = g.contiguous()
        drows = torch.empty(shape, dtype=torch.bfloat16, device=g.device)
        L.call("tf_regroup_bwd", _patch_args(g, drows, B, Cc, H, W, ph, pw), _stream())
        return drows, None, None, None, None, None, None, None


def regroup(patches, init_h, init_w, patch_h, patch_w, out_dtype=None):
    B, Nv, CK = patches.shape
    Cc = CK // (patch_h * patch_w)
    if (init_h // patch_h) * (init_w // patch_w) != Nv:
        raise RuntimeError(f"regroup_patches: {Nv} tokens do not tile a {init_h}x{init_w} map with {patch_h}x{patch_w} patches")
    return _RegroupFn.apply(patches.reshape(B * Nv, CK), B, Cc, init_h, init_w, patch_h, patch_w, out_dtype or patches.dtype)


# ------------------------------------------------------------------------------------------------------
# Linear (K1's conv-as-GEMM and K9's back-projection)
# ------------------------------------------------------------------------------------------------------
import weakref

_shadow_cache = {}      # (id(first owner), #owners, planes) -> (weakrefs to the owning Parameters, key, shadows)


def _weight_shadows(weight, N8, Kp, Np, planes=False, sources=None):
    """bf16 shadows [N8, Kp] and [Kp, Np] of an fp32 weight, re-packed only when the parameter changed (its storage moved or its
    version was bumped -- FusedRAdam / FusionTrainStep do that for their raw-pointer updates).  Keyed on the parameter OBJECT (id + a
    weak reference that must still point at it), never on a bare data_ptr: a freed tensor's address is reused by its successor.
    ``sources``: the Parameters a derived ``weight`` (e.g. the noun | verb | ttc concatenation of the RoI heads) was built from -- the
    cache entry is then keyed on THEM, so the temporary is packed once per optimiser step, not once per forward; a derived tensor
    without ``sources`` is packed every call and never cached (it would only leave a dead entry behind).
    ``planes``: also the lo planes (w - bf16(w), the fp32-accuracy mode's second operand plane) -> (W, W^T, W_lo, W^T_lo)."""
    w2 = weight.reshape(weight.shape[0], -1)
    owners = list(sources) if sources else [weight]
    cacheable = all(isinstance(o, torch.nn.Parameter) for o in owners)
    key = tuple((id(o), o.data_ptr(), o._version) for o in owners) + (tuple(weight.shape), N8, Kp, Np, bool(planes))
    slot = (id(owners[0]), len(owners), bool(planes))
    hit = _shadow_cache.get(slot) if cacheable else None
    if hit is not None and all(r() is o for r, o in zip(hit[0], owners)) and hit[1] == key:
        return hit[2]
    # same owners, new version (the optimiser stepped): re-pack into the entry's tensors instead of zero-filling two fresh ones per
    # weight and step.  (A forward whose backward is still pending must not be followed by an optimiser step and another forward of the
    # same weight before that backward runs: pack_weight bumps the shadows' autograd version, so that backward RAISES.)
    reuse = hit[2] if (hit is not None and all(r() is o for r, o in zip(hit[0], owners)) and hit[2][0].device == w2.device) else None
    wsh, wsh_t = pack_weight(w2, N8, Kp, Kp, Np, into=reuse[:2] if reuse else None)
    out = (wsh, wsh_t)
    if planes:
        # the lo planes, bf16(w - bf16(w)), by the same kernel (round 5; before: three elementwise torch passes over the weight and four
        # zero-filled shadows per weight and optimiser step -- the K1 / K9 weights of the wrapper are 4 - 8 M elements each)
        out = out + pack_weight(w2, N8, Kp, Kp, Np, into=reuse[2:4] if (reuse and len(reuse) == 4) else None, residual=True)
    if not cacheable:
        return out
    if len(_shadow_cache) > 256:                                   # drop entries whose parameter is gone
        for k in [k for k, v in _shadow_cache.items() if any(r() is None for r in v[0])]:
            del _shadow_cache[k]
    _shadow_cache[slot] = (tuple(weakref.ref(o) for o in owners), key, out)
    return out


def _planes_padded(x2d: torch.Tensor, ld: int):
    """[M, K] (fp32 or bf16) -> (hi, lo) bf16 [M, ld], zero padded: hi + lo == x to 16 significant bits (tf_split_planes; a plane width
    that is not a multiple of 8 -- the loss kernel's class-count logits -- takes the torch spelling)."""
    if ld % 8 == 0:
        return planes_of(x2d, ld)
    xf = x2d.float()
    hi = xf.to(torch.bfloat16)
    lo = (xf - hi.float()).to(torch.bfloat16)
    K = x2d.shape[1]
    if ld != K:
        hi, lo = torch.nn.functional.pad(hi, (0, ld - K)), torch.nn.functional.pad(lo, (0, ld - K))
    return hi.contiguous(), lo.contiguous()


def _x3_forward(xh, xl, wsh, wsh_lo, bias, N, N8, Kp, Np):
    """fp32 [M, Np] = (xh + xl) @ (W_hi + W_lo)^T + bias on three bf16 MFMA passes (columns >= N8 are never written)."""
    M = xh.shape[0]
    y = torch.empty(M, Np, dtype=torch.float32, device=xh.device)
    bf = None if bias is None else bias.detach().float().contiguous()
    if bf is not None and N8 != N:
        bf = torch.nn.functional.pad(bf, (0, N8 - N))
    gemm(xh, wsh, y, N8, Kp, L.TF_EPI_BIAS if bias is not None else L.TF_EPI_NONE, bias=bf, A_lo=xl, W_lo=wsh_lo)
    return y


class _LinearX3Fn(torch.autograd.Function):
    """``_LinearFn`` in the fp32-accuracy mode (include/tfusion.h, TfGemmArgs.A_lo): operands as hi + lo bf16 planes, three MFMA passes,
    fp32 results -- K1 / K9 / the RoI heads when ``run.precision`` is 32.  Every step is a kernel of the library: the planes (and the
    input dropout) come from tf_split_planes, the GEMMs write fp32 directly (TfGemmArgs.c_is_f32), the backward's mask is applied in
    place by the same kernel -- no elementwise torch op, no float keep-mask tensor."""

    @staticmethod
    def forward(ctx, x2d, weight, bias, p_drop_in, seed, wsh, wsh_t, wsh_lo, wsh_t_lo):
        _require_cuda(x2d, weight)
        M, K = x2d.shape
        N = weight.shape[0]
        N8, Kp, Np = _up(N, 8), _up(K, 64), _up(N, 64)
        drop = drop_params(p_drop_in, seed, 7)
        # dropout BEFORE the split (scaling each plane separately would round hi * 1/(1-p) to bf16 and lose the 16-bit property);
        # the keep mask is the same function of the element index (row * Kp + col) the bf16 path's tf_dropout_apply uses
        xh, xl = planes_of(x2d, Kp, drop, Kp)
        y = _x3_forward(xh, xl, wsh, wsh_lo, bias, N, N8, Kp, Np)
        ctx.save_for_backward(xh, xl, wsh_t, wsh_t_lo)
        ctx.meta = (M, K, N, N8, Kp, Np, bias is not None, weight.shape, x2d.dtype, drop)
        return y[:, :N] if Np != N else y

    @staticmethod
    def backward(ctx, gy):
        xh, xl, wsh_t, wsh_t_lo = ctx.saved_tensors
        M, K, N, N8, Kp, Np, has_bias, wshape, xdtype, drop = ctx.meta
        gh, gl = planes_of(gy.reshape(M, N), Np)              # (columns [N, Np) zero: the weight shadows' pad rows see nothing)
        dx = None
        if ctx.needs_input_grad[0]:
            dxf = torch.empty(M, Kp, dtype=torch.float32, device=gy.device)
            gemm(gh, wsh_t, dxf, Kp, Np, L.TF_EPI_NONE, A_lo=gl, W_lo=wsh_t_lo)
            if drop[0]:
                dropout_f32_(dxf, K, drop, Kp)
            dx = dxf[:, :K] if Kp != K else dxf
            if xdtype != torch.float32:
                dx = dx.to(xdtype)
        dW = torch.zeros(wshape, dtype=torch.float32, device=gy.device)
        db = torch.zeros(N, dtype=torch.float32, device=gy.device) if has_bias else None
        wgrad(gh, N8, xh, Kp, dW.view(N, -1), db, dY_lo=gl, X_lo=xl)
        return dx, dW, db, None, None, None, None, None, None


def _patch_args_planes(feat, hi, lo, B, Cc, H, W, ph, pw):
    return L.TfPatchArgs(feat=L.ptr(feat), feat_is_f32=_is_f32(feat), cols=L.ptr(hi), ld_cols=hi.stride(0), B=B, C=Cc, H=H, W=W, ph=ph, pw=pw,
                         cols_lo=L.ptr(lo))


class _PatchEmbedX3Fn(torch.autograd.Function):
    """K1 at run.precision 32 (cross_f_box_wrapper.py:266-274, 183-185): the k = s = p Conv2d as im2col + GEMM with the fp32 feature map
    split into hi + lo planes BY THE GATHER (TfPatchArgs.cols_lo) -- no fp32 im2col matrix, no torch permute -- three MFMA passes, fp32
    tokens out; the backward folds the plane pair of d(cols) straight back into an fp32 d(feat)."""

    @staticmethod
    def forward(ctx, feat, weight, ph, pw, wsh, wsh_t, wsh_lo, wsh_t_lo):
        _require_cuda(feat, weight)
        feat = feat.contiguous()
        B, Cc, H, W = feat.shape
        Hp, Wp = H // ph, W // pw
        K, N = Cc * ph * pw, weight.shape[0]
        N8, Kp, Np = _up(N, 8), _up(K, 64), _up(N, 64)
        M = B * Hp * Wp
        xh = torch.empty(M, Kp, dtype=torch.bfloat16, device=feat.device)
        xl = torch.empty(M, Kp, dtype=torch.bfloat16, device=feat.device)
        L.call("tf_patchify_fwd", _patch_args_planes(feat, xh, xl, B, Cc, H, W, ph, pw), _stream())
        y = _x3_forward(xh, xl, wsh, wsh_lo, None, N, N8, Kp, Np)
        ctx.save_for_backward(xh, xl, wsh_t, wsh_t_lo)
        ctx.meta = (feat.shape, feat.dtype, ph, pw, M, K, N, N8, Kp, Np, weight.shape)
        return (y[:, :N] if Np != N else y).view(B, Hp * Wp, N)

    @staticmethod
    def backward(ctx, gy):
        xh, xl, wsh_t, wsh_t_lo = ctx.saved_tensors
        shape, dtype, ph, pw, M, K, N, N8, Kp, Np, wshape = ctx.meta
        gh, gl = planes_of(gy.reshape(M, N), Np)
        dfeat = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty(M, Kp, dtype=torch.bfloat16, device=gy.device)
            dl = torch.empty(M, Kp, dtype=torch.bfloat16, device=gy.device)
            gemm(gh, wsh_t, dh, Kp, Np, L.TF_EPI_NONE, A_lo=gl, W_lo=wsh_t_lo, C_lo=dl)
            dfeat = torch.empty(shape, dtype=dtype, device=gy.device)
            L.call("tf_patchify_bwd", _patch_args_planes(dfeat, dh, dl, shape[0], shape[1], shape[2], shape[3], ph, pw), _stream(), _is_f32(dfeat))
        dW = torch.zeros(wshape, dtype=torch.float32, device=gy.device)
        wgrad(gh, N8, xh, Kp, dW.view(N, -1), None, dY_lo=gl, X_lo=xl)
        return dfeat, dW, None, None, None, None, None, None


def patch_embed_fp32(feat, weight, ph, pw):
    """[B, C, H, W] -> fp32 tokens [B, H' W', d] in the fp32-accuracy mode, entirely on the library's kernels."""
    N = weight.shape[0]
    K = feat.shape[1] * ph * pw
    wsh, wsh_t, wsh_lo, wsh_t_lo = _weight_shadows(weight, _up(N, 8), _up(K, 64), _up(N, 64), planes=True)
    return _PatchEmbedX3Fn.apply(feat, weight, ph, pw, wsh, wsh_t, wsh_lo, wsh_t_lo)


class _BackProjectX3Fn(torch.autograd.Function):
    """K9 at run.precision 32 (utils.py:84-119, 42-46): dropout -> Linear(d -> p^2 C) -> fold, with the GEMM result as a hi + lo plane
    pair that the fold adds back into the fp32 feature map (TfPatchArgs.cols_lo), and the backward's gather splitting the fp32
    cotangent into the plane pair the weight gradient and dgrad consume -- no fp32 [M, p^2 C] matrix, no torch permute."""

    @staticmethod
    def forward(ctx, x, weight, bias, p_drop, seed, geom, wsh, wsh_t, wsh_lo, wsh_t_lo):
        _require_cuda(x, weight)
        B, Nv, d = x.shape
        Cc, H, W, ph, pw = geom
        N = weight.shape[0]
        N8, Kp, Np = _up(N, 8), _up(d, 64), _up(N, 64)
        M = B * Nv
        drop = drop_params(p_drop, seed, 7)
        xh, xl = planes_of(x.reshape(M, d), Kp, drop, Kp)
        yh = torch.empty(M, Np, dtype=torch.bfloat16, device=x.device)
        yl = torch.empty(M, Np, dtype=torch.bfloat16, device=x.device)
        bf = bias.detach().float().contiguous()
        if N8 != N:
            bf = torch.nn.functional.pad(bf, (0, N8 - N))
        gemm(xh, wsh, yh, N8, Kp, L.TF_EPI_BIAS, bias=bf, A_lo=xl, W_lo=wsh_lo, C_lo=yl)
        out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device)
        L.call("tf_regroup_fwd", _patch_args_planes(out, yh, yl, B, Cc, H, W, ph, pw), _stream(), 1)
        ctx.save_for_backward(xh, xl, wsh_t, wsh_t_lo)
        ctx.meta = (B, Nv, d, geom, M, N, N8, Kp, Np, weight.shape, x.dtype, drop)
        return out

    @staticmethod
    def backward(ctx, g):
        xh, xl, wsh_t, wsh_t_lo = ctx.saved_tensors
        B, Nv, d, geom, M, N, N8, Kp, Np, wshape, xdtype, drop = ctx.meta
        Cc, H, W, ph, pw = geom
        g = g.contiguous()
        gh = torch.empty(M, Np, dtype=torch.bfloat16, device=g.device)
        gl = torch.empty(M, Np, dtype=torch.bfloat16, device=g.device)
        L.call("tf_regroup_bwd", _patch_args_planes(g, gh, gl, B, Cc, H, W, ph, pw), _stream())
        dx = None
        if ctx.needs_input_grad[0]:
            dxf = torch.empty(M, Kp, dtype=torch.float32, device=g.device)
            gemm(gh, wsh_t, dxf, Kp, Np, L.TF_EPI_NONE, A_lo=gl, W_lo=wsh_t_lo)
            if drop[0]:
                dropout_f32_(dxf, d, drop, Kp)
            dx = (dxf[:, :d] if Kp != d else dxf).reshape(B, Nv, d)
            if xdtype != torch.float32:
                dx = dx.to(xdtype)
        dW = torch.zeros(wshape, dtype=torch.float32, device=g.device)
        db = torch.zeros(N, dtype=torch.float32, device=g.device)
        wgrad(gh, N8, xh, Kp, dW.view(N, -1), db, dY_lo=gl, X_lo=xl)
        return dx, dW, db, None, None, None, None, None, None, None


def back_project_fp32(x, weight, bias, p_drop, init_h, init_w, ph, pw):
    """tokens [B, Nv, d] -> fp32 feature map [B, C, init_h, init_w] in the fp32-accuracy mode (zero border where the patches do not
    tile the map), entirely on the library's kernels."""
    B, Nv, d = x.shape
    N = weight.shape[0]
    Cc = N // (ph * pw)
    if (init_h // ph) * (init_w // pw) != Nv:
        raise RuntimeError(f"regroup_patches: {Nv} tokens do not tile a {init_h}x{init_w} map with {ph}x{pw} patches")
    seed = next_seed() if p_drop > 0 else 0
    wsh, wsh_t, wsh_lo, wsh_t_lo = _weight_shadows(weight, _up(N, 8), _up(d, 64), _up(N, 64), planes=True)
    return _BackProjectX3Fn.apply(x, weight, bias, float(p_drop), seed, (Cc, init_h, init_w, ph, pw), wsh, wsh_t, wsh_lo, wsh_t_lo)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, weight, bias, p_drop_in, seed, wsh, wsh_t, into=None):
        _require_cuda(x2d, weight)
        ctx.into = into     # (weight.grad, bias.grad or None): the weight gradient is accumulated THERE by the kernel (see linear)
        M, K = x2d.shape
        N = weight.shape[0]
        N8 = _up(N, 8)      # class-count heads (87 nouns, 74 verbs): zero weight rows up to the 16-B store width
        Kp, Np = _up(K, 64), _up(N, 64)
        xb = to_bf16_padded(x2d, Kp)
        drop = drop_params(p_drop_in, seed, 7)
        if drop[0]:
            xd = torch.empty_like(xb)
            L.check(L.load().tf_dropout_apply(L.ptr(xb), L.ptr(xd), xb.numel(), drop[1], drop[0], drop[2], _stream()), "tf_dropout_apply")
            xb = xd
        # xb is SAVED for the backward's weight gradient, and it may be the caller's own tensor (bf16, contiguous, no pad): one that was
        # allocated on ANOTHER stream -- the grouped encoder's output (main) handed to a level's back-projection (level stream).  Autograd
        # replays this node on the stream it ran on and drops the saved tensor as soon as its backward has been ENQUEUED; the allocator
        # would then hand the block to the next allocation on its home stream while the weight gradient has not read it yet (round 4's
        # one-off 0.37 gradient mismatch; DESIGN.md "Round 5").  record_stream defers the reuse behind this stream's work.
        if xb.is_cuda:
            xb.record_stream(torch.cuda.current_stream(xb.device))
        y = torch.zeros(M, Np, dtype=torch.bfloat16, device=x2d.device) if Np != N else torch.empty(M, N, dtype=torch.bfloat16, device=x2d.device)
        bf = None if bias is None else bias.detach().float().contiguous()
        if bf is not None and N8 != N:
            bf = torch.nn.functional.pad(bf, (0, N8 - N))
        gemm(xb, wsh, y, N8, Kp, L.TF_EPI_BIAS if bias is not None else L.TF_EPI_NONE, bias=bf)
        ctx.save_for_backward(xb, wsh_t)
        ctx.meta = (M, K, N, N8, Kp, Np, x2d.dtype, drop, bias is not None, weight.shape)
        return y[:, :N] if Np != N else y

    @staticmethod
    def backward(ctx, gy):
        xb, wsh_t = ctx.saved_tensors
        M, K, N, N8, Kp, Np, xdtype, drop, has_bias, wshape = ctx.meta
        gy = gy.reshape(M, N)
        if N8 != N:
            gy = torch.nn.functional.pad(gy, (0, N8 - N))
        gyb = to_bf16_padded(gy, Np)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, Kp, dtype=torch.bfloat16, device=gy.device)
            gemm(gyb, wsh_t, dx, Kp, Np, L.TF_EPI_NONE)
            if drop[0]:
                L.check(L.load().tf_dropout_apply(L.ptr(dx), L.ptr(dx), dx.numel(), drop[1], drop[0], drop[2], _stream()), "tf_dropout_apply")
            dx = from_padded(dx, K, xdtype)
        if ctx.into is not None:
            gw, gb = ctx.into
            wgrad(gyb, N8, xb, Kp, gw.view(N, -1), gb)
            note_grad_writer(gy.device)
            return dx, None, None, None, None, None, None, None
        dW = torch.zeros(wshape, dtype=torch.float32, device=gy.device)
        db = torch.zeros(N, dtype=torch.float32, device=gy.device) if has_bias else None
        wgrad(gyb, N8, xb, Kp, dW.view(N, -1), db)      # rows >= N are masked by n_src = N
        return dx, dW, db, None, None, None, None, None


def linear(x, weight, bias=None, p_drop_in: float = 0.0, precision: str = "bf16", weight_sources=None, accumulate: bool = False):
    """y = dropout(x) @ W^T + b on the MFMA GEMM; x [..., K] -> bf16 [..., N] (``precision="fp32"``: the fp32-accuracy mode, fp32 out).
    ``weight`` may have any trailing shape (the k = s = p Conv2d weight [d, C, p, p] of K1): it is used as [N, prod(rest)].
    ``weight_sources``: the Parameters a derived ``weight`` (a concatenation) was built from -- see ``_weight_shadows``.
    ``accumulate`` (bf16 path, ``weight`` / ``bias`` leaf Parameters with preallocated contiguous fp32 ``.grad``): the backward adds the
    weight / bias gradient straight into ``.grad`` -- what the encoders do under FusionTrainStep -- instead of handing autograd a fresh
    zero-filled tensor per call to add (one fill, one add and two autograd nodes per weight and step less)."""
    lead = x.shape[:-1]
    K = x.shape[-1]
    N = weight.shape[0]
    if K % 8:
        raise L.TfError(f"linear: K={K} must be a multiple of 8")
    if precision not in ("bf16", "fp32"):
        raise ValueError(f"precision={precision!r}: 'bf16' or 'fp32'")
    _require_cuda(x, weight)
    w2 = weight.reshape(N, -1)
    seed = next_seed() if p_drop_in > 0 else 0
    if precision == "fp32":
        wsh, wsh_t, wsh_lo, wsh_t_lo = _weight_shadows(weight, _up(N, 8), _up(K, 64), _up(N, 64), planes=True, sources=weight_sources)
        y = _LinearX3Fn.apply(x.reshape(-1, K), w2, bias, float(p_drop_in), seed, wsh, wsh_t, wsh_lo, wsh_t_lo)
    else:
        wsh, wsh_t = _weight_shadows(weight, _up(N, 8), _up(K, 64), _up(N, 64), sources=weight_sources)
        into = None
        if accumulate and weight_sources is None and torch.is_grad_enabled() and weight.requires_grad:
            gw, gb = weight.grad, None if bias is None else bias.grad
            ok = gw is not None and gw.dtype == torch.float32 and gw.is_contiguous() and gw.shape == weight.shape
            ok = ok and (bias is None or (bias.requires_grad and gb is not None and gb.dtype == torch.float32 and gb.is_contiguous()))
            if ok:
                into = (gw, gb)
        y = _LinearFn.apply(x.reshape(-1, K), w2, bias, float(p_drop_in), seed, wsh, wsh_t, into)
    return y.reshape(*lead, N)


# ------------------------------------------------------------------------------------------------------
# language auxiliary head: masked pooling + LayerNorm + GELU (lm_layers.py:59-72)
# ------------------------------------------------------------------------------------------------------
class _LmPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask_u8, pool_type, ln_w, ln_b, eps, gelu):
        _require_cuda(x)
        B, Lt, d = x.shape
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise L.TfError(f"lm_pool: dtype {x.dtype}")
        x = x.contiguous()
        pooled = torch.empty(B, d, dtype=torch.float32, device=x.device)
        feat = torch.empty(B, d, dtype=torch.float32, device=x.device)
        arg = torch.empty(B, d, dtype=torch.int32, device=x.device) if pool_type == 1 else None
        lw = None if ln_w is None else ln_w.detach().float().contiguous()
        lb = None if ln_b is None else ln_b.detach().float().contiguous()
        a = L.TfLmPoolArgs(x=L.ptr(x), x_is_f32=_is_f32(x), mask=L.ptr(mask_u8), B=B, L=Lt, d=d, type=pool_type,
                           ln_w=L.ptr(lw), ln_b=L.ptr(lb), eps=eps, gelu=1 if gelu else 0,
                           pooled=L.ptr(pooled), arg=L.ptr(arg), feat=L.ptr(feat))
        L.call("tf_lm_pool_fwd", a, _stream())
        ctx.save_for_backward(pooled, arg, mask_u8, lw, lb)
        ctx.meta = (B, Lt, d, pool_type, eps, gelu, x.dtype)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        pooled, arg, mask_u8, lw, lb = ctx.saved_tensors
        B, Lt, d, pool_type, eps, gelu, xdtype = ctx.meta
        dfeat = dfeat.float().contiguous()
        dx = torch.empty(B, Lt, d, dtype=xdtype, device=dfeat.device)
        dlw = dlb = scratch = None
        if lw is not None:
            dlw = torch.empty(d, dtype=torch.float32, device=dfeat.device)
            dlb = torch.empty(d, dtype=torch.float32, device=dfeat.device)
            scratch = torch.empty(2, B, d, dtype=torch.float32, device=dfeat.device)
        a = L.TfLmPoolArgs(mask=L.ptr(mask_u8), B=B, L=Lt, d=d, type=pool_type, ln_w=L.ptr(lw), ln_b=L.ptr(lb), eps=eps,
                           gelu=1 if gelu else 0, pooled=L.ptr(pooled), arg=L.ptr(arg), dfeat=L.ptr(dfeat),
                           dx=L.ptr(dx), dx_is_f32=_is_f32(dx), dln_w=L.ptr(dlw), dln_b=L.ptr(dlb), scratch=L.ptr(scratch))
        L.call("tf_lm_pool_bwd", a, _stream())
        return dx, None, None, dlw, dlb, None, None


def lm_pool(x, att_mask, pool_type: str, ln_w=None, ln_b=None, eps: float = 1e-5, gelu: bool = False):
    """[B, L, d] language tokens -> [B, d] fp32 features: GELU(LN(mean|max over L of x * att_mask)); att_mask is the
    HF-convention mask (True / 1 = real token) or None."""
    if pool_type not in ("mean", "max"):
        raise NotImplementedError(pool_type)
    m = None if att_mask is None else att_mask.to(torch.uint8).contiguous()
    return _LmPoolFn.apply(x, m, 1 if pool_type == "max" else 0, ln_w, ln_b, float(eps), bool(gelu))


# ------------------------------------------------------------------------------------------------------
# narration pooling tail: tanh, token-axis L2 normalisation, out_dropout in one kernel each way (slowfast_features_dsets.py:229-235)
# ------------------------------------------------------------------------------------------------------
class _PoolNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lens, use_tanh, p_drop, seed):
        _require_cuda(x)
        if x.dim() != 3 or x.dtype not in (torch.float32, torch.bfloat16) or x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
            x = x.contiguous()
        B, T, d = x.shape
        drop = drop_params(p_drop, seed, 9)
        z = torch.empty(B, T, d, dtype=torch.float32, device=x.device)
        y = torch.empty_like(z) if drop[0] else z
        n = torch.empty(B, d, dtype=torch.float32, device=x.device)
        a = L.TfPoolNormArgs(x=L.ptr(x), x_is_f32=_is_f32(x), ldx=x.stride(1), lens=L.ptr(lens), B=B, T=T, d=d, use_tanh=1 if use_tanh else 0,
                             drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2], y=L.ptr(y), z=L.ptr(z), n=L.ptr(n))
        L.call("tf_pool_norm_fwd", a, _stream())
        ctx.save_for_backward(z, n, lens)
        ctx.meta = (B, T, d, use_tanh, drop, x.dtype, x.stride(1))
        return y

    @staticmethod
    def backward(ctx, gy):
        z, n, lens = ctx.saved_tensors
        B, T, d, use_tanh, drop, xdtype, ldx = ctx.meta
        gy = gy.float().contiguous()
        ldg = ldx if xdtype == torch.bfloat16 else d
        gx = torch.empty(B, T, ldg, dtype=xdtype, device=gy.device)
        a = L.TfPoolNormArgs(lens=L.ptr(lens), B=B, T=T, d=d, use_tanh=1 if use_tanh else 0, drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2],
                             z=L.ptr(z), n=L.ptr(n), gy=L.ptr(gy), gx=L.ptr(gx), gx_is_f32=_is_f32(gx), ldgx=ldg)
        L.call("tf_pool_norm_bwd", a, _stream())
        return (gx[..., :d] if ldg != d else gx), None, None, None, None


def pool_norm(x, lens=None, use_tanh: bool = False, p_drop: float = 0.0):
    """[B, T, d] (bf16 / fp32) -> fp32 [B, T, d]: tanh (optional), rows t >= lens[b] zeroed (``lens``: int32 device tensor or None),
    L2 normalisation over the token axis when T > 1, dropout -- SlowFastPooling.forward's tail behind its out_mlp GEMM."""
    seed = next_seed() if p_drop > 0 else 0
    return _PoolNormFn.apply(x, lens, bool(use_tanh), float(p_drop), seed)


# ------------------------------------------------------------------------------------------------------
# RoI heads: softplus TTC output and the four losses (csrc/heads.hip)
# ------------------------------------------------------------------------------------------------------
class _SoftplusColFn(torch.autograd.Function):
    """bf16 logits, or fp32 logits (fp32-accuracy mode): those go to the kernel as hi + lo bf16 planes."""

    @staticmethod
    def forward(ctx, cls, col):
        _require_cuda(cls)
        if cls.dtype not in (torch.bfloat16, torch.float32) or cls.dim() != 2 or cls.stride(1) != 1:
            raise L.TfError("softplus_col: a bf16 or fp32 [R, N] tensor (row stride free) is expected")
        R = cls.shape[0]
        hi, lo = (cls, None) if cls.dtype == torch.bfloat16 else _planes_padded(cls, cls.shape[1])
        y = torch.empty(R, dtype=torch.float32, device=cls.device)
        L.check(L.load().tf_softplus_col(hi.data_ptr(), L.ptr(lo), hi.stride(0), col, y.data_ptr(), None, None, None, R, _stream()), "tf_softplus_col")
        ctx.save_for_backward(hi, lo)
        ctx.col, ctx.ncols = col, cls.shape[1]
        return y

    @staticmethod
    def backward(ctx, gy):
        hi, lo = ctx.saved_tensors
        R = hi.shape[0]
        dh = torch.zeros(R, hi.stride(0), dtype=torch.bfloat16, device=hi.device)
        dl = None if lo is None else torch.zeros(R, hi.stride(0), dtype=torch.bfloat16, device=hi.device)
        gy = gy.float().contiguous()
        L.check(L.load().tf_softplus_col(hi.data_ptr(), L.ptr(lo), hi.stride(0), ctx.col, None, gy.data_ptr(), dh.data_ptr(), L.ptr(dl), R, _stream()),
                "tf_softplus_col")
        if lo is None:
            return dh[:, :ctx.ncols], None
        return dh[:, :ctx.ncols].float() + dl[:, :ctx.ncols].float(), None


def softplus_col(cls, col: int):
    """ttcs = F.softplus(cls[:, col]) as fp32 [R] (roi_wrappers.py:228-229), cls the bf16 logits of the concatenated heads."""
    return _SoftplusColFn.apply(cls, col)


class _NaoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls, box, ttcs, Cn, Cv, noun, verb, ttc_t, reg_t, noun_w, verb_w, verb_ignore, verb_bg, ttc_bg, ttc_bg_val, ttc_beta):
        _require_cuda(cls, box, noun)
        for t in (cls, box):
            if t is not None and (t.dtype not in (torch.bfloat16, torch.float32) or t.dim() != 2 or t.stride(1) != 1):
                raise L.TfError("nao_head_losses: logits must be bf16 or fp32 [R, N] tensors (row stride free)")
        # fp32 logits (fp32-accuracy mode) reach the kernels as hi + lo bf16 planes (TfHeadsLossArgs.cls_lo / box_lo)
        cls_shape, box_shape = cls.shape, None if box is None else box.shape
        cls_lo = box_lo = None
        if cls.dtype == torch.float32:
            cls, cls_lo = _planes_padded(cls, cls.shape[1])
        if box is not None and box.dtype == torch.float32:
            box, box_lo = _planes_padded(box, box.shape[1])
        R = cls.shape[0]
        dev = cls.device
        i64 = lambda t: None if t is None else t.to(device=dev, dtype=torch.int64).contiguous()
        f32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()
        noun, verb, ttc_t, reg_t, noun_w, verb_w = i64(noun), i64(verb), f32(ttc_t), f32(reg_t), f32(noun_w), f32(verb_w)
        ttcs = None if ttcs is None else ttcs.detach().float().contiguous()
        if noun.numel() != R or (verb is not None and verb.numel() != R) or (reg_t is not None and tuple(reg_t.shape) != (R, 4)):
            raise RuntimeError(f"labels / targets do not match the {R} RoIs")
        if ttc_t is not None and ttc_t.numel() != R:
            raise RuntimeError(f"ttc_targets has {ttc_t.numel()} entries for {R} RoIs")
        if noun_w is not None and noun_w.numel() < Cn:
            raise RuntimeError(f"noun class weights: {noun_w.numel()} entries for {Cn} classes")
        if verb is not None and verb_w is not None and verb_w.numel() < Cv:
            raise RuntimeError(f"verb class weights: {verb_w.numel()} entries for {Cv} classes")
        # label ranges are checked by the kernel (out-of-range labels select nothing and raise sums[7]); the flag of an EARLIER call is
        # turned into torch's IndexError here, once its copy has landed -- no host synchronisation in the step (csrc/heads.hip)
        check_label_errors(sync=False)
        sums = torch.zeros(8, dtype=torch.float32, device=dev)
        lse = torch.empty(2 * R, dtype=torch.float32, device=dev)
        losses = torch.empty(4, dtype=torch.float32, device=dev)
        a = L.TfHeadsLossArgs(cls=L.ptr(cls), cls_lo=L.ptr(cls_lo), ld_cls=cls.stride(0), box=L.ptr(box) if reg_t is not None else 0,
                              box_lo=L.ptr(box_lo) if reg_t is not None else 0, ld_box=0 if box is None else box.stride(0),
                              ttcs=L.ptr(ttcs) if ttc_t is not None else 0, R=R, Cn=Cn, Cv=Cv, noun_labels=L.ptr(noun), verb_labels=L.ptr(verb),
                              ttc_targets=L.ptr(ttc_t), reg_targets=L.ptr(reg_t), noun_w=L.ptr(noun_w), verb_w=L.ptr(verb_w) if verb is not None else 0,
                              verb_ignore=int(verb_ignore), verb_bg=int(bool(verb_bg)), ttc_bg=int(bool(ttc_bg)), ttc_bg_val=float(ttc_bg_val),
                              ttc_beta=float(ttc_beta), box_beta=1.0 / 9, sums=L.ptr(sums), lse=L.ptr(lse), losses=L.ptr(losses))
        L.call("tf_heads_loss_fwd", a, _stream())
        _watch_label_flag(sums)
        ctx.args, ctx.keep = a, (cls, box, ttcs, noun, verb, ttc_t, reg_t, noun_w, verb_w, sums, lse, cls_lo, box_lo)
        ctx.shapes = (cls_shape, box_shape)
        return losses

    @staticmethod
    def backward(ctx, g):
        a = ctx.args
        cls, box, ttcs = ctx.keep[0], ctx.keep[1], ctx.keep[2]
        R, dev = cls.shape[0], cls.device
        gscale = g.float().contiguous()
        cls_lo, box_lo = ctx.keep[11], ctx.keep[12]
        d_cls = torch.empty(R, cls.stride(0), dtype=torch.bfloat16, device=dev)
        d_box = None if box is None else torch.empty(R, box.stride(0), dtype=torch.bfloat16, device=dev)
        d_cls_lo = None if cls_lo is None else torch.empty_like(d_cls)
        d_box_lo = None if box_lo is None else torch.empty_like(d_box)
        d_ttcs = None if ttcs is None else torch.empty(R, dtype=torch.float32, device=dev)
        a.gscale, a.d_cls, a.d_box, a.d_ttcs = L.ptr(gscale), L.ptr(d_cls), L.ptr(d_box), L.ptr(d_ttcs)
        a.d_cls_lo, a.d_box_lo = L.ptr(d_cls_lo), L.ptr(d_box_lo)
        L.call("tf_heads_loss_bwd", a, _stream())
        (cs, bs) = ctx.shapes
        g_cls = d_cls[:, :cs[1]] if d_cls_lo is None else d_cls[:, :cs[1]].float() + d_cls_lo[:, :cs[1]].float()
        g_box = None if box is None else (d_box[:, :bs[1]] if d_box_lo is None else d_box[:, :bs[1]].float() + d_box_lo[:, :bs[1]].float())
        return (g_cls, g_box, d_ttcs) + (None,) * 13


_label_flags = []        # (pinned host copy of sums[7], event recorded behind the copy) of recent loss calls


def _watch_label_flag(sums):
    if torch.cuda.is_current_stream_capturing():      # a pinned copy + an event query do not belong in a captured sequence
        return
    host = torch.empty(1, dtype=torch.float32, pin_memory=True)
    host.copy_(sums[7:8], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(sums.device))
    _label_flags.append((host, ev))
    if len(_label_flags) > 64:            # nobody ever asked: keep the queue bounded (the oldest copies have long landed)
        check_label_errors(sync=False)
        del _label_flags[:-64]


def check_label_errors(sync: bool = True):
    """Raises the IndexError torch's cross_entropy raises for a target outside [0, C) -- for any earlier ``nao_head_losses`` call whose
    labels the kernel found out of range (such labels contributed nothing to that call's losses and gradients).  ``sync=False`` looks
    only at calls whose flag has already reached the host; ``sync=True`` waits for all of them (end of an epoch, tests)."""
    bad = 0.0
    while _label_flags:
        host, ev = _label_flags[0]
        if not sync and not ev.query():
            break
        if sync:
            ev.synchronize()
        _label_flags.pop(0)
        bad += float(host[0])
    if bad:
        raise IndexError(f"Target out of bounds: {int(bad)} noun / verb labels outside their class range in an earlier nao_head_losses call")


def nao_head_losses(cls, box, ttcs, Cn, Cv, noun, verb, ttc_targets, reg_targets, noun_w, verb_w, verb_ignore=999, verb_bg=False, ttc_bg=False,
                    ttc_bg_val=0.0, ttc_beta=1.0):
    """-> fp32 [4]: box, noun, verb, ttc losses (include/tfusion.h TfHeadsLossArgs).  cls = noun | verb | ttc logits [R, >= Cn + Cv (+1)],
    box = box_regression [R, 4*Cn], ttcs = softplus outputs [R] (its gradient flows back through ``softplus_col``)."""
    return _NaoLossFn.apply(cls, box, ttcs, Cn, Cv, noun, verb, ttc_targets, reg_targets, noun_w, verb_w, verb_ignore, verb_bg, ttc_bg,
                            ttc_bg_val, ttc_beta)


def attn_dropmask(B: int, H: int, S: int, p: float, seed: int, site: int, device) -> torch.Tensor:
    """Keep-bitmask of the attention-probability dropout site ([B*H*S, ceil(S/64)] u64, as int64 storage)."""
    lib = L.load()
    thr, key, _ = drop_params(p, seed, site)
    bits = torch.empty(lib.tf_attn_dropmask_bytes(B, H, S) // 8, dtype=torch.int64, device=device)
    L.check(lib.tf_attn_dropmask(L.ptr(bits), B, H, S, key, thr, _stream()), "tf_attn_dropmask")
    return bits


def dropout_mask(n: int, p: float, seed: int, site: int, device) -> torch.Tensor:
    """Test hook: the keep-mask a dropout site draws for element indices 0..n-1."""
    thr, key, _ = drop_params(p, seed, site)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    L.check(L.load().tf_dropout_mask(L.ptr(out), n, key, thr, _stream()), "tf_dropout_mask")
    return out
