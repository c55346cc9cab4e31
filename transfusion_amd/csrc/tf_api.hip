// C-ABI layer of libtfusion_hip.so (see include/tfusion.h) and the native encoder runtime: the launch
// sequence of one CrossTransformerModuleBox forward / backward, enqueued on the caller's stream (plus, with a
// TfOverlap handle, the caller-owned side stream) with no allocation and no host synchronisation; callable from any
// thread.  State: none of its own except the opt-in launch tracer; a TfOverlap handle carries its pending-join bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "tf_kernels.h"
#include <atomic>
#include <mutex>
#include <vector>

// Flags of the events that order the chain and its side stream.  They are waited for by streams of the SAME device only
// (hipStreamWaitEvent; never hipEventSynchronize / hipEventQuery followed by a host read), so the system-scope fence a default event
// adds when it is recorded -- the write-back that makes device memory visible to the host and to other devices -- buys nothing: every
// kernel still ends with its own device-scope release (which is what makes its stores visible to the other XCDs' L2s, same stream or
// not).  Same box, three runs each (tools/experiments/event_flags_ab.sh): 4.002 -> 3.979 ms at B = 32, 1.367 -> 1.348 at B = 4;
// hipEventReleaseToDevice alone: 3.992 / 1.363.
#ifndef TF_EVENT_FLAGS
#define TF_EVENT_FLAGS (hipEventDisableTiming | hipEventDisableSystemFence)
#endif
namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* what) {
  if (code > 0) snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", what, code, hipGetErrorString((hipError_t)code));
  else snprintf(g_err, sizeof(g_err), "%s: invalid argument (code %d)", what, code);
  return code;
}
#define TF_TRY(expr, what) do { const int rc__ = (expr); if (rc__ != 0) return fail(rc__, what); } while (0)

struct TraceRec { char name[56]; hipEvent_t e0, e1; hipStream_t st; double flops, bytes; };
std::mutex g_trace_mu;
bool g_trace_on = false;
std::vector<TraceRec> g_trace;
std::vector<hipStream_t> g_trace_sides;

inline size_t up(size_t x, size_t a) { return (x + a - 1) / a * a; }
constexpr int BIG = 1 << 28;

// ---- stream-ordering probes (tf_debug_spin / tf_debug_delay_wgrad; tests) ----
// A spin of a bounded number of microseconds, enqueued like any kernel: a consumer that lacks an edge to the producer behind the spin
// reads the producer's target too early -- deterministically, instead of once in fifteen runs.
std::atomic<int> g_delay_side_wgrad_us{0};
constexpr int TF_SPIN_MAX_US = 50000;
__global__ void spin_kernel(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
int launch_spin(int us, hipStream_t st) {
  if (us <= 0) return 0;
  if (us > TF_SPIN_MAX_US) us = TF_SPIN_MAX_US;
  static const int khz = [] {
    int dev = 0, r = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&r, hipDeviceAttributeWallClockRate, dev) != hipSuccess || r <= 0) r = 100000;
    return r;
  }();
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, (long long)us * khz / 1000);
  return (int)hipGetLastError();
}

uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// ---- encoder memory layout ------------------------------------------------------------------------
struct Dims {
  int B, Nv, Nl, d, H, L, ff;
  int S, M, hd, hdp, dp, ffp, nqkv, ldq;
  int Md;           // B * S: token rows of the dense layout.  M = rows that take part: Md, or TfEncoderDesc.packed_rows (packed batches)
  int split;        // fp32-accuracy mode: every bf16 tensor of the workspace and of the weight shadows is a hi + lo plane pair
  int G;            // TfEncoderDesc.groups: G encoders' samples (B / G each) in one batch, each with its own parameters
  // ragged groups (TfEncoderDesc.group_nv): visual tokens per sample of group g, the group's rows inside the concatenated visual tokens
  // (B / G * nv[g]) and its packed rows (those + an equal share of the un-masked language tokens); vis_total = all visual rows
  bool ragged;
  int nv[TF_MAX_GROUPS], vrows[TF_MAX_GROUPS], grows[TF_MAX_GROUPS], vis_total;
};
bool make_dims(int B, int Nv, int Nl, int d, int H, int L, int ff, Dims* o, int split = 0, int packed_rows = 0, int groups = 1,
               const int* group_nv = nullptr) {
  o->split = split ? 1 : 0;
  o->G = groups > 1 ? groups : 1;
  o->ragged = o->G > 1 && group_nv != nullptr && group_nv[0] > 0;
  if (B <= 0 || B % o->G != 0) return false;
  if (!o->ragged && packed_rows % o->G != 0) return false;          // equal groups (packed: every group drops the same tokens)
  if (B <= 0 || Nv < 0 || Nl < 0 || Nv + Nl <= 0 || d <= 0 || H <= 0 || L <= 0 || L > TF_MAX_LAYERS || ff <= 0) return false;
  if (d % H != 0 || d % 8 != 0) return false;
  o->B = B; o->Nv = Nv; o->Nl = Nl; o->d = d; o->H = H; o->L = L; o->ff = ff;
  o->S = Nv + Nl; o->Md = B * o->S; o->M = o->Md;
  o->vis_total = B * Nv;
  for (int g = 0; g < TF_MAX_GROUPS; ++g) { o->nv[g] = g < o->G ? Nv : 0; o->vrows[g] = g < o->G ? B / o->G * Nv : 0; o->grows[g] = 0; }
  if (o->ragged) {
    if (o->G > TF_MAX_GROUPS) return false;
    o->vis_total = 0;
    int nv_max = 0;
    for (int g = 0; g < o->G; ++g) {
      if (group_nv[g] <= 0 || group_nv[g] > Nv) return false;
      o->nv[g] = group_nv[g]; o->vrows[g] = B / o->G * group_nv[g]; o->vis_total += o->vrows[g];
      nv_max = group_nv[g] > nv_max ? group_nv[g] : nv_max;
    }
    if (nv_max != Nv) return false;                    // Nv is the LARGEST group's token count
  }
  if (packed_rows != 0) {                              // packed batches: every visual row and the un-masked language tokens
    if (packed_rows < o->vis_total || packed_rows > o->Md || packed_rows <= 0) return false;
    o->M = packed_rows;
    if (o->ragged) {                                   // every group drops the same language tokens: an equal share each
      const int lang = packed_rows - o->vis_total;
      if (lang % o->G != 0) return false;
      for (int g = 0; g < o->G; ++g) o->grows[g] = o->vrows[g] + lang / o->G;
    }
  }
  o->hd = d / H;
  o->hdp = (int)up(o->hd, 32);
  if (o->hdp > 256) return false;
  o->dp = (int)up((size_t)H * o->hdp, 64);
  o->ffp = (int)up(ff, 64);
  o->nqkv = 3 * H * o->hdp;
  o->ldq = (int)up(o->nqkv, 64);
  if (o->d > 2048) return false;
  if (o->split && o->hdp > 224) return false;      // attn_x3.hip: head dims up to 224
  return true;
}
struct WOff {   // byte offsets inside wpack, per layer
  size_t win, winT, wo, woT, w1, w1T, w2, w2T, bin, bo, b1, b2;
  size_t win8, w18, w28, s_in, s_w1, s_w2;        // fp8 (e4m3) shadows of the three forward projections + per-output-channel scales
  size_t stride;
};
// bytes of ONE plane of a bf16 tensor inside wpack / work: its lo plane (split mode) starts this many bytes after its hi plane
inline size_t plane(size_t bytes) { return up(bytes, 256); }
WOff make_woff(const Dims& D) {
  WOff w; size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += up(bytes, 256); return r; };
  auto take2 = [&](size_t bytes) { size_t r = o; o += plane(bytes) * (size_t)(1 + D.split); return r; };   // bf16 tensor (+ lo plane)
  w.win = take2((size_t)D.nqkv * D.dp * 2);   w.winT = take2((size_t)D.dp * D.ldq * 2);
  w.wo = take2((size_t)D.dp * D.dp * 2);      w.woT = take2((size_t)D.dp * D.dp * 2);
  w.w1 = take2((size_t)D.ffp * D.dp * 2);     w.w1T = take2((size_t)D.dp * D.ffp * 2);
  w.w2 = take2((size_t)D.dp * D.ffp * 2);     w.w2T = take2((size_t)D.ffp * D.dp * 2);
  w.bin = take((size_t)D.ldq * 4); w.bo = take((size_t)D.dp * 4); w.b1 = take((size_t)D.ffp * 4); w.b2 = take((size_t)D.dp * 4);
  w.win8 = take((size_t)D.nqkv * D.dp); w.w18 = take((size_t)D.ffp * D.dp); w.w28 = take((size_t)D.dp * D.ffp);
  w.s_in = take((size_t)D.ldq * 4); w.s_w1 = take((size_t)D.ffp * 4); w.s_w2 = take((size_t)D.dp * 4);
  w.stride = o;
  return w;
}
struct AOff {   // byte offsets inside work
  size_t keymask, zeros, x0, x_stride;            // X[l] = x0 + l * x_stride, l = 0..L
  size_t perr, cu, starts, dense_of, pol;         // packed batches: mismatch word, cu[B+1] (by position), start_of[B] (by sample), dense_of[Md],
                                                  // packed_of_lang[B*Nl] (int32)
  size_t layer0, layer_stride;                    // per-layer block
  size_t qkv, o, lse, z1, mean1, rstd1, x1, u, h, z2, mean2, rstd2, dbits;   // offsets inside a layer block
  size_t meanf, rstdf;
  size_t dxa, dxb, dz, dy, dzb, dyb, du, d_o, dqkv, delta;
  size_t dz1, dy1, dzb1, dyb1, du1, dqkv1;        // the same six for odd layers
  size_t dsw;                                     // dS tiles of the attention backward (TfAttnArgs.ds_work)
  size_t a8, sa8;                                 // fp8 copy of the current GEMM input [M, max(dp, ffp)] bytes + per-token scales
  size_t bskip;                                   // block-sparse tile maps of attn_block_bits: skip_q | skip_k, ceil(S/128) u64 words each
  size_t visrows;                                 // ragged groups: packed row of every token of the concatenated visual tokens (int32, <= B * Nv)
  size_t total;
};
AOff make_aoff(const Dims& D) {
  AOff a; size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += up(bytes, 256); return r; };
  auto take2 = [&](size_t bytes) { size_t r = o; o += plane(bytes) * (size_t)(1 + D.split); return r; };      // bf16 tensor (+ lo plane)
  const size_t md = (size_t)D.M * D.dp * 2, mf = (size_t)D.M * D.ffp * 2, mq = (size_t)D.M * D.ldq * 2;
  const size_t st = (size_t)D.B * D.H * D.S * 4, mr = (size_t)D.M * 4;
  // (everything up to x0 is sized by the DENSE shape: these offsets do not move with the packed row count)
  a.zeros = take(256);
  a.perr = take(256);
  a.bskip = take((size_t)2 * ((D.S + 127) / 128) * 8);
  a.cu = take((size_t)(D.B + 1) * 4); a.starts = take((size_t)D.B * 4); a.dense_of = take((size_t)D.Md * 4); a.pol = take((size_t)D.B * (D.Nl > 0 ? D.Nl : 1) * 4);
  a.visrows = take((size_t)D.B * (D.Nv > 0 ? D.Nv : 1) * 4);
  a.keymask = take((size_t)D.Md);
  a.x0 = o; a.x_stride = plane(md) * (size_t)(1 + D.split); o += a.x_stride * (D.L + 1);
  a.layer0 = o;
  {
    size_t lo = 0;
    auto ltake = [&](size_t bytes) { size_t r = lo; lo += up(bytes, 256); return r; };
    auto ltake2 = [&](size_t bytes) { size_t r = lo; lo += plane(bytes) * (size_t)(1 + D.split); return r; };
    a.qkv = ltake2(mq); a.o = ltake2(md); a.lse = ltake(st); a.z1 = ltake2(md); a.mean1 = ltake(mr); a.rstd1 = ltake(mr);
    a.x1 = ltake2(md); a.u = ltake2(mf); a.h = ltake2(mf); a.z2 = ltake2(md); a.mean2 = ltake(mr); a.rstd2 = ltake(mr);
    a.dbits = ltake(tf_attn_dropmask_bytes(D.B, D.H, D.S));
    a.layer_stride = lo;
  }
  o += a.layer_stride * D.L;
  a.meanf = take((size_t)D.B * (D.Nv > 0 ? D.Nv : 1) * 4); a.rstdf = take((size_t)D.B * (D.Nv > 0 ? D.Nv : 1) * 4);
  a.dxa = take2(md); a.dxb = take2(md); a.dz = take2(md); a.dy = take2(md); a.dzb = take2(md); a.dyb = take2(md); a.du = take2(mf); a.d_o = take2(md);
  a.dqkv = take2(mq);
  a.delta = take(st);
  // What a layer's weight gradients read -- dz / dy, dzb / dyb, du, dqkv -- exists twice (set 0: even layers, set 1: odd layers): the
  // gradients of layer l are launched together at the end of the layer's backward and run beside the chain of layer l - 1, which
  // fills the other set.
  a.dz1 = take2(md); a.dy1 = take2(md); a.dzb1 = take2(md); a.dyb1 = take2(md); a.du1 = take2(mf); a.dqkv1 = take2(mq);
  a.dsw = take((D.split ? 4 : 1) * tf_attn_ds_bytes(D.B, D.H, D.S));      // fp32-accuracy mode: hi + lo planes of dS and of Pd
  a.a8 = take((size_t)D.M * (D.ffp > D.dp ? D.ffp : D.dp)); a.sa8 = take(mr);
  a.total = o;
  return a;
}

// a bf16 tensor of the runtime: hi plane, lo plane (null in bf16 mode), leading dimension
struct Buf { const void* p; const void* lo; int ld; };
struct Ctx {
  Dims D; WOff W; AOff A;
  const TfEncoderDesc* e;
  unsigned char* wp; unsigned char* wk;
  hipStream_t st;
  size_t pd, pf, pq;                                  // plane sizes of [M, dp], [M, ffp], [M, ldq] activations
  const void* lo(const void* p, size_t plane_bytes) const { return D.split ? (const unsigned char*)p + plane_bytes : nullptr; }
  Buf act_d(const void* p) const { return Buf{p, lo(p, pd), D.dp}; }
  Buf act_f(const void* p) const { return Buf{p, lo(p, pf), D.ffp}; }
  Buf act_q(const void* p) const { return Buf{p, lo(p, pq), D.ldq}; }
  // weight shadow [rows, ld] bf16 inside a layer's wpack block
  Buf wgt(const unsigned char* p, int rows, int ld) const { return Buf{p, lo(p, plane((size_t)rows * ld * 2)), ld}; }
  void* X(int l) const { return wk + A.x0 + (size_t)l * A.x_stride; }
  bool packed() const { return e->packed_rows > 0; }
  const int* cu() const { return packed() ? (const int*)(wk + A.cu) : nullptr; }                // attention: rows of the sample in position p (longest first)
  const int* starts() const { return packed() ? (const int*)(wk + A.starts) : nullptr; }        // first packed row of sample b
  const int* dense_of() const { return packed() ? (const int*)(wk + A.dense_of) : nullptr; }    // packed row -> b * S + s
  const int* pol() const { return packed() ? (const int*)(wk + A.pol) : nullptr; }              // language token (b, j) -> packed row or -1
  unsigned char* LB(int l) const { return wk + A.layer0 + (size_t)l * A.layer_stride; }
  unsigned char* WB(int l) const { return wp + (size_t)l * W.stride; }
  long long wg() const { return (long long)(W.stride * (size_t)D.L); }        // bytes between the shadow blocks of two groups
  long long pg() const { return D.G > 1 ? e->param_gstride : 0; }             // bytes between the parameters (and gradients) of two groups
  // ragged groups: the groups' packed row counts / visual row counts into a launch's group_rows (left all zero otherwise: equal ranges)
  void rows_into(int (&dst)[TF_MAX_GROUPS]) const { if (D.ragged) for (int g = 0; g < TF_MAX_GROUPS; ++g) dst[g] = D.grows[g]; }
  void vrows_into(int (&dst)[TF_MAX_GROUPS]) const { if (D.ragged) for (int g = 0; g < TF_MAX_GROUPS; ++g) dst[g] = D.vrows[g]; }
  const int* vis_rows() const { return D.ragged ? (const int*)(wk + A.visrows) : nullptr; }   // concatenated visual token -> packed row
};
bool make_ctx(const TfEncoderDesc* e, hipStream_t st, Ctx* c) {
  if (e == nullptr || e->wpack == nullptr || e->work == nullptr) return false;
  if (!make_dims(e->B, e->Nv, e->Nl, e->d, e->H, e->L, e->ff, &c->D, e->precision, e->packed_rows, e->groups, e->group_nv)) return false;
  if (c->D.G > 1 && (e->param_gstride <= 0 || (e->param_gstride & 15) != 0 || e->attn_block_bits != nullptr)) return false;
  if (c->D.ragged && e->packed_rows <= 0) return false;               // ragged groups exist on packed rows only
  if (e->precision != 0 && e->precision != 1) return false;
  if (e->precision && e->fp8_proj) return false;      // fp8 operands have no lo plane
  if (e->act != 0 && e->act != 1) return false;
  c->W = make_woff(c->D); c->A = make_aoff(c->D);
  c->e = e; c->wp = (unsigned char*)e->wpack; c->wk = (unsigned char*)e->work; c->st = st;
  c->pd = plane((size_t)c->D.M * c->D.dp * 2); c->pf = plane((size_t)c->D.M * c->D.ffp * 2); c->pq = plane((size_t)c->D.M * c->D.ldq * 2);
  return true;
}

struct Drop { unsigned thr, key; float scale; };
Drop drop_for(const TfEncoderDesc* e, float p, unsigned site) {
  Drop d{0u, 0u, 1.f};
  if (e->training && p > 0.f) { d.thr = tf_drop_threshold(p); d.key = tf_drop_key(e->seed, site); d.scale = tf_drop_scale(p); }
  return d;
}
enum Site { SITE_PATCH = 0, SITE_ATTN = 1, SITE_DROP1 = 2, SITE_FFN = 3, SITE_DROP2 = 4 };
inline unsigned site_of(int layer, int which) { return 16u + (unsigned)layer * 8u + (unsigned)which; }

const Buf NOBUF{nullptr, nullptr, 0};
int gemm(const Ctx& c, Buf A, Buf W, Buf C, const float* bias, Buf R, Buf C2, int N, int K, int epi, Drop dr) {
  TfGemmArgs g{};
  g.A = A.p; g.lda = A.ld; g.W = W.p; g.ldw = W.ld; g.C = (void*)C.p; g.ldc = C.ld; g.bias = bias; g.R = R.p; g.ldr = R.ld;
  g.C2 = (void*)C2.p; g.ldc2 = C2.ld;
  g.A_lo = A.lo; g.W_lo = W.lo; g.C_lo = (void*)C.lo; g.R_lo = R.lo; g.C2_lo = (void*)C2.lo;
  g.M = c.D.M; g.N = N; g.K = K; g.epilogue = epi; g.drop_thr = dr.thr; g.drop_key = dr.key; g.drop_scale = dr.scale; g.act = c.e->act;
  g.groups = c.D.G; g.w_gstride = c.wg();            // (every W / bias of the runtime lives in a layer's shadow block)
  c.rows_into(g.group_rows);
  return tf_launch_gemm_nt(&g, c.st);
}
// forward projection with fp8 operands: quantise the bf16 activation per token, then the fp8 large-tile GEMM
int gemm_fp8(const Ctx& c, const void* A, int lda, int K, const void* W8, const float* sw, void* C, int ldc, const float* bias, const void* R,
             int ldr, void* C2, int ldc2, int N, int epi, Drop dr) {
  unsigned char* a8 = c.wk + c.A.a8; float* sa = (float*)(c.wk + c.A.sa8);
  int rc = tf_launch_quant_rows_fp8(A, lda, a8, K, sa, c.D.M, K, c.st);
  if (rc != 0) return rc;
  TfGemmArgs g{};
  g.A = a8; g.lda = K; g.W = W8; g.ldw = K; g.C = C; g.ldc = ldc; g.bias = bias; g.R = R; g.ldr = ldr; g.C2 = C2; g.ldc2 = ldc2;
  g.M = c.D.M; g.N = N; g.K = K; g.epilogue = epi; g.drop_thr = dr.thr; g.drop_key = dr.key; g.drop_scale = dr.scale; g.act = c.e->act;
  g.fp8 = 1; g.scale_a = sa; g.scale_w = sw;
  g.groups = c.D.G; g.w_gstride = c.wg();
  c.rows_into(g.group_rows);
  return tf_launch_gemm_nt(&g, c.st);
}
// Side stream of one tf_encoder_bwd call: the weight gradients.  A layer's four weight-gradient products read tensors the chain has
// produced by the end of the layer's backward (dy / dz and h, du and x1, dyb / dzb and o, dqkv and x); they are launched TOGETHER
// (tf_launch_wgrad_multi: 144 output tiles at d = 768 fill the chip at three row chunks, where each product alone needed 5 - 14 and
// merged them with that many times |dW| of fp32 atomics) on the side stream and run beside the chain of the layer below, which
// writes the OTHER set of dy / dz / du / dyb / dzb / dqkv (AOff: set = layer parity).  An event record or wait on the chain is a
// barrier packet (~5 us of dispatch overlap), so a layer has one fork, one done event (side stream, by parity) and one guard: layer l
// waits for the gradients of layer l + 2 before it overwrites their operands.  The LAST layer of a backward has no chain left to hide
// behind: its products start as early as their operands exist (wgrad_plan).
enum { EV_FORK0 = 0, EV_FORK1 = 1, EV_FORK2 = 2, EV_DONE0 = 4 };           // fork events by position, done events EV_DONE0 + parity
enum { W_W2 = 1, W_W1 = 2, W_WO = 4, W_WI = 8 };
struct Side {
  hipStream_t st = nullptr; hipEvent_t* ev = nullptr;
  bool pending[2] = {false, false};       // a done event of this parity is recorded and the chain has not waited for it
  int last = 0;                           // parity of the most recently recorded done event (FIFO: it covers the other one)
};
// which products start at which point of a layer's backward chain: [0] after the FFN-down dgrad, [1] after the LN1 backward,
// [2] after the attention backward
struct WPlan { int at[3]; };
int ncus() {
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
  }();
  return n;
}
WPlan wgrad_plan(bool last_layer) {
  // hex digits, lowest first: the products started at positions 0, 1, 2.  A product can start only where its operands exist (W2 | W1
  // from position 0, WO from position 1, WI at position 2); whatever the digits leave out starts at position 2.
  static const int body = TF_ENV_INT("TF_WGM_BODY", 0x000), tail = TF_ENV_INT("TF_WGM_TAIL", 0x843);
  const int v = last_layer ? tail : body;
  WPlan p{{v & (W_W2 | W_W1), (v >> 4) & (W_W2 | W_W1 | W_WO), 0}};
  p.at[1] &= ~p.at[0];
  p.at[2] = (W_W2 | W_W1 | W_WO | W_WI) & ~(p.at[0] | p.at[1]);
  return p;
}
int side_fork(const Ctx& c, Side& sd, int fork_ev) {            // what the chain has produced so far is visible to the side stream
  if (sd.st == nullptr) return 0;
  int rc = (int)hipEventRecord(sd.ev[fork_ev], c.st);
  if (rc == 0) rc = (int)hipStreamWaitEvent(sd.st, sd.ev[fork_ev], 0);
  return rc;
}
int side_done(Side& sd, int parity) {
  if (sd.st == nullptr) return 0;
  sd.pending[parity] = true; sd.last = parity;
  return (int)hipEventRecord(sd.ev[EV_DONE0 + parity], sd.st);
}
int guard(const Ctx& c, Side& sd, int parity) {                 // the chain is about to overwrite what the gradients of this parity read
  if (sd.st != nullptr && sd.pending[parity]) { sd.pending[parity] = false; return (int)hipStreamWaitEvent(c.st, sd.ev[EV_DONE0 + parity], 0); }
  return 0;
}
TfWgradArgs wjob(const Ctx& c, Buf dY, int N, Buf X, int K, float* dW, int lddw, float* db, int rg, int rgp, int n_src, int cg, int cgp, int k_src) {
  TfWgradArgs w{};
  w.dY = dY.p; w.ldy = dY.ld; w.X = X.p; w.ldx = X.ld; w.dW = dW; w.lddw = lddw; w.db = db; w.zeros = c.wk + c.A.zeros;
  w.dY_lo = dY.lo; w.X_lo = X.lo;
  w.M = c.D.M; w.N = N; w.K = K; w.rg = rg; w.rgp = rgp; w.n_src = n_src; w.cg = cg; w.cgp = cgp; w.k_src = k_src; w.m_chunk = 0;
  w.groups = c.D.G; w.dw_gstride = c.pg();
  c.rows_into(w.group_rows);
  return w;
}
// the products `mask` selects out of jobs[0..3] (W_W2, W_W1, W_WO, W_WI) as one launch on the side stream (or on the chain without one)
int wgrad_launch(const Ctx& c, Side& sd, const TfWgradArgs* jobs, int mask, bool alone) {
  TfWgradArgs sel[4]; int n = 0;
  for (int i = 0; i < 4; ++i) if ((mask >> i) & 1) sel[n++] = jobs[i];
  if (n == 0) return 0;
  hipStream_t st = sd.st != nullptr ? sd.st : c.st;
  if (sd.st != nullptr) { const int rc = launch_spin(g_delay_side_wgrad_us.load(std::memory_order_relaxed), st); if (rc != 0) return rc; }
  // sizing is the launcher's: 0 = the launch has the chip to itself, -1 = it runs beside the chain
  const int blocks = (sd.st != nullptr && !alone) ? -1 : 0;
  if (n * c.D.G > TF_WGRAD_MULTI_MAX) {                        // more groups than one launch takes: one launch per product
    for (int i = 0; i < n; ++i) { const int rc = tf_launch_wgrad_multi(&sel[i], 1, blocks, st); if (rc != 0) return rc; }
    return 0;
  }
  return tf_launch_wgrad_multi(sel, n, blocks, st);
}

}  // namespace

// =====================================================================================================
extern "C" {

int tf_version(void) { return TF_ABI_VERSION; }

int tf_trace_start(void) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  for (auto& r : g_trace) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_trace.clear();
  g_trace_on = true;
  return 0;
}
long long tf_trace_stop(TfTraceRecord* out, long long cap) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace_on = false;
  TF_TRY((int)hipDeviceSynchronize(), "tf_trace_stop(sync)");
  long long n = 0;
  for (auto& r : g_trace) {
    if (out != nullptr && n < cap) {
      TfTraceRecord& o = out[n];
      memset(&o, 0, sizeof(o));
      strncpy(o.name, r.name, sizeof(o.name) - 1);
      float ms = 0.f, ms0 = 0.f;
      (void)hipEventElapsedTime(&ms, r.e0, r.e1);
      (void)hipEventElapsedTime(&ms0, g_trace.front().e0, r.e0);
      o.us = ms * 1e3f; o.start_us = ms0 * 1e3f; o.flops = r.flops; o.bytes = r.bytes;
      for (hipStream_t sd : g_trace_sides) if (sd == r.st) o.side = 1;
    }
    ++n;
  }
  for (auto& r : g_trace) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }   // (the first record's e0 is every record's time base)
  g_trace.clear();
  return n;
}

int tf_overlap_create(TfOverlap* o) {
  if (o == nullptr) return fail(-1, "tf_overlap_create");
  memset(o, 0, sizeof(*o));
  hipStream_t st = nullptr;
  // TF_SIDE_PRIORITY (experiment): -1 = the device's highest stream priority, 1 = its lowest, unset / 0 = default.  The lowest priority
  // is worth -0.6 % on the single-encoder benchmark (4.235 vs 4.26 ms, same box) and is a disaster for the wrapper, whose four level
  // streams then starve their four side streams (B = 4: 11.5 vs 6.35 ms per step): the default priority stays.
  static const int prio_sel = TF_ENV_INT("TF_SIDE_PRIORITY", 0);
  // TF_SIDE_CUS (experiment): the side stream confined to n of the 256 CUs by a CU mask (the KFD deals mask bits round-robin over
  // the XCDs: the low n bits are n / 8 CUs of every XCD), so that the weight-gradient workgroups leave whole CUs to the chain
  static const int side_cus = TF_ENV_INT("TF_SIDE_CUS", 0), side_cu0 = TF_ENV_INT("TF_SIDE_CU0", 0);
  if (side_cus > 0 && side_cus < 256) {
    uint32_t mask[8] = {};
    for (int i = 0; i < side_cus; ++i) { const int b = (i + side_cu0) % 256; mask[b / 32] |= 1u << (b % 32); }
    TF_TRY((int)hipExtStreamCreateWithCUMask(&st, 8, mask), "tf_overlap_create(CU-masked stream)");
  } else if (prio_sel != 0) {
    int least = 0, greatest = 0;
    TF_TRY((int)hipDeviceGetStreamPriorityRange(&least, &greatest), "tf_overlap_create(priority range)");
    TF_TRY((int)hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_sel < 0 ? greatest : least), "tf_overlap_create(stream)");
  } else {
    TF_TRY((int)hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "tf_overlap_create(stream)");
  }
  o->stream = st;
  tf_trace_mark_side(st);
  for (int i = 0; i < 8; ++i) {
    hipEvent_t ev = nullptr;
    TF_TRY((int)hipEventCreateWithFlags(&ev, TF_EVENT_FLAGS), "tf_overlap_create(event)");
    o->ev[i] = ev;
  }
  return 0;
}
int tf_overlap_join(TfOverlap* o, tf_stream_t s) {
  if (o == nullptr) return fail(-1, "tf_overlap_join");
  if (o->stream == nullptr || o->pending == 0u) return 0;
  // FIFO side stream: the most recently recorded done event (its parity is kept in `reserved`) covers every earlier one
  TF_TRY((int)hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)o->ev[EV_DONE0 + (int)(o->reserved & 1u)], 0), "tf_overlap_join");
  o->pending = 0u;
  return 0;
}
int tf_overlap_destroy(TfOverlap* o) {
  if (o == nullptr) return fail(-1, "tf_overlap_destroy");
  for (int i = 0; i < 8; ++i) if (o->ev[i] != nullptr) { (void)hipEventDestroy((hipEvent_t)o->ev[i]); o->ev[i] = nullptr; }
  if (o->stream != nullptr) { (void)hipStreamSynchronize((hipStream_t)o->stream); (void)hipStreamDestroy((hipStream_t)o->stream); o->stream = nullptr; }
  return 0;
}
int tf_debug_spin(int us, tf_stream_t s) {
  if (us < 0) return fail(-1, "tf_debug_spin");
  TF_TRY(launch_spin(us, (hipStream_t)s), "tf_debug_spin");
  return 0;
}
int tf_debug_delay_wgrad(int us) {
  if (us < 0) return fail(-1, "tf_debug_delay_wgrad");
  return g_delay_side_wgrad_us.exchange(us > TF_SPIN_MAX_US ? TF_SPIN_MAX_US : us, std::memory_order_relaxed);
}
const char* tf_last_error(void) { return g_err; }
void tf_set_error_msg(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

uint32_t tf_drop_key(uint64_t seed, uint32_t site) { return (uint32_t)(splitmix64(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(site + 1))) >> 32); }
uint32_t tf_drop_threshold(float p) {
  if (!(p > 0.f)) return 0u;
  double t = floor((double)p * 65536.0 + 0.5);
  if (t >= 65535.0) return 65535u;
  if (t < 1.0) return 1u;
  return (uint32_t)t;
}
float tf_drop_scale(float p) {
  const uint32_t t = tf_drop_threshold(p);
  return t ? (float)(1.0 / (1.0 - (double)t / 65536.0)) : 1.0f;
}
size_t tf_attn_dropmask_bytes(int B, int H, int S) { return (size_t)B * H * S * ((S + 63) / 64) * 8; }
int tf_attn_dropmask(void* bits, int B, int H, int S, uint32_t key, uint32_t thr, tf_stream_t s) {
  if (bits == nullptr) return fail(-1, "tf_attn_dropmask");
  TF_TRY(tf_launch_attn_dropmask(bits, B, H, S, key, thr, (hipStream_t)s), "tf_attn_dropmask");
  return 0;
}

int tf_attn_dropmask_rows(void* bits, long long nrows, int S, uint32_t key, uint32_t thr, tf_stream_t s) {
  if (bits == nullptr || nrows < 0 || S <= 0) return fail(-1, "tf_attn_dropmask_rows");
  TF_TRY(tf_launch_attn_dropmask_rows(bits, nrows, S, key, thr, (hipStream_t)s), "tf_attn_dropmask_rows");
  return 0;
}

#define TF_WRAP(name, call) do { if (a == nullptr) return fail(-1, name); TF_TRY(call, name); return 0; } while (0)
int tf_quant_rows_fp8(const void* src, int ld_src, void* dst, int ld_dst, float* scale, int rows, int cols, tf_stream_t s) {
  if (src == nullptr || dst == nullptr || scale == nullptr) return fail(-1, "tf_quant_rows_fp8");
  TF_TRY(tf_launch_quant_rows_fp8(src, ld_src, dst, ld_dst, scale, rows, cols, (hipStream_t)s), "tf_quant_rows_fp8");
  return 0;
}
int tf_gemm_fwd(const TfGemmArgs* a, tf_stream_t s) { TF_WRAP("tf_gemm_fwd", tf_launch_gemm_nt(a, (hipStream_t)s)); }
int tf_gemm_wgrad(const TfWgradArgs* a, tf_stream_t s) { TF_WRAP("tf_gemm_wgrad", tf_launch_wgrad_tn(a, (hipStream_t)s)); }
int tf_gemm_wgrad_multi(const TfWgradArgs* probs, int count, int blocks, tf_stream_t s) {
  if (probs == nullptr || count <= 0) return fail(-1, "tf_gemm_wgrad_multi");
  TF_TRY(tf_launch_wgrad_multi(probs, count, blocks, (hipStream_t)s), "tf_gemm_wgrad_multi");
  return 0;
}
int tf_attn_fwd(const TfAttnArgs* a, tf_stream_t s) { TF_WRAP("tf_attn_fwd", tf_launch_attn_fwd(a, (hipStream_t)s)); }
int tf_attn_bwd(const TfAttnArgs* a, tf_stream_t s) {
  if (a == nullptr) return fail(-1, "tf_attn_bwd");
  TF_TRY(tf_launch_attn_bwd(a, (hipStream_t)s), "tf_attn_bwd");
  return 0;
}
int tf_layernorm_fwd(const TfLnArgs* a, tf_stream_t s) { TF_WRAP("tf_layernorm_fwd", tf_launch_ln_fwd(a, (hipStream_t)s)); }
int tf_layernorm_bwd(const TfLnArgs* a, tf_stream_t s) { TF_WRAP("tf_layernorm_bwd", tf_launch_ln_bwd(a, (hipStream_t)s)); }
int tf_assemble_fwd(const TfAssembleArgs* a, tf_stream_t s) { TF_WRAP("tf_assemble_fwd", tf_launch_assemble_fwd(a, (hipStream_t)s)); }
int tf_assemble_bwd(const TfAssembleArgs* a, tf_stream_t s) { TF_WRAP("tf_assemble_bwd", tf_launch_assemble_bwd(a, (hipStream_t)s)); }
int tf_patchify_fwd(const TfPatchArgs* a, tf_stream_t s) { TF_WRAP("tf_patchify_fwd", tf_launch_im2col(a, (hipStream_t)s)); }
int tf_patchify_bwd(const TfPatchArgs* a, int f32, tf_stream_t s) { TF_WRAP("tf_patchify_bwd", tf_launch_col2im(a, f32, (hipStream_t)s)); }
int tf_regroup_fwd(const TfPatchArgs* a, int f32, tf_stream_t s) { TF_WRAP("tf_regroup_fwd", tf_launch_col2im(a, f32, (hipStream_t)s)); }
int tf_regroup_bwd(const TfPatchArgs* a, tf_stream_t s) { TF_WRAP("tf_regroup_bwd", tf_launch_im2col(a, (hipStream_t)s)); }
int tf_attn_block_skip(const void* bits, int S, void* skip_q, void* skip_k, tf_stream_t s) {
  TF_TRY(tf_launch_attn_block_skip(bits, S, skip_q, skip_k, (hipStream_t)s), "tf_attn_block_skip");
  return 0;
}
int tf_split_planes(const TfPlanesArgs* a, tf_stream_t s) { TF_WRAP("tf_split_planes", tf_launch_split_planes(a, (hipStream_t)s)); }
int tf_pack_weight(const TfPackArgs* a, tf_stream_t s) { TF_WRAP("tf_pack_weight", tf_launch_pack(a, (hipStream_t)s)); }
int tf_copy_rows(const TfCopyRowsArgs* a, tf_stream_t s) { TF_WRAP("tf_copy_rows", tf_launch_copy_rows(a, (hipStream_t)s)); }
// ---- the step clock -------------------------------------------------------------------------------------------------------
namespace {
unsigned* g_clock_word = nullptr;
std::mutex g_clock_mu;
}
const uint32_t* tf_clock_ptr(void) {
  std::lock_guard<std::mutex> lk(g_clock_mu);
  if (g_clock_word == nullptr) {
    unsigned* w = nullptr;
    if (hipMalloc((void**)&w, 256) != hipSuccess) { (void)fail(-9, "tf_clock_ptr (hipMalloc)"); return nullptr; }
    if (hipMemset(w, 0, 256) != hipSuccess || tf_tu_set_clock_gemm(w) != 0 || tf_tu_set_clock_rowops(w) != 0) {
      (void)hipFree(w);
      (void)fail(-9, "tf_clock_ptr (publishing the clock word)");
      return nullptr;
    }
    g_clock_word = w;
  }
  return g_clock_word;
}
int tf_clock_advance(uint32_t by, tf_stream_t s) {
  unsigned* w = (unsigned*)tf_clock_ptr();
  if (w == nullptr) return -9;
  TF_TRY(tf_launch_clock_advance(w, by, (hipStream_t)s), "tf_clock_advance");
  return 0;
}
int tf_clock_set(uint32_t value, tf_stream_t s) {
  unsigned* w = (unsigned*)tf_clock_ptr();
  if (w == nullptr) return -9;
  TF_TRY((int)hipMemsetAsync(w, 0, sizeof(unsigned), (hipStream_t)s), "tf_clock_set");
  if (value != 0) TF_TRY(tf_launch_clock_advance(w, value, (hipStream_t)s), "tf_clock_set");
  return 0;
}
int tf_radam_step(const TfRadamArgs* a, tf_stream_t s) { TF_WRAP("tf_radam_step", tf_launch_radam(a, (hipStream_t)s)); }
int tf_heads_loss_fwd(const TfHeadsLossArgs* a, tf_stream_t s) { TF_WRAP("tf_heads_loss_fwd", tf_launch_heads_loss_fwd(a, (hipStream_t)s)); }
int tf_heads_loss_bwd(const TfHeadsLossArgs* a, tf_stream_t s) { TF_WRAP("tf_heads_loss_bwd", tf_launch_heads_loss_bwd(a, (hipStream_t)s)); }
int tf_softplus_col(const void* x, const void* x_lo, int ld, int col, float* y, const float* dy, void* dx, void* dx_lo, int R, tf_stream_t s) {
  TF_TRY(tf_launch_softplus_col(x, x_lo, ld, col, y, dy, dx, dx_lo, R, (hipStream_t)s), "tf_softplus_col");
  return 0;
}
int tf_pool_norm_fwd(const TfPoolNormArgs* a, tf_stream_t s) { TF_WRAP("tf_pool_norm_fwd", tf_launch_pool_norm_fwd(a, (hipStream_t)s)); }
int tf_pool_norm_bwd(const TfPoolNormArgs* a, tf_stream_t s) { TF_WRAP("tf_pool_norm_bwd", tf_launch_pool_norm_bwd(a, (hipStream_t)s)); }
int tf_lm_pool_fwd(const TfLmPoolArgs* a, tf_stream_t s) { TF_TRY(tf_launch_lm_pool_fwd(a, (hipStream_t)s), "tf_lm_pool_fwd"); return 0; }
int tf_lm_pool_bwd(const TfLmPoolArgs* a, tf_stream_t s) { TF_TRY(tf_launch_lm_pool_bwd(a, (hipStream_t)s), "tf_lm_pool_bwd"); return 0; }
int tf_sumsq(const float* x, long long n, float* out, tf_stream_t s) { TF_TRY(tf_launch_sumsq(x, n, out, (hipStream_t)s), "tf_sumsq"); return 0; }
int tf_sumsq_set(const float* x, long long n, float* out, tf_stream_t s) { TF_TRY(tf_launch_sumsq_ex(x, n, out, 0, (hipStream_t)s), "tf_sumsq_set"); return 0; }
int tf_sq_loss_fwd(const TfSqLossArgs* a, tf_stream_t s) {
  if (a == nullptr) return fail(-1, "tf_sq_loss_fwd");
  TF_TRY(tf_launch_sq_loss(a, 0, (hipStream_t)s), "tf_sq_loss_fwd");
  return 0;
}
int tf_sq_loss_bwd(const TfSqLossArgs* a, tf_stream_t s) {
  if (a == nullptr) return fail(-1, "tf_sq_loss_bwd");
  TF_TRY(tf_launch_sq_loss(a, 1, (hipStream_t)s), "tf_sq_loss_bwd");
  return 0;
}
int tf_dropout_mask(uint8_t* out, long long n, uint32_t key, uint32_t thr, tf_stream_t s) {
  TF_TRY(tf_launch_dropout_mask(out, n, key, thr, (hipStream_t)s), "tf_dropout_mask"); return 0;
}
int tf_dropout_apply(const void* x, void* y, long long n, uint32_t key, uint32_t thr, float scale, tf_stream_t s) {
  TF_TRY(tf_launch_dropout_apply(x, y, n, key, thr, scale, (hipStream_t)s), "tf_dropout_apply"); return 0;
}
int tf_cast_f32_bf16(const float* src, void* dst, long long n, tf_stream_t s) { TF_TRY(tf_launch_cast_f32_bf16(src, dst, n, (hipStream_t)s), "tf_cast_f32_bf16"); return 0; }
int tf_cast_bf16_f32(const void* src, float* dst, long long n, tf_stream_t s) { TF_TRY(tf_launch_cast_bf16_f32(src, dst, n, (hipStream_t)s), "tf_cast_bf16_f32"); return 0; }

// ---- encoder runtime --------------------------------------------------------------------------------
int tf_encoder_plan(int B, int Nv, int Nl, int d, int H, int L, int ff, TfEncoderPlan* out) {
  Dims D;
  if (out == nullptr || !make_dims(B, Nv, Nl, d, H, L, ff, &D)) return fail(-1, "tf_encoder_plan");
  const WOff W = make_woff(D); const AOff A = make_aoff(D);
  out->hd = D.hd; out->hdp = D.hdp; out->dp = D.dp; out->ffp = D.ffp; out->ldq = D.ldq; out->S = D.S; out->M = D.M;
  out->wpack_bytes = W.stride * (size_t)D.L; out->work_bytes = A.total;
  return 0;
}

}  // extern "C"
TfTraceScope::TfTraceScope(const char* name, hipStream_t stream, double flops, double bytes) : idx(-1), st(stream) {
  if (!g_trace_on) return;
  TraceRec r{};
  strncpy(r.name, name, sizeof(r.name) - 1);
  r.st = stream; r.flops = flops; r.bytes = bytes;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  (void)hipEventRecord(r.e0, stream);
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace.push_back(r);
  idx = (long long)g_trace.size() - 1;
}
TfTraceScope::~TfTraceScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  if (idx < (long long)g_trace.size()) (void)hipEventRecord(g_trace[idx].e1, st);
}
void tf_trace_mark_side(hipStream_t side) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  for (hipStream_t s : g_trace_sides) if (s == side) return;
  g_trace_sides.push_back(side);
}
namespace {
// bf16 shadows of layer l, for ALL groups of a grouped call in one launch (blockIdx.y = group: the same tensors, param_gstride bytes
// further in the parameters, wg() bytes further in wpack)
int pack_layer(const Ctx& c, int l, hipStream_t st) {
  const Dims& D = c.D;
  unsigned char* w = c.WB(l);
  const TfLayerParams p = c.e->p[l];
  for (int residual = 0; residual <= D.split; ++residual) {      // split mode: a second launch writes the lo planes, bf16(w - bf16(w))
    TfPackArgs batch[8];
    int nb = 0;
    auto pack = [&](const float* src, int rows, int cols, unsigned char* dst, int ld, unsigned char* dstT, int ldT, int rows_p, int cols_p, int rg,
                    int rgp, int cg, int cgp, int f32) {
      if (f32 && residual) return;                               // biases are fp32: no lo plane
      TfPackArgs a{};
      a.src = src; a.rows = rows; a.cols = cols; a.ld_dst = ld; a.ld_dst_t = ldT; a.rows_p = rows_p;
      a.cols_p = cols_p; a.rg = rg; a.rgp = rgp; a.cg = cg; a.cgp = cgp; a.dst_is_f32 = f32; a.residual = residual;
      a.dst = residual ? dst + plane((size_t)rows_p * ld * 2) : dst;
      a.dst_t = dstT == nullptr ? nullptr : (residual ? dstT + plane((size_t)cols_p * ldT * 2) : dstT);
      batch[nb++] = a;
    };
    pack(p.in_w, 3 * D.d, D.d, w + c.W.win, D.dp, w + c.W.winT, D.ldq, D.nqkv, D.dp, D.hd, D.hdp, BIG, BIG, 0);
    pack(p.out_w, D.d, D.d, w + c.W.wo, D.dp, w + c.W.woT, D.dp, D.dp, D.dp, BIG, BIG, D.hd, D.hdp, 0);
    pack(p.w1, D.ff, D.d, w + c.W.w1, D.dp, w + c.W.w1T, D.ffp, D.ffp, D.dp, BIG, BIG, BIG, BIG, 0);
    pack(p.w2, D.d, D.ff, w + c.W.w2, D.ffp, w + c.W.w2T, D.dp, D.dp, D.ffp, BIG, BIG, BIG, BIG, 0);
    pack(p.in_b, 1, 3 * D.d, w + c.W.bin, D.ldq, nullptr, 0, 1, D.nqkv, BIG, BIG, D.hd, D.hdp, 1);
    pack(p.out_b, 1, D.d, w + c.W.bo, D.dp, nullptr, 0, 1, D.dp, BIG, BIG, BIG, BIG, 1);
    pack(p.b1, 1, D.ff, w + c.W.b1, D.ffp, nullptr, 0, 1, D.ffp, BIG, BIG, BIG, BIG, 1);
    pack(p.b2, 1, D.d, w + c.W.b2, D.dp, nullptr, 0, 1, D.dp, BIG, BIG, BIG, BIG, 1);
    const int rc = tf_launch_pack_batch_groups(batch, nb, D.G, c.pg(), c.wg(), st);
    if (rc != 0) return rc;
  }
  int rc = 0;
  for (int grp = 0; grp < D.G && c.e->fp8_proj && rc == 0; ++grp) {   // fp8 shadows of the forward projections, one scale per output channel
    unsigned char* wg = w + (size_t)grp * c.wg();
    rc = tf_launch_quant_rows_fp8(wg + c.W.win, D.dp, wg + c.W.win8, D.dp, (float*)(wg + c.W.s_in), D.nqkv, D.dp, st);
    if (rc == 0) rc = tf_launch_quant_rows_fp8(wg + c.W.w1, D.dp, wg + c.W.w18, D.dp, (float*)(wg + c.W.s_w1), D.ffp, D.dp, st);
    if (rc == 0) rc = tf_launch_quant_rows_fp8(wg + c.W.w2, D.ffp, wg + c.W.w28, D.ffp, (float*)(wg + c.W.s_w2), D.dp, D.ffp, st);
  }
  return rc;
}
// LayerNorm over all M rows of a [M, dp] activation
void ln_rows(const Ctx& c, TfLnArgs& n, Buf x, const float* gamma, float* mean, float* rstd) {
  n.x = x.p; n.x_lo = x.lo; n.ldx = x.ld; n.gamma = gamma; n.mean = mean; n.rstd = rstd;
  n.rows = c.D.M; n.d = c.D.d; n.rows_per_group = c.D.M; n.x_group_stride = c.D.M; n.y_group_stride = c.D.M; n.eps = 1e-5f;
  n.pgroups = c.D.G; n.p_gstride = c.pg();
  c.rows_into(n.group_rows);
}
// final LayerNorm / copy over the visual rows: with ragged groups the output rows are the concatenated visual tokens, found through the
// row map of this forward (the x side), in ragged parameter groups
void ln_vis_rows(const Ctx& c, TfLnArgs& n) {
  const Dims& D = c.D;
  n.rows = D.B * D.Nv; n.d = D.d; n.rows_per_group = D.Nv; n.x_group_stride = D.S; n.y_group_stride = D.Nv; n.eps = 1e-5f;
  n.x_group_row0 = c.starts();
  n.pgroups = D.G; n.p_gstride = c.pg();
  if (D.ragged) {
    n.rows = D.vis_total; n.rows_per_group = D.vis_total; n.y_group_stride = D.vis_total; n.x_group_row0 = nullptr;
    n.x_row_map = c.vis_rows();
    c.vrows_into(n.group_rows);
  }
}
}  // namespace
extern "C" {

int tf_encoder_plan_ex(const TfEncoderDesc* e, TfEncoderPlan* out) {
  Dims D;
  if (e == nullptr || out == nullptr || (e->precision != 0 && e->precision != 1) ||
      !make_dims(e->B, e->Nv, e->Nl, e->d, e->H, e->L, e->ff, &D, e->precision, 0, e->groups, e->group_nv)) return fail(-1, "tf_encoder_plan_ex");
  const WOff W = make_woff(D); const AOff A = make_aoff(D);
  out->hd = D.hd; out->hdp = D.hdp; out->dp = D.dp; out->ffp = D.ffp; out->ldq = D.ldq; out->S = D.S; out->M = D.M;
  out->wpack_bytes = W.stride * (size_t)D.L * (size_t)D.G; out->work_bytes = A.total;       // one shadow block per group, back to back
  return 0;
}

int tf_encoder_pack(const TfEncoderDesc* e, tf_stream_t s) {
  Ctx c;
  if (!make_ctx(e, (hipStream_t)s, &c)) return fail(-1, "tf_encoder_pack");
  for (int l = 0; l < c.D.L; ++l) TF_TRY(pack_layer(c, l, c.st), "pack layer");
  return 0;
}

int tf_encoder_fwd(const TfEncoderDesc* e, tf_stream_t s) {
  Ctx c;
  if (!make_ctx(e, (hipStream_t)s, &c)) return fail(-1, "tf_encoder_fwd");
  const Dims& D = c.D;
  if (e->vis == nullptr || e->lang == nullptr || e->vis_out == nullptr) return fail(-1, "tf_encoder_fwd(null io)");
  // ---- side stream: everything of this forward that does not depend on activations (see TfOverlap) -- forked FIRST, so that layer 0's
  // weight re-pack runs under the row map and the token assemble instead of ahead of the first GEMM ----
  hipStream_t side = nullptr; hipEvent_t* ev = nullptr;
  if (e->overlap != nullptr && e->overlap->stream != nullptr) { side = (hipStream_t)e->overlap->stream; ev = (hipEvent_t*)e->overlap->ev; }
  // one event per layer on the side stream, recorded after that layer's re-pack AND dropout mask (FIFO: it covers both);
  // layers >= 3 share an event: a wait on it then covers every later record too (correct, less overlap); ev[1]: layer 0's re-pack alone
  auto evi = [](int l) { return 4 + (l < 3 ? l : 3); };
  bool side_work[TF_MAX_LAYERS] = {};
  static const int one_wait_sel = TF_ENV_INT("TF_FWD_ONE_WAIT", 1);
  const bool one_wait = one_wait_sel != 0 && c.D.G == 1 && c.D.L >= 3;
  bool side_later = false;                                 // layers >= 1 have side-stream work
  if (side != nullptr && e->overlap->pending != 0u) TF_TRY(tf_overlap_join(e->overlap, s), "fwd join");   // events are about to be reused
  // the row map (packed batches) comes BEFORE the fork: the side stream's attention dropout masks are generated for the rows and key
  // tiles the samples really have (tf_launch_attn_dropmask_packed reads cu), so they wait for it through the fork event
  uint8_t* km = (uint8_t*)(c.wk + c.A.keymask);
  if (c.packed()) {
    // packed batches: the masked language tokens are never gathered, so every row that exists is a real token -- no key mask
    TF_TRY(tf_launch_row_map(e->lang_pad_mask, D.B, D.Nv, D.Nl, (int*)(c.wk + c.A.cu), (int*)(c.wk + c.A.starts), (int*)(c.wk + c.A.dense_of), (int*)(c.wk + c.A.pol),
                             e->packed_rows, (int*)(c.wk + c.A.perr), D.G, D.ragged ? D.nv : nullptr, (int*)(c.wk + c.A.visrows), e->packed_error_host, c.st), "row_map");
    km = nullptr;
  } else {
    TF_TRY(tf_launch_key_mask(e->lang_pad_mask, km, D.B, D.Nv, D.Nl, c.st), "key_mask");
  }
  if (side != nullptr) {
    TF_TRY((int)hipEventRecord(ev[0], c.st), "fwd fork");        // earlier work on the chain may still use these buffers
    TF_TRY((int)hipStreamWaitEvent(side, ev[0], 0), "fwd fork");
    if (e->repack) {
      TF_TRY(pack_layer(c, 0, side), "pack layer 0");
      TF_TRY((int)hipEventRecord(ev[1], side), "pack 0 event");
    }
    for (int l = 0; l < c.D.L; ++l) {
      const Drop dr = drop_for(e, e->p_token, site_of(l, SITE_ATTN));
      if (e->repack && l >= 1) { TF_TRY(pack_layer(c, l, side), "pack layer"); side_work[l] = true; }
      if (dr.thr) { TF_TRY(tf_launch_attn_dropmask_packed(c.LB(l) + c.A.dbits, c.D.B, c.D.H, c.D.S, c.cu(), dr.key, dr.thr, side), "attn_dropmask"); side_work[l] = true; }
      // one encoder (G = 1): ONE event behind the side work of ALL layers >= 1, waited for once at the top of layer 1 (the side stream is
      // through with it ~150 us before the chain gets there); grouped calls keep an event per layer (their re-packs take longer)
      if (side_work[l] && (!one_wait || l == 0)) TF_TRY((int)hipEventRecord(ev[evi(l)], side), "side event");
      if (l >= 1 && side_work[l]) side_later = true;
    }
    if (one_wait && side_later) TF_TRY((int)hipEventRecord(ev[evi(1)], side), "side event (layers >= 1)");
  } else if (e->repack) {
    for (int l = 0; l < c.D.L; ++l) TF_TRY(pack_layer(c, l, c.st), "pack layer");
  }
  if (e->attn_block_bits != nullptr)              // the block mask's skippable tiles, once per forward (the backward reads them too)
    TF_TRY(tf_launch_attn_block_skip(e->attn_block_bits, D.S, c.wk + c.A.bskip, c.wk + c.A.bskip + (size_t)((D.S + 127) / 128) * 8, c.st), "block_skip");
  {
    TfAssembleArgs a{};
    const Buf x0 = c.act_d(c.X(0));
    a.vis = e->vis; a.vis_is_f32 = e->vis_is_f32; a.ld_vis = D.d; a.lang = e->lang; a.lang_is_f32 = e->lang_is_f32; a.ld_lang = D.d;
    a.pe = e->pe; a.pe_lang = e->pe_lang; a.kind_v = e->kind_v; a.kind_l = e->kind_l; a.out = (void*)x0.p; a.out_lo = (void*)x0.lo; a.ld_out = D.dp;
    a.B = D.B; a.Nv = D.Nv; a.Nl = D.Nl; a.d = D.d;
    a.row_map = c.dense_of(); a.rows = c.packed() ? D.M : 0;
    a.pgroups = D.G; a.p_gstride = c.pg();
    if (D.ragged) { for (int g = 0; g < TF_MAX_GROUPS; ++g) a.group_nv[g] = D.nv[g]; c.rows_into(a.group_rows); }
    const Drop dr = drop_for(e, e->p_patch, SITE_PATCH);
    a.drop_thr = dr.thr; a.drop_key = dr.key; a.drop_scale = dr.scale;
    TF_TRY(tf_launch_assemble_fwd(&a, c.st), "assemble_fwd");
  }
  if (side != nullptr && e->repack) TF_TRY((int)hipStreamWaitEvent(c.st, ev[1], 0), "pack 0 wait");     // before the first GEMM
  const float scale = 1.0f / sqrtf((float)D.hd);
  for (int l = 0; l < D.L; ++l) {
    // layer 0 needs only its mask (before the attention); later layers wait once, at the top, for pack + mask
    if (side != nullptr && l >= 1 && (one_wait ? (l == 1 && side_later) : side_work[l])) TF_TRY((int)hipStreamWaitEvent(c.st, ev[evi(l)], 0), "side wait");
    unsigned char* w = c.WB(l); unsigned char* b = c.LB(l);
    const TfLayerParams& p = e->p[l];
    const Drop none{0u, 0u, 1.f};
    const Buf x = c.act_d(c.X(l)), qkv = c.act_q(b + c.A.qkv), o = c.act_d(b + c.A.o), z1 = c.act_d(b + c.A.z1), x1 = c.act_d(b + c.A.x1);
    const Buf u = c.act_f(b + c.A.u), hh = c.act_f(b + c.A.h), z2 = c.act_d(b + c.A.z2), xn = c.act_d(c.X(l + 1));
    if (e->fp8_proj)
      TF_TRY(gemm_fp8(c, x.p, D.dp, D.dp, w + c.W.win8, (const float*)(w + c.W.s_in), b + c.A.qkv, D.ldq, (const float*)(w + c.W.bin),
                      nullptr, 0, nullptr, 0, D.nqkv, TF_EPI_BIAS, none), "gemm qkv (fp8)");
    else
      TF_TRY(gemm(c, x, c.wgt(w + c.W.win, D.nqkv, D.dp), qkv, (const float*)(w + c.W.bin), NOBUF, NOBUF, D.nqkv, D.dp, TF_EPI_BIAS, none), "gemm qkv");
    {
      TfAttnArgs a{};
      a.qkv = qkv.p; a.qkv_lo = qkv.lo; a.ld_qkv = D.ldq; a.out = (void*)o.p; a.out_lo = (void*)o.lo; a.ld_out = D.dp; a.lse = (float*)(b + c.A.lse);
      a.key_mask = km; a.cu_rows = c.cu(); a.B = D.B; a.S = D.S; a.H = D.H; a.HDP = D.hdp; a.scale = scale;
      const Drop dr = drop_for(e, e->p_token, site_of(l, SITE_ATTN));
      a.drop_thr = dr.thr; a.drop_key = dr.key; a.drop_scale = dr.scale; a.drop_bits = b + c.A.dbits; a.block_bits = e->attn_block_bits;
      if (e->attn_block_bits != nullptr) { a.block_skip_q = c.wk + c.A.bskip; a.block_skip_k = c.wk + c.A.bskip + (size_t)((D.S + 127) / 128) * 8; }
      if (dr.thr && side == nullptr) TF_TRY(tf_launch_attn_dropmask_packed(b + c.A.dbits, D.B, D.H, D.S, c.cu(), dr.key, dr.thr, c.st), "attn_dropmask");
      if (dr.thr && side != nullptr && l == 0) TF_TRY((int)hipStreamWaitEvent(c.st, ev[evi(0)], 0), "mask wait");
      TF_TRY(tf_launch_attn_fwd(&a, c.st), "attn_fwd");
    }
    TF_TRY(gemm(c, o, c.wgt(w + c.W.wo, D.dp, D.dp), z1, (const float*)(w + c.W.bo), x, NOBUF, D.dp, D.dp, TF_EPI_BIAS_DROP_RES,
                drop_for(e, e->p_token, site_of(l, SITE_DROP1))), "gemm out_proj");
    {
      TfLnArgs n{};
      ln_rows(c, n, z1, p.n1_w, (float*)(b + c.A.mean1), (float*)(b + c.A.rstd1));
      n.y = (void*)x1.p; n.y_lo = (void*)x1.lo; n.ldy = D.dp; n.y_is_f32 = 0; n.beta = p.n1_b;
      TF_TRY(tf_launch_ln_fwd(&n, c.st), "ln1_fwd");
    }
    if (e->fp8_proj) {
      TF_TRY(gemm_fp8(c, x1.p, D.dp, D.dp, w + c.W.w18, (const float*)(w + c.W.s_w1), b + c.A.u, D.ffp, (const float*)(w + c.W.b1),
                      nullptr, 0, b + c.A.h, D.ffp, D.ffp, TF_EPI_BIAS_GELU_DROP_G, drop_for(e, e->p_token, site_of(l, SITE_FFN))), "gemm ffn_up (fp8)");
      TF_TRY(gemm_fp8(c, hh.p, D.ffp, D.ffp, w + c.W.w28, (const float*)(w + c.W.s_w2), b + c.A.z2, D.dp, (const float*)(w + c.W.b2),
                      x1.p, D.dp, nullptr, 0, D.dp, TF_EPI_BIAS_DROP_RES, drop_for(e, e->p_token, site_of(l, SITE_DROP2))), "gemm ffn_down (fp8)");
    } else {
      TF_TRY(gemm(c, x1, c.wgt(w + c.W.w1, D.ffp, D.dp), u, (const float*)(w + c.W.b1), NOBUF, hh, D.ffp, D.dp, TF_EPI_BIAS_GELU_DROP_G,
                  drop_for(e, e->p_token, site_of(l, SITE_FFN))), "gemm ffn_up");   // slot "u" holds G = d h / d u
      TF_TRY(gemm(c, hh, c.wgt(w + c.W.w2, D.dp, D.ffp), z2, (const float*)(w + c.W.b2), x1, NOBUF, D.dp, D.ffp, TF_EPI_BIAS_DROP_RES,
                  drop_for(e, e->p_token, site_of(l, SITE_DROP2))), "gemm ffn_down");
    }
    {
      TfLnArgs n{};
      ln_rows(c, n, z2, p.n2_w, (float*)(b + c.A.mean2), (float*)(b + c.A.rstd2));
      n.y = (void*)xn.p; n.y_lo = (void*)xn.lo; n.ldy = D.dp; n.y_is_f32 = 0; n.beta = p.n2_b;
      TF_TRY(tf_launch_ln_fwd(&n, c.st), "ln2_fwd");
    }
  }
  // visual rows: final LayerNorm (cross_f_box_layers.py:104-107) or plain copy
  const Buf xl = c.act_d(c.X(D.L));
  if (D.Nv > 0) {
    if (e->final_norm) {
      TfLnArgs n{};
      n.x = xl.p; n.x_lo = xl.lo; n.ldx = D.dp; n.y = e->vis_out; n.ldy = D.d; n.y_is_f32 = e->vis_out_is_f32; n.gamma = e->fn_w; n.beta = e->fn_b;
      n.mean = (float*)(c.wk + c.A.meanf); n.rstd = (float*)(c.wk + c.A.rstdf);
      ln_vis_rows(c, n);
      TF_TRY(tf_launch_ln_fwd(&n, c.st), "final_ln_fwd");
    } else {
      TfCopyRowsArgs r{};
      r.src = xl.p; r.src_lo = xl.lo; r.src_is_f32 = 0; r.ld_src = D.dp; r.src_rpg = D.Nv; r.src_gstride = D.S; r.src_group_row0 = c.starts();
      r.dst = e->vis_out; r.dst_is_f32 = e->vis_out_is_f32; r.ld_dst = D.d; r.dst_rpg = D.Nv; r.dst_gstride = D.Nv; r.rows = D.B * D.Nv; r.cols = D.d;
      if (D.ragged) { r.src_row_map = c.vis_rows(); r.rows = D.vis_total; r.dst_rpg = D.vis_total; r.dst_gstride = D.vis_total; }
      TF_TRY(tf_launch_copy_rows(&r, c.st), "vis_copy");
    }
  }
  if (e->lang_out != nullptr && D.Nl > 0) {
    TfCopyRowsArgs r{};
    const size_t lang_off = (size_t)D.Nv * D.dp * 2;
    r.src = (const unsigned char*)xl.p + lang_off; r.src_lo = xl.lo ? (const unsigned char*)xl.lo + lang_off : nullptr;
    r.src_is_f32 = 0; r.ld_src = D.dp; r.src_rpg = D.Nl; r.src_gstride = D.S;
    if (c.packed()) {                   // language token (b, j) sits in packed row pol[b * Nl + j]; masked tokens (-1) get zero rows
      r.src = xl.p; r.src_lo = xl.lo; r.src_row_map = c.pol();
    }
    r.dst = e->lang_out; r.dst_is_f32 = e->lang_out_is_f32; r.ld_dst = D.d; r.dst_rpg = D.Nl; r.dst_gstride = D.Nl; r.rows = D.B * D.Nl; r.cols = D.d;
    TF_TRY(tf_launch_copy_rows(&r, c.st), "lang_copy");
  }
  return 0;
}

int tf_encoder_bwd(const TfEncoderDesc* e, tf_stream_t s) {
  Ctx c;
  if (!make_ctx(e, (hipStream_t)s, &c)) return fail(-1, "tf_encoder_bwd");
  const Dims& D = c.D;
  if (e->d_vis_out == nullptr && e->d_lang_out == nullptr) return fail(-1, "tf_encoder_bwd(no cotangent)");
  const Buf dxa = c.act_d(c.wk + c.A.dxa), dxb = c.act_d(c.wk + c.A.dxb), d_o = c.act_d(c.wk + c.A.d_o);
  // the tensors a layer's weight gradients read, by layer parity (see Side)
  const Buf dz_[2] = {c.act_d(c.wk + c.A.dz), c.act_d(c.wk + c.A.dz1)}, dy_[2] = {c.act_d(c.wk + c.A.dy), c.act_d(c.wk + c.A.dy1)};
  const Buf dzb_[2] = {c.act_d(c.wk + c.A.dzb), c.act_d(c.wk + c.A.dzb1)}, dyb_[2] = {c.act_d(c.wk + c.A.dyb), c.act_d(c.wk + c.A.dyb1)};
  const Buf du_[2] = {c.act_f(c.wk + c.A.du), c.act_f(c.wk + c.A.du1)}, dqkv_[2] = {c.act_q(c.wk + c.A.dqkv), c.act_q(c.wk + c.A.dqkv1)};
  float* delta = (float*)(c.wk + c.A.delta);
  const uint8_t* km = c.packed() ? nullptr : (const uint8_t*)(c.wk + c.A.keymask);
  const int l_hi = e->bwd_nlayers > 0 ? e->bwd_hi : D.L - 1;
  const int l_lo = e->bwd_nlayers > 0 ? e->bwd_hi - e->bwd_nlayers + 1 : 0;
  if (l_hi >= D.L || l_lo < 0 || l_lo > l_hi) return fail(-1, "tf_encoder_bwd(layer range)");
  const bool head = l_hi == D.L - 1, tail = l_lo == 0;
  Side sd;
  if (e->overlap != nullptr && e->overlap->stream != nullptr) {
    sd.st = (hipStream_t)e->overlap->stream; sd.ev = (hipEvent_t*)e->overlap->ev;
    sd.pending[0] = (e->overlap->pending & 1u) != 0; sd.pending[1] = (e->overlap->pending & 2u) != 0;   // left by a defer_join call
    sd.last = (int)(e->overlap->reserved & 1u);
  }
  // ---- gradient w.r.t. the last layer's output X[L] -> dxa ----
  if (head && D.Nv > 0) {
    if (e->final_norm && e->d_vis_out != nullptr) {
      TfLnArgs n{};
      const Buf xl = c.act_d(c.X(D.L));
      n.x = xl.p; n.x_lo = xl.lo; n.ldx = D.dp; n.gamma = e->fn_w; n.mean = (float*)(c.wk + c.A.meanf); n.rstd = (float*)(c.wk + c.A.rstdf);
      ln_vis_rows(c, n);
      n.dy = e->d_vis_out; n.lddy = D.d; n.dy_is_f32 = e->d_vis_out_is_f32; n.dx = (void*)dxa.p; n.dx_lo = (void*)dxa.lo; n.lddx = D.dp;
      n.dgamma = e->g_fn_w; n.dbeta = e->g_fn_b;
      TF_TRY(tf_launch_ln_bwd(&n, c.st), "final_ln_bwd");
    } else {
      TfCopyRowsArgs r{};
      r.src = e->d_vis_out; r.src_is_f32 = e->d_vis_out_is_f32; r.ld_src = D.d; r.src_rpg = D.Nv; r.src_gstride = D.Nv;
      r.dst = (void*)dxa.p; r.dst_lo = (void*)dxa.lo; r.dst_is_f32 = 0; r.ld_dst = D.dp; r.dst_rpg = D.Nv; r.dst_gstride = D.S; r.rows = D.B * D.Nv; r.cols = D.d;
      r.dst_group_row0 = c.starts();
      if (D.ragged) { r.dst_row_map = c.vis_rows(); r.rows = D.vis_total; r.src_rpg = D.vis_total; r.src_gstride = D.vis_total; }
      TF_TRY(tf_launch_copy_rows(&r, c.st), "dvis_copy");
    }
  }
  if (head && D.Nl > 0) {
    TfCopyRowsArgs r{};
    const size_t lang_off = (size_t)D.Nv * D.dp * 2;
    r.src = e->d_lang_out; r.src_is_f32 = e->d_lang_out_is_f32; r.ld_src = D.d; r.src_rpg = D.Nl; r.src_gstride = D.Nl;
    r.dst = (unsigned char*)dxa.p + lang_off; r.dst_lo = dxa.lo ? (unsigned char*)dxa.lo + lang_off : nullptr;
    r.dst_is_f32 = 0; r.ld_dst = D.dp; r.dst_rpg = D.Nl; r.dst_gstride = D.S; r.rows = D.B * D.Nl; r.cols = D.d;
    if (c.packed()) {                   // the cotangents of masked tokens are dropped with their rows (TfEncoderDesc.packed_rows)
      r.dst = (void*)dxa.p; r.dst_lo = (void*)dxa.lo; r.dst_row_map = c.pol();
    }
    TF_TRY(tf_launch_copy_rows(&r, c.st), "dlang_copy");
  }
  const float scale = 1.0f / sqrtf((float)D.hd);
  const Drop none{0u, 0u, 1.f};
  for (int l = l_hi; l >= l_lo; --l) {
    unsigned char* w = c.WB(l); unsigned char* b = c.LB(l);
    const TfLayerParams& p = e->p[l]; const TfLayerParams& g = e->g[l];
    const Buf x = c.act_d(c.X(l)), qkv = c.act_q(b + c.A.qkv), o = c.act_d(b + c.A.o), z1 = c.act_d(b + c.A.z1), x1 = c.act_d(b + c.A.x1);
    const Buf u = c.act_f(b + c.A.u), hh = c.act_f(b + c.A.h), z2 = c.act_d(b + c.A.z2);
    const int par = l & 1;
    const Buf dz = dz_[par], dy = dy_[par], dzb = dzb_[par], dyb = dyb_[par], du = du_[par], dqkv = dqkv_[par];
    const WPlan plan = wgrad_plan(tail && l == l_lo && sd.st != nullptr);
    TfWgradArgs jobs[4] = {};
    // ---- LN2 backward: dxa -> dz (= d z2), dy (= dropout2-masked) ----
    const Drop d2 = drop_for(e, e->p_token, site_of(l, SITE_DROP2));
    TF_TRY(guard(c, sd, par), "guard");                      // the weight gradients of layer l + 2 read this set
    {
      TfLnArgs n{};
      ln_rows(c, n, z2, p.n2_w, (float*)(b + c.A.mean2), (float*)(b + c.A.rstd2));
      n.dy = dxa.p; n.dy_lo = dxa.lo; n.lddy = D.dp; n.dy_is_f32 = 0; n.dx = (void*)dz.p; n.dx_lo = (void*)dz.lo; n.lddx = D.dp;
      n.dgamma = g.n2_w; n.dbeta = g.n2_b;
      if (d2.thr) { n.dx_drop = (void*)dy.p; n.dx_drop_lo = (void*)dy.lo; n.lddxd = D.dp; n.drop_thr = d2.thr; n.drop_key = d2.key; n.drop_scale = d2.scale; n.drop_ld = D.dp; }
      TF_TRY(tf_launch_ln_bwd(&n, c.st), "ln2_bwd");
    }
    const Buf dy2 = d2.thr ? dy : dz;
    TF_TRY(gemm(c, dy2, c.wgt(w + c.W.w2T, D.ffp, D.dp), du, nullptr, u, NOBUF, D.ffp, D.dp, TF_EPI_MUL, none),
           "dgrad ffn_down");                                 // dU = dH . G (G stored by the forward FFN-up epilogue)
    jobs[0] = wjob(c, dy2, D.dp, hh, D.ffp, g.w2, D.ff, g.b2, BIG, BIG, D.d, BIG, BIG, D.ff);
    jobs[1] = wjob(c, du, D.ffp, x1, D.dp, g.w1, D.d, g.b1, BIG, BIG, D.ff, BIG, BIG, D.d);
    if (plan.at[0]) {
      TF_TRY(side_fork(c, sd, EV_FORK0), "fork 0");
      TF_TRY(wgrad_launch(c, sd, jobs, plan.at[0], false), "wgrad (after the FFN-down dgrad)");
    }
    TF_TRY(gemm(c, du, c.wgt(w + c.W.w1T, D.dp, D.ffp), dxb, nullptr, dz, NOBUF, D.dp, D.ffp, TF_EPI_ADD, none), "dgrad ffn_up");
    // ---- LN1 backward: dxb -> dzb (= d z1), dyb (= dropout1-masked) ----
    const Drop d1 = drop_for(e, e->p_token, site_of(l, SITE_DROP1));
    {
      TfLnArgs n{};
      ln_rows(c, n, z1, p.n1_w, (float*)(b + c.A.mean1), (float*)(b + c.A.rstd1));
      n.dy = dxb.p; n.dy_lo = dxb.lo; n.lddy = D.dp; n.dy_is_f32 = 0; n.dx = (void*)dzb.p; n.dx_lo = (void*)dzb.lo; n.lddx = D.dp;
      n.dgamma = g.n1_w; n.dbeta = g.n1_b;
      if (d1.thr) { n.dx_drop = (void*)dyb.p; n.dx_drop_lo = (void*)dyb.lo; n.lddxd = D.dp; n.drop_thr = d1.thr; n.drop_key = d1.key; n.drop_scale = d1.scale; n.drop_ld = D.dp; }
      TF_TRY(tf_launch_ln_bwd(&n, c.st), "ln1_bwd");
    }
    const Buf dy1 = d1.thr ? dyb : dzb;
    jobs[2] = wjob(c, dy1, D.dp, o, D.dp, g.out_w, D.d, g.out_b, BIG, BIG, D.d, D.hd, D.hdp, D.d);
    if (plan.at[1]) {
      TF_TRY(side_fork(c, sd, EV_FORK1), "fork 1");
      TF_TRY(wgrad_launch(c, sd, jobs, plan.at[1], false), "wgrad (after the LN1 backward)");
    }
    TF_TRY(gemm(c, dy1, c.wgt(w + c.W.woT, D.dp, D.dp), d_o, nullptr, NOBUF, NOBUF, D.dp, D.dp, TF_EPI_NONE, none), "dgrad out_proj");
    {
      TfAttnArgs a{};
      a.qkv = qkv.p; a.qkv_lo = qkv.lo; a.ld_qkv = D.ldq; a.out = (void*)o.p; a.out_lo = (void*)o.lo; a.ld_out = D.dp; a.lse = (float*)(b + c.A.lse);
      a.key_mask = km; a.cu_rows = c.cu(); a.B = D.B; a.S = D.S; a.H = D.H; a.HDP = D.hdp; a.scale = scale;
      const Drop dr = drop_for(e, e->p_token, site_of(l, SITE_ATTN));
      a.drop_thr = dr.thr; a.drop_key = dr.key; a.drop_scale = dr.scale; a.drop_bits = b + c.A.dbits; a.block_bits = e->attn_block_bits;
      if (e->attn_block_bits != nullptr) { a.block_skip_q = c.wk + c.A.bskip; a.block_skip_k = c.wk + c.A.bskip + (size_t)((D.S + 127) / 128) * 8; }
      a.dout = d_o.p; a.dout_lo = d_o.lo; a.ld_dout = D.dp; a.dqkv = (void*)dqkv.p; a.dqkv_lo = (void*)dqkv.lo; a.ld_dqkv = D.ldq; a.delta = delta;
      a.ds_work = (void*)(c.wk + c.A.dsw); a.ds_planes = D.split ? 4 : 1;
      TF_TRY(tf_launch_attn_bwd(&a, c.st), "attn_bwd");
    }
    jobs[3] = wjob(c, dqkv, D.nqkv, x, D.dp, g.in_w, D.d, g.in_b, D.hd, D.hdp, 3 * D.d, BIG, BIG, D.d);
    if (plan.at[2]) {
      TF_TRY(side_fork(c, sd, EV_FORK2), "fork 2");
      // (the last launch of a whole backward has the chip to itself apart from one dgrad and the assemble kernel)
      TF_TRY(wgrad_launch(c, sd, jobs, plan.at[2], tail && l == l_lo), "wgrad (after the attention backward)");
    }
    TF_TRY(side_done(sd, par), "done");
    TF_TRY(gemm(c, dqkv, c.wgt(w + c.W.winT, D.dp, D.ldq), dxa, nullptr, dzb, NOBUF, D.dp, D.ldq, TF_EPI_ADD, none), "dgrad in_proj");
  }
  if (tail) {
    TfAssembleArgs a{};
    a.B = D.B; a.Nv = D.Nv; a.Nl = D.Nl; a.d = D.d; a.dout = dxa.p; a.dout_lo = dxa.lo; a.ld_dout = D.dp;
    a.dvis = e->d_vis; a.dvis_is_f32 = e->d_vis_is_f32; a.ld_dvis = D.d; a.dlang = e->d_lang; a.dlang_is_f32 = e->d_lang_is_f32; a.ld_dlang = D.d;
    a.dkind_v = e->g_kind_v; a.dkind_l = e->g_kind_l;
    a.pgroups = D.G; a.p_gstride = c.pg();
    a.row_map = c.dense_of(); a.rows = c.packed() ? D.M : 0;
    if (D.ragged) { for (int g = 0; g < TF_MAX_GROUPS; ++g) a.group_nv[g] = D.nv[g]; c.rows_into(a.group_rows); }
    if (c.packed() && e->d_lang != nullptr)          // rows of masked language tokens are not visited: their gradient is zero
      TF_TRY((int)hipMemsetAsync(e->d_lang, 0, (size_t)D.B * D.Nl * D.d * (e->d_lang_is_f32 ? 4 : 2), c.st), "d_lang zero fill");
    const Drop dr = drop_for(e, e->p_patch, SITE_PATCH);
    a.drop_thr = dr.thr; a.drop_key = dr.key; a.drop_scale = dr.scale;
    TF_TRY(tf_launch_assemble_bwd(&a, c.st), "assemble_bwd");
  }
  if (sd.st != nullptr && e->defer_join) {
    e->overlap->pending = (sd.pending[0] ? 1u : 0u) | (sd.pending[1] ? 2u : 0u);      // the next call / tf_overlap_join takes over
    e->overlap->reserved = (unsigned)sd.last;
    return 0;
  }
  // join: the side stream is FIFO, so its last recorded event covers everything before it -- one wait, not one per layer
  if (sd.pending[0] || sd.pending[1]) { sd.pending[sd.last] = true; sd.pending[sd.last ^ 1] = false; TF_TRY(guard(c, sd, sd.last), "join"); }
  if (sd.st != nullptr) e->overlap->pending = 0u;
  return 0;
}

int tf_encoder_packed_error(const TfEncoderDesc* e, int* out, tf_stream_t s) {
  Ctx c;
  if (out == nullptr || !make_ctx(e, (hipStream_t)s, &c)) return fail(-1, "tf_encoder_packed_error");
  TF_TRY((int)hipMemcpyAsync(out, c.wk + c.A.perr, sizeof(int), hipMemcpyDeviceToHost, c.st), "tf_encoder_packed_error");
  return 0;
}

long long tf_encoder_peek(const TfEncoderDesc* e, const char* name, float* dst, long long cap, tf_stream_t s) {
  Ctx c;
  if (name == nullptr || dst == nullptr || !make_ctx(e, (hipStream_t)s, &c)) return fail(-1, "tf_encoder_peek");
  const Dims& D = c.D;
  const void* src = nullptr; long long cols = 0; size_t pl = 0;
  int l = 0;
  auto lb = [&](size_t off) { return (const void*)(c.LB(l) + off); };
  if (sscanf(name, "qkv%d", &l) == 1) { src = lb(c.A.qkv); cols = D.ldq; pl = c.pq; }
  else if (sscanf(name, "z1_%d", &l) == 1) { src = lb(c.A.z1); cols = D.dp; pl = c.pd; }
  else if (sscanf(name, "x1_%d", &l) == 1) { src = lb(c.A.x1); cols = D.dp; pl = c.pd; }
  else if (sscanf(name, "z2_%d", &l) == 1) { src = lb(c.A.z2); cols = D.dp; pl = c.pd; }
  else if (sscanf(name, "x%d", &l) == 1) { if (l < 0 || l > D.L) return fail(-1, "peek"); src = c.X(l); cols = D.dp; pl = c.pd; l = 0; }
  else if (sscanf(name, "o%d", &l) == 1) { src = lb(c.A.o); cols = D.dp; pl = c.pd; }
  else if (sscanf(name, "u%d", &l) == 1) { src = lb(c.A.u); cols = D.ffp; pl = c.pf; }
  else if (sscanf(name, "h%d", &l) == 1) { src = lb(c.A.h); cols = D.ffp; pl = c.pf; }
  else if (strcmp(name, "dqkv") == 0) { src = c.wk + c.A.dqkv; cols = D.ldq; pl = c.pq; }
  else if (strcmp(name, "dxa") == 0) { src = c.wk + c.A.dxa; cols = D.dp; pl = c.pd; }
  else if (strcmp(name, "do") == 0) { src = c.wk + c.A.d_o; cols = D.dp; pl = c.pd; }
  else return fail(-1, "tf_encoder_peek(name)");
  if (l < 0 || l >= D.L) return fail(-1, "tf_encoder_peek(layer)");
  const long long n = (long long)D.M * cols;
  if (n > cap) return fail(-1, "tf_encoder_peek(cap)");
  TfCopyRowsArgs r{};                                   // bf16 (hi + lo in the fp32-accuracy mode) -> fp32
  r.src = src; r.src_lo = c.lo(src, pl); r.src_is_f32 = 0; r.ld_src = (int)cols; r.src_rpg = D.M; r.src_gstride = D.M;
  r.dst = dst; r.dst_is_f32 = 1; r.ld_dst = (int)cols; r.dst_rpg = D.M; r.dst_gstride = D.M; r.rows = D.M; r.cols = (int)cols;
  TF_TRY(tf_launch_copy_rows(&r, c.st), "peek copy");
  return n;
}

}  // extern "C"
