// RoI heads of the NAO detector and their losses: what sits between the fused features and the scalar the
// optimiser sees.  Reference:
//   modeling/obj_detection/faster_rcnn_wrapper.py:93-100   box_regressor = Dropout -> Linear(repr, 4*Cn); noun / verb classifiers
//   modeling/obj_detection/roi_wrappers.py:204-231         logits; ttcs = softplus(ttc_pred_layer(box_features))
//   runner/metrics_losses/losses.py:98-135                 box_loss: smooth-L1 (beta 1/9, sum) over positive RoIs / number of RoIs
//   runner/nao/ego_nao_trainer.py:307-359                  noun / verb class-weighted CE on (logits + 1e-6), verb background handling,
//                                                          TTC smooth-L1 on the softplus outputs of the non-background RoIs
//   runner/abc_nao_trainer.py:53-56                        criterion objects (class weights, reduction mean, ttc_beta)
// The four Linears run on the MFMA GEMM (tf_gemm_fwd: box head, and noun | verb | ttc concatenated into one GEMM).  This file holds
// the row-wise part: one wave per RoI reads its logits once and produces every loss term (forward) or every logit gradient
// (backward); the loss normalisers (sum of class weights, selected-row counts) stay on the device, so there is no host sync.
#include "tf_common.h"
#include "tf_kernels.h"

namespace {

constexpr int MAXK = 8;                     // classes per lane: up to 512 classes per head

__device__ __forceinline__ float ld_logit(const u16* hi, const u16* lo, size_t i) {
  float v = bf2f(hi[i]);
  if (lo != nullptr) v += bf2f(lo[i]);
  return v;
}
__device__ __forceinline__ float smooth_l1(float d, float beta) {
  const float a = fabsf(d);
  return (beta > 0.f && a < beta) ? 0.5f * d * d / beta : a - 0.5f * beta;
}
__device__ __forceinline__ float smooth_l1_grad(float d, float beta) {
  const float a = fabsf(d);
  if (beta > 0.f && a < beta) return d / beta;
  return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
}

// log-sum-exp of one row of C classes held MAXK per lane (class c = lane + 64 k); returns lse, leaves the shifted logits in v
__device__ __forceinline__ float row_lse(const u16* hi, const u16* lo, size_t base, int C, int lane, float (&v)[MAXK]) {
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < C ? ld_logit(hi, lo, base + c) + 1e-6f : -INFINITY;      // the reference adds 1e-6 to every logit (:310, :322)
    m = fmaxf(m, v[k]);
  }
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) s += (lane + 64 * k < C) ? __expf(v[k] - m) : 0.f;
  return m + __logf(wave_sum(s));
}

// Labels are range-checked HERE, not by host-side min / max reductions (two blocking syncs per step): a noun label outside [0, Cn) or
// a verb label that is neither verb_ignore nor inside [0, Cv) selects nothing -- no class weight, no logit, no box slot is read
// through it -- and, unless it is torch's ignore_index (-100, which nn.CrossEntropyLoss skips silently), raises sums[7], which the
// host reads lazily and turns into the IndexError torch would have raised.
constexpr long long kTorchIgnoreIndex = -100;
__device__ __forceinline__ bool label_ok(long long y, int C) { return y >= 0 && y < (long long)C; }

// sums: [0] noun num  [1] noun den  [2] verb num  [3] verb den  [4] ttc num  [5] ttc count  [6] box num  [7] count of out-of-range labels
__global__ __launch_bounds__(256) void heads_loss_fwd_kernel(const TfHeadsLossArgs a) {
  __shared__ float red[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < a.R) {
    const u16* cls = (const u16*)a.cls; const u16* cls_lo = (const u16*)a.cls_lo;
    const size_t rb = (size_t)r * a.ld_cls;
    const long long yn = a.noun_labels[r];
    const bool yn_ok = label_ok(yn, a.Cn);
    if (!yn_ok && yn != kTorchIgnoreIndex && lane == 0) acc[7] += 1.f;
    float v[MAXK];
    if (a.noun_w != nullptr) {                       // noun head (criterion.noun > 0)
      const float lse = row_lse(cls, cls_lo, rb, a.Cn, lane, v);
      if (lane == 0) {
        a.lse[r] = lse;
        if (yn_ok) {
          const float w = a.noun_w[yn];
          acc[0] = w * (lse - (ld_logit(cls, cls_lo, rb + yn) + 1e-6f));
          acc[1] = w;
        }
      }
    }
    const long long tv = a.verb_labels != nullptr ? a.verb_labels[r] : 0;
    const bool bg = a.verb_labels != nullptr && tv == a.verb_ignore;
    const bool tv_ok = bg || label_ok(tv, a.Cv);
    if (a.verb_w != nullptr && a.Cv > 0) {           // verb head; background RoIs: last class (verb_bg) or dropped (:316-320)
      const float lse = row_lse(cls, cls_lo, rb + a.Cn, a.Cv, lane, v);
      if (!tv_ok && tv != kTorchIgnoreIndex && lane == 0) acc[7] += 1.f;
      if (lane == 0) {
        a.lse[a.R + r] = lse;
        if (tv_ok && (a.verb_bg || !bg)) {
          const long long yv = bg ? a.Cv - 1 : tv;
          const float w = a.verb_w[yv];
          acc[2] = w * (lse - (ld_logit(cls, cls_lo, rb + a.Cn + yv) + 1e-6f));
          acc[3] = w;
        }
      }
    }
    if (a.ttcs != nullptr && lane == 0) {            // TTC: smooth-L1(beta) on the softplus outputs, mean over the selected RoIs (:347-359)
      float tt = a.ttc_targets[r];
      bool sel = true;
      if (!a.ttc_bg) sel = !bg;
      else if (tt == (float)a.verb_ignore) tt = a.ttc_bg_val;
      if (sel) { acc[4] = smooth_l1(a.ttcs[r] - tt, a.ttc_beta); acc[5] = 1.f; }
    }
    if (a.box != nullptr && yn > 0 && yn_ok && lane < 4) {    // box regression of the label's class, positives only (losses.py:119-131)
      const float d = ld_logit((const u16*)a.box, (const u16*)a.box_lo, (size_t)r * a.ld_box + 4 * yn + lane) - a.reg_targets[(size_t)r * 4 + lane];
      float l = smooth_l1(d, a.box_beta);
      l += __shfl_xor(l, 1, 64);
      l += __shfl_xor(l, 2, 64);
      if (lane == 0) acc[6] = l;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) red[wave][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (t != 0.f) atomicAdd(a.sums + threadIdx.x, t);
  }
}

// losses[0..3] = box, noun, verb, ttc
__global__ void heads_loss_finalize_kernel(const float* __restrict__ sums, float* __restrict__ losses, int R) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  losses[0] = sums[6] / (float)(R > 1 ? R : 1);                    // box_loss / max(labels.numel(), 1)
  losses[1] = sums[1] > 0.f ? sums[0] / sums[1] : 0.f;             // weighted mean (CrossEntropyLoss(weight, reduction="mean"))
  losses[2] = sums[3] > 0.f ? sums[2] / sums[3] : 0.f;
  losses[3] = sums[5] > 0.f ? sums[4] / sums[5] : 0.f;             // no selected RoI: the reference leaves ttc_loss at 0 (:358)
}

__global__ __launch_bounds__(256) void heads_loss_bwd_kernel(const TfHeadsLossArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= a.R) return;
  const u16* cls = (const u16*)a.cls; const u16* cls_lo = (const u16*)a.cls_lo;
  const size_t rb = (size_t)r * a.ld_cls;
  const long long yn = a.noun_labels[r];
  const float g_box = a.gscale[0], g_noun = a.gscale[1], g_verb = a.gscale[2], g_ttc = a.gscale[3];
  u16* dc = (u16*)a.d_cls; u16* dc_lo = (u16*)a.d_cls_lo;
  auto put = [&](u16* hi, u16* lo, size_t i, float g) {
    const u16 hb = f2bf(g);
    hi[i] = hb;
    if (lo != nullptr) lo[i] = f2bf(g - bf2f(hb));
  };
  const bool yn_ok = label_ok(yn, a.Cn);
  // noun columns
  {
    const float coef = (a.noun_w != nullptr && yn_ok && a.sums[1] > 0.f) ? g_noun * a.noun_w[yn] / a.sums[1] : 0.f;
    const float lse = a.noun_w != nullptr ? a.lse[r] : 0.f;
    for (int c = lane; c < a.Cn; c += 64) {
      float g = 0.f;
      if (coef != 0.f) g = coef * (__expf(ld_logit(cls, cls_lo, rb + c) + 1e-6f - lse) - (c == yn ? 1.f : 0.f));
      put(dc, dc_lo, rb + c, g);
    }
  }
  const long long tv = a.verb_labels != nullptr ? a.verb_labels[r] : 0;
  const bool bg = a.verb_labels != nullptr && tv == a.verb_ignore;
  {
    const bool on = a.verb_w != nullptr && a.Cv > 0 && (a.verb_bg || !bg) && (bg || label_ok(tv, a.Cv)) && a.sums[3] > 0.f;
    const long long yv = bg ? a.Cv - 1 : tv;
    const float coef = on ? g_verb * a.verb_w[yv] / a.sums[3] : 0.f;
    const float lse = on ? a.lse[a.R + r] : 0.f;
    for (int c = lane; c < a.Cv; c += 64) {
      float g = 0.f;
      if (on) g = coef * (__expf(ld_logit(cls, cls_lo, rb + a.Cn + c) + 1e-6f - lse) - (c == yv ? 1.f : 0.f));
      put(dc, dc_lo, rb + a.Cn + c, g);
    }
  }
  for (int c = a.Cn + a.Cv + lane; c < a.ld_cls; c += 64) put(dc, dc_lo, rb + c, 0.f);     // ttc pre-activation column and padding: the
                                                                                           // TTC gradient leaves through d_ttcs (softplus)
  if (a.d_ttcs != nullptr && lane == 0) {
    float g = 0.f;
    if (a.ttcs != nullptr && a.sums[5] > 0.f) {
      float tt = a.ttc_targets[r];
      bool sel = true;
      if (!a.ttc_bg) sel = !bg;
      else if (tt == (float)a.verb_ignore) tt = a.ttc_bg_val;
      if (sel) g = g_ttc / a.sums[5] * smooth_l1_grad(a.ttcs[r] - tt, a.ttc_beta);
    }
    a.d_ttcs[r] = g;
  }
  if (a.d_box != nullptr) {
    u16* db = (u16*)a.d_box; u16* db_lo = (u16*)a.d_box_lo;
    const size_t bb = (size_t)r * a.ld_box;
    const float coef = g_box / (float)(a.R > 1 ? a.R : 1);
    for (int c = lane; c < a.ld_box; c += 64) {
      float g = 0.f;
      if (a.box != nullptr && yn > 0 && yn_ok && c >= 4 * yn && c < 4 * yn + 4) {
        const float d = ld_logit((const u16*)a.box, (const u16*)a.box_lo, bb + c) - a.reg_targets[(size_t)r * 4 + (c - 4 * yn)];
        g = coef * smooth_l1_grad(d, a.box_beta);
      }
      put(db, db_lo, bb + c, g);
    }
  }
}

// y = softplus(x) (beta 1, threshold 20: F.softplus defaults, roi_wrappers.py:229) over column `col` of a bf16 [R, ld] tensor -> fp32 [R]
__global__ __launch_bounds__(256) void softplus_col_fwd_kernel(const u16* __restrict__ x, const u16* __restrict__ x_lo, int ld, int col, float* __restrict__ y, int R) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float z = ld_logit(x, x_lo, (size_t)r * ld + col);
  y[r] = z > 20.f ? z : log1pf(__expf(z));
}
// dx[r, col] += dy[r] * sigmoid(x[r, col])  (bf16 (+lo) gradient tensor of the logits, already written by heads_loss_bwd_kernel or zero)
__global__ __launch_bounds__(256) void softplus_col_bwd_kernel(const u16* __restrict__ x, const u16* __restrict__ x_lo, int ld, int col,
                                                               const float* __restrict__ dy, u16* __restrict__ dx, u16* __restrict__ dx_lo, int R) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const size_t i = (size_t)r * ld + col;
  const float z = ld_logit(x, x_lo, i);
  const float g = dy[r] * (z > 20.f ? 1.f : 1.f / (1.f + __expf(-z))) + ld_logit(dx, dx_lo, i);
  const u16 hb = f2bf(g);
  dx[i] = hb;
  if (dx_lo != nullptr) dx_lo[i] = f2bf(g - bf2f(hb));
}

int check(const TfHeadsLossArgs* a) {
  if (a->R <= 0) return 1;
  if (a->cls == nullptr || a->noun_labels == nullptr || a->sums == nullptr || a->lse == nullptr) return -2;
  if (a->Cn <= 0 || a->Cn > 64 * MAXK || a->Cv < 0 || a->Cv > 64 * MAXK || a->ld_cls < a->Cn + a->Cv) return -2;
  if (a->box != nullptr && (a->reg_targets == nullptr || a->ld_box < 4 * a->Cn)) return -2;
  if (a->ttcs != nullptr && a->ttc_targets == nullptr) return -2;
  if (a->verb_w != nullptr && a->Cv > 0 && a->verb_labels == nullptr) return -2;
  return 0;
}

}  // namespace

extern "C" int tf_launch_heads_loss_fwd(const TfHeadsLossArgs* a, hipStream_t st) {
  const int c = check(a);
  if (c) return c > 0 ? 0 : c;
  if (a->losses == nullptr) return -2;
  {
    TfTraceScope tr("heads_loss_fwd_kernel", st);
    hipLaunchKernelGGL(heads_loss_fwd_kernel, dim3((a->R + 3) / 4), dim3(256), 0, st, *a);
  }
  hipLaunchKernelGGL(heads_loss_finalize_kernel, dim3(1), dim3(64), 0, st, a->sums, a->losses, a->R);
  return (int)hipGetLastError();
}
extern "C" int tf_launch_heads_loss_bwd(const TfHeadsLossArgs* a, hipStream_t st) {
  const int c = check(a);
  if (c) return c > 0 ? 0 : c;
  if (a->gscale == nullptr || a->d_cls == nullptr) return -2;
  TfTraceScope tr("heads_loss_bwd_kernel", st);
  hipLaunchKernelGGL(heads_loss_bwd_kernel, dim3((a->R + 3) / 4), dim3(256), 0, st, *a);
  return (int)hipGetLastError();
}
extern "C" int tf_launch_softplus_col(const void* x, const void* x_lo, int ld, int col, float* y, const float* dy, void* dx, void* dx_lo, int R,
                                      hipStream_t st) {
  if (R <= 0) return 0;
  if (x == nullptr || col < 0 || col >= ld) return -2;
  if (dy == nullptr) {
    if (y == nullptr) return -2;
    hipLaunchKernelGGL(softplus_col_fwd_kernel, dim3((R + 255) / 256), dim3(256), 0, st, (const u16*)x, (const u16*)x_lo, ld, col, y, R);
  } else {
    if (dx == nullptr) return -2;
    hipLaunchKernelGGL(softplus_col_bwd_kernel, dim3((R + 255) / 256), dim3(256), 0, st, (const u16*)x, (const u16*)x_lo, ld, col, dy, (u16*)dx,
                       (u16*)dx_lo, R);
  }
  return (int)hipGetLastError();
}
