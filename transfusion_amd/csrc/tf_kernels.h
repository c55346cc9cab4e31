// Internal launcher interface between the .hip translation units and the C-ABI layer (tf_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tfusion.h"

#ifdef __cplusplus
extern "C" {
#endif
void tf_set_error_msg(const char* msg);                             // sets this thread's tf_last_error() text (tf_api.hip)
int tf_launch_gemm_nt(const TfGemmArgs* a, hipStream_t stream);
int tf_launch_wgrad_tn(const TfWgradArgs* a, hipStream_t stream);
int tf_launch_wgrad_multi(const TfWgradArgs* probs, int count, int blocks, hipStream_t stream);   // wgrad_multi.hip
int tf_wgrad_tiles(int N, int K, int caller_sized);                // output tiles of the wgrad kernel that will run
int tf_launch_attn_fwd(const TfAttnArgs* a, hipStream_t stream);
int tf_launch_attn_bwd(const TfAttnArgs* a, hipStream_t stream);
int tf_launch_attn_bwd_part(const TfAttnArgs* a, int part, hipStream_t stream);   // 1: dQ (+ delta), 2: dK / dV
int tf_launch_attn_fwd_x3(const TfAttnArgs* a, hipStream_t stream);    // fp32-accuracy mode (attn_x3.hip); reached through the two above
int tf_launch_attn_bwd_x3(const TfAttnArgs* a, hipStream_t stream);
int tf_launch_ln_fwd(const TfLnArgs* a, hipStream_t stream);
int tf_launch_ln_bwd(const TfLnArgs* a, hipStream_t stream);
int tf_launch_assemble_fwd(const TfAssembleArgs* a, hipStream_t stream);
int tf_launch_assemble_bwd(const TfAssembleArgs* a, hipStream_t stream);
int tf_launch_pack(const TfPackArgs* a, hipStream_t stream);
int tf_launch_pack_batch(const TfPackArgs* a, int n, hipStream_t stream);   // n <= 8 tensors, one launch
// ... of `groups` parameter sets whose tensors sit src_gstride (sources) / dst_gstride (shadows) BYTES apart
int tf_launch_pack_batch_groups(const TfPackArgs* a, int n, int groups, long long src_gstride, long long dst_gstride, hipStream_t stream);
int tf_launch_copy_rows(const TfCopyRowsArgs* a, hipStream_t stream);
// packed batches: cu [B+1] (by position, longest sample first), start_of [B] (by sample), dense_of [B*S], packed_of_lang [B*Nl] from the
// language padding mask; err[0] = the mask's row total when != expected
// ragged groups: group_nv [TF_MAX_GROUPS] host ints (visual tokens per sample of group g; null / [0] == 0: Nv everywhere) and vis_rows
// [sum_g (B / groups) group_nv[g]] (packed row of every token of the concatenated visual tokens; null unless ragged)
int tf_launch_row_map(const uint8_t* lang_pad_mask, int B, int Nv, int Nl, int* cu, int* start_of, int* dense_of, int* packed_of_lang, int expected,
                      int* err, int groups, const int* group_nv, int* vis_rows, int* err_host /* pinned host word or null */, hipStream_t stream);
int tf_launch_key_mask(const uint8_t* lang_pad_mask, uint8_t* key_mask, int B, int Nv, int Nl, hipStream_t stream);
int tf_launch_dropout_apply(const void* x, void* y, long long n, unsigned key, unsigned thr, float scale, hipStream_t stream);
int tf_launch_attn_dropmask(void* bits, int B, int H, int S, unsigned key, unsigned thr, hipStream_t stream);
int tf_launch_attn_dropmask_packed(void* bits, int B, int H, int S, const int* cu, unsigned key, unsigned thr, hipStream_t stream);
int tf_launch_attn_block_skip(const void* block_bits, int S, void* skip_q, void* skip_k, hipStream_t stream);
int tf_launch_attn_dropmask_rows(void* bits, long long nrows, int S, unsigned key, unsigned thr, hipStream_t stream);
int tf_launch_dropout_mask(uint8_t* out, long long n, unsigned key, unsigned thr, hipStream_t stream);
int tf_launch_cast_f32_bf16(const float* src, void* dst, long long n, hipStream_t stream);
int tf_launch_cast_bf16_f32(const void* src, float* dst, long long n, hipStream_t stream);
int tf_launch_quant_rows_fp8(const void* src, int ld_src, void* dst, int ld_dst, float* scale, int rows, int cols, hipStream_t stream);
int tf_launch_radam(const TfRadamArgs* a, hipStream_t stream);
int tf_launch_heads_loss_fwd(const TfHeadsLossArgs* a, hipStream_t stream);
int tf_launch_heads_loss_bwd(const TfHeadsLossArgs* a, hipStream_t stream);
int tf_launch_softplus_col(const void* x, const void* x_lo, int ld, int col, float* y, const float* dy, void* dx, void* dx_lo, int R, hipStream_t stream);
int tf_launch_pool_norm_fwd(const TfPoolNormArgs* a, hipStream_t stream);
int tf_launch_pool_norm_bwd(const TfPoolNormArgs* a, hipStream_t stream);
int tf_launch_lm_pool_fwd(const TfLmPoolArgs* a, hipStream_t stream);
int tf_launch_lm_pool_bwd(const TfLmPoolArgs* a, hipStream_t stream);
int tf_launch_sumsq(const float* x, long long n, float* out /* atomically accumulated */, hipStream_t stream);
int tf_launch_sumsq_ex(const float* x, long long n, float* out, int accumulate /* 0: out is overwritten */, hipStream_t stream);
int tf_launch_sq_loss(const TfSqLossArgs* a, int backward, hipStream_t stream);
int tf_launch_clock_advance(unsigned* clock, unsigned by, hipStream_t stream);
int tf_tu_set_clock_gemm(const unsigned* clock);                       // per translation unit: where its kernels find the step clock
int tf_tu_set_clock_rowops(const unsigned* clock);
int tf_launch_im2col(const TfPatchArgs* a, hipStream_t stream);       // feat -> cols
int tf_launch_col2im(const TfPatchArgs* a, int out_is_f32, hipStream_t stream);  // cols -> feat (fold; border zero)
int tf_launch_split_planes(const TfPlanesArgs* a, hipStream_t st);

#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
// ---- ragged row ranges (TfGemmArgs.group_rows and its siblings) ----
// The TF_MAX_GROUPS counts are kernel-argument words; every loop below is fully unrolled over CONSTANT indices, so they stay scalar
// registers (a dynamic index would send the array through scratch memory).
struct TfGroupTab { int v[TF_MAX_GROUPS]; };
struct TfRange { int g, lo, n; };                                     // range index, its first row, its row count
__device__ __forceinline__ TfRange tf_range_of_row(const int (&gr)[TF_MAX_GROUPS], int G, int r) {
  TfRange t{0, 0, gr[0]};
  int start = 0;
#pragma unroll
  for (int i = 0; i < TF_MAX_GROUPS; ++i) {
    if (i < G && r >= start) { t.g = i; t.lo = start; t.n = gr[i]; }
    start += gr[i];
  }
  return t;
}
__device__ __forceinline__ TfRange tf_range_of_group(const int (&gr)[TF_MAX_GROUPS], int g) {
  TfRange t{g, 0, 0};
  int start = 0;
#pragma unroll
  for (int i = 0; i < TF_MAX_GROUPS; ++i) {
    if (i == g) { t.lo = start; t.n = gr[i]; }
    start += gr[i];
  }
  return t;
}
// host: are these ragged counts well-formed (G of them, all positive, adding up to M)?
inline bool tf_ragged_ok(const int (&gr)[TF_MAX_GROUPS], int G, int M) {
  if (G < 1 || G > TF_MAX_GROUPS) return false;
  long long sum = 0;
  for (int i = 0; i < TF_MAX_GROUPS; ++i) {
    if (i < G ? gr[i] <= 0 : gr[i] != 0) return false;
    sum += gr[i];
  }
  return sum == M;
}

// Experiment switches (TF_* environment variables read by the launch planners) exist in EXPERIMENTS builds only (-DTF_EXPERIMENTS:
// `python -m transfusion_amd.build --exp`, tools/build_variant.sh -> build/variants/, selected with TFUSION_LIB).  In the shipped
// library the macros are their defaults: no environment variable can change what it computes, and the names are not even in the binary.
#ifdef TF_EXPERIMENTS
#include <cstdlib>
static inline int tf_env_int_(const char* name, int dflt) { const char* v = getenv(name); return v != nullptr ? (int)strtol(v, nullptr, 0) : dflt; }
static inline double tf_env_dbl_(const char* name, double dflt) { const char* v = getenv(name); return v != nullptr ? atof(v) : dflt; }
#define TF_ENV_INT(name, dflt) tf_env_int_(name, dflt)
#define TF_ENV_DBL(name, dflt) tf_env_dbl_(name, dflt)
#else
#define TF_ENV_INT(name, dflt) (dflt)
#define TF_ENV_DBL(name, dflt) (dflt)
#endif
// Launch tracer hook (see tf_trace_start in tfusion.h): a scope object around one kernel launch.  Costs one predictable
// branch when tracing is off.
struct TfTraceScope {
  long long idx; hipStream_t st;
  TfTraceScope(const char* name, hipStream_t stream, double flops = 0.0, double bytes = 0.0);
  ~TfTraceScope();
};
void tf_trace_mark_side(hipStream_t side);
#endif
