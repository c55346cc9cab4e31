// Attention for the fp32-accuracy mode (TfEncoderDesc.precision = 1, BASELINE configs[2] "fp32"): same algorithm and the same
// reference lines as attn_bf16.hip (torch18_adapters.py:756-799, mask merge :578-597), on tensors that are hi + lo bf16 plane
// pairs.  Every S x S x hd contraction is three bf16 MFMA passes into one fp32 accumulator,
//        X . Y  ~=  X_lo . Y_hi  +  X_hi . Y_lo  +  X_hi . Y_hi          (error ~2^-17 per product),
// probabilities / dS leave the softmax as fp32 registers and are split there into their own hi + lo operand fragments; all
// softmax statistics (max, sum, LSE, delta) are fp32.  gfx950 has f32-input MFMA at 1/16 of the bf16 rate; three bf16 passes
// cost 3/16 of that.
//
// Register budget decides the structure: with lo planes the resident operand fragments double, so
//   * every kernel runs one wave per SIMD (512 registers), 4 waves x 32 rows per workgroup, tiles staged load -> LDS directly;
//   * dK and dV are two launches of one kernel (WHICH): a wave that kept K, V (hi + lo) resident AND both accumulators would need
//     ~450 registers at head dim 192.  The dV launch needs only K resident, the dK launch K and V.
// The bf16 kernels' tuning (two waves per SIMD, register prefetch, read-ahead pipelines) is deliberately absent here.
//
// With a dS workspace (TfAttnArgs.ds_work, 2 x tf_attn_ds_bytes: a hi and a lo plane; self attention) the backward computes S and dP
// for dQ no longer: a small kernel forms delta, the dK launch writes its dS tiles -- per (batch, head) a [key][query] bf16 matrix of
// side 128 ceil(S / 128) stored as 32 x 32 tiles of 2 KiB (a wave of the dK launch fills one per query tile, contiguously), twice --
// and attn_bwd_dq_ds_x3_kernel forms dQ = scale dS . K from them with one product instead of three (eight products per layer -> six).
// With 4 x tf_attn_ds_bytes the dK launch (which then runs FIRST) also writes Pd = P . keep / (1 - p), in the register layout the
// dV product consumes it in (a lane's 16-byte B fragments side by side: 1-KiB stores, no transposition), and attn_bwd_dv_pd_x3_kernel
// forms dV^T += dO^T . Pd from it without recomputing S: five products per layer.
#include "tf_common.h"
#include <cstdio>
#include "tf_kernels.h"
#include "attn_common.h"

namespace {

typedef const u16* __restrict__ cu16p;

__device__ __forceinline__ f32x16 mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);       // small terms first
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}
// registers 8s..8s+7 of a 32x32 fp32 accumulator -> hi + lo bf16 B-operand fragments of k-step s
__device__ __forceinline__ void acc_frag_split(const f32x16& x, int s, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = x[8 * s + j];
    const __bf16 hb = (__bf16)v;
    hi[j] = hb;
    lo[j] = (__bf16)(v - (float)hb);
  }
}
// 4 consecutive fp32 values -> 8-B hi and lo stores
__device__ __forceinline__ void store4_split(u16* hi, u16* lo, float a, float b, float c, float d) {
  u32x2 h, l;
  h[0] = pack2bf(a, b); h[1] = pack2bf(c, d);
  l[0] = pack2bf(a - bf2f((u16)(h[0] & 0xffffu)), b - bf2f((u16)(h[0] >> 16)));
  l[1] = pack2bf(c - bf2f((u16)(h[1] & 0xffffu)), d - bf2f((u16)(h[1] >> 16)));
  *(u32x2*)hi = h;
  *(u32x2*)lo = l;
}
// The rows of a set of 32 x 32 accumulator tiles (lane = row, lane half h owns 4 of every 8 columns) as hi + lo planes.  Stored as they
// stand that is two 8-byte stores per group and lane; here the halves swap one group per pair (v_permlane32_swap_b32: see
// attn_fwd_kernel's epilogue, attn_bf16.hip) and store 16 bytes per plane.  Every lane takes part in the swaps; `pred` guards the stores
// (rows past the sample's end), so the row pointers must be computed from a clamped row.
template <int DBLK>
__device__ __forceinline__ void store_acc_rows_split(u16* row_h, u16* row_l, const f32x16 (&acc)[DBLK], float scale, int h, bool pred) {
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      unsigned wh[2][2], wl[2][2];                                  // [even / odd group][word]: hi and lo plane words
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int g4 = 2 * pr + q;
        const float a = acc[d][4 * g4] * scale, b = acc[d][4 * g4 + 1] * scale, c = acc[d][4 * g4 + 2] * scale, e = acc[d][4 * g4 + 3] * scale;
        wh[q][0] = pack2bf(a, b); wh[q][1] = pack2bf(c, e);
        wl[q][0] = pack2bf(a - bf2f((u16)(wh[q][0] & 0xffffu)), b - bf2f((u16)(wh[q][0] >> 16)));
        wl[q][1] = pack2bf(c - bf2f((u16)(wh[q][1] & 0xffffu)), e - bf2f((u16)(wh[q][1] >> 16)));
      }
      const u32x2 h0 = __builtin_amdgcn_permlane32_swap(wh[0][0], wh[1][0], false, false), h1 = __builtin_amdgcn_permlane32_swap(wh[0][1], wh[1][1], false, false);
      const u32x2 l0 = __builtin_amdgcn_permlane32_swap(wl[0][0], wl[1][0], false, false), l1 = __builtin_amdgcn_permlane32_swap(wl[0][1], wl[1][1], false, false);
      if (pred) {
        const int c0 = d * 32 + 8 * (2 * pr + h);
        *(u32x4*)(row_h + c0) = u32x4{h0[0], h1[0], h0[1], h1[1]};
        *(u32x4*)(row_l + c0) = u32x4{l0[0], l1[0], l0[1], l1[1]};
      }
    }
}
// one 64- or 32-row tile of both planes: global -> registers -> LDS (256 threads)
template <int ROWS, int HDP>
__device__ __forceinline__ void stage_pair(cu16p hi, cu16p lo, size_t ld, int row0, int row_max, bool zero_fill, unsigned char* lds_hi,
                                           unsigned char* lds_lo, int tid) {
  TileRegs<ROWS, HDP> r;
  r.load(hi, ld, row0, row_max, zero_fill, tid);
  r.store(lds_hi, tid);
  r.load(lo, ld, row0, row_max, zero_fill, tid);
  r.store(lds_lo, tid);
}

// two tiles, both planes each: ALL four global loads are issued before the first LDS store, so a loop iteration pays one memory
// round trip instead of four (stage_pair x 2 serialises load -> store -> load -> store ...).  dK / dV kernels: 458 -> 386 us and
// 408 -> 334 us.  (Holding the NEXT block's four tiles in registers during the matrix work on top of that changes nothing: 331 / 406 us.)
template <int ROWS, int HDP>
__device__ __forceinline__ void stage_quad(cu16p a_hi, cu16p a_lo, size_t lda, bool a_zero, unsigned char* la_hi, unsigned char* la_lo,
                                           cu16p b_hi, cu16p b_lo, size_t ldb, bool b_zero, unsigned char* lb_hi, unsigned char* lb_lo,
                                           int row0, int row_max, int tid) {
  TileRegs<ROWS, HDP> r0, r1, r2, r3;
  r0.load(a_hi, lda, row0, row_max, a_zero, tid);
  r1.load(a_lo, lda, row0, row_max, a_zero, tid);
  r2.load(b_hi, ldb, row0, row_max, b_zero, tid);
  r3.load(b_lo, ldb, row0, row_max, b_zero, tid);
  r0.store(la_hi, tid);
  r1.store(la_lo, tid);
  r2.store(lb_hi, tid);
  r3.store(lb_lo, tid);
}

// dS workspace of the fp32-accuracy backward: per (batch, head) a [side][side] bf16 matrix, side = 128 ceil(S / 128); the lo plane
// follows the hi plane (plane = tf_attn_ds_bytes(B, H, S) bytes = B H side^2 elements + slack)
__host__ __device__ inline size_t ds_side_x3(int S) { return (size_t)128 * ((S + 127) / 128); }
__host__ __device__ inline size_t ds_plane_x3(int B, int H, int S) { return (size_t)B * H * ds_side_x3(S) * ds_side_x3(S) + 2048; }   // elements

// ================================================================================================
// forward: St[key][q] = K . Q^T (3 passes), online softmax in fp32, O^T += V^T . Pt (3 passes)
// ================================================================================================
template <int HDP>
__global__ __launch_bounds__(256, 1) void attn_fwd_x3_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kt_h = smem;
  unsigned char* kt_l = smem + 64 * G::TSTR;
  unsigned char* vt_h = smem + 128 * G::TSTR;
  unsigned char* vt_l = smem + 192 * G::TSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int S = a.S;                                   // keys
  const int Sq = a.q != nullptr ? a.Sq : S;            // queries
  const int nqb = (Sq + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nqb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int q0 = (logical % nqb) * 128 + wave * 32;
  const size_t ld = a.ld_qkv;
  // S / Sq keep indexing lse / delta / the dropout rows and shaping the grid; Sb / Sqb are THIS sample's row counts and row0 its first
  // row (packed batches: TfAttnArgs.cu_rows; dense: b * S and S)
  const SampleRows sr = sample_rows(a.cu_rows, b, S);
  const bool cross = a.q != nullptr;                            // own query set (TfAttnArgs.q): Sq rows per (batch, head)
  const int Sb = sr.len, Sqb = cross ? Sq : Sb;
  if ((logical % nqb) * 128 >= Sqb) return;                     // query blocks past the sample's end (workgroup-uniform)
  const size_t boff = sr.row0 * ld;
  const size_t qrow0 = cross ? (size_t)b * Sq : sr.row0;        // first query row of this sample in q / out / dout / dq
  const size_t ldq = cross ? (size_t)a.ld_q : ld;
  const size_t qoff = cross ? qrow0 * ldq + (size_t)head * HDP : boff + (size_t)(0 * a.H + head) * HDP;
  cu16p q_h = (cross ? (const u16*)a.q : (const u16*)a.qkv) + qoff, q_l = (cross ? (const u16*)a.q_lo : (const u16*)a.qkv_lo) + qoff;
  cu16p k_h = (const u16*)a.qkv + boff + (size_t)(1 * a.H + head) * HDP, k_l = (const u16*)a.qkv_lo + boff + (size_t)(1 * a.H + head) * HDP;
  cu16p v_h = (const u16*)a.qkv + boff + (size_t)(2 * a.H + head) * HDP, v_l = (const u16*)a.qkv_lo + boff + (size_t)(2 * a.H + head) * HDP;

  bf16x8 qf_h[G::KSTEPS], qf_l[G::KSTEPS];
  {
    const int qr = min(q0 + (lane & 31), Sqb - 1);
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) {
      qf_h[ks] = as_bf16x8(*(const u32x4*)(q_h + (size_t)qr * ldq + ks * 16 + 8 * h));
      qf_l[ks] = as_bf16x8(*(const u32x4*)(q_l + (size_t)qr * ldq + ks * 16 + 8 * h));
    }
  }
  f32x16 o[G::DBLK];
#pragma unroll
  for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  const float sc = a.scale * LOG2E;
  const int qrow = q0 + (lane & 31);
  const int SW = (S + 63) / 64;
  const unsigned long long* drow = a.drop_thr ? (const unsigned long long*)a.drop_bits + ((size_t)bh * Sq + min(qrow, Sqb - 1)) * SW : nullptr;
  const unsigned long long* brow = a.block_bits ? (const unsigned long long*)a.block_bits + (size_t)min(qrow, Sqb - 1) * SW : nullptr;

  const int ntiles = (valid_key_limit(a.key_mask, b, Sb, lane) + 63) / 64;
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * 64;
    __syncthreads();                       // previous tile fully consumed
    if constexpr (HDP <= 192) {
      stage_quad<64, HDP>(k_h, k_l, ld, false, kt_h, kt_l, v_h, v_l, ld, false, vt_h, vt_l, kv0, Sb - 1, tid);
    } else {                                 // head dim 224: four 64-row tiles in flight at once do not fit the register file
      stage_pair<64, HDP>(k_h, k_l, ld, kv0, Sb - 1, false, kt_h, kt_l, tid);
      stage_pair<64, HDP>(v_h, v_l, ld, kv0, Sb - 1, false, vt_h, vt_l, tid);
    }
    __syncthreads();
    const unsigned long long dm = a.drop_thr ? (drow[t] >> (4 * h)) : ~0ull;
    const unsigned long long vall = key_bits(a.key_mask, b, Sb, kv0, lane);
    const unsigned long long blk = brow ? brow[t] : 0ull;
    const unsigned long long vbits = (vall & ~blk) >> (4 * h);

    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks)
        st[kb] = mfma3(row_frag<HDP>(kt_h, kb * 32, ks, lane), row_frag<HDP>(kt_l, kb * 32, ks, lane), qf_h[ks], qf_l[ks], st[kb]);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (!((vbits >> (kb * 32 + (r & 3) + 8 * (r >> 2))) & 1ull)) st[kb][r] = -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc;
    const bool need = mx - m_run > RESCALE_THR;   // per row (see attn_fwd_kernel): rows that keep their max multiply by exactly 1
    if (__any(need)) {
      const float m_new = need ? fmaxf(m_run, mx) : m_run;
      const float alpha = need ? fast_exp2(m_run - m_new) : 1.0f;
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(st[kb][r], sc, -m_run));
        psum += p;
        st[kb][r] = ((dm >> (kb * 32 + (r & 3) + 8 * (r >> 2))) & 1ull) ? p : 0.f;
      }
    l_run += psum;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf_h, pf_l;
        acc_frag_split(st[kb], s, pf_h, pf_l);
#pragma unroll
        for (int d = 0; d < G::DBLK; ++d)
          o[d] = mfma3(tr_frag<HDP>(vt_h, kb * 32 + 16 * s, d * 32, lane), tr_frag<HDP>(vt_l, kb * 32 + 16 * s, d * 32, lane), pf_h, pf_l, o[d]);
      }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = (a.drop_thr ? a.drop_scale : 1.0f) / l_tot;
  {
    const size_t off = (qrow0 + min(qrow, Sqb - 1)) * a.ld_out + (size_t)head * HDP;
    store_acc_rows_split<G::DBLK>((u16*)a.out + off, (u16*)a.out_lo + off, o, inv, h, qrow < Sqb);
    if (qrow < Sqb && h == 0 && a.lse != nullptr) a.lse[(size_t)bh * Sq + qrow] = m_run + log2f(l_tot);
  }
}

// ================================================================================================
// backward, dQ: query on the lane, loop over key tiles (see attn_bwd_dq_kernel)
// ================================================================================================
template <int HDP>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_x3_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kt_h = smem;
  unsigned char* kt_l = smem + 64 * G::TSTR;
  unsigned char* vt_h = smem + 128 * G::TSTR;
  unsigned char* vt_l = smem + 192 * G::TSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int S = a.S;                                   // keys
  const int Sq = a.q != nullptr ? a.Sq : S;            // queries
  const int nqb = (Sq + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nqb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int q0 = (logical % nqb) * 128 + wave * 32;
  const size_t ld = a.ld_qkv;
  // S / Sq keep indexing lse / delta / the dropout rows and shaping the grid; Sb / Sqb are THIS sample's row counts and row0 its first
  // row (packed batches: TfAttnArgs.cu_rows; dense: b * S and S)
  const SampleRows sr = sample_rows(a.cu_rows, b, S);
  const bool cross = a.q != nullptr;                            // own query set (TfAttnArgs.q): Sq rows per (batch, head)
  const int Sb = sr.len, Sqb = cross ? Sq : Sb;
  if ((logical % nqb) * 128 >= Sqb) return;                     // query blocks past the sample's end (workgroup-uniform)
  const size_t boff = sr.row0 * ld;
  const size_t qrow0 = cross ? (size_t)b * Sq : sr.row0;        // first query row of this sample in q / out / dout / dq
  const size_t ldq = cross ? (size_t)a.ld_q : ld;
  const size_t qoff = cross ? qrow0 * ldq + (size_t)head * HDP : boff + (size_t)(0 * a.H + head) * HDP;
  cu16p q_h = (cross ? (const u16*)a.q : (const u16*)a.qkv) + qoff, q_l = (cross ? (const u16*)a.q_lo : (const u16*)a.qkv_lo) + qoff;
  cu16p k_h = (const u16*)a.qkv + boff + (size_t)(1 * a.H + head) * HDP, k_l = (const u16*)a.qkv_lo + boff + (size_t)(1 * a.H + head) * HDP;
  cu16p v_h = (const u16*)a.qkv + boff + (size_t)(2 * a.H + head) * HDP, v_l = (const u16*)a.qkv_lo + boff + (size_t)(2 * a.H + head) * HDP;
  const size_t dooff = qrow0 * a.ld_dout + (size_t)head * HDP;
  cu16p do_h = (const u16*)a.dout + dooff, do_l = (const u16*)a.dout_lo + dooff;

  const int qrow = q0 + (lane & 31);
  const int qr = min(qrow, Sqb - 1);
  bf16x8 qf_h[G::KSTEPS], qf_l[G::KSTEPS], dof_h[G::KSTEPS], dof_l[G::KSTEPS];
#pragma unroll
  for (int ks = 0; ks < G::KSTEPS; ++ks) {
    qf_h[ks] = as_bf16x8(*(const u32x4*)(q_h + (size_t)qr * ldq + ks * 16 + 8 * h));
    qf_l[ks] = as_bf16x8(*(const u32x4*)(q_l + (size_t)qr * ldq + ks * 16 + 8 * h));
    dof_h[ks] = as_bf16x8(*(const u32x4*)(do_h + (size_t)qr * a.ld_dout + ks * 16 + 8 * h));
    dof_l[ks] = as_bf16x8(*(const u32x4*)(do_l + (size_t)qr * a.ld_dout + ks * 16 + 8 * h));
  }
  const float lse = a.lse[(size_t)bh * Sq + qr];
  float delta = 0.f;                       // rowsum(dO . O) in fp32 from both planes of both tensors
  {
    const size_t ooff = (qrow0 + qr) * a.ld_out + (size_t)head * HDP;
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) {
      float of[8], df[8];
      load8_split(a.out, a.out_lo, ooff + ks * 16 + 8 * h, of);
      join8(__builtin_bit_cast(u32x4, dof_h[ks]), __builtin_bit_cast(u32x4, dof_l[ks]), df);
#pragma unroll
      for (int e = 0; e < 8; ++e) delta = fmaf(of[e], df[e], delta);
    }
    delta += __shfl_xor(delta, 32, 64);
    if (h == 0 && qrow < Sqb) a.delta[(size_t)bh * Sq + qrow] = delta;
  }
  f32x16 dq[G::DBLK];
#pragma unroll
  for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
  const float sc = a.scale * LOG2E;
  const int SW = (S + 63) / 64;
  const unsigned long long* drow = a.drop_thr ? (const unsigned long long*)a.drop_bits + ((size_t)bh * Sq + qr) * SW : nullptr;
  const unsigned long long* brow = a.block_bits ? (const unsigned long long*)a.block_bits + (size_t)qr * SW : nullptr;
  const float dscale = a.drop_thr ? a.drop_scale : 1.0f;

  const int ntiles = (valid_key_limit(a.key_mask, b, Sb, lane) + 63) / 64;
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * 64;
    __syncthreads();
    if constexpr (HDP <= 192) {
      stage_quad<64, HDP>(k_h, k_l, ld, false, kt_h, kt_l, v_h, v_l, ld, false, vt_h, vt_l, kv0, Sb - 1, tid);
    } else {                                 // head dim 224: four 64-row tiles in flight at once do not fit the register file
      stage_pair<64, HDP>(k_h, k_l, ld, kv0, Sb - 1, false, kt_h, kt_l, tid);
      stage_pair<64, HDP>(v_h, v_l, ld, kv0, Sb - 1, false, vt_h, vt_l, tid);
    }
    __syncthreads();
    const unsigned long long dm = a.drop_thr ? (drow[t] >> (4 * h)) : ~0ull;
    const unsigned long long vall = key_bits(a.key_mask, b, Sb, kv0, lane);
    const unsigned long long blk = brow ? brow[t] : 0ull;
    const unsigned long long vbits = (vall & ~blk) >> (4 * h);
#pragma unroll 1                     // the two 32-key halves one after the other: interleaved by the unroller the kernel spills at head dim 192
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks)
        st = mfma3(row_frag<HDP>(kt_h, kb * 32, ks, lane), row_frag<HDP>(kt_l, kb * 32, ks, lane), qf_h[ks], qf_l[ks], st);
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks)
        dp = mfma3(row_frag<HDP>(vt_h, kb * 32, ks, lane), row_frag<HDP>(vt_l, kb * 32, ks, lane), dof_h[ks], dof_l[ks], dp);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int bit = kb * 32 + (r & 3) + 8 * (r >> 2);
        const float p = ((vbits >> bit) & 1ull) ? fast_exp2(fmaf(st[r], sc, -lse)) : 0.f;
        const float ks = ((dm >> bit) & 1ull) ? dscale : 0.f;
        st[r] = p * fmaf(dp[r], ks, -delta);          // dSt
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 ds_h, ds_l;
        acc_frag_split(st, s, ds_h, ds_l);
#pragma unroll
        for (int d = 0; d < G::DBLK; ++d)
          dq[d] = mfma3(tr_frag<HDP>(kt_h, kb * 32 + 16 * s, d * 32, lane), tr_frag<HDP>(kt_l, kb * 32 + 16 * s, d * 32, lane), ds_h, ds_l, dq[d]);
      }
    }
  }
  {
    const size_t qc = (size_t)min(qrow, Sqb - 1);
    const size_t off = cross ? (qrow0 + qc) * a.ld_dq + (size_t)head * HDP : (sr.row0 + qc) * a.ld_dqkv + (size_t)(0 * a.H + head) * HDP;
    store_acc_rows_split<G::DBLK>((cross ? (u16*)a.dq : (u16*)a.dqkv) + off, (cross ? (u16*)a.dq_lo : (u16*)a.dqkv_lo) + off, dq, a.scale, h, qrow < Sqb);
  }
}

// ================================================================================================
// backward, dV (WHICH = 0) or dK (WHICH = 1): key on the lane, loop over query tiles of 32 (see attn_bwd_dkv_kernel)
// (dV and dK in ONE pass -- S and dP once for both, four products per tile instead of five -- was built and works up to head dim 160:
// two accumulator sets + the K fragments are 288 registers at head dim 192, so the V rows have to come from LDS, and 128 V rows in
// both planes (147 KB at the dual-use tile stride, 100 KB packed) do not fit beside the 74 KB of Q / dO tiles.  The benchmark's and
// the reference's fp32 configurations have head dims 192 and 224: the variant is not in the tree.)
//   S[q][key] = Q.K^T -> P ;  dP = dO.V^T ;  Pd = P*keep/(1-p) ;  dS = P*(keep/(1-p)*dP - delta)
//   dV^T[d][key] += dO^T[d][q] . Pd[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key] ; dK *= scale
// ================================================================================================
template <int HDP, int WHICH>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_x3_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* qt_h = smem;
  unsigned char* qt_l = smem + 32 * G::TSTR;
  unsigned char* dot_h = smem + 64 * G::TSTR;
  unsigned char* dot_l = smem + 96 * G::TSTR;
  float* lse_s = (float*)(smem + 128 * G::TSTR);
  float* del_s = lse_s + 32;
  unsigned* dw_s = (unsigned*)(del_s + 32);                       // [4 waves][32 query rows] keep-bit words of the wave's 32 keys
  unsigned* bw_s = dw_s + 128;                                    // [4 waves][32 query rows] block-bit words
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int S = a.S;                                   // keys
  const int Sq = a.q != nullptr ? a.Sq : S;            // queries
  const int nkb = (S + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nkb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int key0 = (logical % nkb) * 128 + wave * 32;
  const size_t ld = a.ld_qkv;
  const SampleRows sr = sample_rows(a.cu_rows, b, S);            // packed batches: see attn_fwd_x3_kernel
  const bool cross = a.q != nullptr;                            // own query set (TfAttnArgs.q): Sq rows per (batch, head)
  const int Sb = sr.len, Sqb = cross ? Sq : Sb;
  if ((logical % nkb) * 128 >= Sb) return;                      // key blocks past the sample's end (workgroup-uniform)
  const size_t boff = sr.row0 * ld;
  const size_t qrow0 = cross ? (size_t)b * Sq : sr.row0;
  const size_t ldq = cross ? (size_t)a.ld_q : ld;
  const size_t qoff = cross ? qrow0 * ldq + (size_t)head * HDP : boff + (size_t)(0 * a.H + head) * HDP;
  cu16p q_h = (cross ? (const u16*)a.q : (const u16*)a.qkv) + qoff, q_l = (cross ? (const u16*)a.q_lo : (const u16*)a.qkv_lo) + qoff;
  cu16p k_h = (const u16*)a.qkv + boff + (size_t)(1 * a.H + head) * HDP, k_l = (const u16*)a.qkv_lo + boff + (size_t)(1 * a.H + head) * HDP;
  cu16p v_h = (const u16*)a.qkv + boff + (size_t)(2 * a.H + head) * HDP, v_l = (const u16*)a.qkv_lo + boff + (size_t)(2 * a.H + head) * HDP;
  const size_t dooff = qrow0 * a.ld_dout + (size_t)head * HDP;
  cu16p do_h = (const u16*)a.dout + dooff, do_l = (const u16*)a.dout_lo + dooff;

  const int key = key0 + (lane & 31);
  const int kr_ = min(key, Sb - 1);
  bool key_ok = key < Sb;
  if (key_ok && a.key_mask != nullptr) key_ok = a.key_mask[(size_t)b * S + key] == 0;
  constexpr int NV = WHICH == 1 ? G::KSTEPS : 1;                  // V fragments are resident only in the dK launch
  bf16x8 kf_h[G::KSTEPS], kf_l[G::KSTEPS], vf_h[NV], vf_l[NV];
#pragma unroll
  for (int ks = 0; ks < G::KSTEPS; ++ks) {
    kf_h[ks] = as_bf16x8(*(const u32x4*)(k_h + (size_t)kr_ * ld + ks * 16 + 8 * h));
    kf_l[ks] = as_bf16x8(*(const u32x4*)(k_l + (size_t)kr_ * ld + ks * 16 + 8 * h));
    if constexpr (WHICH == 1) {
      vf_h[ks] = as_bf16x8(*(const u32x4*)(v_h + (size_t)kr_ * ld + ks * 16 + 8 * h));
      vf_l[ks] = as_bf16x8(*(const u32x4*)(v_l + (size_t)kr_ * ld + ks * 16 + 8 * h));
    }
  }
  f32x16 acc[G::DBLK];
#pragma unroll
  for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  const float sc = a.scale * LOG2E;
  const float dscale = a.drop_thr ? a.drop_scale : 1.0f;
  const bool blk = a.block_bits != nullptr;
  // dS rows of this lane's key in the workspace planes (WHICH == 1 with TfAttnArgs.ds_work, self attention): see the file header
  u16* ds_h = nullptr; u16* ds_l = nullptr; u16* pd_h = nullptr; u16* pd_l = nullptr;
  if constexpr (WHICH == 1) {
    if (a.ds_work != nullptr && !cross) {
      const size_t nb = ds_side_x3(S) / 32;                        // 32 x 32 tiles per side
      // tile (key0 / 32, query tile t) of this (batch, head); inside it row (lane & 31), the 16-B chunk this lane stores (see the loop)
      ds_h = (u16*)a.ds_work + (((size_t)bh * nb + (key0 >> 5)) * nb) * 1024 + (lane & 31) * 32;
      ds_l = ds_h + ds_plane_x3(a.B, a.H, S);
      if (a.ds_planes >= 4) {               // Pd tiles: [k-step s2][lane][8 values], hi plane 2, lo plane 3
        pd_h = (u16*)a.ds_work + 2 * ds_plane_x3(a.B, a.H, S) + (((size_t)bh * nb + (key0 >> 5)) * nb) * 1024 + lane * 8;
        pd_l = pd_h + ds_plane_x3(a.B, a.H, S);
      }
    }
  }

  const int ntiles = ((logical % nkb) * 128 >= valid_key_limit(a.key_mask, b, Sb, lane)) ? 0 : (Sqb + 31) / 32;
  const int dw_ld = 2 * ((S + 63) / 64);
  const int wsel = min(key0 >> 5, dw_ld - 1);      // (waves whose keys all lie past S: the row's last word, never used -- attn_bf16.hip)
  const unsigned* dbits = (const unsigned*)a.drop_bits + (size_t)bh * Sq * dw_ld + wsel;
  const unsigned* bbits = blk ? (const unsigned*)a.block_bits + wsel : nullptr;
  // the per-row scalars of a query block (LSE, delta, keep / block bit words) are fetched ONE BLOCK AHEAD into four registers: read
  // in place they were a dependent global load between the two barriers of every block
  auto row_scalars = [&](int q0, float& r_lse, float& r_del, unsigned& r_dw, unsigned& r_bw) {
    const int row = q0 + (lane & 31);
    const bool in = row < Sqb;
    const int q = min(row, Sqb - 1);
    r_lse = in ? a.lse[(size_t)bh * Sq + q] : 1.0e30f;             // P = 0 for rows past the end
    r_del = in ? a.delta[(size_t)bh * Sq + q] : 0.f;
    r_dw = a.drop_thr ? (in ? dbits[(size_t)q * dw_ld] : 0u) : 0xffffffffu;
    r_bw = blk ? bbits[(size_t)q * dw_ld] : 0u;
  };
  float n_lse = 0.f, n_del = 0.f;
  unsigned n_dw = 0u, n_bw = 0u;
  if (ntiles > 0) row_scalars(0, n_lse, n_del, n_dw, n_bw);
  // DEFER (the dK launch that stores dS / Pd tiles, head dims <= 192): the 8 tile stores of tile t are ISSUED at the start of tile
  // t + 1's matrix phase.  vmcnt retires in issue order, so the staging wait of the next tile (global loads -> LDS) also waits for
  // every older store: issued right behind the softmax they had only the 36 MFMAs of the dK product to complete in, and the wait
  // exposed the rest -- 391 us per launch against 293 with the stores ablated.  Deferred they get a whole tile's matrix work.
  // 32 registers carry the packed tiles across the staging (head dim 224 has none to spare: it stores at once).
  constexpr bool DEFER = WHICH == 1 && HDP <= 192;
  u32x4 pend[8];
  auto flush_pending = [&](int tp) {          // tile tp's dS chunks (hi, lo) x 2 and Pd fragments (hi, lo) x 2
    u16* th = ds_h + (size_t)tp * 1024;
    u16* tl = ds_l + (size_t)tp * 1024;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      *(u32x4*)(th + 8 * (2 * c2 + h)) = pend[2 * c2];
      *(u32x4*)(tl + 8 * (2 * c2 + h)) = pend[2 * c2 + 1];
    }
    if (pd_h != nullptr) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        *(u32x4*)(pd_h + (size_t)tp * 1024 + s2 * 512) = pend[4 + 2 * s2];
        *(u32x4*)(pd_l + (size_t)tp * 1024 + s2 * 512) = pend[5 + 2 * s2];
      }
    }
  };
  for (int t = 0; t < ntiles; ++t) {
    const int q0 = t * 32;
    __syncthreads();
    stage_quad<32, HDP>(q_h, q_l, ldq, false, qt_h, qt_l, do_h, do_l, a.ld_dout, true, dot_h, dot_l, q0, Sqb - 1, tid);   // dO rows >= Sqb: zero
    if (tid < 32) { lse_s[tid] = n_lse; del_s[tid] = n_del; }
    if (lane < 32) { dw_s[wave * 32 + lane] = n_dw; bw_s[wave * 32 + lane] = n_bw; }
    if (t + 1 < ntiles) row_scalars(q0 + 32, n_lse, n_del, n_dw, n_bw);
    __syncthreads();
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks)
      st = mfma3(row_frag<HDP>(qt_h, 0, ks, lane), row_frag<HDP>(qt_l, 0, ks, lane), kf_h[ks], kf_l[ks], st);
    if constexpr (DEFER) { if (ds_h != nullptr && t > 0) flush_pending(t - 1); }      // (between the S and the dP product: see DEFER)
    if constexpr (WHICH == 1) {
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks)
        dp = mfma3(row_frag<HDP>(dot_h, 0, ks, lane), row_frag<HDP>(dot_l, 0, ks, lane), vf_h[ks], vf_l[ks], dp);
    }
    // registers 4g..4g+3 are query rows 8g + 4h + (0..3)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 l4 = *(const f32x4*)(lse_s + 8 * g4 + 4 * h);
      const f32x4 d4 = *(const f32x4*)(del_s + 8 * g4 + 4 * h);
      const u32x4 w4 = *(const u32x4*)(dw_s + wave * 32 + 8 * g4 + 4 * h);
      const u32x4 b4 = *(const u32x4*)(bw_s + wave * 32 + 8 * g4 + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g4 + i;
        const bool att = key_ok && !((b4[i] >> (lane & 31)) & 1u);
        const float p = att ? fast_exp2(fmaf(st[r], sc, -l4[i])) : 0.f;
        const float keep_scale = ((w4[i] >> (lane & 31)) & 1u) ? dscale : 0.f;
        if constexpr (WHICH == 0) st[r] = p * keep_scale;                              // Pd
        else {
          st[r] = p * fmaf(dp[r], keep_scale, -d4[i]);                                 // dS
          dp[r] = p * keep_scale;                                                      // Pd (dP is spent): stored below for the dV kernel
        }
      }
    }
    if constexpr (WHICH == 1) {
      if (pd_h != nullptr) {              // (workgroup-uniform) the two B fragments of the dV product, as they sit in this lane
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 f_h, f_l;
          acc_frag_split(dp, s2, f_h, f_l);
          if constexpr (DEFER) {
            pend[4 + 2 * s2] = __builtin_bit_cast(u32x4, f_h); pend[5 + 2 * s2] = __builtin_bit_cast(u32x4, f_l);
          } else {
            *(bf16x8*)(pd_h + (size_t)t * 1024 + s2 * 512) = f_h;
            *(bf16x8*)(pd_l + (size_t)t * 1024 + s2 * 512) = f_l;
          }
        }
      }
    }
    if constexpr (WHICH == 1) {
      if (ds_h != nullptr) {              // (workgroup-uniform) this wave's 32 keys x 32 queries of dS: one contiguous 2-KiB tile per plane
        // registers 4g..4g+3 are queries 8g + 4h + (0..3): lanes l and l ^ 32 hold the two halves of every 16-B chunk (8 queries) of
        // their key's row.  Lane half h keeps chunks h and h + 2 and gets the other lane's half of them: 16-B stores, whole rows.
        u32x2 hi[4], lo[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          hi[g4][0] = pack2bf(st[4 * g4], st[4 * g4 + 1]); hi[g4][1] = pack2bf(st[4 * g4 + 2], st[4 * g4 + 3]);
          lo[g4][0] = pack2bf(st[4 * g4] - bf2f((u16)(hi[g4][0] & 0xffffu)), st[4 * g4 + 1] - bf2f((u16)(hi[g4][0] >> 16)));
          lo[g4][1] = pack2bf(st[4 * g4 + 2] - bf2f((u16)(hi[g4][1] & 0xffffu)), st[4 * g4 + 3] - bf2f((u16)(hi[g4][1] >> 16)));
        }
        u16* th = ds_h + (size_t)t * 1024;
        u16* tl = ds_l + (size_t)t * 1024;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {                              // chunk 2 c2 + h is this lane's; it gives away chunk 2 c2 + (1 - h)
          const u32x2 mine_h = h ? hi[2 * c2 + 1] : hi[2 * c2], give_h = h ? hi[2 * c2] : hi[2 * c2 + 1];
          const u32x2 mine_l = h ? lo[2 * c2 + 1] : lo[2 * c2], give_l = h ? lo[2 * c2] : lo[2 * c2 + 1];
          u32x2 got_h, got_l;
          got_h[0] = (unsigned)__shfl_xor((int)give_h[0], 32, 64); got_h[1] = (unsigned)__shfl_xor((int)give_h[1], 32, 64);
          got_l[0] = (unsigned)__shfl_xor((int)give_l[0], 32, 64); got_l[1] = (unsigned)__shfl_xor((int)give_l[1], 32, 64);
          const u32x4 vh = h ? u32x4{got_h[0], got_h[1], mine_h[0], mine_h[1]} : u32x4{mine_h[0], mine_h[1], got_h[0], got_h[1]};
          const u32x4 vl = h ? u32x4{got_l[0], got_l[1], mine_l[0], mine_l[1]} : u32x4{mine_l[0], mine_l[1], got_l[0], got_l[1]};
          if constexpr (DEFER) { pend[2 * c2] = vh; pend[2 * c2 + 1] = vl; }
          else {
            *(u32x4*)(th + 8 * (2 * c2 + h)) = vh;
            *(u32x4*)(tl + 8 * (2 * c2 + h)) = vl;
          }
        }
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 f_h, f_l;
      acc_frag_split(st, s2, f_h, f_l);
      const unsigned char* th = WHICH == 0 ? dot_h : qt_h;
      const unsigned char* tl = WHICH == 0 ? dot_l : qt_l;
#pragma unroll
      for (int d = 0; d < G::DBLK; ++d)
        acc[d] = mfma3(tr_frag<HDP>(th, 16 * s2, d * 32, lane), tr_frag<HDP>(tl, 16 * s2, d * 32, lane), f_h, f_l, acc[d]);
    }
  }
  if constexpr (DEFER) { if (ds_h != nullptr && ntiles > 0) flush_pending(ntiles - 1); }
  {
    const size_t off = (sr.row0 + min(key, Sb - 1)) * a.ld_dqkv + (size_t)((WHICH == 0 ? 2 : 1) * a.H + head) * HDP;
    store_acc_rows_split<G::DBLK>((u16*)a.dqkv + off, (u16*)a.dqkv_lo + off, acc, WHICH == 0 ? 1.0f : a.scale, h, key < Sb);
  }
}

// ================================================================================================
// backward, dV from the Pd tiles the dK launch wrote:  dV^T[d][key] += dO^T[d][q] . Pd[q][key]  (3 passes), key on the lane.
// One product instead of the two of attn_bwd_dkv_x3_kernel<.., 0> (no S, no softmax, no Q tile, no K fragments): per 32-query tile
// the dO tile (both planes) is staged for the workgroup, the next tile's loads in flight during the matrix work, and a lane's two B
// fragments per plane are two 16-byte loads from the tile the producer lane of the same index stored.
// ================================================================================================
template <int HDP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dv_pd_x3_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* dot_h = smem;
  unsigned char* dot_l = smem + 32 * G::TSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int S = a.S;
  const int nkb = (S + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nkb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int key0 = (logical % nkb) * 128 + wave * 32;
  const SampleRows sr = sample_rows(a.cu_rows, b, S);
  const int Sb = sr.len;
  if ((logical % nkb) * 128 >= Sb) return;                      // key blocks past the sample's end (workgroup-uniform)
  const size_t dooff = sr.row0 * a.ld_dout + (size_t)head * HDP;
  cu16p do_h = (const u16*)a.dout + dooff, do_l = (const u16*)a.dout_lo + dooff;
  const size_t nb = ds_side_x3(S) / 32;
  cu16p pd_h = (const u16*)a.ds_work + 2 * ds_plane_x3(a.B, a.H, S) + (((size_t)bh * nb + (key0 >> 5)) * nb) * 1024 + lane * 8;
  cu16p pd_l = pd_h + ds_plane_x3(a.B, a.H, S);
  f32x16 acc[G::DBLK];
#pragma unroll
  for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  // (the dK launch ran the same tile loop: it wrote a tile for every t below, zeros where keys or queries are masked or past the end)
  const int ntiles = ((logical % nkb) * 128 >= valid_key_limit(a.key_mask, b, Sb, lane)) ? 0 : (Sb + 31) / 32;
  TileRegs<32, HDP> r_h, r_l;
  bf16x8 f_h[2], f_l[2];
  auto fetch = [&](int t) {
    r_h.load(do_h, a.ld_dout, t * 32, Sb - 1, true, tid);       // dO rows past the end: zeros
    r_l.load(do_l, a.ld_dout, t * 32, Sb - 1, true, tid);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f_h[s2] = *(const bf16x8*)(pd_h + (size_t)t * 1024 + s2 * 512);
      f_l[s2] = *(const bf16x8*)(pd_l + (size_t)t * 1024 + s2 * 512);
    }
  };
  if (ntiles > 0) fetch(0);
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();                                            // every wave is done with the previous tile
    r_h.store(dot_h, tid); r_l.store(dot_l, tid);
    const bf16x8 c_h0 = f_h[0], c_h1 = f_h[1], c_l0 = f_l[0], c_l1 = f_l[1];
    __syncthreads();
    if (t + 1 < ntiles) fetch(t + 1);
#pragma unroll
    for (int d = 0; d < G::DBLK; ++d) {
      acc[d] = mfma3(tr_frag<HDP>(dot_h, 0, d * 32, lane), tr_frag<HDP>(dot_l, 0, d * 32, lane), c_h0, c_l0, acc[d]);
      acc[d] = mfma3(tr_frag<HDP>(dot_h, 16, d * 32, lane), tr_frag<HDP>(dot_l, 16, d * 32, lane), c_h1, c_l1, acc[d]);
    }
  }
  const int key = key0 + (lane & 31);
  {
    const size_t off = (sr.row0 + min(key, Sb - 1)) * a.ld_dqkv + (size_t)(2 * a.H + head) * HDP;
    store_acc_rows_split<G::DBLK>((u16*)a.dqkv + off, (u16*)a.dqkv_lo + off, acc, 1.0f, h, key < Sb);
  }
}

// ================================================================================================
// delta[b, h, q] = rowsum(dO . O) in fp32 from both planes of both tensors, one wave per token row (the dS path: the dQ kernel that
// used to form it is not launched)
// ================================================================================================
__global__ __launch_bounds__(256) void attn_delta_x3_kernel(const TfAttnArgs a) {
  __shared__ float part[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = a.S, nrb = (S + 3) / 4;
  const int b = blockIdx.x / nrb, q = (blockIdx.x % nrb) * 4 + wave;
  const SampleRows sr = sample_rows(a.cu_rows, b, S);
  if (q >= sr.len) return;                                 // (no barrier below: waves are independent)
  const int cph = a.HDP / 8, nch = a.H * cph;             // 16-B chunks per head / per row (<= 256)
  const size_t oo = (sr.row0 + q) * a.ld_out, od = (sr.row0 + q) * a.ld_dout;
  for (int c = lane; c < nch; c += 64) {
    float of[8], df[8];
    load8_split(a.out, a.out_lo, oo + c * 8, of);
    load8_split(a.dout, a.dout_lo, od + c * 8, df);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(of[e], df[e], s);
    part[wave][c] = s;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's LDS writes have landed (wave-private rows of `part`)
  __builtin_amdgcn_wave_barrier();
  if (lane < a.H) {
    float s = 0.f;
    for (int c = 0; c < cph; ++c) s += part[wave][lane * cph + c];
    a.delta[((size_t)b * a.H + lane) * S + q] = s;
  }
}

// ================================================================================================
// backward, dQ from the dS planes the dK launch wrote:  dQ^T[d][q] += K^T[d][key] . dS^T[key][q]  (3 passes);  dQ = scale * dQ^T^T.
// 4 waves x 32 queries per workgroup; per 64-key tile the K tile (both planes) and the [64 keys][128 queries] block of dS (both planes)
// are staged for the workgroup, the next tile's loads in flight during the matrix work; a wave's B operand is the transposed read of
// ITS 32 query columns -- tr_frag hands lane (query n) the 8 keys of a k-step in the order the A operand's K^T fragment uses.
// ================================================================================================
// KT = keys per staged tile: 32 halves the LDS of a workgroup (57 KB at head dim 192) so that TWO share a CU (252 registers)
template <int HDP, int KT>
__global__ __launch_bounds__(256, (KT == 32 && HDP <= 192) ? 2 : 1) void attn_bwd_dq_ds_x3_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  using GD = Geo<128>;                                          // the dS block as a tile of 128 columns
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kt_h = smem;
  unsigned char* kt_l = smem + KT * G::TSTR;
  unsigned char* dt_h = smem + 2 * KT * G::TSTR;
  unsigned char* dt_l = dt_h + KT * GD::TSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int S = a.S;
  const int nqb = (S + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nqb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int qblk = logical % nqb;
  const SampleRows sr = sample_rows(a.cu_rows, b, S);
  const int Sb = sr.len;
  if (qblk * 128 >= Sb) return;                                 // query blocks past the sample's end (workgroup-uniform)
  const size_t ld = a.ld_qkv;
  cu16p k_h = (const u16*)a.qkv + sr.row0 * ld + (size_t)(1 * a.H + head) * HDP, k_l = (const u16*)a.qkv_lo + sr.row0 * ld + (size_t)(1 * a.H + head) * HDP;
  const size_t nb = ds_side_x3(S) / 32;                          // 32 x 32 tiles (2 KiB) per side
  cu16p d_h = (const u16*)a.ds_work + ((size_t)bh * nb * nb + (size_t)qblk * 4) * 1024, d_l = d_h + ds_plane_x3(a.B, a.H, S);
  const int ntiles = (valid_key_limit(a.key_mask, b, Sb, lane) + KT - 1) / KT;

  f32x16 dq[G::DBLK];
#pragma unroll
  for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

  TileRegs<KT, HDP> rk_h, rk_l;
  constexpr int NP = KT / 16;                                   // 16-byte pieces of the dS block per thread and plane
  // the [64 keys][128 queries] block of dS = 2 x 4 tiles of 2 KiB per plane: piece id = i * 256 + tid -> tile id >> 7 (key half kbl = tile
  // >> 2, query tile qt = tile & 3), row (id & 127) >> 2, 16-B piece id & 3: a tile is read by 128 consecutive threads, contiguously
  u32x4 rd_h[NP], rd_l[NP];
  unsigned d_off[NP];                                            // element offset of piece i inside key tile 0 (a key tile further: 2 nb tiles)
  int d_lds[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int id = i * 256 + tid, tile = id >> 7, kbl = tile >> 2, qt = tile & 3, row = (id & 127) >> 2, pc = id & 3;
    d_off[i] = (unsigned)((kbl * nb + qt) * 1024 + row * 32 + pc * 8);
    d_lds[i] = tile_off(kbl * 32 + row, qt * 4 + pc, GD::TSTR);
  }
  auto fetch = [&](int t) {
    rk_h.load(k_h, ld, t * KT, Sb - 1, false, tid);
    rk_l.load(k_l, ld, t * KT, Sb - 1, false, tid);
    const size_t tb = (size_t)t * (KT / 32) * nb * 1024;                // (the dK launch wrote every tile a key tile reads: whole 128-key workgroups)
#pragma unroll
    for (int i = 0; i < NP; ++i) { rd_h[i] = *(const u32x4*)(d_h + tb + d_off[i]); rd_l[i] = *(const u32x4*)(d_l + tb + d_off[i]); }
  };
  if (ntiles > 0) fetch(0);
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();                                            // every wave is done with the previous tiles
    rk_h.store(kt_h, tid); rk_l.store(kt_l, tid);
#pragma unroll
    for (int i = 0; i < NP; ++i) { *(u32x4*)(dt_h + d_lds[i]) = rd_h[i]; *(u32x4*)(dt_l + d_lds[i]) = rd_l[i]; }
    __syncthreads();
    if (t + 1 < ntiles) fetch(t + 1);
#pragma unroll 1
    for (int kb = 0; kb < KT / 32; ++kb) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 ds_hi = tr_frag<128>(dt_h, kb * 32 + 16 * s, wave * 32, lane);
        const bf16x8 ds_lo = tr_frag<128>(dt_l, kb * 32 + 16 * s, wave * 32, lane);
#pragma unroll
        for (int d = 0; d < G::DBLK; ++d)
          dq[d] = mfma3(tr_frag<HDP>(kt_h, kb * 32 + 16 * s, d * 32, lane), tr_frag<HDP>(kt_l, kb * 32 + 16 * s, d * 32, lane), ds_hi, ds_lo, dq[d]);
      }
    }
  }
  const int qrow = qblk * 128 + wave * 32 + (lane & 31);
  {
    const size_t off = (sr.row0 + min(qrow, Sb - 1)) * a.ld_dqkv + (size_t)(0 * a.H + head) * HDP;
    store_acc_rows_split<G::DBLK>((u16*)a.dqkv + off, (u16*)a.dqkv_lo + off, dq, a.scale, h, qrow < Sb);
  }
}

template <int HDP> int launch_fwd_x3(const TfAttnArgs* a, hipStream_t st) {
  const size_t lds = 256 * Geo<HDP>::TSTR;
  static const hipError_t once = hipFuncSetAttribute((const void*)attn_fwd_x3_kernel<HDP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)once;
  char nm[56];
  snprintf(nm, sizeof(nm), "attn_fwd_x3_kernel<%d>", HDP);
  const int Sq = a->q != nullptr ? a->Sq : a->S;
  TfTraceScope tr(nm, st, 4.0 * a->B * a->H * (double)Sq * a->S * HDP);
  hipLaunchKernelGGL(attn_fwd_x3_kernel<HDP>, dim3(((Sq + 127) / 128) * a->B * a->H), dim3(256), lds, st, *a);
  return (int)hipGetLastError();
}
template <int HDP> int launch_bwd_x3(const TfAttnArgs* a, hipStream_t st) {
  const size_t lds_q = 256 * Geo<HDP>::TSTR, lds_kv = 128 * Geo<HDP>::TSTR + 256 + 1024;
  static const hipError_t o1 = hipFuncSetAttribute((const void*)attn_bwd_dq_x3_kernel<HDP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
  static const hipError_t o2 = hipFuncSetAttribute((const void*)attn_bwd_dkv_x3_kernel<HDP, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
  static const hipError_t o3 = hipFuncSetAttribute((const void*)attn_bwd_dkv_x3_kernel<HDP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
  (void)o1; (void)o2; (void)o3;
  const int Sq = a->q != nullptr ? a->Sq : a->S;
  const dim3 grid(((a->S + 127) / 128) * a->B * a->H), grid_q(((Sq + 127) / 128) * a->B * a->H);
  const double fl = 4.0 * a->B * a->H * (double)Sq * a->S * HDP;       // credited as in attn_bf16.hip: backward = 2x forward over dq + dkv
  char nm[56];
  const bool ds = a->ds_work != nullptr && a->q == nullptr;            // S and dP once: delta, dV, dK (+ dS planes), dQ from dS
  if (ds) {
    TfTraceScope tr("attn_delta_x3_kernel", st);
    hipLaunchKernelGGL(attn_delta_x3_kernel, dim3(a->B * ((a->S + 3) / 4)), dim3(256), 0, st, *a);
  } else {
    snprintf(nm, sizeof(nm), "attn_bwd_dq_x3_kernel<%d>", HDP);
    TfTraceScope tr(nm, st, fl);
    hipLaunchKernelGGL(attn_bwd_dq_x3_kernel<HDP>, grid_q, dim3(256), lds_q, st, *a);
  }
  const bool pd = ds && a->ds_planes >= 4;                             // ... and Pd for the dV product: the dK launch goes first
  if (!pd) {
    snprintf(nm, sizeof(nm), "attn_bwd_dkv_x3_kernel<%d, dV>", HDP);
    TfTraceScope tr(nm, st, fl / 2);
    hipLaunchKernelGGL((attn_bwd_dkv_x3_kernel<HDP, 0>), grid, dim3(256), lds_kv, st, *a);
  }
  {
    snprintf(nm, sizeof(nm), "attn_bwd_dkv_x3_kernel<%d, dK>", HDP);
    TfTraceScope tr(nm, st, fl / 2);
    hipLaunchKernelGGL((attn_bwd_dkv_x3_kernel<HDP, 1>), grid, dim3(256), lds_kv, st, *a);
  }
  if (pd) {
    const size_t lds_pd = 64 * Geo<HDP>::TSTR;
    static const hipError_t o6 = hipFuncSetAttribute((const void*)attn_bwd_dv_pd_x3_kernel<HDP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pd);
    (void)o6;
    snprintf(nm, sizeof(nm), "attn_bwd_dv_pd_x3_kernel<%d>", HDP);
    TfTraceScope tr(nm, st, fl / 2);
    hipLaunchKernelGGL(attn_bwd_dv_pd_x3_kernel<HDP>, grid, dim3(256), lds_pd, st, *a);
  }
  if (ds) {
    static const int kt = TF_ENV_INT("TF_X3_DQ_KT", 32);              // experiment switch: 64 = one workgroup per CU
    snprintf(nm, sizeof(nm), "attn_bwd_dq_ds_x3_kernel<%d>", HDP);
    TfTraceScope tr(nm, st, fl / 2);
    if (kt == 64 || HDP > 192) {                                    // (head dim 224: 300 registers, one workgroup per CU either way)
      const size_t lds_ds = 128 * Geo<HDP>::TSTR + 128 * Geo<128>::TSTR;
      static const hipError_t o4 = hipFuncSetAttribute((const void*)attn_bwd_dq_ds_x3_kernel<HDP, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ds);
      (void)o4;
      hipLaunchKernelGGL((attn_bwd_dq_ds_x3_kernel<HDP, 64>), grid_q, dim3(256), lds_ds, st, *a);
    } else {
      const size_t lds_ds = 64 * Geo<HDP>::TSTR + 64 * Geo<128>::TSTR;
      static const hipError_t o7 = hipFuncSetAttribute((const void*)attn_bwd_dq_ds_x3_kernel<HDP, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ds);
      (void)o7;
      hipLaunchKernelGGL((attn_bwd_dq_ds_x3_kernel<HDP, 32>), grid_q, dim3(256), lds_ds, st, *a);
    }
  }
  return (int)hipGetLastError();
}

}  // namespace

// head dims up to 224 (d = 896, 4 heads): 256 would need > 160 KiB of LDS for the four 64-row tiles of the dQ kernel
#define TF_ATTN_X3_DISPATCH(FN)                  \
  switch (a->HDP) {                              \
    case 32: return FN<32>(a, st);               \
    case 64: return FN<64>(a, st);               \
    case 96: return FN<96>(a, st);               \
    case 128: return FN<128>(a, st);             \
    case 160: return FN<160>(a, st);             \
    case 192: return FN<192>(a, st);             \
    case 224: return FN<224>(a, st);             \
    default: return -3;                          \
  }

extern "C" int tf_launch_attn_fwd_x3(const TfAttnArgs* a, hipStream_t st) {
  if (a->qkv_lo == nullptr || a->out_lo == nullptr || (a->q != nullptr && a->q_lo == nullptr)) return -6;
  TF_ATTN_X3_DISPATCH(launch_fwd_x3)
}
extern "C" int tf_launch_attn_bwd_x3(const TfAttnArgs* a, hipStream_t st) {
  if (a->qkv_lo == nullptr || a->out_lo == nullptr || a->dout_lo == nullptr || a->dqkv_lo == nullptr) return -6;
  if (a->q != nullptr && (a->q_lo == nullptr || a->dq_lo == nullptr)) return -6;
  TF_ATTN_X3_DISPATCH(launch_bwd_x3)
}
