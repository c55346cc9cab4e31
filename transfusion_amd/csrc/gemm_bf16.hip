// bf16 MFMA GEMMs for the fusion block (gfx950).
//
//   gemm_nt   C[M,N] = A[M,K] . W[N,K]^T  (+ fused epilogue)      -- nn.Linear layout, no transposes
//             used for QKV / out-proj / FFN (K3,K5,K6,K7), the dgrad GEMMs (with W^T shadows) and K1/K9.
//   wgrad_tn  dW[N,K] += dY[M,N]^T . X[M,K]  (fp32 atomics, split over M; bias grad fused)
//
// Structure of gemm_nt: 128x128 output tile, BK = 64, 256 threads = 4 waves (2x2), each wave a 64x64
// sub-tile as 4x4 v_mfma_f32_16x16x32_bf16 accumulators.  Operand tiles are staged global->LDS by
// LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), double-buffered, one barrier per
// K-tile.  The LDS image is lane-linear, so the bank swizzle (16-B chunk ^= (row>>1)&7 on 128-B rows,
// conflict-free for ds_read_b128) is applied to the per-lane SOURCE address and again on the read.
// The product is computed transposed (D'[n][m] = W.X^T) so that each lane owns 4 consecutive output
// columns; the tile is then passed through LDS once so that the epilogue (bias / GELU / dropout /
// residual) works on 16-B row chunks and HBM sees full 256-B row segments.
#include "tf_common.h"
#include "tf_kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;          // 16 KiB per operand tile
constexpr int CT_STRIDE = 272;                    // C-tile row stride in LDS (256 B + 16 B pad)

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds(TF_GLB_PTR(gsrc), TF_LDS_PTR(lds_dst), 16, 0, 0);
}

// bijective XCD-aware remap: blocks b, b+8, b+16.. share an XCD (round-robin dispatch); give each
// XCD a contiguous range of logical tiles so that tiles sharing an A panel hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const TfGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const u16* __restrict__ A = (const u16*)g.A;
  const u16* __restrict__ W = (const u16*)g.W;

  auto stage = [&](int buf, int kt) {
    unsigned char* abase = smem + buf * (2 * TILE_BYTES);
    unsigned char* bbase = abase + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int R0 = (i * 4 + wave) * 8;                   // 8 rows x 128 B per wave-instruction
      const int r = R0 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);            // swizzle on the source chunk
      const int gm = min(m0 + r, g.M - 1), gn = min(n0 + r, g.N - 1);
      glds16(A + (size_t)gm * g.lda + kt * BK + c * 8, abase + R0 * 128);
      glds16(W + (size_t)gn * g.ldw + kt * BK + c * 8, bbase + R0 * 128);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  stage(0, 0);
  __syncthreads();
  const int frow = lane & 15, fch = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const unsigned char* abase = smem + cur * (2 * TILE_BYTES);
    const unsigned char* bbase = abase + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rn = wc * 64 + i * 16 + frow;
        wf[i] = *(const bf16x8*)(bbase + rn * 128 + (((ks * 4 + fch) ^ ((rn >> 1) & 7)) << 4));
        const int rm = wr * 64 + i * 16 + frow;
        xf[i] = *(const bf16x8*)(abase + rm * 128 + (((ks * 4 + fch) ^ ((rm >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue phase 1: (acc + bias) -> bf16 -> LDS C tile [128][CT_STRIDE] ----
  unsigned char* ct = smem;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int nl = wc * 64 + ni * 16 + (lane >> 4) * 4;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (g.bias != nullptr && n0 + nl < g.N) b = *(const f32x4*)(g.bias + n0 + nl);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int ml = wr * 64 + mi * 16 + (lane & 15);
      u32x2 v;
      v[0] = pack2bf(acc[ni][mi][0] + b[0], acc[ni][mi][1] + b[1]);
      v[1] = pack2bf(acc[ni][mi][2] + b[2], acc[ni][mi][3] + b[3]);
      *(u32x2*)(ct + ml * CT_STRIDE + nl * 2) = v;
    }
  }
  __syncthreads();
  // ---- phase 2: row-contiguous 16-B chunks, elementwise epilogue, coalesced stores ----
  u16* __restrict__ C = (u16*)g.C;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int id = i * 256 + tid;
    const int row = id >> 4, c = id & 15;
    const int gm = m0 + row, gn = n0 + c * 8;
    if (gm >= g.M || gn >= g.N) continue;
    u32x4 v = *(const u32x4*)(ct + row * CT_STRIDE + c * 16);
    if constexpr (EPI == TF_EPI_BIAS || EPI == TF_EPI_NONE) {
      *(u32x4*)(C + (size_t)gm * g.ldc + gn) = v;
    } else {
      float f[8];
      unpack8(v, f);
      if constexpr (EPI == TF_EPI_BIAS_GELU_DROP) {
        *(u32x4*)(C + (size_t)gm * g.ldc + gn) = v;       // pre-activation U (saved for backward)
        const unsigned km = g.drop_thr ? tf_keep8((unsigned)gm * (unsigned)g.ldc2 + (unsigned)gn, g.drop_key, g.drop_thr) : 0xffu;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = ((km >> e) & 1u) ? gelu_f(f[e]) * g.drop_scale : 0.f;
        *(u32x4*)((u16*)g.C2 + (size_t)gm * g.ldc2 + gn) = pack8(f);
      } else if constexpr (EPI == TF_EPI_BIAS_DROP_RES) {
        float r[8];
        unpack8(*(const u32x4*)((const u16*)g.R + (size_t)gm * g.ldr + gn), r);
        const unsigned km = g.drop_thr ? tf_keep8((unsigned)gm * (unsigned)g.ldc + (unsigned)gn, g.drop_key, g.drop_thr) : 0xffu;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = r[e] + (((km >> e) & 1u) ? f[e] * g.drop_scale : 0.f);
        *(u32x4*)(C + (size_t)gm * g.ldc + gn) = pack8(f);
      } else if constexpr (EPI == TF_EPI_ADD) {
        float r[8];
        unpack8(*(const u32x4*)((const u16*)g.R + (size_t)gm * g.ldr + gn), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += r[e];
        *(u32x4*)(C + (size_t)gm * g.ldc + gn) = pack8(f);
      } else if constexpr (EPI == TF_EPI_DGELU_DROP) {
        // dU = dH . mask/(1-p) . gelu'(U); R = U, dropout index space = that of H (ldr == ld of H)
        float u[8];
        unpack8(*(const u32x4*)((const u16*)g.R + (size_t)gm * g.ldr + gn), u);
        const unsigned km = g.drop_thr ? tf_keep8((unsigned)gm * (unsigned)g.ldr + (unsigned)gn, g.drop_key, g.drop_thr) : 0xffu;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = ((km >> e) & 1u) ? f[e] * g.drop_scale * gelu_grad_f(u[e]) : 0.f;
        *(u32x4*)(C + (size_t)gm * g.ldc + gn) = pack8(f);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[n][k] += sum_m dY[m][n] X[m][k].  Tile 128(n) x 128(k), reduction step 64 rows of m,
// 4 waves (2x2) of 64x64 as 2x2 v_mfma_f32_32x32x16_bf16.  Both operands have the reduction index
// on the slow axis in memory, so fragments come from ds_read_b64_tr_b16 (hardware transpose) on
// row-major [64][128] LDS tiles; swizzle: 16-B chunk ^= (row&3)<<2 (conflict-free for the tr reads).
// The 32x32 accumulator's register r is 128 contiguous bytes of one dW row per half-wave, which is the
// full-rate shape for global_atomic_add_f32.  Bias grad (column sums of dY) = one extra MFMA against a
// ones operand in the k-tile-0 blocks.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(const TfWgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_k = (g.K + 127) / 128;
  const int n0 = (blockIdx.x / tiles_k) * 128, k0 = (blockIdx.x % tiles_k) * 128;
  const int m_begin = blockIdx.y * g.m_chunk;
  const int m_end = min(g.M, m_begin + g.m_chunk);
  const int nsteps = (m_end - m_begin + 63) / 64;
  const u16* __restrict__ dY = (const u16*)g.dY;
  const u16* __restrict__ X = (const u16*)g.X;
  const u16* __restrict__ Z = (const u16*)g.zeros;

  auto stage = [&](int buf, int st) {
    unsigned char* ybase = smem + buf * (2 * TILE_BYTES);
    unsigned char* xbase = ybase + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int R0 = (i * 4 + wave) * 4;                   // 4 rows x 256 B per wave-instruction
      const int r = R0 + (lane >> 4);
      const int c = (lane & 15) ^ ((r & 3) << 2);
      const int gm = m_begin + st * 64 + r;
      const bool ok = gm < m_end;
      const int cn = min(n0 + c * 8, g.N - 8), ck = min(k0 + c * 8, g.K - 8);   // clamp: never stored
      const u16* sy = ok ? dY + (size_t)gm * g.ldy + cn : Z + c * 8;
      const u16* sx = ok ? X + (size_t)gm * g.ldx + ck : Z + c * 8;
      glds16(sy, ybase + R0 * 256);
      glds16(sx, xbase + R0 * 256);
    }
  };

  f32x16 acc[2][2], accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; accb[i][r] = 0.f; }
  }
  // Bias grad = column sums of dY = one extra MFMA of the dY fragment against a ones operand.  The duty is spread
  // evenly: the 2*tiles_k waves that share this n-range (tiles_k k-tile blocks x 2 wave columns) each take every
  // (2*tiles_k)-th 16-row step, so no block carries more than ~1/(2*tiles_k) extra MFMA work (a k0==0-only scheme
  // made those blocks 1.5x longer and they set the kernel time).
  const bool has_bias = g.db != nullptr;
  const int bias_mod = 2 * tiles_k;
  int bias_cnt = 2 * (blockIdx.x % tiles_k) + wc;              // counts down to this wave's next duty step
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  // tr-read lane geometry
  const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int h = grp >> 1, cb = grp & 1;

  if (nsteps > 0) {
    stage(0, 0);
    __syncthreads();
  }
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    if (st + 1 < nsteps) stage(cur ^ 1, st + 1);
    const unsigned char* ybase = smem + cur * (2 * TILE_BYTES);
    const unsigned char* xbase = ybase + TILE_BYTES;
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int row0 = ms * 16 + 8 * h + q, row1 = row0 + 4;   // row&3 == q for both
        const int chy = wr * 8 + b * 4 + cb * 2 + (p >> 1);
        const int chx = wc * 8 + b * 4 + cb * 2 + (p >> 1);
        const int sw = q << 2, o8 = (p & 1) * 8;
        af[b] = join_tr(lds_read_tr16(ybase + row0 * 256 + ((chy ^ sw) << 4) + o8),
                        lds_read_tr16(ybase + row1 * 256 + ((chy ^ sw) << 4) + o8));
        bfr[b] = join_tr(lds_read_tr16(xbase + row0 * 256 + ((chx ^ sw) << 4) + o8),
                         lds_read_tr16(xbase + row1 * 256 + ((chx ^ sw) << 4) + o8));
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[nb], bfr[kb], acc[nb][kb], 0, 0, 0);
      }
      const bool my_turn = has_bias && bias_cnt == 0;             // wave-uniform
      bias_cnt = bias_cnt == 0 ? bias_mod - 1 : bias_cnt - 1;
      if (my_turn) {
        accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], ones, accb[0], 0, 0, 0);
        accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], ones, accb[1], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: fp32 atomics into the (unpadded) parameter-layout gradient ----
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int np = n0 + wr * 64 + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int ng = np / g.rgp, ne = np - ng * g.rgp;
      const int ns = ng * g.rg + ne;
      const bool nok = (np < g.N) && (ne < g.rg) && (ns < g.n_src);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int kp = k0 + wc * 64 + kb * 32 + (lane & 31);
        const int kg = kp / g.cgp, ke = kp - kg * g.cgp;
        const int ks = kg * g.cg + ke;
        if (nok && kp < g.K && ke < g.cg && ks < g.k_src)
          atomicAdd(g.dW + (size_t)ns * g.lddw + ks, acc[nb][kb][r]);
      }
      if (has_bias && (lane & 31) == 0 && nok) atomicAdd(g.db + ns, accb[nb][r]);
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers (host)
// ------------------------------------------------------------------------------------------------
extern "C" int tf_launch_gemm_nt(const TfGemmArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->N <= 0) return 0;
  if (a->K <= 0 || a->K % BK != 0 || a->N % 8 != 0) return -2;
  if ((a->lda % 8) || (a->ldw % 8) || (a->ldc % 8)) return -3;
  const int tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
  const size_t lds = 4 * TILE_BYTES;   // 64 KiB (>= the 34 KiB C tile)
  dim3 grid(tiles), block(256);
#define TF_GEMM_CASE(E) case E: hipLaunchKernelGGL(gemm_nt_kernel<E>, grid, block, lds, stream, *a); break;
  switch (a->epilogue) {
    TF_GEMM_CASE(TF_EPI_NONE)
    TF_GEMM_CASE(TF_EPI_BIAS)
    TF_GEMM_CASE(TF_EPI_BIAS_GELU_DROP)
    TF_GEMM_CASE(TF_EPI_BIAS_DROP_RES)
    TF_GEMM_CASE(TF_EPI_ADD)
    TF_GEMM_CASE(TF_EPI_DGELU_DROP)
    default: return -4;
  }
#undef TF_GEMM_CASE
  return (int)hipGetLastError();
}

extern "C" int tf_launch_wgrad_tn(const TfWgradArgs* a_in, hipStream_t stream) {
  TfWgradArgs a = *a_in;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return 0;
  if ((a.N % 8) || (a.K % 8) || (a.ldy % 8) || (a.ldx % 8) || a.zeros == nullptr) return -2;
  if (a.rgp < a.rg || a.cgp < a.cg || a.rg <= 0 || a.cg <= 0) return -3;
  const int tiles = ((a.N + 127) / 128) * ((a.K + 127) / 128);
  const int steps = (a.M + 63) / 64;
  // Every split adds one full fp32 |dW| of atomic traffic (chip-wide ~1.3 TB/s), so use the FEWEST splits that
  // still give one resident wave of blocks: 256 CUs x 2 blocks (64 KiB LDS each).
  int splits = a.m_chunk > 0 ? (a.M + a.m_chunk - 1) / a.m_chunk : (512 / tiles);
  if (splits > steps) splits = steps;
  if (splits < 1) splits = 1;
  if (a.m_chunk <= 0) a.m_chunk = ((steps + splits - 1) / splits) * 64;
  splits = (a.M + a.m_chunk - 1) / a.m_chunk;
  dim3 grid(tiles, splits), block(256);
  hipLaunchKernelGGL(wgrad_tn_kernel, grid, block, 4 * TILE_BYTES, stream, a);
  return (int)hipGetLastError();
}
