// Weight gradients of SEVERAL linears in ONE launch (gfx950):  dW_p[N_p, K_p] += dY_p[M_p, N_p]^T . X_p[M_p, K_p],  p = 0 .. count-1.
//
// Why one launch: a layer's four weight gradients (in-proj, out-proj, FFN-up, FFN-down: 54 + 18 + 36 + 36 output tiles of 256 x 128 at
// d = 768) each had to fill the chip BY ITSELF, which meant splitting the reduction (the token rows M) 5 - 14 ways and merging the
// partial tiles with fp32 atomics: 34 MB of atomic traffic per launch against 2.4 - 7.1 MB of gradient, at the chip's ~1.3 TB/s atomic
// rate.  Together the 144 tiles fill 256 CUs at THREE row chunks each (432 workgroups, ~1.7 per CU: 172 us for a d = 768 layer against
// 210 at two chunks and 207 at four): well under half the atomic bytes, a quarter of the launches, one prologue and one flush per ~170
// reduction steps instead of per 40 - 100.
//
// Work item = (row chunk c, output tile t of problem p), chunk-major: the workgroups that run together walk the SAME rows of dY / X
// (an XCD owns a contiguous range of items, k-tile fastest, so the tiles that share a dY column panel share an L2).
// Tile 256 (n) x 128 (k), 4 waves (2 x 2) of 128 x 64 = 4 x 2 v_mfma_f32_32x32x16_bf16 per 16-row substep.  Both operands have the
// reduction index on the slow axis in memory, so fragments come from ds_read_b64_tr_b16 (hardware transpose) on row-major LDS tiles
// (dY [32][256] | X [32][128] per 32-row step, 16-B chunk ^= (row & 3) << 2: conflict-free for the transposed reads).
// Staging: LDS-DMA through BUFFER loads (buffer_load_dwordx4 ... lds): per lane six loop-invariant 32-bit offsets, the step advances
// one scalar offset per operand, and rows past the chunk's end arrive as zeros (no ragged path, no clamps, no 64-bit address
// arithmetic in the loop).  The six transfers of a step are issued BETWEEN the step's MFMAs (an LDS-DMA costs its wave 60 - 180 issue
// cycles: in front of the MFMAs, at one wave per SIMD, the matrix pipe idles for all of them).  3-slot ring, counted vmcnt, raw
// s_barrier; transposed reads from inline asm (hipcc otherwise orders every LDS-DMA before any later ds_read with vmcnt(0)).
// Bias gradients (column sums of dY): one extra accumulator per wave, D += SEL_nb . dYfrag_nb with SEL_nb[i][m] = (i >> 3 == nb); the
// duty rotates over the waves that share an n-range (k-tile blocks x wave columns), every (2 tiles_k)-th 16-row substep each.
// SPLIT (fp32-accuracy mode): three passes over the chunk's rows -- (dY_hi, X_hi), (dY_lo, X_hi), (dY_hi, X_lo) -- into the same
// accumulators; the bias gradient takes passes 0 and 1.
#include <cstdio>
#include <cstdlib>
#include <utility>
#include "tf_common.h"
#include "tf_kernels.h"

// timing ablations of the K loop (experiments builds only: wrong results): 2 one MFMA in eight, 4 fragment reads in the first step
// only, 16 no flush.  Measured with them (tools/experiments/wgrad_multi_kloop_ablations.txt): see DESIGN.md, "What bounds the step now".
#ifndef TF_EXPERIMENTS
#undef TF_ABL_WGM
#endif
#ifndef TF_ABL_WGM
#define TF_ABL_WGM 0
#endif

namespace {

constexpr int WG_MAX = TF_WGRAD_MULTI_MAX;

struct WgProb {                                   // one problem as the kernel sees it (112 B)
  const unsigned char* dY; const unsigned char* X; float* dW; float* db;
  const unsigned char* dY_lo; const unsigned char* X_lo;
  int ldy, ldx, lddw, M, N, K;
  int rg, rgp, n_src, cg, cgp, k_src;
  int tiles_k, tile0, m_chunk, pad;               // k-tiles per n-tile row; first global tile index; rows per chunk (multiple of 32)
};
struct WgMulti { int count, chunks, tiles_total, pad; WgProb p[WG_MAX]; };

template <int N, class F, int... I> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// (a __device__ function: the LDS-DMA builtin inside a lambda of a kernel TEMPLATE can make hipcc's host pass drop the instantiation
// silently -- the library then fails to load with an undefined __device_stub__; it did when a fourth template parameter was tried)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_dst, int voff, int soff) {
#if defined(TF_EXPERIMENTS) && defined(TF_WGM_AUX)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, TF_LDS_PTR(lds_dst), 16, voff, soff, 0, TF_WGM_AUX);      // cache-policy experiment (bit 1: nt)
#else
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, TF_LDS_PTR(lds_dst), 16, voff, soff, 0, 0);
#endif
}
// INTL: 0 = the step's transfers in front of its MFMAs (the form of wgrad_tn2_kernel), 1 = one transfer after every second MFMA
// MF16: the products as v_mfma_f32_16x16x32_bf16 (8 x 4 blocks of 16 x 16 per wave, one instruction spans the step's 32 rows) instead of
// v_mfma_f32_32x32x16_bf16 (4 x 2 blocks of 32 x 32, two 16-row substeps): the same matrix cycles, fragment bytes and accumulator
// registers -- the 16 x 16 shape holds a higher clock under load (MI355X_MICROARCH.md, DVFS give-back (7); timing ablation with the
// reads unchanged: 170 -> 164 us alone, step -0.9 %).  Its A / B fragments put the step's four row OCTETS on the four 16-lane groups
// (the 32 x 32 shape: two octets x two column halves), so the tile swizzle also folds row bit 3 into the chunk index (conflict-free
// for both read patterns).
template <bool SPLIT, int NS, int INTL, bool MF16>
__global__ __launch_bounds__(256, 2) void wgrad_multi_kernel(const WgMulti a) {
  constexpr int WN = 2, WK = 2, NB = 4, KB = 2;
  constexpr int NWV = WN * WK, TN = WN * NB * 32, TK = WK * KB * 32;
  constexpr int STEP = 32, YROW = TN * 2, XROW = TK * 2, YB = STEP * YROW, XB = STEP * XROW, SLOT = YB + XB;
  constexpr int NPY = YB / 1024 / NWV, NPX = XB / 1024 / NWV, NPW = NPY + NPX;     // 1-KiB transfers per wave and step: 4 + 2
  constexpr int YLPR = YROW / 16, XLPR = XROW / 16;                                 // lanes per tile row of a transfer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WK, wc = wave % WK;

  // ---- work item -> (chunk, problem, tile) ----
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int chunk = logical / a.tiles_total, gt = logical - chunk * a.tiles_total;
  int pi = 0;
  for (int i = 1; i < a.count; ++i) pi = gt >= a.p[i].tile0 ? i : pi;
  const WgProb& P = a.p[pi];
  const int tiles_k = P.tiles_k, tile = gt - P.tile0;
  const int n0 = (tile / tiles_k) * TN, k0 = (tile % tiles_k) * TK;
  const int m_begin = chunk * P.m_chunk;
  const int m_end = min(P.M, m_begin + P.m_chunk);
  if (m_begin >= m_end) return;                                   // a problem with fewer rows than the launch's chunk count covers
  const int rows = m_end - m_begin;
  const int nsteps0 = (rows + STEP - 1) / STEP;
  const int nsteps = SPLIT ? 3 * nsteps0 : nsteps0;
  const int ldy = P.ldy, ldx = P.ldx, N = P.N, K = P.K;

  // ---- buffer resources over the chunk's rows: offsets past the last row read zeros ----
  const __amdgpu_buffer_rsrc_t rYh = __builtin_amdgcn_make_buffer_rsrc((void*)(P.dY + (size_t)m_begin * ldy * 2), 0, rows * ldy * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rXh = __builtin_amdgcn_make_buffer_rsrc((void*)(P.X + (size_t)m_begin * ldx * 2), 0, rows * ldx * 2, 0x00020000);
  __amdgpu_buffer_rsrc_t rYl = rYh, rXl = rXh;
  if constexpr (SPLIT) {
    rYl = __builtin_amdgcn_make_buffer_rsrc((void*)(P.dY_lo + (size_t)m_begin * ldy * 2), 0, rows * ldy * 2, 0x00020000);
    rXl = __builtin_amdgcn_make_buffer_rsrc((void*)(P.X_lo + (size_t)m_begin * ldx * 2), 0, rows * ldx * 2, 0x00020000);
  }
  // per-lane source offsets of this wave's transfers inside one 32-row step (tile swizzle folded in; columns clamped: never stored)
  int vy[NPY], vx[NPX];
#pragma unroll
  for (int i = 0; i < NPY; ++i) {
    const int j = i * NWV + wave;
    const int r = j * (1024 / YROW) + lane / YLPR;
    const int c = (lane % YLPR) ^ ((r & 3) << 2) ^ (MF16 ? ((r >> 3) & 1) << 1 : 0);
    vy[i] = r * ldy * 2 + min(n0 + c * 8, N - 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int j = i * NWV + wave;
    const int r = j * (1024 / XROW) + lane / XLPR;
    const int c = (lane % XLPR) ^ ((r & 3) << 2) ^ (MF16 ? ((r >> 3) & 1) << 1 : 0);
    vx[i] = r * ldx * 2 + min(k0 + c * 8, K - 8) * 2;
  }
  const int step_y = STEP * ldy * 2, step_x = STEP * ldx * 2;
  // transfer q (0 .. NPW-1: first the dY pieces, then the X pieces) of reduction step `step` into ring slot `slot`
  auto piece = [&](auto Q, int slot, int step) {
    constexpr int q = decltype(Q)::value;
    int st = step;
    bool ylo = false, xlo = false;
    if constexpr (SPLIT) {
      const int seg = step >= 2 * nsteps0 ? 2 : (step >= nsteps0 ? 1 : 0);
      st = step - seg * nsteps0;
      ylo = seg == 1; xlo = seg == 2;
    }
    unsigned char* base = smem + slot * SLOT;
    if constexpr (q < NPY) {
      dma16(SPLIT && ylo ? rYl : rYh, base + (q * NWV + wave) * 1024, vy[q], st * step_y);
    } else {
      constexpr int i = q - NPY;
      dma16(SPLIT && xlo ? rXl : rXh, base + YB + (i * NWV + wave) * 1024, vx[i], st * step_x);
    }
  };
  auto stage = [&](int slot, int step) { sfor<NPW>([&](auto Q) { piece(Q, slot, step); }); };

  const bool has_bias = P.db != nullptr;
  const int bias_mod = WK * tiles_k;                           // k-tile blocks x wave columns share one n-range: take turns
  int bias_cnt = WK * (tile % tiles_k) + wc;
  const int grp = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
  const unsigned lds_base = lds_addr_of(smem);
  float* __restrict__ dW = P.dW;
  const int lddw = P.lddw, rg = P.rg, rgp = P.rgp, n_src = P.n_src, cg = P.cg, cgp = P.cgp, k_src = P.k_src;
  stage(0, 0);
  if constexpr (NS >= 3) stage(1, 1);
  if constexpr (NS >= 4) stage(2, 2);
  int slot = 0;
  if constexpr (MF16) {
    f32x4 acc4[32];                                            // block (nb, kb), nb = 0 .. 7 (16 n each), kb = 0 .. 3: acc4[nb * 4 + kb]
    f32x4 accb4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane = 16 g + 4 q4 + p4 supplies row 8 g + q4 (+ 4: second read), columns 4 p4 .. 4 p4 + 3 of a block's 16; it receives column
    // (lane & 15), rows 8 g .. 8 g + 7 = the A (or B) fragment of v_mfma_f32_16x16x32_bf16
    const int sw = (q4 << 2) | ((grp & 1) << 1), o8 = (p4 & 1) * 8;
    const int yoff = (8 * grp + q4) * YROW + ((((wr * 16) | (p4 >> 1)) ^ sw) << 4) + o8;              // n-block nb (16 columns): ^ (nb * 32)
    const int xoff = YB + (8 * grp + q4) * XROW + ((((wc * 8) | (p4 >> 1)) ^ sw) << 4) + o8;          // k-block kb: ^ (kb * 32)
    for (int st = 0; st < nsteps; ++st) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * NPW) : "memory");
      const unsigned sl = lds_base + slot * SLOT;
      u64 xb[4][2], ya[8][2];
#define TF_RDX(kb) xb[kb][0] = tr_read_asm<0>(sl + (xoff ^ ((kb) * 32))); xb[kb][1] = tr_read_asm<4 * XROW>(sl + (xoff ^ ((kb) * 32)));
#define TF_RDY(nb) ya[nb][0] = tr_read_asm<0>(sl + (yoff ^ ((nb) * 32))); ya[nb][1] = tr_read_asm<4 * YROW>(sl + (yoff ^ ((nb) * 32)));
      TF_RDX(0) TF_RDX(1) TF_RDX(2) TF_RDX(3)
      TF_RDY(0) TF_RDY(1) TF_RDY(2) TF_RDY(3)
      TF_RDY(4) TF_RDY(5) TF_RDY(6) TF_RDY(7)
#undef TF_RDX
#undef TF_RDY
      const int nslot = slot == 0 ? NS - 1 : slot - 1;
      const int nstep = st + NS - 1;
      slot = slot == NS - 1 ? 0 : slot + 1;
      if constexpr (INTL == 0) stage(nslot, nstep);
      bf16x8 bfr[4], af[8];
      auto half = [&](auto H) {
        constexpr int h = decltype(H)::value;
        sfor<16>([&](auto I) {
          constexpr int i = decltype(I)::value, nb = 4 * h + i / 4, kb = i % 4;
          acc4[nb * 4 + kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nb], bfr[kb], acc4[nb * 4 + kb], 0, 0, 0);
          if constexpr (INTL == 1) {
            constexpr int done = h * 16 + i + 1;                 // one transfer behind every fourth MFMA of the step (64 matrix cycles apart)
            if constexpr (done % 4 == 0 && done / 4 - 1 < NPW) {
              __builtin_amdgcn_sched_barrier(0);
              piece(std::integral_constant<int, done / 4 - 1>{}, nslot, nstep);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        });
      };
      // the first 16 products need the four X fragments and dY blocks 0 - 3 (16 reads; the 8 of blocks 4 - 7 still in flight)
      asm volatile("s_waitcnt lgkmcnt(8)"
                   : "+v"(xb[0][0]), "+v"(xb[0][1]), "+v"(xb[1][0]), "+v"(xb[1][1]), "+v"(xb[2][0]), "+v"(xb[2][1]), "+v"(xb[3][0]), "+v"(xb[3][1]),
                     "+v"(ya[0][0]), "+v"(ya[0][1]), "+v"(ya[1][0]), "+v"(ya[1][1]), "+v"(ya[2][0]), "+v"(ya[2][1]), "+v"(ya[3][0]), "+v"(ya[3][1]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) bfr[kb] = join_tr64(xb[kb][0], xb[kb][1]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) af[nb] = join_tr64(ya[nb][0], ya[nb][1]);
      half(std::integral_constant<int, 0>{});
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(ya[4][0]), "+v"(ya[4][1]), "+v"(ya[5][0]), "+v"(ya[5][1]), "+v"(ya[6][0]), "+v"(ya[6][1]), "+v"(ya[7][0]), "+v"(ya[7][1]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 4; nb < 8; ++nb) af[nb] = join_tr64(ya[nb][0], ya[nb][1]);
      half(std::integral_constant<int, 1>{});
      // bias gradient: D[i][j] += sum_m SEL[i][m] dY[m][16 nb + j], SEL row i = ones iff i == nb: row nb of the one extra accumulator
      // collects block nb's column sums; a wave takes every (2 tiles_k)-th STEP of the chunk (the waves that share its n-range the others)
      const bool my_turn = has_bias && bias_cnt == 0 && (!SPLIT || st < 2 * nsteps0);          // wave-uniform
      bias_cnt = bias_cnt == 0 ? bias_mod - 1 : bias_cnt - 1;
      if (my_turn) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          const unsigned w = li == nb ? 0x3F803F80u : 0u;          // two bf16 ones
          const u32x4 sv = {w, w, w, w};
          accb4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(sv), af[nb], accb4, 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // surplus transfers land before the workgroup retires
    // ---- flush: register r of block (nb, kb) holds rows 4 (lane >> 4) + r, column lane & 15: four 64-B row segments per instruction ----
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int np = n0 + wr * 128 + nb * 16 + 4 * grp + r;
        const int ng = np / rgp, ne = np - ng * rgp;
        const int ns = ng * rg + ne;
        const bool nok = (np < N) && (ne < rg) && (ns < n_src);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const int kp = k0 + wc * 64 + kb * 16 + li;
          const int kg = kp / cgp, ke = kp - kg * cgp;
          const int ks = kg * cg + ke;
          if (nok && kp < K && ke < cg && ks < k_src) atomicAdd(dW + (size_t)ns * lddw + ks, acc4[nb * 4 + kb][r]);
        }
      }
      if (has_bias && grp == (nb >> 2)) {                        // row nb of the bias accumulator: lanes 16 (nb >> 2) + j, register nb & 3
        const int np = n0 + wr * 128 + nb * 16 + li;
        const int ng = np / rgp, ne = np - ng * rgp;
        const int ns = ng * rg + ne;
        if (np < N && ne < rg && ns < n_src) atomicAdd(P.db + ns, accb4[nb & 3]);
      }
    }
  } else {
  f32x16 acc[NB][KB], accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    accb[r] = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int k = 0; k < KB; ++k) acc[i][k][r] = 0.f;
  }
  const int sel_nb = (lane & 31) / (32 / NB);
  const int h = grp >> 1, cb = grp & 1;
  const int sw = q4 << 2, o8 = (p4 & 1) * 8;
  const int yoff = (8 * h + q4) * YROW + (((wr * NB * 4 + cb * 2 + (p4 >> 1)) ^ sw) << 4) + o8;         // n-block nb: ^ (nb * 64)
  const int xoff = YB + (8 * h + q4) * XROW + (((wc * KB * 4 + cb * 2 + (p4 >> 1)) ^ sw) << 4) + o8;    // k-block kb: ^ (kb * 64)
  for (int st = 0; st < nsteps; ++st) {
    // retire step st's transfers (issued NS - 1 steps ago), make them visible, recycle slot (st + NS - 1) % NS
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * NPW) : "memory");
    const unsigned sl = lds_base + slot * SLOT;
    u64 y0[NB][2], x0[KB][2], y1[NB][2], x1[KB][2];             // [block][row half] of substep 0 / 1
#define TF_RD(Y, Xv, MS)                                                                                          \
    Y[0][0] = tr_read_asm<(MS) * 16 * YROW>(sl + yoff);           Y[0][1] = tr_read_asm<(MS) * 16 * YROW + 4 * YROW>(sl + yoff);           \
    Xv[0][0] = tr_read_asm<(MS) * 16 * XROW>(sl + xoff);          Xv[0][1] = tr_read_asm<(MS) * 16 * XROW + 4 * XROW>(sl + xoff);          \
    Y[1][0] = tr_read_asm<(MS) * 16 * YROW>(sl + (yoff ^ 64));    Y[1][1] = tr_read_asm<(MS) * 16 * YROW + 4 * YROW>(sl + (yoff ^ 64));    \
    Xv[1][0] = tr_read_asm<(MS) * 16 * XROW>(sl + (xoff ^ 64));   Xv[1][1] = tr_read_asm<(MS) * 16 * XROW + 4 * XROW>(sl + (xoff ^ 64));   \
    Y[2][0] = tr_read_asm<(MS) * 16 * YROW>(sl + (yoff ^ 128));   Y[2][1] = tr_read_asm<(MS) * 16 * YROW + 4 * YROW>(sl + (yoff ^ 128));   \
    Y[3][0] = tr_read_asm<(MS) * 16 * YROW>(sl + (yoff ^ 192));   Y[3][1] = tr_read_asm<(MS) * 16 * YROW + 4 * YROW>(sl + (yoff ^ 192));
    if (!(TF_ABL_WGM & 4) || st == 0) {
    TF_RD(y0, x0, 0)
    TF_RD(y1, x1, 1)
    }
#undef TF_RD
    const int nslot = slot == 0 ? NS - 1 : slot - 1;           // (st + NS - 1) % NS given slot = st % NS
    const int nstep = st + NS - 1;                             // steps past the end read zeros (never consumed)
    slot = slot == NS - 1 ? 0 : slot + 1;
    if constexpr (INTL == 0) stage(nslot, nstep);
    int pq = 0;                                                // MFMAs issued so far in this step (compile-time after unrolling)
    auto substep = [&](u64 (&Y)[NB][2], u64 (&Xv)[KB][2], auto MS) {
      constexpr int ms = decltype(MS)::value;
      bf16x8 af[NB], bfr[KB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) af[nb] = join_tr64(Y[nb][0], Y[nb][1]);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) bfr[kb] = join_tr64(Xv[kb][0], Xv[kb][1]);
      sfor<NB * KB>([&](auto I) {
        constexpr int i = decltype(I)::value, nb = i / KB, kb = i % KB;
        if constexpr ((TF_ABL_WGM & 32) != 0) {       // timing only: the same matrix cycles as two 16x16x32 instructions (wrong values)
          f32x4 t0 = {acc[nb][kb][4 * ms], acc[nb][kb][4 * ms + 1], acc[nb][kb][4 * ms + 2], acc[nb][kb][4 * ms + 3]};
          f32x4 t1 = {acc[nb][kb][8 + 4 * ms], acc[nb][kb][9 + 4 * ms], acc[nb][kb][10 + 4 * ms], acc[nb][kb][11 + 4 * ms]};
          t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nb], bfr[kb], t0, 0, 0, 0);
          t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nb], bfr[kb], t1, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) { acc[nb][kb][4 * ms + q] = t0[q]; acc[nb][kb][8 + 4 * ms + q] = t1[q]; }
        } else
        if constexpr (!(TF_ABL_WGM & 2) || i == 0) acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[nb], bfr[kb], acc[nb][kb], 0, 0, 0);
        else asm volatile("" :: "v"(af[nb]), "v"(bfr[kb]));
        if constexpr (INTL == 1) {
          constexpr int done = ms * NB * KB + i + 1;             // one transfer behind every second MFMA of the step
          if constexpr (done % 2 == 0 && done / 2 - 1 < NPW) {
            __builtin_amdgcn_sched_barrier(0);
            piece(std::integral_constant<int, done / 2 - 1>{}, nslot, nstep);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      });
      const bool my_turn = has_bias && bias_cnt == 0 && (!SPLIT || st < 2 * nsteps0);          // wave-uniform
      bias_cnt = bias_cnt == 0 ? bias_mod - 1 : bias_cnt - 1;
      if (my_turn) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const unsigned w = sel_nb == nb ? 0x3F803F80u : 0u;          // two bf16 ones
          const u32x4 sv = {w, w, w, w};
          accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(sv), af[nb], accb, 0, 0, 0);
        }
      }
    };
    (void)pq;
    // substep 0 may start when its 12 reads are back (the 12 of substep 1 still in flight)
    asm volatile("s_waitcnt lgkmcnt(12)"
                 : "+v"(y0[0][0]), "+v"(y0[0][1]), "+v"(y0[1][0]), "+v"(y0[1][1]), "+v"(y0[2][0]), "+v"(y0[2][1]), "+v"(y0[3][0]),
                   "+v"(y0[3][1]), "+v"(x0[0][0]), "+v"(x0[0][1]), "+v"(x0[1][0]), "+v"(x0[1][1]));
    __builtin_amdgcn_sched_barrier(0);
    substep(y0, x0, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(y1[0][0]), "+v"(y1[0][1]), "+v"(y1[1][0]), "+v"(y1[1][1]), "+v"(y1[2][0]), "+v"(y1[2][1]), "+v"(y1[3][0]),
                   "+v"(y1[3][1]), "+v"(x1[0][0]), "+v"(x1[0][1]), "+v"(x1[1][0]), "+v"(x1[1][1]));
    __builtin_amdgcn_sched_barrier(0);
    substep(y1, x1, std::integral_constant<int, 1>{});
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // surplus transfers land before the workgroup retires
  // ---- flush: fp32 atomics into the (unpadded) parameter-layout gradient; a register of a 32x32 accumulator is two 128-B row segments ----
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int np = n0 + wr * (NB * 32) + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int ng = np / rgp, ne = np - ng * rgp;
      const int ns = ng * rg + ne;
      const bool nok = (np < N) && (ne < rg) && (ns < n_src);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int kp = k0 + wc * (KB * 32) + kb * 32 + (lane & 31);
        const int kg = kp / cgp, ke = kp - kg * cgp;
        const int ks = kg * cg + ke;
        if ((TF_ABL_WGM & 16) && (r != 0 || (lane & 31) != 0)) { asm volatile("" :: "v"(acc[nb][kb][r])); continue; }
        if (nok && kp < K && ke < cg && ks < k_src) atomicAdd(dW + (size_t)ns * lddw + ks, acc[nb][kb][r]);
      }
    }
    if (has_bias && lane < 32) {                                 // column sums of n-block nb: accumulator rows (32 / NB) nb .. = register 16 nb / NB
      const int np = n0 + wr * (NB * 32) + nb * 32 + lane;
      const int ng = np / rgp, ne = np - ng * rgp;
      const int ns = ng * rg + ne;
      if (np < N && ne < rg && ns < n_src) atomicAdd(P.db + ns, accb[(16 / NB) * nb]);
    }
  }  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Form 2: 192 x 192 output tiles, EIGHT waves = two quads.  A d = 768 layer's four products are 48 + 16 + 32 + 32 = 128 such tiles: at two
// row chunks 256 workgroups, one per CU, every CU the same work (the 256 x 128 form: 144 tiles x 3 chunks = 432 workgroups, 176 CUs with
// two and 80 with one).  The two quads of a workgroup take the two HALVES of its row chunk with a 3-slot ring each, and their partial
// tiles meet in LDS before ONE atomic pass: 256 x 147 KB = 37.7 MB of fp32 atomics per layer launch instead of 432 x 128 KB = 56.6 MB.
// Quad = 2 x 2 waves of 96 x 96 = 6 x 6 blocks of v_mfma_f32_16x16x32_bf16 (144 accumulator registers); a step stages 2 x [32][192]
// = 24 KB per quad for 2.36 MFLOP (96 FLOP per staged byte; the 256 x 128 form: 85), and a wave reads 24 transposed fragments halves
// for 36 instructions (the other form: 24 for 32).
// LDS rows are 384 B (24 chunks of 16 B): row r holds source chunk c at position c ^ (s(r) << 1), s(r) = ((r >> 1) & 1) | ((r >> 3) & 1) << 1,
// i.e. 16-column block b of row r sits at block b ^ s(r) of its aligned group of four -- the eight rows a transposed read touches per
// 32-lane half (r = 8 g + q, g = 0 / 1, q = 0 .. 3) then fall on eight different 32-byte bank groups (rows of odd r start 32 banks on).
// A wave's six blocks per operand are blocks 2 w, 2 w + 1 (w = its row / column in the quad) of each of the three aligned groups, so a
// lane needs two base addresses per operand and the group is an instruction offset.
template <bool SPLIT, int NS>
__global__ __launch_bounds__(512, 2) void wgrad_multi192_kernel(const WgMulti a) {
  constexpr int T = 192, STEP = 32, ROWB = T * 2, OPB = STEP * ROWB, SLOT = 2 * OPB;        // 384 B rows, 12 KB per operand, 24 KB per slot
  constexpr int NPO = OPB / 1024 / 4, NPW = 2 * NPO;                                         // 1-KiB transfers per wave, operand and step: 3 (+ 3)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quad = wave8 >> 2, wave = wave8 & 3;
  const int wr = wave >> 1, wc = wave & 1;

  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int chunk = logical / a.tiles_total, gt = logical - chunk * a.tiles_total;
  int pi = 0;
  for (int i = 1; i < a.count; ++i) pi = gt >= a.p[i].tile0 ? i : pi;
  const WgProb& P = a.p[pi];
  const int tiles_k = P.tiles_k, tile = gt - P.tile0;
  const int n0 = (tile / tiles_k) * T, k0 = (tile % tiles_k) * T;
  const int m_begin = chunk * P.m_chunk;                           // m_chunk: a multiple of 64 rows
  if (m_begin >= P.M) return;
  const int half = P.m_chunk >> 1;
  const int rows_wg = min(P.M, m_begin + P.m_chunk) - m_begin;
  const int nsteps0 = (min(rows_wg, half) + STEP - 1) / STEP;       // quad 0 has the longer half: both quads walk its step count
  const int nsteps = SPLIT ? 3 * nsteps0 : nsteps0;
  const int qbeg = m_begin + quad * half;
  const int qrows = max(0, min(P.M, qbeg + half) - qbeg);           // 0: every transfer of this quad reads zeros
  const int ldy = P.ldy, ldx = P.ldx, N = P.N, K = P.K;
  const size_t qoff = (size_t)min(qbeg, P.M - 1);
  const __amdgpu_buffer_rsrc_t rYh = __builtin_amdgcn_make_buffer_rsrc((void*)(P.dY + qoff * ldy * 2), 0, qrows * ldy * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rXh = __builtin_amdgcn_make_buffer_rsrc((void*)(P.X + qoff * ldx * 2), 0, qrows * ldx * 2, 0x00020000);
  __amdgpu_buffer_rsrc_t rYl = rYh, rXl = rXh;
  if constexpr (SPLIT) {
    rYl = __builtin_amdgcn_make_buffer_rsrc((void*)(P.dY_lo + qoff * ldy * 2), 0, qrows * ldy * 2, 0x00020000);
    rXl = __builtin_amdgcn_make_buffer_rsrc((void*)(P.X_lo + qoff * ldx * 2), 0, qrows * ldx * 2, 0x00020000);
  }
  // transfer j (0 .. 11) of an operand fills LDS bytes [1024 j, 1024 j + 1024) of its [32][384 B] image; wave w issues j = w, w + 4, w + 8
  int vy[NPO], vx[NPO];
#pragma unroll
  for (int i = 0; i < NPO; ++i) {
    const int off = (i * 4 + wave) * 1024 + lane * 16;
    const int r = off / ROWB, cp = (off - r * ROWB) >> 4;
    const int c = cp ^ ((((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1);
    vy[i] = r * ldy * 2 + min(n0 + c * 8, N - 8) * 2;
    vx[i] = r * ldx * 2 + min(k0 + c * 8, K - 8) * 2;
  }
  const int step_y = STEP * ldy * 2, step_x = STEP * ldx * 2;
  unsigned char* ring = smem + quad * NS * SLOT;
  auto piece = [&](auto Q, int slot, int step) {
    constexpr int q = decltype(Q)::value;
    int st = step;
    bool ylo = false, xlo = false;
    if constexpr (SPLIT) {
      const int seg = step >= 2 * nsteps0 ? 2 : (step >= nsteps0 ? 1 : 0);
      st = step - seg * nsteps0;
      ylo = seg == 1; xlo = seg == 2;
    }
    unsigned char* base = ring + slot * SLOT;
    if constexpr (q < NPO) dma16(SPLIT && ylo ? rYl : rYh, base + (q * 4 + wave) * 1024, vy[q], st * step_y);
    else dma16(SPLIT && xlo ? rXl : rXh, base + OPB + ((q - NPO) * 4 + wave) * 1024, vx[q - NPO], st * step_x);
  };
  auto stage = [&](int slot, int step) { sfor<NPW>([&](auto Q) { piece(Q, slot, step); }); };

  const bool has_bias = P.db != nullptr;
  const int bias_mod = 2 * tiles_k;                            // k-tile workgroups x wave columns share one n-range (per quad): take turns
  int bias_cnt = 2 * (tile % tiles_k) + wc;
  const int grp = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
  float* __restrict__ dW = P.dW;
  const int lddw = P.lddw, rg = P.rg, rgp = P.rgp, n_src = P.n_src, cg = P.cg, cgp = P.cgp, k_src = P.k_src;
  // fragment addresses: lane = 16 g + 4 q4 + p4 supplies row 8 g + q4 (+ 4), columns 4 p4 .. 4 p4 + 3 of a 16-column block
  const int sw = ((q4 >> 1) & 1) | ((grp & 1) << 1);
  const int rowb = (8 * grp + q4) * ROWB + (p4 >> 1) * 16 + (p4 & 1) * 8;
  const unsigned lds_q = lds_addr_of(ring);
  const unsigned ya0 = rowb + 32 * ((2 * wr) ^ sw), ya1 = rowb + 32 * ((2 * wr + 1) ^ sw);             // block i: (i & 1 ? ya1 : ya0) + 128 (i >> 1)
  const unsigned xa0 = OPB + rowb + 32 * ((2 * wc) ^ sw), xa1 = OPB + rowb + 32 * ((2 * wc + 1) ^ sw);

  f32x4 acc4[36];                                              // block (i, k): acc4[i * 6 + k]
  f32x4 accb4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 36; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  stage(0, 0);
  if constexpr (NS >= 3) stage(1, 1);
  int slot = 0;
  for (int st = 0; st < nsteps; ++st) {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * NPW) : "memory");
    const unsigned sl = lds_q + slot * SLOT;
    u64 xb[6][2], ya[6][2];
#define TF_RDX(k) xb[k][0] = tr_read_asm<128 * ((k) >> 1)>(sl + (((k) & 1) ? xa1 : xa0)); xb[k][1] = tr_read_asm<128 * ((k) >> 1) + 4 * ROWB>(sl + (((k) & 1) ? xa1 : xa0));
#define TF_RDY(i) ya[i][0] = tr_read_asm<128 * ((i) >> 1)>(sl + (((i) & 1) ? ya1 : ya0)); ya[i][1] = tr_read_asm<128 * ((i) >> 1) + 4 * ROWB>(sl + (((i) & 1) ? ya1 : ya0));
    TF_RDX(0) TF_RDX(1) TF_RDX(2) TF_RDX(3) TF_RDX(4) TF_RDX(5)
    TF_RDY(0) TF_RDY(1) TF_RDY(2)
    TF_RDY(3) TF_RDY(4) TF_RDY(5)
#undef TF_RDX
#undef TF_RDY
    const int nslot = slot == 0 ? NS - 1 : slot - 1;
    const int nstep = st + NS - 1;
    slot = slot == NS - 1 ? 0 : slot + 1;
    bf16x8 bfr[6], af[6];
    auto part = [&](auto H) {
      constexpr int h = decltype(H)::value;
      sfor<18>([&](auto I) {
        constexpr int j = decltype(I)::value, i = 3 * h + j / 6, k = j % 6;
        acc4[i * 6 + k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[k], acc4[i * 6 + k], 0, 0, 0);
        constexpr int done = h * 18 + j + 1;                   // one transfer behind every sixth MFMA of the step
        if constexpr (done % 6 == 0 && done / 6 - 1 < NPW) {
          __builtin_amdgcn_sched_barrier(0);
          piece(std::integral_constant<int, done / 6 - 1>{}, nslot, nstep);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    };
    asm volatile("s_waitcnt lgkmcnt(6)"
                 : "+v"(xb[0][0]), "+v"(xb[0][1]), "+v"(xb[1][0]), "+v"(xb[1][1]), "+v"(xb[2][0]), "+v"(xb[2][1]), "+v"(xb[3][0]), "+v"(xb[3][1]),
                   "+v"(xb[4][0]), "+v"(xb[4][1]), "+v"(xb[5][0]), "+v"(xb[5][1]),
                   "+v"(ya[0][0]), "+v"(ya[0][1]), "+v"(ya[1][0]), "+v"(ya[1][1]), "+v"(ya[2][0]), "+v"(ya[2][1]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) bfr[k] = join_tr64(xb[k][0], xb[k][1]);
#pragma unroll
    for (int i = 0; i < 3; ++i) af[i] = join_tr64(ya[i][0], ya[i][1]);
    part(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ya[3][0]), "+v"(ya[3][1]), "+v"(ya[4][0]), "+v"(ya[4][1]), "+v"(ya[5][0]), "+v"(ya[5][1]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 3; i < 6; ++i) af[i] = join_tr64(ya[i][0], ya[i][1]);
    part(std::integral_constant<int, 1>{});
    const bool my_turn = has_bias && bias_cnt == 0 && (!SPLIT || st < 2 * nsteps0);          // wave-uniform
    bias_cnt = bias_cnt == 0 ? bias_mod - 1 : bias_cnt - 1;
    if (my_turn) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const unsigned w = li == i ? 0x3F803F80u : 0u;             // row i of the bias accumulator collects block i's column sums
        const u32x4 sv = {w, w, w, w};
        accb4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(sv), af[i], accb4, 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // surplus transfers have landed, every wave is past its last fragment read
  // ---- the two quads' partial tiles meet in LDS: a wave hands the partner wave of the other quad the 18 blocks that one flushes (quad 0:
  // i = 0 .. 2, quad 1: i = 3 .. 5) and adds what it receives; image [wave][quad][block][lane] of 16 B, lane-contiguous ----
  {
    float* xch = (float*)smem;
    const int give0 = quad == 0 ? 18 : 0;                      // first of the 18 accumulators handed over
    float* mine = xch + (((wave * 2 + quad) * 18) * 64 + lane) * 4;
#pragma unroll
    for (int b = 0; b < 18; ++b) {
      const f32x4 v = quad == 0 ? acc4[18 + b] : acc4[b];
      *(f32x4*)(mine + b * 256) = v;
    }
    (void)give0;
    __syncthreads();
    const float* theirs = xch + (((wave * 2 + (quad ^ 1)) * 18) * 64 + lane) * 4;
#pragma unroll
    for (int b = 0; b < 18; ++b) {
      const f32x4 v = *(const f32x4*)(theirs + b * 256);
      if (quad == 0) acc4[b] += v; else acc4[18 + b] += v;
    }
  }
  // ---- flush: this quad's 3 x 6 blocks; register r of a block holds rows 4 (lane >> 4) + r, column lane & 15 ----
#pragma unroll
  for (int ii = 0; ii < 3; ++ii) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
          if (qd != quad) continue;                              // (wave-uniform; keeps the accumulator indices compile-time)
          const int i = 3 * qd + ii;
          const int np = n0 + 16 * (4 * (i >> 1) + 2 * wr + (i & 1)) + 4 * grp + r;
          const int ng = np / rgp, ne = np - ng * rgp;
          const int ns = ng * rg + ne;
          const int kp = k0 + 16 * (4 * (k >> 1) + 2 * wc + (k & 1)) + li;
          const int kg = kp / cgp, ke = kp - kg * cgp;
          const int ks = kg * cg + ke;
          if (np < N && ne < rg && ns < n_src && kp < K && ke < cg && ks < k_src) atomicAdd(dW + (size_t)ns * lddw + ks, acc4[i * 6 + k][r]);
        }
      }
    }
  }
  if (has_bias) {                                                // each quad adds the column sums of its own rows
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (grp == (i >> 2)) {
        const int np = n0 + 16 * (4 * (i >> 1) + 2 * wr + (i & 1)) + li;
        const int ng = np / rgp, ne = np - ng * rgp;
        const int ns = ng * rg + ne;
        if (np < N && ne < rg && ns < n_src) atomicAdd(P.db + ns, accb4[i & 3]);
      }
    }
  }
}

int cu_count() {
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
  }();
  return n;
}

}  // namespace

// count problems (each a TfWgradArgs; `groups` expands into that many problems) as one launch.  blocks > 0: that many workgroups in
// flight (the row chunks follow from it); 0: sized here for a launch that has the chip to itself (two workgroups per CU); -1: for a
// launch that runs beside a dependent chain of other kernels.
extern "C" int tf_launch_wgrad_multi(const TfWgradArgs* probs, int count, int blocks, hipStream_t stream) {
  if (probs == nullptr || count <= 0) return -1;
  WgMulti m{};
  int n = 0, tiles = 0, min_steps = 1 << 30, max_steps = 0;
  bool split = false, any = false;
  double flops = 0.0, abytes = 0.0;      // algorithmic work: 2 M N K; operands read once + the fp32 gradient
  for (int i = 0; i < count; ++i) {
    const TfWgradArgs& a = probs[i];
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) continue;
    if (a.dY == nullptr || a.X == nullptr || a.dW == nullptr) return -1;
    const int G = a.groups > 1 ? a.groups : 1;
    const bool ragged = G > 1 && a.group_rows[0] > 0;              // TfWgradArgs.group_rows: range g holds group_rows[g] rows
    if (ragged ? !tf_ragged_ok(a.group_rows, G, a.M) : (a.M % G) != 0) return -7;
    if ((a.N % 8) || (a.K % 8) || (a.ldy % 8) || (a.ldx % 8)) return -2;
    if (a.rgp < a.rg || a.cgp < a.cg || a.rg <= 0 || a.cg <= 0) return -3;
    const bool sp = a.dY_lo != nullptr;
    if (sp && a.X_lo == nullptr) return -6;
    if (any && sp != split) return -8;                        // one arithmetic mode per launch
    split = sp; any = true;
    size_t row0 = 0;
    for (int g = 0; g < G; ++g) {
      if (n >= WG_MAX) return -9;
      const int Mg = ragged ? a.group_rows[g] : a.M / G;
      WgProb& p = m.p[n++];
      p.dY = (const unsigned char*)a.dY + row0 * a.ldy * 2; p.X = (const unsigned char*)a.X + row0 * a.ldx * 2;
      p.dY_lo = sp ? (const unsigned char*)a.dY_lo + row0 * a.ldy * 2 : nullptr;
      p.X_lo = sp ? (const unsigned char*)a.X_lo + row0 * a.ldx * 2 : nullptr;
      row0 += (size_t)Mg;
      p.dW = (float*)((unsigned char*)a.dW + (long long)g * a.dw_gstride);
      p.db = a.db != nullptr ? (float*)((unsigned char*)a.db + (long long)g * a.dw_gstride) : nullptr;
      p.ldy = a.ldy; p.ldx = a.ldx; p.lddw = a.lddw; p.M = Mg; p.N = a.N; p.K = a.K;
      p.rg = a.rg; p.rgp = a.rgp; p.n_src = a.n_src; p.cg = a.cg; p.cgp = a.cgp; p.k_src = a.k_src;
      p.tiles_k = (a.K + 127) / 128;
      p.tile0 = tiles;
      tiles += p.tiles_k * ((a.N + 255) / 256);
      const int steps = (Mg + 31) / 32;
      min_steps = steps < min_steps ? steps : min_steps;
      max_steps = steps > max_steps ? steps : max_steps;
      flops += 2.0 * Mg * a.N * a.K;
      abytes += ((double)Mg * a.N + (double)Mg * a.K) * 2.0 * (sp ? 2.0 : 1.0) + (double)a.N * a.K * 4.0;
    }
  }
  if (n == 0) return 0;
  // Form 2 (192 x 192 tiles, two quads per workgroup, one workgroup per CU) where it makes ONE well-filled round of workgroups that keep
  // a few dozen steps per quad: a d = 768 layer's four products (128 tiles x 2 chunks = 256), its FFN pair (64 x 4), its in-proj (48 x 5)
  // It is taken by launches that have the chip to themselves (blocks = 0: the last launch of a backward).  Beside the chain (-1) its
  // full-CU workgroups leave the chain's row kernels no half-free CUs: 156 against 161 us alone, and the STEP +0.3 % (3.909 against
  // 3.899 ms, tools/experiments/wgm192_ab.sh) -- the 256 x 128 form stays there.
  static const int f192 = TF_ENV_INT("TF_WGM_192", -1);       // experiments: 0 never, 1 whenever the shapes allow, 2 the rule for -1 launches too
  const bool force192 = blocks == -2;                         // tf_gemm_wgrad_multi(blocks = -2): form 2 wherever a quad gets two steps (tests)
  if (force192) blocks = -1;
  if (f192 != 0 && (blocks == 0 || force192 || f192 > 0)) {
    int t192 = 0;
    for (int i = 0; i < n; ++i) t192 += ((m.p[i].N + 191) / 192) * ((m.p[i].K + 191) / 192);
    const int cus = cu_count();
    const int G = t192 >= cus ? 1 : (cus + t192 / 2) / t192;
    const long wgs = (long)t192 * G;
    const int per_quad = ((min_steps + G - 1) / G + 1) / 2;
    const bool fits = wgs <= cus && wgs * 100 >= 85L * cus && per_quad >= 24;
    if ((f192 == 1 || force192) ? per_quad >= 2 : fits) {
      WgMulti m2 = m;
      int tl = 0;
      bool ok32 = true;
      for (int i = 0; i < n; ++i) {
        WgProb& q = m2.p[i];
        q.tiles_k = (q.K + 191) / 192;
        q.tile0 = tl;
        tl += q.tiles_k * ((q.N + 191) / 192);
        int cs = ((q.M + 31) / 32 + G - 1) / G;
        cs += cs & 1;
        q.m_chunk = cs * 32;
        const int ld = q.ldy > q.ldx ? q.ldy : q.ldx;
        if ((long long)q.m_chunk * ld * 2 >= (1ll << 31)) ok32 = false;
      }
      if (ok32) {
        m2.count = n; m2.chunks = G; m2.tiles_total = tl;
        constexpr int NS2 = 3, LDS2 = 2 * NS2 * 2 * 32 * 384;
        TfTraceScope tr(split ? "wgrad_multi192_kernel<x3>" : "wgrad_multi192_kernel", stream, flops, abytes);
        if (split) {
          static const hipError_t o = hipFuncSetAttribute((const void*)wgrad_multi192_kernel<true, NS2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
          (void)o; hipLaunchKernelGGL((wgrad_multi192_kernel<true, NS2>), dim3((unsigned)(tl * G)), dim3(512), LDS2, stream, m2);
        } else {
          static const hipError_t o = hipFuncSetAttribute((const void*)wgrad_multi192_kernel<false, NS2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
          (void)o; hipLaunchKernelGGL((wgrad_multi192_kernel<false, NS2>), dim3((unsigned)(tl * G)), dim3(512), LDS2, stream, m2);
        }
        return (int)hipGetLastError();
      }
    }
  }
  if (blocks == 0) blocks = 2 * cu_count();
  int chunks;
  if (blocks > 0) chunks = (blocks + tiles / 2) / tiles;
  else {
    // Beside the chain: ~1.7 workgroups per CU (three row chunks for a d = 768 layer's 144 tiles: 172 us alone against 210 at two and 207
    // at four), and fewer when M is small, so that a workgroup keeps a few dozen 32-row steps to amortise its prologue and flush
    static const int per_cu_x10 = TF_ENV_INT("TF_WGM_BLOCKS_X10", 17), steps_per_chunk = TF_ENV_INT("TF_WGM_MIN_STEPS", 40);
    chunks = (cu_count() * per_cu_x10 / 10 + tiles / 2) / tiles;
    if (chunks > min_steps / steps_per_chunk) chunks = min_steps / steps_per_chunk;
  }
  if (chunks < 1) chunks = 1;
  if (chunks > min_steps) chunks = min_steps;
  // a chunk's rows are addressed with 32-bit byte offsets inside a buffer resource: keep every chunk under 2 GiB
  for (;;) {
    bool ok = true;
    for (int i = 0; i < n; ++i) {
      const long long rows = (((long long)(m.p[i].M + 31) / 32 + chunks - 1) / chunks) * 32;
      const int ld = m.p[i].ldy > m.p[i].ldx ? m.p[i].ldy : m.p[i].ldx;
      if (rows * ld * 2 >= (1ll << 31)) ok = false;
    }
    if (ok) break;
    chunks *= 2;
  }
  for (int i = 0; i < n; ++i) m.p[i].m_chunk = (((m.p[i].M + 31) / 32 + chunks - 1) / chunks) * 32;
  m.count = n; m.chunks = chunks; m.tiles_total = tiles;
  (void)max_steps;
  constexpr int NS = 3, LDS = NS * (32 * 512 + 32 * 256);
  static const int intl = TF_ENV_INT("TF_WGM_INTL", 1), mf16 = TF_ENV_INT("TF_WGM_MF16", 1);
  dim3 grid((unsigned)tiles * (unsigned)chunks), block(256);
  TfTraceScope tr(split ? "wgrad_multi_kernel<x3>" : "wgrad_multi_kernel", stream, flops, abytes);
#define TF_WGM_LAUNCH(S, I, F) do { \
    static const hipError_t once = hipFuncSetAttribute((const void*)wgrad_multi_kernel<S, NS, I, F>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
    (void)once; hipLaunchKernelGGL((wgrad_multi_kernel<S, NS, I, F>), grid, block, LDS, stream, m); } while (0)
  if (split) { if (mf16) TF_WGM_LAUNCH(true, 1, true); else TF_WGM_LAUNCH(true, 1, false); }
  else if (intl == 0) { if (mf16) TF_WGM_LAUNCH(false, 0, true); else TF_WGM_LAUNCH(false, 0, false); }
  else { if (mf16) TF_WGM_LAUNCH(false, 1, true); else TF_WGM_LAUNCH(false, 1, false); }
#undef TF_WGM_LAUNCH
  return (int)hipGetLastError();
}
