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
#include <cstdlib>
#include <type_traits>
#include <atomic>
#include "tf_common.h"
#include <cstdio>
#include "tf_kernels.h"

namespace {

constexpr int BN = 128;
int num_cus();
// row tiles of a launch: TfGemmArgs.groups independent row ranges of M / groups rows each (tiles never straddle two groups)
// gr: ragged row ranges (TfGemmArgs.group_rows) or null / [0] == 0 for equal ones
inline long row_tiles(int M, int G, int BM, const int* gr = nullptr) {
  const int g = G > 1 ? G : 1;
  if (g > 1 && gr != nullptr && gr[0] > 0) {
    long t = 0;
    for (int i = 0; i < g; ++i) t += (gr[i] + BM - 1) / BM;
    return t;
  }
  return (long)g * ((M / g + BM - 1) / BM);
}
inline long row_tiles(const TfGemmArgs* a, int BM) { return row_tiles(a->M, a->groups, BM, a->group_rows); }
// Device side of the grouping: logical tile -> (group, tile inside the group); the local copy of the arguments is narrowed to the
// group -- rows [m_lo, M) of the global row space, the group's weight / bias / scale tensors -- so that every later bound and clamp of
// the kernel (which all read g.M) holds unchanged.  Returns the tile index inside the group.
__device__ __forceinline__ int enter_group(TfGemmArgs& g, int logical, int BM, int tiles_n, int& m_lo) {
  m_lo = 0;
  if (g.groups <= 1) return logical;
  int grp, Mg, inside;
  if (g.group_rows[0] > 0) {                      // ragged ranges: walk the (at most TF_MAX_GROUPS) tile counts; constant indices, scalar work
    int t = logical, lo = 0, left = logical;
    grp = 0; Mg = g.group_rows[0];
    bool found = false;
#pragma unroll
    for (int i = 0; i < TF_MAX_GROUPS; ++i) {
      const int tg = ((g.group_rows[i] + BM - 1) / BM) * tiles_n;
      if (!found && i < g.groups) {
        if (t < tg || i == g.groups - 1) { found = true; grp = i; Mg = g.group_rows[i]; m_lo = lo; left = t; }
        t -= tg; lo += g.group_rows[i];
      }
    }
    inside = left;
  } else {
    Mg = g.M / g.groups;
    const int tiles_g = ((Mg + BM - 1) / BM) * tiles_n;
    grp = logical / tiles_g;
    m_lo = grp * Mg;
    inside = logical - grp * tiles_g;
  }
  const long long off = (long long)grp * g.w_gstride;
  g.M = m_lo + Mg;
  g.W = (const unsigned char*)g.W + off;
  if (g.W_lo != nullptr) g.W_lo = (const unsigned char*)g.W_lo + off;
  if (g.bias != nullptr) g.bias = (const float*)((const unsigned char*)g.bias + off);
  if (g.scale_w != nullptr) g.scale_w = (const float*)((const unsigned char*)g.scale_w + off);
  return inside;
}
constexpr int TILE_BYTES = 128 * 64 * 2;          // 16 KiB per 128-row x 64-deep operand tile (wgrad; the W operand at BK = 64)
constexpr int CT_STRIDE = 272;                    // C-tile row stride in LDS (256 B + 16 B pad)

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds(TF_GLB_PTR(gsrc), TF_LDS_PTR(lds_dst), 16, 0, 0);
}

// FFN activation (TfGemmArgs.act): h = act(u), dh = act'(u); 0 = exact GELU, 1 = ReLU.  The choice is uniform over a launch:
// act_loop8 branches ONCE per 8-element chunk (a per-element branch splits the chunk into basic blocks and serialises it).
template <bool RELU>
__device__ __forceinline__ void act_parts(float u, float& h, float& dh) {
  if constexpr (RELU) { h = fmaxf(u, 0.f); dh = u > 0.f ? 1.f : 0.f; return; }
  float cdf, ex;
  gelu_parts(u, cdf, ex);
  h = u * cdf;
  dh = cdf + u * 0.39894228040143268f * ex;
}
template <typename F>
__device__ __forceinline__ void act_loop8(int act, F&& body) {      // body(e, relu_tag) for e = 0..7
  if (act == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) body(e, std::true_type{});
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) body(e, std::false_type{});
  }
}
#define TF_ACT(tag, u, h, dh) act_parts<decltype(tag)::value>(u, h, dh)

// 16-byte store of an output chunk; NT: nontemporal (a streaming store: the launch's output burst does not displace what the NEXT kernel
// is about to read from the L2 / the memory-side cache).  Which epilogues store that way: TF_NT_MASK, bit EPI for C, bit 8 + EPI for C2
// (measured per epilogue in the step: DESIGN.md "Round 6"; experiments builds may override the mask).
#ifndef TF_EXPERIMENTS
#undef TF_NT_MASK
#endif
#ifndef TF_NT_MASK
// shipped: the FFN-up outputs (G: read again only by the backward; H) and the residual-add dgrads (EPI_ADD).  Same-box A/B of the step,
// three runs each (gpurun_out/r6_nt_ab3.txt): none 4.181 ms; G 4.162; G + H 4.159; ADD 4.162; every epilogue 4.152; BIAS (QKV, read at
// once by the attention), BIAS_DROP_RES (read at once by LayerNorm), NONE, MUL: within +-0.1 % of none.  The same hint on the LayerNorm
// outputs cost +1.2 %, on the attention outputs +4.6 %: what the next kernel reads at once should stay cached.
#define TF_NT_MASK 0x4050
#endif
template <bool NT> __device__ __forceinline__ void st_c16(void* p, u32x4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, (u32x4*)p);
  else *(u32x4*)p = v;
}
// the fp32-accuracy mode's plane-pair store with the same hint (TF_NT_MASK3: bit EPI for C, bit 8 + EPI for C2)
#ifndef TF_EXPERIMENTS
#undef TF_NT_MASK3
#endif
#ifndef TF_NT_MASK3
#define TF_NT_MASK3 0x4050      // as TF_NT_MASK (the fp32 leg: 9.98 -> 9.91 ms, two runs each, gpurun_out/r6_nt3.txt: inside the noise, same policy)
#endif
template <bool NT> __device__ __forceinline__ void store8_split_nt(void* hi, void* lo, size_t off, const float (&f)[8]) {
  if constexpr (!NT) { store8_split(hi, lo, off, f); return; }
  if (lo != nullptr) {
    u32x4 h, l;
    split8(f, h, l);
    __builtin_nontemporal_store(h, (u32x4*)((u16*)hi + off));
    __builtin_nontemporal_store(l, (u32x4*)((u16*)lo + off));
  } else {
    __builtin_nontemporal_store(pack8(f), (u32x4*)((u16*)hi + off));
  }
}
// elementwise epilogue of one 16-B chunk (8 consecutive columns of one output row) -- shared by both GEMM kernels
// rpre: the chunk of R already in registers (the large-tile kernel fetches every R chunk of its tile before the C tile goes
// through LDS, so the HBM latency is paid once per tile instead of once per chunk), or null = load it here
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_chunk(const TfGemmArgs& g, u16* __restrict__ C, u32x4 v, int gm, int gn,
                                                    const u32x4* rpre = nullptr) {
  constexpr bool NTC = ((TF_NT_MASK >> EPI) & 1) != 0;
  [[maybe_unused]] constexpr bool NTC2 = ((TF_NT_MASK >> (8 + EPI)) & 1) != 0;
  [[maybe_unused]] auto load_r = [&]() -> u32x4 {
    return rpre != nullptr ? *rpre : *(const u32x4*)((const u16*)g.R + (size_t)gm * g.ldr + gn);
  };
  if constexpr (EPI == TF_EPI_BIAS || EPI == TF_EPI_NONE) {
    st_c16<NTC>(C + (size_t)gm * g.ldc + gn, v);
  } else {
    float f[8];
    unpack8(v, f);
    if constexpr (EPI == TF_EPI_BIAS_GELU_DROP) {
      st_c16<NTC>(C + (size_t)gm * g.ldc + gn, v);       // pre-activation U (saved for backward)
      const unsigned km = g.drop_thr ? tf_keep8((unsigned)gm * (unsigned)g.ldc2 + (unsigned)gn, g.drop_key, g.drop_thr) : 0xffu;
      act_loop8(g.act, [&](int e, auto relu) {
        float hh, dh;
        TF_ACT(relu, f[e], hh, dh);
        f[e] = ((km >> e) & 1u) ? hh * g.drop_scale : 0.f;
      });
      st_c16<NTC2>((u16*)g.C2 + (size_t)gm * g.ldc2 + gn, pack8(f));
    } else if constexpr (EPI == TF_EPI_BIAS_GELU_DROP_G) {
      const unsigned km = g.drop_thr ? tf_keep8((unsigned)gm * (unsigned)g.ldc2 + (unsigned)gn, g.drop_key, g.drop_thr) : 0xffu;
      float gd[8];
      act_loop8(g.act, [&](int e, auto relu) {
        float hh, dh;
        TF_ACT(relu, f[e], hh, dh);
        const float keep = ((km >> e) & 1u) ? g.drop_scale : 0.f;
        gd[e] = keep * dh;                                           // d dropout(act(u)) / du
        f[e] = keep * hh;
      });
      st_c16<NTC>(C + (size_t)gm * g.ldc + gn, pack8(gd));
      st_c16<NTC2>((u16*)g.C2 + (size_t)gm * g.ldc2 + gn, pack8(f));
    } else if constexpr (EPI == TF_EPI_MUL) {
      float r[8];
      unpack8(load_r(), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= r[e];
      st_c16<NTC>(C + (size_t)gm * g.ldc + gn, pack8(f));
    } else if constexpr (EPI == TF_EPI_BIAS_DROP_RES) {
      float r[8];
      unpack8(load_r(), r);
      const unsigned km = g.drop_thr ? tf_keep8((unsigned)gm * (unsigned)g.ldc + (unsigned)gn, g.drop_key, g.drop_thr) : 0xffu;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = r[e] + (((km >> e) & 1u) ? f[e] * g.drop_scale : 0.f);
      st_c16<NTC>(C + (size_t)gm * g.ldc + gn, pack8(f));
    } else if constexpr (EPI == TF_EPI_ADD) {
      float r[8];
      unpack8(load_r(), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += r[e];
      st_c16<NTC>(C + (size_t)gm * g.ldc + gn, pack8(f));
    } else if constexpr (EPI == TF_EPI_DGELU_DROP) {
      // dU = dH . mask/(1-p) . gelu'(U); R = U, dropout index space = that of H (ldr == ld of H)
      float u[8];
      unpack8(load_r(), u);
      const unsigned km = g.drop_thr ? tf_keep8((unsigned)gm * (unsigned)g.ldr + (unsigned)gn, g.drop_key, g.drop_thr) : 0xffu;
      act_loop8(g.act, [&](int e, auto relu) {
        float hh, dh;
        TF_ACT(relu, u[e], hh, dh);
        f[e] = ((km >> e) & 1u) ? f[e] * g.drop_scale * dh : 0.f;
      });
      st_c16<NTC>(C + (size_t)gm * g.ldc + gn, pack8(f));
    }
  }
}

// ---- fp32-accuracy mode (TfGemmArgs.A_lo != null): the same epilogues on fp32 values, every bf16 tensor a hi + lo plane pair ----
// rhi / rlo: the chunk's R planes already in registers (split_epilogue fetches a whole group's before its LDS pass), or null
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_chunk_f32(const TfGemmArgs& g, float (&f)[8], int gm, int gn,
                                                        const u32x4* rhi = nullptr, const u32x4* rlo = nullptr) {
  const size_t oc = (size_t)gm * g.ldc + gn;
  if constexpr (EPI == TF_EPI_BIAS || EPI == TF_EPI_NONE) {
    if (g.c_is_f32) {                                    // (uniform over the launch) fp32 result, no planes
      *(f32x4*)((float*)g.C + oc) = f32x4{f[0], f[1], f[2], f[3]};
      *(f32x4*)((float*)g.C + oc + 4) = f32x4{f[4], f[5], f[6], f[7]};
    } else {
      store8_split(g.C, g.C_lo, oc, f);
    }
  } else if constexpr (EPI == TF_EPI_BIAS_GELU_DROP || EPI == TF_EPI_BIAS_GELU_DROP_G) {
    const unsigned km = g.drop_thr ? tf_keep8((unsigned)gm * (unsigned)g.ldc2 + (unsigned)gn, g.drop_key, g.drop_thr) : 0xffu;
    float gd[8], hv[8];
    act_loop8(g.act, [&](int e, auto relu) {
      float hh, dh;
      TF_ACT(relu, f[e], hh, dh);
      const float keep = ((km >> e) & 1u) ? g.drop_scale : 0.f;
      gd[e] = keep * dh;
      hv[e] = keep * hh;
    });
    constexpr bool NT3 = ((TF_NT_MASK3 >> EPI) & 1) != 0, NT3B = ((TF_NT_MASK3 >> (8 + EPI)) & 1) != 0;
    if constexpr (EPI == TF_EPI_BIAS_GELU_DROP) store8_split_nt<NT3>(g.C, g.C_lo, oc, f);      // pre-activation U
    else store8_split_nt<NT3>(g.C, g.C_lo, oc, gd);                                             // G = d h / d u
    store8_split_nt<NT3B>(g.C2, g.C2_lo, (size_t)gm * g.ldc2 + gn, hv);
  } else {
    float r[8];
    if (rhi != nullptr) join8(*rhi, *rlo, r);
    else load8_split(g.R, g.R_lo, (size_t)gm * g.ldr + gn, r);
    if constexpr (EPI == TF_EPI_MUL) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= r[e];
    } else if constexpr (EPI == TF_EPI_BIAS_DROP_RES) {
      const unsigned km = g.drop_thr ? tf_keep8((unsigned)gm * (unsigned)g.ldc + (unsigned)gn, g.drop_key, g.drop_thr) : 0xffu;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = r[e] + (((km >> e) & 1u) ? f[e] * g.drop_scale : 0.f);
    } else if constexpr (EPI == TF_EPI_ADD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += r[e];
    } else if constexpr (EPI == TF_EPI_DGELU_DROP) {
      const unsigned km = g.drop_thr ? tf_keep8((unsigned)gm * (unsigned)g.ldr + (unsigned)gn, g.drop_key, g.drop_thr) : 0xffu;
      act_loop8(g.act, [&](int e, auto relu) {
        float hh, dh;
        TF_ACT(relu, r[e], hh, dh);
        f[e] = ((km >> e) & 1u) ? f[e] * g.drop_scale * dh : 0.f;
      });
    }
    store8_split_nt<((TF_NT_MASK3 >> EPI) & 1) != 0>(g.C, g.C_lo, oc, f);
  }
}

// The accumulators of a workgroup (waves NWR x NWC, each 64 columns x 16*MBLK rows as f32x4 blocks acc[ni][mi]: row
// 16 mi + (lane & 15), columns 16 ni + 4 (lane >> 4) .. +3) pass through LDS as FP32, PM 16-row blocks of every wave at a time,
// and leave as row-contiguous 8-column chunks for gemm_epilogue_chunk_f32.
template <int EPI, int MBLK, int PM, int TBN, int NT>
__device__ __forceinline__ void split_epilogue(const TfGemmArgs& g, f32x4 (&acc)[4][MBLK], unsigned char* ct, int m0, int n0, int wave_rows,
                                               int wr, int wc, int lane, int tid) {
  constexpr int RS = TBN * 4 + 16, CH = TBN / 8;
  constexpr bool HAS_R = EPI == TF_EPI_MUL || EPI == TF_EPI_ADD || EPI == TF_EPI_BIAS_DROP_RES || EPI == TF_EPI_DGELU_DROP;
  constexpr int PER = (2 * PM * 16 * CH + NT - 1) / NT;              // chunks of a group per thread
#pragma unroll
  for (int p0 = 0; p0 < MBLK; p0 += PM) {
    // this thread's R chunks of the group, both planes, fetched before the group passes through LDS (one HBM latency per group
    // instead of one per chunk)
    u32x4 rhi[HAS_R ? PER : 1], rlo[HAS_R ? PER : 1];
    if constexpr (HAS_R) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int id = min(k * NT + tid, 2 * PM * 16 * CH - 1);
        const int row_l = id / CH, c = id - row_l * CH;
        const int w = row_l / (PM * 16), rem = row_l - w * (PM * 16);
        const int gm = min(m0 + w * wave_rows + p0 * 16 + rem, g.M - 1), gn = min(n0 + c * 8, g.N - 8);   // clamped: unused when out of range
        const size_t off = (size_t)gm * g.ldr + gn;
        rhi[k] = *(const u32x4*)((const u16*)g.R + off);
        rlo[k] = *(const u32x4*)((const u16*)g.R_lo + off);
      }
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int nl = wc * 64 + ni * 16 + (lane >> 4) * 4;
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (g.bias != nullptr && n0 + nl < g.N) b = *(const f32x4*)(g.bias + n0 + nl);
#pragma unroll
      for (int pm = 0; pm < PM; ++pm) {
        if (p0 + pm < MBLK) {
          const int row_l = (wr * PM + pm) * 16 + (lane & 15);
          *(f32x4*)(ct + row_l * RS + nl * 4) = acc[ni][p0 + pm] + b;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int id = k * NT + tid;
      if (id >= 2 * PM * 16 * CH) continue;
      const int row_l = id / CH, c = id - row_l * CH;
      const int w = row_l / (PM * 16), rem = row_l - w * (PM * 16);
      if (p0 + rem / 16 >= MBLK) continue;
      const int gm = m0 + w * wave_rows + p0 * 16 + rem, gn = n0 + c * 8;
      if (gm >= g.M || gn >= g.N) continue;
      const f32x4 lo4 = *(const f32x4*)(ct + row_l * RS + c * 32), hi4 = *(const f32x4*)(ct + row_l * RS + c * 32 + 16);
      float f[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      gemm_epilogue_chunk_f32<EPI>(g, f, gm, gn, HAS_R ? &rhi[k] : nullptr, HAS_R ? &rlo[k] : nullptr);
    }
    __syncthreads();
  }
}

// chunk swizzle of a tile row: BK = 64 (128-B rows, 8 chunks): ^ (r>>1)&7 ; BK = 32 (64-B rows, 4 chunks; four rows
// span one 256-B bank row): ^ g[(r>>2)&3], g = {0,2,3,1}.  Both make the 16 rows of a ds_read_b128 lane group hit 16 distinct slots.
template <int BK> __device__ __forceinline__ int swz(int r) { return BK == 64 ? ((r >> 1) & 7) : ((0x78 >> ((r >> 1) & 6)) & 3); }

// SPLIT (fp32-accuracy mode), this 128x128 kernel: the K loop runs three times over the operands' planes -- (A_hi, W_hi), (A_lo, W_hi), (A_hi, W_lo) -- into
// the same fp32 accumulators; the epilogue works on fp32 values and writes hi + lo planes.
// NSLOT > 2 ("ring"): for grids of at most about one workgroup per CU (small row counts: the reference's per-GPU batch of 4 - 5
// samples, the wrapper's upper levels).  There no second workgroup hides the DMA latency and the double-buffered loop runs at one
// L2 round trip per K-tile (M = 2083, N = 768, K = 1536: 34 us for 5 GFLOP); a 4-slot ring with counted waits keeps three K-tiles
// of transfers in flight, as in the large-tile kernel.
template <int EPI, int MI, int BK, bool SPLIT = false, int NSLOT = 2>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const TfGemmArgs g_in) {
  TfGemmArgs g = g_in;
  g.drop_key = tf_salted(g.drop_key);                            // the step clock (tf_common.h)
  static_assert(NSLOT == 2 || (!SPLIT && BK == 64), "the ring form exists for bf16 operands, BK = 64");
  constexpr int BM = 32 * MI;
  constexpr int ROWB = BK * 2;                                   // bytes per tile row (128 or 64)
  constexpr int RPI = 1024 / ROWB;                               // rows per 1-KiB DMA instruction (8 or 16)
  constexpr int CPR = ROWB / 16;                                 // 16-B chunks per row (8 or 4)
  constexpr int A_BYTES = BM * ROWB, W_BYTES = 128 * ROWB, BUF_BYTES = A_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (g.N + BN - 1) / BN;
  int m_lo;
  const int logical = enter_group(g, xcd_remap(blockIdx.x, gridDim.x), BM, tiles_n, m_lo);
  const int m0 = m_lo + (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const int nk0 = g.K / BK;

  auto stage = [&](int buf, int kstep) {
    const u16* __restrict__ A = (const u16*)g.A;
    const u16* __restrict__ W = (const u16*)g.W;
    int kt = kstep;
    if constexpr (SPLIT) {                                    // wave-uniform plane selection
      const int seg = kstep >= 2 * nk0 ? 2 : (kstep >= nk0 ? 1 : 0);
      kt = kstep - seg * nk0;
      if (seg == 1) A = (const u16*)g.A_lo;
      if (seg == 2) W = (const u16*)g.W_lo;
    }
    unsigned char* abase = smem + buf * BUF_BYTES;
    unsigned char* bbase = abase + A_BYTES;
#pragma unroll
    for (int i = 0; i < (BM / RPI + 3) / 4; ++i) {          // A tile: BM rows, RPI rows per 1-KiB wave-instruction
      const int inst = i * 4 + wave;
      if (inst < BM / RPI) {
        const int R0 = inst * RPI;
        const int r = R0 + lane / CPR;
        const int c = (lane % CPR) ^ swz<BK>(r);            // swizzle on the source chunk
        const int gm = min(m0 + r, g.M - 1);
        glds16(A + (size_t)gm * g.lda + kt * BK + c * 8, abase + R0 * ROWB);
      }
    }
#pragma unroll
    for (int i = 0; i < (128 / RPI + 3) / 4; ++i) {         // W tile: 128 rows
      const int inst = i * 4 + wave;
      if (inst < 128 / RPI) {
        const int R0 = inst * RPI;
        const int r = R0 + lane / CPR;
        const int c = (lane % CPR) ^ swz<BK>(r);
        const int gn = min(n0 + r, g.N - 1);
        glds16(W + (size_t)gn * g.ldw + kt * BK + c * 8, bbase + R0 * ROWB);
      }
    }
  };

  f32x4 acc[4][MI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = SPLIT ? 3 * nk0 : nk0;
  const int frow = lane & 15, fch = lane >> 4;
  if constexpr (NSLOT > 2) {
    constexpr int DIST = NSLOT - 1, PER_WAVE = MI + 4;             // DMA instructions per wave and K-tile: 4 MI (A) + 16 (W) over 4 waves
#pragma unroll
    for (int d = 0; d < DIST; ++d) stage(d, min(d, nk - 1));       // (past the end the last tile is re-fetched: one counted wait fits all phases)
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DIST - 1) * PER_WAVE) : "memory");
      const unsigned char* abase = smem + cur * BUF_BYTES;
      const unsigned char* bbase = abase + A_BYTES;
      bf16x8 wf[2][4], xf[2][MI];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rn = wc * 64 + i * 16 + frow;
          wf[ks][i] = *(const bf16x8*)(bbase + rn * ROWB + (((ks * 4 + fch) ^ swz<BK>(rn)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int rm = wr * (BM / 2) + i * 16 + frow;
          xf[ks][i] = *(const bf16x8*)(abase + rm * ROWB + (((ks * 4 + fch) ^ swz<BK>(rm)) << 4));
        }
      }
      // (after the reads in program order: hipcc cannot tell the DMA's LDS destination from the slot being read)
      // slot of tile kt + DIST == slot of tile kt - 1: every wave finished reading it before this barrier
      stage(cur == 0 ? NSLOT - 1 : cur - 1, min(kt + DIST, nk - 1));
      cur = cur == NSLOT - 1 ? 0 : cur + 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][ni], xf[ks][mi], acc[ni][mi], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // surplus transfers land before the C tile reuses LDS
  } else {
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const unsigned char* abase = smem + cur * BUF_BYTES;
    const unsigned char* bbase = abase + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 wf[4], xf[MI];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rn = wc * 64 + i * 16 + frow;
        wf[i] = *(const bf16x8*)(bbase + rn * ROWB + (((ks * 4 + fch) ^ swz<BK>(rn)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int rm = wr * (BM / 2) + i * 16 + frow;
        xf[i] = *(const bf16x8*)(abase + rm * ROWB + (((ks * 4 + fch) ^ swz<BK>(rm)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    }
    __syncthreads();
  }
  }

  if constexpr (SPLIT) {
    split_epilogue<EPI, MI, 2, BN, 256>(g, acc, smem, m0, n0, BM / 2, wr, wc, lane, tid);
    return;
  }
  // ---- epilogue phase 1: (acc + bias) -> bf16 -> LDS C tile [BM][CT_STRIDE] ----
  unsigned char* ct = smem;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int nl = wc * 64 + ni * 16 + (lane >> 4) * 4;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (g.bias != nullptr && n0 + nl < g.N) b = *(const f32x4*)(g.bias + n0 + nl);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int ml = wr * (BM / 2) + mi * 16 + (lane & 15);
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
  for (int i = 0; i < 2 * MI; ++i) {
    const int id = i * 256 + tid;
    const int row = id >> 4, c = id & 15;
    const int gm = m0 + row, gn = n0 + c * 8;
    if (gm >= g.M || gn >= g.N) continue;
    u32x4 v = *(const u32x4*)(ct + row * CT_STRIDE + c * 16);
    gemm_epilogue_chunk<EPI>(g, C, v, gm, gn);
  }
}

// ------------------------------------------------------------------------------------------------
// Large-tile variant for the big launches: (32*MF) x 256 output tile (MF = 8 / 9 / 10 -> 256 / 288 / 320 rows),
// 512 threads = 8 waves as 2 (M) x 4 (N), each wave (16*MF) x 64, one workgroup per CU.  Twice the MFMA work per
// staged byte of the 128-wide kernel (whose waves spend their issue slots on LDS-DMA), and:
//   * K advances in 32-deep steps through a 4-SLOT LDS ring; the DMA of step i+3 is issued in phase i, right after the
//     barrier that proves slot (i+3)%4 was fully read, so every transfer has three phases (~3.5k cycles) to land;
//   * waits are COUNTED (s_waitcnt vmcnt(8): everything except the two youngest phases' transfers) and the barrier is a
//     raw s_barrier -- never __syncthreads(), which would drain the in-flight transfers;
//   * a slot is read one phase AFTER the wait + barrier that retire its transfers.
// ------------------------------------------------------------------------------------------------
constexpr int BIG_BN = 256, BIG_BK = 32, BIG_ROWB = 64;          // 64-B tile rows, 16 rows per 1-KiB DMA instruction
constexpr int BIG_CT_STRIDE = 528;                               // C-tile row stride in LDS (512 B + 16 B pad)

__device__ __forceinline__ void wait_vm_barrier8() { asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_vm_barrier4() { asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_vm_barrier0() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }

// FP8: operands are OCP e4m3 bytes.  The byte geometry is unchanged (64-B tile rows = 64 values, one K-step = 64 values); the K loop
// pairs two K-steps per v_mfma_scale_f32_16x16x128_f8f6f4 (see the loop), and the per-row / per-output-channel scales are applied with
// the bias.
// NWR = 1 ("duo"): half the rows -- (16*MF) x 256, 256 threads = 4 waves of the same 144 x 64 wave tile, 3-slot ring (76.8 KB), TWO
// workgroups per CU.  Costs 47 % more DMA bytes per MFMA (the W tile is staged once per 144 rows instead of once per 288) and is no
// faster with the chip to itself (N = 1536, K = 768: 70 vs 66 us), but inside the step the launches of two or more rounds gain 2-4 %
// and so do the kernels that follow them (finer-grained tail; a 256-thread / 77 KB workgroup can share a CU with a weight-gradient
// workgroup of the side stream, a 512-thread / 139 KB one cannot): 5.46 -> 5.36 ms per step, same box.  Single-round launches lose.
// Experiment switch kept: the workgroups with blockIdx in [stagger_lo, stagger_hi) (the second slot of each CU in the first round)
// can start `stagger_ticks` (100 MHz) late, to put one workgroup's store burst under the other's K loop -- measured neutral to
// negative at 5 / 15 / 30 us, alone and in the step, so the default is 0.
template <int EPI, int MF, bool FP8 = false, bool SPLIT = false, int NWR = 2>
__global__ __launch_bounds__(256 * NWR, 2) void gemm_nt_big_kernel(const TfGemmArgs g_in, int stagger_lo, int stagger_hi, int stagger_ticks) {
  TfGemmArgs g = g_in;
  g.drop_key = tf_salted(g.drop_key);                            // the step clock (tf_common.h)
  static_assert(!(FP8 && SPLIT), "fp8 operands have no lo plane");
  static_assert(NWR == 2 || !(FP8 || SPLIT), "the two-per-CU form exists for bf16 operands only");
  constexpr int ES = FP8 ? 1 : 2;                                  // bytes per operand element
  constexpr int NT = 256 * NWR, NWAVES = 4 * NWR;
  constexpr int NSLOT = NWR == 2 ? 4 : 3, DIST = NSLOT - 1;        // ring slots; the DMA of step i + DIST is issued in phase i
  constexpr int BM = 16 * MF * NWR;
  constexpr int A_BYTES = BM * BIG_ROWB, W_BYTES = BIG_BN * BIG_ROWB, SLOT = A_BYTES + W_BYTES;
  constexpr int NA = BM / 16, NW = BIG_BN / 16, NINST = NA + NW;   // DMA instructions per K-step (34 / 36 / 36; duo 25)
  constexpr int PER_WAVE = (NINST + NWAVES - 1) / NWAVES;          // <= 5 (duo: 7)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = NWR == 2 ? wave >> 2 : 0, wc = wave & 3;
  if constexpr (NWR == 1) {
    if (stagger_ticks > 0 && (int)blockIdx.x >= stagger_lo && (int)blockIdx.x < stagger_hi) {
      const unsigned long long r0 = wall_clock64();                 // s_memrealtime: 100 MHz, independent of the core clock
      while ((long long)(wall_clock64() - r0) < (long long)stagger_ticks) __builtin_amdgcn_s_sleep(32);
    }
  }
  const int tiles_n = (g.N + BIG_BN - 1) / BIG_BN;
  int m_lo;
  const int logical = enter_group(g, xcd_remap(blockIdx.x, gridDim.x), BM, tiles_n, m_lo);
  const int m0 = m_lo + (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BIG_BN;
  const unsigned char* __restrict__ A = (const unsigned char*)g.A;
  const unsigned char* __restrict__ W = (const unsigned char*)g.W;

  // per-lane byte offsets of this wave's DMA sources (loop invariant); instruction j = i*8 + wave covers 16 rows x 64 B
  size_t src_off[PER_WAVE];
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int j = min(i * NWAVES + wave, NINST - 1);          // surplus slots repeat the last transfer (same bytes, same place)
    const bool isW = j >= NA;
    const int r = (isW ? j - NA : j) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz<32>(r);
    const int grow = isW ? min(n0 + r, g.N - 1) : min(m0 + r, g.M - 1);
    src_off[i] = (size_t)grow * (isW ? g.ldw : g.lda) * ES + c * 16;
  }
  const int nk0 = g.K * ES / (BIG_BK * 2);                         // K-steps of 64 operand bytes in one pass over K
  auto stage = [&](int slot, int step) {
    unsigned char* sl = smem + slot * SLOT;
    int kstep = step;
    const unsigned char* Ap = A;
    const unsigned char* Wp = W;
    if constexpr (SPLIT) {                                        // phase 2k: the hi planes of K-step k, phase 2k+1: its lo planes (wave-uniform)
      kstep = step >> 1;
      Ap = (step & 1) ? (const unsigned char*)g.A_lo : A;
      Wp = (step & 1) ? (const unsigned char*)g.W_lo : W;
    }
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int j = min(i * NWAVES + wave, NINST - 1);          // wave-uniform, branch-free (keeps the K loop one basic block)
      const bool isW = j >= NA;
      const unsigned char* base = (isW ? Wp : Ap) + (size_t)kstep * (BIG_BK * 2);
      unsigned char* dst = sl + (isW ? A_BYTES + (j - NA) * 1024 : j * 1024);
      glds16(base + src_off[i], dst);
    }
  };

  f32x4 acc[4][MF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = SPLIT ? 2 * nk0 : nk0;
  const int frow = lane & 15, fch = lane >> 4;
  // fragment j sits 16 rows = 1024 B after fragment 0 with the SAME swizzle: one base register each plus immediates
  const int rn0 = wc * 64 + frow, rm0 = wr * (16 * MF) + frow;
  const int woff0 = A_BYTES + rn0 * BIG_ROWB + ((fch ^ swz<32>(rn0)) << 4);
  const int xoff0 = rm0 * BIG_ROWB + ((fch ^ swz<32>(rm0)) << 4);
  if constexpr (SPLIT) {
    // fp32-accuracy mode: A_hi.W_hi + A_lo.W_hi + A_hi.W_lo per K-step in TWO ring phases instead of three passes over K.  Phase 2k
    // stages (A_hi, W_hi) of K-step k and multiplies them; phase 2k+1 stages (A_lo, W_lo) and forms A_lo.W_hi and A_hi.W_lo with the hi
    // planes still sitting in the previous slot -- a third fewer DMA bytes and barriers for the same matrix work.  Four slots, the DMA
    // of phase p+2 issued in phase p: slot (p+2)%4 == (p-2)%4 was last read in phase p-1, which every wave has left behind the barrier.
    // In an odd phase the second fragment set is read BEFORE the DMA issue in program order (hipcc would otherwise put a vmcnt(0)
    // between an LDS-DMA and any later LDS read).
    stage(0, 0);
    stage(1, min(1, nk - 1));
    auto product = [&](const bf16x8 (&wf)[4], const bf16x8 (&xf)[MF]) {
#pragma unroll
      for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    };
    for (int p = 0; p < nk; p += 2) {
      const int s0 = p & 3, s1 = (p + 1) & 3;                       // slots of the hi and the lo phase of this K-step
      {                                                             // ---- even phase: hi x hi ----
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PER_WAVE) : "memory");
        bf16x8 wf[4], xf[MF];
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(smem + s0 * SLOT + woff0 + j * 1024);
#pragma unroll
        for (int j = 0; j < MF; ++j) xf[j] = *(const bf16x8*)(smem + s0 * SLOT + xoff0 + j * 1024);
        stage((p + 2) & 3, min(p + 2, nk - 1));
        product(wf, xf);
      }
      {                                                             // ---- odd phase: lo x hi, then hi x lo ----
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PER_WAVE) : "memory");
        bf16x8 wf[4], xf[MF];
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(smem + s0 * SLOT + woff0 + j * 1024);      // W_hi (previous slot)
#pragma unroll
        for (int j = 0; j < MF; ++j) xf[j] = *(const bf16x8*)(smem + s1 * SLOT + xoff0 + j * 1024);     // A_lo
        product(wf, xf);
        bf16x8 wl[4], xh[MF];
#pragma unroll
        for (int j = 0; j < 4; ++j) wl[j] = *(const bf16x8*)(smem + s1 * SLOT + woff0 + j * 1024);      // W_lo
#pragma unroll
        for (int j = 0; j < MF; ++j) xh[j] = *(const bf16x8*)(smem + s0 * SLOT + xoff0 + j * 1024);     // A_hi (previous slot)
        stage((p + 3) & 3, min(p + 3, nk - 1));
        product(wl, xh);
      }
    }
  } else if constexpr (FP8) {
    // fp8 operands at the DOUBLE rate: v_mfma_scale_f32_16x16x128_f8f6f4 (BASELINE configs[4]).  One instruction contracts 128 values:
    // a lane's 32 operand bytes are its 16-B chunk of K-step 2j followed by the same chunk of K-step 2j+1 -- both operands use the same
    // byte <-> k map, so every k meets its partner exactly once.  The four ring slots work as two PAIRS: pair j is consumed while the
    // transfers of pair j+1 land in the other two slots (issued after this pair's fragment reads in program order, behind the barrier
    // that proves the other pair was fully read).  The block scales of the instruction are 2^0 (E8M0 127): the per-row / per-channel
    // scales of TfGemmArgs are applied with the bias, as before.  An odd K-step count: the missing half contributes zeros.
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    const int npair = (nk + 1) >> 1;
    stage(0, 0);
    stage(1, min(1, nk - 1));
    for (int j = 0; j < npair; ++j) {
      const int s0 = (2 * j) & 3, s1 = s0 + 1;
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      const bool full = 2 * j + 1 < nk;                              // wave-uniform
      i32x8 wf[4], xf[MF];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const i32x4 lo = *(const i32x4*)(smem + s0 * SLOT + woff0 + q * 1024);
        i32x4 hi = *(const i32x4*)(smem + s1 * SLOT + woff0 + q * 1024);
        if (!full) hi = i32x4{0, 0, 0, 0};
        wf[q] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int q = 0; q < MF; ++q) {
        const i32x4 lo = *(const i32x4*)(smem + s0 * SLOT + xoff0 + q * 1024);
        i32x4 hi = *(const i32x4*)(smem + s1 * SLOT + xoff0 + q * 1024);
        if (!full) hi = i32x4{0, 0, 0, 0};
        xf[q] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      if (j + 1 < npair) {                                           // (wave-uniform) the next pair into the other two slots
        stage((2 * j + 2) & 3, min(2 * j + 2, nk - 1));
        stage((2 * j + 3) & 3, min(2 * j + 3, nk - 1));
      }
#pragma unroll
      for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
  } else {
  // prologue: steps 0, 1, 2 in flight (past the end the last step is re-fetched: the loop body is branch-free, and
    // every phase always has exactly two younger steps' transfers in flight, so one counted wait fits all phases)
  #pragma unroll
    for (int d = 0; d < DIST; ++d) stage(d, min(d, nk - 1));
    int cur = 0;                                                     // ring slot of step i
    for (int i = 0; i < nk; ++i) {
      // retire step i's transfers (issued DIST phases ago) on every wave, then make them visible
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DIST - 1) * PER_WAVE) : "memory");
      const unsigned char* slw = smem + cur * SLOT + woff0;
      const unsigned char* slx = smem + cur * SLOT + xoff0;
      bf16x8 wf[4], xf[MF];
  #pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(slw + j * 1024);
  #pragma unroll
      for (int j = 0; j < MF; ++j) xf[j] = *(const bf16x8*)(slx + j * 1024);
      // (after the reads in program order: the compiler cannot tell the DMA's LDS destination from the slot being read)
      // slot of step i + DIST == slot of step i - 1: every wave finished reading it before this barrier
      stage(cur == 0 ? NSLOT - 1 : cur - 1, min(i + DIST, nk - 1));
      cur = cur == NSLOT - 1 ? 0 : cur + 1;
  #pragma unroll
      for (int mi = 0; mi < MF; ++mi)
  #pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
      // schedule: fragment reads first, then the step's DMA instructions spread one per 7 MFMAs, so that a wave's DMA
      // issue (tens of cycles each) overlaps its own and its SIMD partner's matrix work instead of preceding it
      constexpr int MM = 1;
      __builtin_amdgcn_sched_group_barrier(0x100, 4 + MF, 0);
      constexpr int GAP = (NWR == 2 && (4 * MF) / PER_WAVE > 7) ? 7 : (4 * MF) / PER_WAVE;      // MFMAs between two DMA instructions
  #pragma unroll
      for (int q = 0; q < PER_WAVE; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, GAP * MM, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, (4 * MF - GAP * PER_WAVE) * MM, 0);
    }
}
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // surplus transfers must land before LDS is reused
#if defined(TF_EXPERIMENTS) && defined(TF_ABL_EPI)
  if (TF_ABL_EPI & 2) {                 // timing ablation: no epilogue at all (the accumulators are kept alive by one conditional store)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MF; ++j) t += acc[i][j][0] + acc[i][j][3];
    if (t == 12345.678f) ((float*)g.C)[0] = t;
    return;
  }
#endif

  if constexpr (SPLIT) {
    split_epilogue<EPI, MF, 3, BIG_BN, 512>(g, acc, smem, m0, n0, 16 * MF, wr, wc, lane, tid);
    return;
  }
  // every chunk of R this thread will need in phase 2 is fetched half-way through phase 1: the loads fly while the C tile
  // passes through LDS, so HBM latency is paid once per tile, not once per chunk (half of the accumulators are dead by then:
  // 2*MF*4 registers of R next to 2*MF*4 of accumulators)
  constexpr bool HAS_R = EPI == TF_EPI_MUL || EPI == TF_EPI_ADD || EPI == TF_EPI_BIAS_DROP_RES || EPI == TF_EPI_DGELU_DROP;
  u32x4 rpre[HAS_R ? 2 * MF : 1];
  // ---- epilogue phase 1: (acc + bias) -> bf16 -> LDS C tile [BM][BIG_CT_STRIDE] ----
  unsigned char* ct = smem;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int nl = wc * 64 + ni * 16 + (lane >> 4) * 4;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (g.bias != nullptr && n0 + nl < g.N) b = *(const f32x4*)(g.bias + n0 + nl);
    f32x4 sw = {1.f, 1.f, 1.f, 1.f};
    if constexpr (FP8) { if (g.scale_w != nullptr && n0 + nl < g.N) sw = *(const f32x4*)(g.scale_w + n0 + nl); }
    if constexpr (HAS_R) {
      if (ni == 2) {
#pragma unroll
        for (int i = 0; i < 2 * MF; ++i) {
          const int id = i * NT + tid;
          const int gm = min(m0 + (id >> 5), g.M - 1), gn = min(n0 + (id & 31) * 8, g.N - 8);     // clamped: never used out of range
          rpre[i] = *(const u32x4*)((const u16*)g.R + (size_t)gm * g.ldr + gn);
        }
      }
    }
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      const int ml = wr * (16 * MF) + mi * 16 + (lane & 15);
      f32x4 c4 = acc[ni][mi];
      if constexpr (FP8) {
        const float sa = g.scale_a != nullptr ? g.scale_a[min(m0 + ml, g.M - 1)] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) c4[e] *= sa * sw[e];
      }
      u32x2 v;
      v[0] = pack2bf(c4[0] + b[0], c4[1] + b[1]);
      v[1] = pack2bf(c4[2] + b[2], c4[3] + b[3]);
      *(u32x2*)(ct + ml * BIG_CT_STRIDE + nl * 2) = v;
    }
  }
  __syncthreads();
  // ---- phase 2: row-contiguous 16-B chunks (32 per row), elementwise epilogue, coalesced stores ----
  u16* __restrict__ C = (u16*)g.C;
  constexpr int UNR = HAS_R ? 2 * MF : 2;          // rpre[] must stay in registers; the long activation bodies stay rolled
#pragma unroll UNR
  for (int i = 0; i < 2 * MF; ++i) {
    const int id = i * NT + tid;
    const int row = id >> 5, c = id & 31;
    const int gm = m0 + row, gn = n0 + c * 8;
    if (gm >= g.M || gn >= g.N) continue;
    u32x4 v = *(const u32x4*)(ct + row * BIG_CT_STRIDE + c * 16);
#if defined(TF_EXPERIMENTS) && defined(TF_ABL_EPI)
    if ((TF_ABL_EPI & 1) && v[0] != 0x12345678u) continue;       // timing ablation: the LDS round trip without the global stores
#endif
    gemm_epilogue_chunk<EPI>(g, C, v, gm, gn, HAS_R ? &rpre[i] : nullptr);
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
// SPLIT (fp32-accuracy mode): three passes over the block's rows -- (dY_hi, X_hi), (dY_lo, X_hi), (dY_hi, X_lo) -- into the same
// accumulators; the bias grad (column sums of dY) takes passes 0 and 1.
// grouped weight gradient (TfWgradArgs.groups, blockIdx.y = group): the local copy of the arguments is moved to the group's row range
// of dY / X (M / groups rows) and to its own dW / db, groups * dw_gstride bytes apart
__device__ __forceinline__ void enter_wgrad_group(TfWgradArgs& g) {
  if (g.groups <= 1) return;
  const int grp = blockIdx.y, Mg = g.M / g.groups;
  g.dY = (const unsigned char*)g.dY + (size_t)grp * Mg * g.ldy * 2;
  g.X = (const unsigned char*)g.X + (size_t)grp * Mg * g.ldx * 2;
  if (g.dY_lo != nullptr) g.dY_lo = (const unsigned char*)g.dY_lo + (size_t)grp * Mg * g.ldy * 2;
  if (g.X_lo != nullptr) g.X_lo = (const unsigned char*)g.X_lo + (size_t)grp * Mg * g.ldx * 2;
  g.dW = (float*)((unsigned char*)g.dW + (long long)grp * g.dw_gstride);
  if (g.db != nullptr) g.db = (float*)((unsigned char*)g.db + (long long)grp * g.dw_gstride);
  g.M = Mg;
}
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(const TfWgradArgs g_in) {
  TfWgradArgs g = g_in;
  enter_wgrad_group(g);
  // K (= rows of M) advances in 32-row steps through a 4-slot LDS ring (16 KiB per slot: dY [32][128] | X [32][128]);
  // the DMA of step i+3 is issued in phase i, so a transfer has three phases to land (with one-step prefetch the loop
  // ran at DMA latency: ~2500 cycles per 64 rows against ~500 cycles of MFMA).  Counted vmcnt + raw s_barrier.
  constexpr int STEP = 32, SLOT = 2 * STEP * 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_k = (g.K + 127) / 128;
  const int tiles = tiles_k * ((g.N + 127) / 128);
  // XCD-aware order: each XCD gets a contiguous range of (split, tile) work items, k-tile fastest, so the blocks that
  // share an L2 walk the SAME rows of dY / X together (measured before: 416 MB fetched per launch for 139 MB of operands,
  // every XCD streaming the whole chunk)
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = logical % tiles, split = logical / tiles;
  const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
  const int m_begin = split * g.m_chunk;
  const int m_end = min(g.M, m_begin + g.m_chunk);
  const int nsteps0 = (m_end - m_begin + STEP - 1) / STEP;
  const int nsteps = SPLIT ? 3 * nsteps0 : nsteps0;
  const unsigned char* __restrict__ Z = (const unsigned char*)g.zeros;

  // one step = 32 rows x 256 B per operand = 8 wave-instructions per operand, 2 per wave: rows R0..R0+3, R0 = (i*4+wave)*4
  const int lr = lane >> 4, lc = lane & 15;
  auto stage = [&](int slot, int step) {
    const unsigned char* __restrict__ dY = (const unsigned char*)g.dY;
    const unsigned char* __restrict__ X = (const unsigned char*)g.X;
    int st = step;
    if constexpr (SPLIT) {
      const int seg = step >= 2 * nsteps0 ? 2 : (step >= nsteps0 ? 1 : 0);
      st = step - seg * nsteps0;
      if (seg == 1) dY = (const unsigned char*)g.dY_lo;
      if (seg == 2) X = (const unsigned char*)g.X_lo;
    }
    unsigned char* ybase = smem + slot * SLOT;
    unsigned char* xbase = ybase + STEP * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int R0 = (i * 4 + wave) * 4;
      const int r = R0 + lr;
      const int c = lc ^ ((r & 3) << 2);
      const int gm = m_begin + st * STEP + r;
      const bool ok = gm < m_end;
      const int cn = min(n0 + c * 8, g.N - 8), ck = min(k0 + c * 8, g.K - 8);   // clamp: never stored
      const unsigned char* sy = ok ? dY + ((size_t)gm * g.ldy + cn) * 2 : Z + c * 16;
      const unsigned char* sx = ok ? X + ((size_t)gm * g.ldx + ck) * 2 : Z + c * 16;
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
  // (2*tiles_k)-th 16-row step, so no block carries more than ~1/(2*tiles_k) extra MFMA work.
  const bool has_bias = g.db != nullptr;
  const int bias_mod = 2 * tiles_k;
  int bias_cnt = 2 * (tile % tiles_k) + wc;                    // counts down to this wave's next duty step
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  // tr-read lane geometry
  const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int h = grp >> 1, cb = grp & 1;
  const int sw = q << 2, o8 = (p & 1) * 8;
  // per-lane offsets inside a slot for (ms, b): row0 = ms*16 + 8h + q (row1 = row0 + 4), chunk = w*8 + b*4 + cb*2 + (p>>1)
  const int yoff = (8 * h + q) * 256 + (((wr * 8 + cb * 2 + (p >> 1)) ^ sw) << 4) + o8;
  const int xoff = STEP * 256 + (8 * h + q) * 256 + (((wc * 8 + cb * 2 + (p >> 1)) ^ sw) << 4) + o8;

  const unsigned lds_base = lds_addr_of(smem);
  if (nsteps > 0) {
    stage(0, 0);
    stage(1, min(1, nsteps - 1));
    stage(2, min(2, nsteps - 1));
  }
  for (int st = 0; st < nsteps; ++st) {
    // retire step st's transfers (issued three phases ago; 4 per wave per step), make them visible, recycle slot (st+3)&3
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    const unsigned sl = lds_base + (st & 3) * SLOT;
    // 16 transposed reads of this step, issued from inline asm (see tf_common.h): ya/yb = dY columns b = 0 / 1, xa/xb = X
    u64 y[2][2][2], x[2][2][2];                               // [ms][b][row half]
#define TF_TR(ms, b)                                                                           \
    y[ms][b][0] = tr_read_asm<(ms) * 4096 + 0>(sl + (yoff ^ ((b) * 64)));                      \
    y[ms][b][1] = tr_read_asm<(ms) * 4096 + 1024>(sl + (yoff ^ ((b) * 64)));                   \
    x[ms][b][0] = tr_read_asm<(ms) * 4096 + 0>(sl + (xoff ^ ((b) * 64)));                      \
    x[ms][b][1] = tr_read_asm<(ms) * 4096 + 1024>(sl + (xoff ^ ((b) * 64)));
    TF_TR(0, 0) TF_TR(0, 1) TF_TR(1, 0) TF_TR(1, 1)
#undef TF_TR
    // (after the reads: hipcc orders an LDS-DMA before any LATER LDS read with a vmcnt(0), it cannot tell the slots apart)
    stage((st + 3) & 3, min(st + 3, nsteps - 1));
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(y[0][0][0]), "+v"(y[0][0][1]), "+v"(y[0][1][0]), "+v"(y[0][1][1]), "+v"(y[1][0][0]), "+v"(y[1][0][1]),
                   "+v"(y[1][1][0]), "+v"(y[1][1][1]), "+v"(x[0][0][0]), "+v"(x[0][0][1]), "+v"(x[0][1][0]), "+v"(x[0][1][1]),
                   "+v"(x[1][0][0]), "+v"(x[1][0][1]), "+v"(x[1][1][0]), "+v"(x[1][1][1]));
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 af[2][2], bfr[2][2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int b = 0; b < 2; ++b) { af[ms][b] = join_tr64(y[ms][b][0], y[ms][b][1]); bfr[ms][b] = join_tr64(x[ms][b][0], x[ms][b][1]); }
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ms][nb], bfr[ms][kb], acc[nb][kb], 0, 0, 0);
      }
      const bool my_turn = has_bias && bias_cnt == 0 && (!SPLIT || st < 2 * nsteps0);             // wave-uniform
      bias_cnt = bias_cnt == 0 ? bias_mod - 1 : bias_cnt - 1;
      if (my_turn) {
        accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ms][0], ones, accb[0], 0, 0, 0);
        accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ms][1], ones, accb[1], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // surplus transfers land before the workgroup retires

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


// ------------------------------------------------------------------------------------------------
// wgrad, 256(n) x 128(k) tile: 4 waves (2x2) of 128x64 = 4x2 v_mfma_f32_32x32x16_bf16 per 16-row substep.
// Why: with the 64x64 wave tile of wgrad_tn_kernel every MFMA needs two operand fragments = 1 KiB of LDS reads, i.e.
// 128 B/clk per CU at MFMA peak -- exactly the LDS limit.  A 128x64 wave tile reads 6 fragments per 8 MFMAs (0.75 KiB
// per MFMA).  Slot = dY [32][256] (512-B rows) | X [32][128] (256-B rows) = 24 KiB, 3-slot ring = 72 KiB -> two
// workgroups per CU.  Same swizzle rule for both row widths: 16-B chunk ^= (row & 3) << 2 (bits 2-3 of the chunk index).
// Bias grad: ONE extra accumulator for all four n-blocks -- D += SEL_nb . dYfrag_nb with SEL_nb[i][m] = (i >> 3 == nb),
// so rows 8nb..8nb+7 of the 32x32 result (= accumulator registers 4nb..4nb+3) hold the column sums of n-block nb.
// ------------------------------------------------------------------------------------------------
// NS = ring slots (3; 2 = 48 KiB: the workgroup then fits on a CU next to an attention-backward workgroup of the chain -- experiment)
template <bool SPLIT, int NS = 3>        // see wgrad_tn_kernel
__global__ __launch_bounds__(256, 2) void wgrad_tn2_kernel(const TfWgradArgs g_in) {
  TfWgradArgs g = g_in;
  enter_wgrad_group(g);
  constexpr int STEP = 32, YB = STEP * 512, XB = STEP * 256, SLOT = YB + XB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_k = (g.K + 127) / 128;
  const int tiles = tiles_k * ((g.N + 255) / 256);
  const int logical = xcd_remap(blockIdx.x, gridDim.x);       // an XCD owns a contiguous (split, tile) range, k-tile fastest
  const int tile = logical % tiles, split = logical / tiles;
  const int n0 = (tile / tiles_k) * 256, k0 = (tile % tiles_k) * 128;
  const int m_begin = split * g.m_chunk;
  const int m_end = min(g.M, m_begin + g.m_chunk);
  const int nsteps0 = (m_end - m_begin + STEP - 1) / STEP;
  const int nsteps = SPLIT ? 3 * nsteps0 : nsteps0;
  const unsigned char* __restrict__ dY = (const unsigned char*)g.dY;
  const unsigned char* __restrict__ X = (const unsigned char*)g.X;
  const unsigned char* __restrict__ Z = (const unsigned char*)g.zeros;
  // byte distance from a hi plane to its lo plane (wave-uniform; planes share the leading dimension)
  const ptrdiff_t lo_y = SPLIT ? (const unsigned char*)g.dY_lo - dY : 0, lo_x = SPLIT ? (const unsigned char*)g.X_lo - X : 0;

  // one step = 16 KiB of dY (16 wave-instructions of 2 rows) + 8 KiB of X (8 of 4 rows): 6 per wave.
  // Per-lane source pointers of the six pieces at step 0 are computed ONCE (columns clamped, swizzle folded in); a step only
  // adds a wave-uniform byte stride.  (Recomputing row index, clamps, the 64-bit multiply-add and the zero-source select for
  // every piece of every step was ~57 VALU instructions per 16 MFMAs.)  Only a ragged final step (rows past m_end must read
  // zeros) takes the general path.
  const unsigned char* py[4];
  const unsigned char* px[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = i * 4 + wave;
    const int r = 2 * j + (lane >> 5);
    const int c = (lane & 31) ^ ((r & 3) << 2);
    const int cn = min(n0 + c * 8, g.N - 8);                                     // clamp: never stored
    py[i] = dY + ((size_t)min(m_begin + r, g.M - 1) * g.ldy + cn) * 2;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = i * 4 + wave;
    const int r = 4 * j + (lane >> 4);
    const int c = (lane & 15) ^ ((r & 3) << 2);
    const int ck = min(k0 + c * 8, g.K - 8);
    px[i] = X + ((size_t)min(m_begin + r, g.M - 1) * g.ldx + ck) * 2;
  }
  const size_t step_y = (size_t)STEP * g.ldy * 2, step_x = (size_t)STEP * g.ldx * 2;
  const bool ragged = ((m_end - m_begin) % STEP) != 0;        // block-uniform; false whenever M is a multiple of 32
  auto stage_general = [&](int slot, int st, ptrdiff_t py_off, ptrdiff_t px_off) {
    unsigned char* ybase = smem + slot * SLOT;
    unsigned char* xbase = ybase + YB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = i * 4 + wave;
      const int r = 2 * j + (lane >> 5);
      const int c = (lane & 31) ^ ((r & 3) << 2);
      const int gm = m_begin + st * STEP + r;
      const int cn = min(n0 + c * 8, g.N - 8);
      const unsigned char* sy = gm < m_end ? dY + py_off + ((size_t)gm * g.ldy + cn) * 2 : Z + (c & 15) * 16;
      glds16(sy, ybase + j * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = i * 4 + wave;
      const int r = 4 * j + (lane >> 4);
      const int c = (lane & 15) ^ ((r & 3) << 2);
      const int gm = m_begin + st * STEP + r;
      const int ck = min(k0 + c * 8, g.K - 8);
      const unsigned char* sx = gm < m_end ? X + px_off + ((size_t)gm * g.ldx + ck) * 2 : Z + c * 16;
      glds16(sx, xbase + j * 1024);
    }
  };
  auto stage = [&](int slot, int step) {
    int st = step;
    ptrdiff_t py_off = 0, px_off = 0;
    if constexpr (SPLIT) {
      const int seg = step >= 2 * nsteps0 ? 2 : (step >= nsteps0 ? 1 : 0);
      st = step - seg * nsteps0;
      py_off = seg == 1 ? lo_y : 0;
      px_off = seg == 2 ? lo_x : 0;
    }
    if (ragged && st == nsteps0 - 1) { stage_general(slot, st, py_off, px_off); return; }
    unsigned char* ybase = smem + slot * SLOT;
    unsigned char* xbase = ybase + YB;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(py[i] + py_off + (size_t)st * step_y, ybase + (i * 4 + wave) * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(px[i] + px_off + (size_t)st * step_x, xbase + (i * 4 + wave) * 1024);
  };

  f32x16 acc[4][2], accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    accb[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; }
  }
  const bool has_bias = g.db != nullptr;
  const int bias_mod = 2 * tiles_k;                            // k-tile blocks x wave columns share one n-range: take turns
  int bias_cnt = 2 * (tile % tiles_k) + wc;
  // (the selector operands of the bias MFMAs -- ones in rows 8 nb .. 8 nb + 7 -- are rebuilt where they are used, on the few steps
  // in which this wave has the bias duty: kept resident, their 16 registers pushed the kernel one register past its 256 and into scratch)
  const int sel_nb = (lane & 31) >> 3;

  const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int h = grp >> 1, cb = grp & 1;
  const int sw = q << 2, o8 = (p & 1) * 8;
  const int yoff = (8 * h + q) * 512 + (((wr * 16 + cb * 2 + (p >> 1)) ^ sw) << 4) + o8;        // n-block nb: ^ (nb * 64)
  const int xoff = YB + (8 * h + q) * 256 + (((wc * 8 + cb * 2 + (p >> 1)) ^ sw) << 4) + o8;    // k-block kb: ^ (kb * 64)

  const unsigned lds_base = lds_addr_of(smem);
  if (nsteps > 0) {
    stage(0, 0);
    if constexpr (NS == 3) stage(1, min(1, nsteps - 1));
  }
  int slot = 0;
  for (int st = 0; st < nsteps; ++st) {
    // retire step st's transfers (issued NS - 1 phases ago; 6 per wave per step), make them visible, recycle slot (st+NS-1)%NS
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * 6) : "memory");
    const unsigned sl = lds_base + slot * SLOT;
    u64 y0[4][2], x0[2][2], y1[4][2], x1[2][2];               // [block][row half] of substep 0 / 1
#define TF_RD(Y, Xv, MS)                                                                                  \
    Y[0][0] = tr_read_asm<(MS) * 8192>(sl + yoff);            Y[0][1] = tr_read_asm<(MS) * 8192 + 2048>(sl + yoff);            \
    Xv[0][0] = tr_read_asm<(MS) * 4096>(sl + xoff);           Xv[0][1] = tr_read_asm<(MS) * 4096 + 1024>(sl + xoff);           \
    Y[1][0] = tr_read_asm<(MS) * 8192>(sl + (yoff ^ 64));     Y[1][1] = tr_read_asm<(MS) * 8192 + 2048>(sl + (yoff ^ 64));     \
    Xv[1][0] = tr_read_asm<(MS) * 4096>(sl + (xoff ^ 64));    Xv[1][1] = tr_read_asm<(MS) * 4096 + 1024>(sl + (xoff ^ 64));    \
    Y[2][0] = tr_read_asm<(MS) * 8192>(sl + (yoff ^ 128));    Y[2][1] = tr_read_asm<(MS) * 8192 + 2048>(sl + (yoff ^ 128));    \
    Y[3][0] = tr_read_asm<(MS) * 8192>(sl + (yoff ^ 192));    Y[3][1] = tr_read_asm<(MS) * 8192 + 2048>(sl + (yoff ^ 192));
    TF_RD(y0, x0, 0)
    TF_RD(y1, x1, 1)
#undef TF_RD
    // (after the reads: hipcc orders an LDS-DMA before any LATER LDS read with a vmcnt(0), it cannot tell the slots apart)
    const int nslot = slot == 0 ? NS - 1 : slot - 1;           // (st + NS - 1) % NS given slot = st % NS
    stage(nslot, min(st + NS - 1, nsteps - 1));
    slot = slot == NS - 1 ? 0 : slot + 1;
    auto substep = [&](u64 (&Y)[4][2], u64 (&Xv)[2][2]) {
      bf16x8 af[4], bfr[2];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) af[nb] = join_tr64(Y[nb][0], Y[nb][1]);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) bfr[kb] = join_tr64(Xv[kb][0], Xv[kb][1]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[nb], bfr[kb], acc[nb][kb], 0, 0, 0);
      const bool my_turn = has_bias && bias_cnt == 0 && (!SPLIT || st < 2 * nsteps0);          // wave-uniform
      bias_cnt = bias_cnt == 0 ? bias_mod - 1 : bias_cnt - 1;
      if (my_turn) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const unsigned w = sel_nb == nb ? 0x3F803F80u : 0u;          // two bf16 ones
          const u32x4 sv = {w, w, w, w};
          accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(sv), af[nb], accb, 0, 0, 0);
        }
      }
    };
    // substep 0 may start when its 12 reads are back (the 12 of substep 1 still in flight)
    asm volatile("s_waitcnt lgkmcnt(12)"
                 : "+v"(y0[0][0]), "+v"(y0[0][1]), "+v"(y0[1][0]), "+v"(y0[1][1]), "+v"(y0[2][0]), "+v"(y0[2][1]), "+v"(y0[3][0]),
                   "+v"(y0[3][1]), "+v"(x0[0][0]), "+v"(x0[0][1]), "+v"(x0[1][0]), "+v"(x0[1][1]));
    __builtin_amdgcn_sched_barrier(0);
    substep(y0, x0);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(y1[0][0]), "+v"(y1[0][1]), "+v"(y1[1][0]), "+v"(y1[1][1]), "+v"(y1[2][0]), "+v"(y1[2][1]), "+v"(y1[3][0]),
                   "+v"(y1[3][1]), "+v"(x1[0][0]), "+v"(x1[0][1]), "+v"(x1[1][0]), "+v"(x1[1][1]));
    __builtin_amdgcn_sched_barrier(0);
    substep(y1, x1);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // surplus transfers land before the workgroup retires

  // ---- epilogue: fp32 atomics into the (unpadded) parameter-layout gradient ----
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int np = n0 + wr * 128 + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int ng = np / g.rgp, ne = np - ng * g.rgp;
      const int ns = ng * g.rg + ne;
      const bool nok = (np < g.N) && (ne < g.rg) && (ns < g.n_src);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int kp = k0 + wc * 64 + kb * 32 + (lane & 31);
        const int kg = kp / g.cgp, ke = kp - kg * g.cgp;
        const int ks = kg * g.cg + ke;
        if (nok && kp < g.K && ke < g.cg && ks < g.k_src) atomicAdd(g.dW + (size_t)ns * g.lddw + ks, acc[nb][kb][r]);
      }
    }
    if (has_bias && lane < 32) {                                 // column sums of n-block nb: accumulator rows 8nb.. = register 4nb
      const int np = n0 + wr * 128 + nb * 32 + lane;
      const int ng = np / g.rgp, ne = np - ng * g.rgp;
      const int ns = ng * g.rg + ne;
      if (np < g.N && ne < g.rg && ns < g.n_src) atomicAdd(g.db + ns, accb[4 * nb]);
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers (host)
// ------------------------------------------------------------------------------------------------
namespace {
template <int MI, int BK, bool SPLIT = false, int NSLOT = 2> int launch_gemm_mi(const TfGemmArgs* a, hipStream_t stream) {
  constexpr int BM = 32 * MI;
  const int tiles = (int)row_tiles(a, BM) * ((a->N + BN - 1) / BN);
  size_t lds = NSLOT * (size_t)(BM + 128) * BK * 2;                // BK 64, two slots: 64 / 72 / 80 KiB
  if (lds < (size_t)BM * CT_STRIDE) lds = (size_t)BM * CT_STRIDE;  // never below the C tile of BM x 272 B
  dim3 grid(tiles), block(256);
#define TF_GEMM_CASE(E)                                                                                           \
  case E: {                                                                                                       \
    static bool attr_set = false;                                                                                 \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<E, MI, BK, SPLIT, NSLOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; } \
    hipLaunchKernelGGL((gemm_nt_kernel<E, MI, BK, SPLIT, NSLOT>), grid, block, lds, stream, *a);                  \
  } break;
  switch (a->epilogue) {
    TF_GEMM_CASE(TF_EPI_NONE)
    TF_GEMM_CASE(TF_EPI_BIAS)
    TF_GEMM_CASE(TF_EPI_BIAS_GELU_DROP)
    TF_GEMM_CASE(TF_EPI_BIAS_DROP_RES)
    TF_GEMM_CASE(TF_EPI_ADD)
    TF_GEMM_CASE(TF_EPI_DGELU_DROP)
    TF_GEMM_CASE(TF_EPI_BIAS_GELU_DROP_G)
    TF_GEMM_CASE(TF_EPI_MUL)
    default: return -4;
  }
#undef TF_GEMM_CASE
  return (int)hipGetLastError();
}
int num_cus();
// tile-height choice: minimise rounds(over 2 workgroups x #CUs) x (height + per-tile fixed cost, ~2 row blocks' worth); ties go to
// the taller tile.  64- and 96-row tiles (MI = 2, 3) exist for SMALL row counts -- the reference's own per-GPU batch of 4 - 5 samples is
// ~2,000 - 3,500 token rows, where a 128-row tiling of an N = 768 GEMM is ~100 - 170 tiles for 512 slots.
int plan_cus();
int pick_mi(int M, int N, int G = 1, const int* gr = nullptr) {
  const int slots = 2 * plan_cus();
  const int tn = (N + BN - 1) / BN;
  static const int env_lo = TF_ENV_INT("TF_GEMM_MI_MIN", 2);       // experiment switch
  // half-filled chips: the time is one tile's, so the shortest tile that still leaves every CU at most one workgroup wins
  int best = 6; double best_cost = -1.0;
  for (int mi = 6; mi >= (env_lo < 2 ? 2 : env_lo); --mi) {
    const long tiles = row_tiles(M, G, 32 * mi, gr) * tn;
    const long rounds = (tiles + slots - 1) / slots;
    // two workgroups share a CU: a round in which at most half the slots are taken runs each workgroup alone on its CU (~1.6x as fast)
    const double share = (tiles - (rounds - 1) * slots) * 2 <= slots ? 0.62 : 1.0;
    const double cost = ((rounds - 1) + share) * (mi + 2);
    if (best_cost < 0 || cost < best_cost - 1e-9) { best = mi; best_cost = cost; }
  }
  return best;
}
}  // namespace

namespace {
template <int MF, bool SPLIT = false> int launch_gemm_big(const TfGemmArgs* a, hipStream_t stream) {
  constexpr int BM = 32 * MF;
  const int tiles = (int)row_tiles(a, BM) * ((a->N + BIG_BN - 1) / BIG_BN);
  size_t lds = 4 * (size_t)(BM + BIG_BN) * BIG_ROWB;               // 4-slot ring
  if (lds < (size_t)BM * BIG_CT_STRIDE) lds = (size_t)BM * BIG_CT_STRIDE;
  dim3 grid(tiles), block(512);
#define TF_GEMM_CASE(E)                                                                                           \
  case E: {                                                                                                       \
    static bool attr_set = false;                                                                                 \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<E, MF, false, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; } \
    hipLaunchKernelGGL((gemm_nt_big_kernel<E, MF, false, SPLIT>), grid, block, lds, stream, *a, 0, 0, 0);         \
  } break;
#define TF_GEMM_CASE8(E)                                                                                          \
  case E: {                                                                                                       \
    static bool attr_set = false;                                                                                 \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<E, MF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; } \
    hipLaunchKernelGGL((gemm_nt_big_kernel<E, MF, true>), grid, block, lds, stream, *a, 0, 0, 0);                 \
  } break;
  if constexpr (!SPLIT && MF == 8) if (a->fp8) {          // fp8 operands: the 256-row tile (the 32-byte fragments of the 128-deep MFMA leave no room for a ninth row block)
    switch (a->epilogue) {
      TF_GEMM_CASE8(TF_EPI_NONE)
      TF_GEMM_CASE8(TF_EPI_BIAS)
      TF_GEMM_CASE8(TF_EPI_BIAS_DROP_RES)
      TF_GEMM_CASE8(TF_EPI_BIAS_GELU_DROP_G)
      default: return -4;
    }
    return (int)hipGetLastError();
  }
#undef TF_GEMM_CASE8
  switch (a->epilogue) {
    TF_GEMM_CASE(TF_EPI_NONE)
    TF_GEMM_CASE(TF_EPI_BIAS)
    TF_GEMM_CASE(TF_EPI_BIAS_GELU_DROP)
    TF_GEMM_CASE(TF_EPI_BIAS_DROP_RES)
    TF_GEMM_CASE(TF_EPI_ADD)
    TF_GEMM_CASE(TF_EPI_DGELU_DROP)
    TF_GEMM_CASE(TF_EPI_BIAS_GELU_DROP_G)
    TF_GEMM_CASE(TF_EPI_MUL)
    default: return -4;
  }
#undef TF_GEMM_CASE
  return (int)hipGetLastError();
}
int num_cus();
// two-per-CU form (NWR = 1): 144 x 256 tiles (MF = 9), 3-slot ring; the second-slot workgroups of the first round start late
int launch_gemm_duo(const TfGemmArgs* a, hipStream_t stream, int stagger_ticks) {
  constexpr int MF = 9, BM = 16 * MF;
  const int tiles = (int)row_tiles(a, BM) * ((a->N + BIG_BN - 1) / BIG_BN);
  size_t lds = 3 * (size_t)(BM + BIG_BN) * BIG_ROWB;               // 76.8 KB: two workgroups per CU
  static_assert((size_t)BM * BIG_CT_STRIDE <= 3 * (size_t)(BM + BIG_BN) * BIG_ROWB, "the C tile must fit the ring");
  dim3 grid(tiles), block(256);
  const int ncu = num_cus();
#define TF_GEMM_CASE(E)                                                                                           \
  case E: {                                                                                                       \
    static bool attr_set = false;                                                                                 \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<E, MF, false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; } \
    hipLaunchKernelGGL((gemm_nt_big_kernel<E, MF, false, false, 1>), grid, block, lds, stream, *a, ncu, 2 * ncu, stagger_ticks); \
  } break;
  switch (a->epilogue) {
    TF_GEMM_CASE(TF_EPI_NONE)
    TF_GEMM_CASE(TF_EPI_BIAS)
    TF_GEMM_CASE(TF_EPI_BIAS_GELU_DROP)
    TF_GEMM_CASE(TF_EPI_BIAS_DROP_RES)
    TF_GEMM_CASE(TF_EPI_ADD)
    TF_GEMM_CASE(TF_EPI_DGELU_DROP)
    TF_GEMM_CASE(TF_EPI_BIAS_GELU_DROP_G)
    TF_GEMM_CASE(TF_EPI_MUL)
    default: return -4;
  }
#undef TF_GEMM_CASE
  return (int)hipGetLastError();
}
// How many independent launch sequences share the chip right now (the wrapper's feature levels run on their own streams): the tile
// choice then plans for its SHARE of the CUs -- with four levels side by side the chip is full anyway, and the short tiles that fill an
// empty chip at small row counts only cost efficiency (wrapper at B = 4: 6.68 ms with 128-row tiles against 7.11 with 64-row ones).
std::atomic<int> g_gemm_concurrency{1};
int plan_cus() { const int c = g_gemm_concurrency.load(std::memory_order_relaxed); const int n = num_cus() / (c < 1 ? 1 : c); return n < 8 ? 8 : n; }
int num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
  }
  return cus;
}
// large tile: 160 .. 288 rows (MF = 5 .. 9), one workgroup per CU: minimise rounds x (height + the per-tile fixed cost, ~3 row
// blocks' worth: fill, the C-tile burst).  The row count is whatever the batch's real tokens add up to (packed batches), so
// the height that makes the tile grid an exact number of rounds changes from step to step; fp8 / fp32-accuracy operands keep 8 / 9.
int pick_mf(int M, int N, int mf_lo = 5, int G = 1, const int* gr = nullptr) {
  const int tn = (N + BIG_BN - 1) / BIG_BN, slots = plan_cus();
  static const int env_lo = TF_ENV_INT("TF_GEMM_MF_MIN", 0);      // experiment switch
  if (env_lo > mf_lo) mf_lo = env_lo > 9 ? 9 : env_lo;
  int best = 9; long best_cost = -1;
  for (int mf = 9; mf >= mf_lo; --mf) {                 // ties go to the taller tile (fewer W re-stagings)
    const long tiles = row_tiles(M, G, 32 * mf, gr) * tn;
    const long cost = ((tiles + slots - 1) / slots) * (mf + 3);
    if (best_cost < 0 || cost < best_cost) { best = mf; best_cost = cost; }
  }
  return best;
}
template <bool SPLIT> int launch_gemm_big_mf(int mf, const TfGemmArgs* a, hipStream_t stream) {
  if constexpr (SPLIT) {
    switch (mf) {                                        // (the fp32-accuracy mode plans its tile height like the bf16 one since round 4)
      case 5: return launch_gemm_big<5, true>(a, stream);
      case 6: return launch_gemm_big<6, true>(a, stream);
      case 7: return launch_gemm_big<7, true>(a, stream);
      case 8: return launch_gemm_big<8, true>(a, stream);
      default: return launch_gemm_big<9, true>(a, stream);
    }
  }
  else {
    switch (mf) {
      case 5: return launch_gemm_big<5>(a, stream);
      case 6: return launch_gemm_big<6>(a, stream);
      case 7: return launch_gemm_big<7>(a, stream);
      case 8: return launch_gemm_big<8>(a, stream);
      default: return launch_gemm_big<9>(a, stream);
    }
  }
}
}  // namespace

TF_TU_SET_CLOCK(tf_tu_set_clock_gemm)
extern "C" void tf_set_gemm_concurrency(int n) { g_gemm_concurrency.store(n < 1 ? 1 : n, std::memory_order_relaxed); }

extern "C" int tf_launch_gemm_nt(const TfGemmArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->N <= 0) return 0;
  if (a->groups > 1 && a->group_rows[0] > 0) { if (!tf_ragged_ok(a->group_rows, a->groups, a->M)) return -7; }   // ragged row ranges
  else if (a->groups > 1 && (a->M % a->groups) != 0) return -7;     // equal row ranges
  if (a->K <= 0 || a->K % 64 != 0 || a->N % 8 != 0) return -2;
  if ((a->lda % 8) || (a->ldw % 8) || (a->ldc % 8)) return -3;
  static const int big = TF_ENV_INT("TF_GEMM_BIG", 1);
  const double fl = 2.0 * a->M * a->N * a->K;          // algorithmic (the fp32-accuracy mode executes three bf16 passes of it)
  const bool split = a->A_lo != nullptr;
  if (split) {
    if (a->fp8 || a->W_lo == nullptr || (a->C_lo == nullptr && !a->c_is_f32)) return -6;
    if (a->c_is_f32 && a->epilogue != TF_EPI_NONE && a->epilogue != TF_EPI_BIAS) return -6;
    const int e = a->epilogue;
    const bool needs_r = e == TF_EPI_MUL || e == TF_EPI_BIAS_DROP_RES || e == TF_EPI_ADD || e == TF_EPI_DGELU_DROP;
    if (needs_r && (a->R == nullptr || a->R_lo == nullptr)) return -6;
    if ((e == TF_EPI_BIAS_GELU_DROP || e == TF_EPI_BIAS_GELU_DROP_G) && (a->C2 == nullptr || a->C2_lo == nullptr)) return -6;
  }
  if (a->fp8) {                                   // fp8 operands: large-tile kernel only (any shape; rows are clamped)
    if ((a->lda % 16) || (a->ldw % 16)) return -3;
    char nm[56];
    snprintf(nm, sizeof(nm), "gemm_nt_big_kernel<%d, 8, fp8>", a->epilogue);
    TfTraceScope tr(nm, stream, fl);
    return launch_gemm_big<8>(a, stream);
  }
  // Which kernel: the large tile (288/256 x 256, one workgroup per CU) is ~1.8x as efficient per CU as the 128-wide one (two
  // per CU) once the chip is full, but at small M its grid is a fraction of a round -- M = 5664, N = 768 is 69 tiles for 256
  // CUs, 76 us for 20 GFLOP.  Estimated time = rounds x (fixed + per-K cost of one tile), constants from the K-sweeps on
  // MI355X (large: 13.3 us + 26.4 us per 1000 K; 128-wide at two per CU: ~8 us + 21 us per 1000 K).
  static const int split_mf_lo = TF_ENV_INT("TF_GEMM_SPLIT_MF_MIN", 5);     // experiment switch (8: the two heights the mode had before)
  bool use_big = big && a->M >= 1024 && a->N >= 256;
  if (use_big) {
    const int mfp = pick_mf(a->M, a->N, split ? split_mf_lo : 5, a->groups, a->group_rows);
    const long tb = row_tiles(a, 32 * mfp) * ((a->N + BIG_BN - 1) / BIG_BN);
    const int mi = split ? 4 : pick_mi(a->M, a->N, a->groups, a->group_rows);            // (the fp32-accuracy mode has one tile height of this kernel)
    const long ts = row_tiles(a, 32 * mi) * ((a->N + BN - 1) / BN);
    const int pc = plan_cus();
    const double t_big = (double)((tb + pc - 1) / pc) * (13.3 + 0.0264 * a->K) * (mfp / 9.0);
    const long rs = (ts + 2 * pc - 1) / (2 * pc);
    const double sh = (ts - (rs - 1) * 2 * pc) * 2 <= 2 * pc ? 0.62 : 1.0;         // (see pick_mi)
    const double t_small = ((double)(rs - 1) + sh) * (8.0 + 0.021 * a->K) * ((mi + 2) / 6.0);
    static const int model = TF_ENV_INT("TF_GEMM_MODEL", 1);      // experiment switch
    if (model && t_small < t_big) use_big = false;
  }
  // Two workgroups per CU (144 x 256 tiles) for launches of two or more rounds (QKV, FFN-up, FFN-down dgrad at the benchmark shape):
  // see the kernel's header.  TF_GEMM_DUO=0 turns it off, TF_GEMM_DUO_MIN sets the tile threshold, TF_GEMM_DUO_US the stagger.
  static const int duo = TF_ENV_INT("TF_GEMM_DUO", 1);
  static const int duo_min = TF_ENV_INT("TF_GEMM_DUO_MIN", 0);
  static const double duo_us = TF_ENV_DBL("TF_GEMM_DUO_US", -1.0);
  if (use_big && duo && !split) {
    const long td = row_tiles(a, 144) * ((a->N + BIG_BN - 1) / BIG_BN);
    static const int duo_1r = TF_ENV_INT("TF_GEMM_DUO_1R", 0);   // experiment: epilogue bitmask -> also single-round launches
    const long need = duo_min > 0 ? duo_min : (((duo_1r >> a->epilogue) & 1) ? num_cus() + 1 : 2L * num_cus() + 1);
    if (td >= need) {
      const double us = duo_us >= 0 ? duo_us : 0.0;
      char nm[56];
      snprintf(nm, sizeof(nm), "gemm_nt_duo_kernel<%d, 9>", a->epilogue);
      TfTraceScope tr(nm, stream, fl);
      return launch_gemm_duo(a, stream, (int)(us * 100.0));
    }
  }
  if (use_big) {
    const int mf = pick_mf(a->M, a->N, split ? split_mf_lo : 5, a->groups, a->group_rows);
    char nm[56];
    snprintf(nm, sizeof(nm), split ? "gemm_nt_big_kernel<%d, %d, x3>" : "gemm_nt_big_kernel<%d, %d>", a->epilogue, mf);
    TfTraceScope tr(nm, stream, fl);
    if (split) return launch_gemm_big_mf<true>(mf, a, stream);
    return launch_gemm_big_mf<false>(mf, a, stream);
  }
  if (split) {                                     // one tile height in this mode (fewer instantiations)
    char nm[56];
    snprintf(nm, sizeof(nm), "gemm_nt_kernel<%d, 4, 64, x3>", a->epilogue);
    TfTraceScope tr(nm, stream, fl);
    return launch_gemm_mi<4, 64, true>(a, stream);
  }
  const int mi_sel = pick_mi(a->M, a->N, a->groups, a->group_rows);
  // at most about one workgroup per CU: the 4-slot ring form (see the kernel's header).  TF_GEMM_RING = percent of the CU count up
  // to which a grid takes it (0 = never).
  static const int ring_pct = TF_ENV_INT("TF_GEMM_RING", 100);
  if (ring_pct > 0 && mi_sel <= 5) {
    const long tiles = row_tiles(a, 32 * mi_sel) * ((a->N + BN - 1) / BN);
    if (tiles * 100 <= (long)ring_pct * num_cus()) {
      char nm[56];
      snprintf(nm, sizeof(nm), "gemm_nt_kernel<%d, %d, 64, ring>", a->epilogue, mi_sel);
      TfTraceScope tr(nm, stream, fl);
      static const int ring_slots = TF_ENV_INT("TF_GEMM_RING_SLOTS", 4);
      if (ring_slots == 3) {
        switch (mi_sel) {
          case 2: return launch_gemm_mi<2, 64, false, 3>(a, stream);
          case 3: return launch_gemm_mi<3, 64, false, 3>(a, stream);
          case 4: return launch_gemm_mi<4, 64, false, 3>(a, stream);
          default: return launch_gemm_mi<5, 64, false, 3>(a, stream);
        }
      }
      switch (mi_sel) {
        case 2: return launch_gemm_mi<2, 64, false, 4>(a, stream);
        case 3: return launch_gemm_mi<3, 64, false, 4>(a, stream);
        case 4: return launch_gemm_mi<4, 64, false, 4>(a, stream);
        default: return launch_gemm_mi<5, 64, false, 4>(a, stream);
      }
    }
  }
  char nm[56];
  snprintf(nm, sizeof(nm), "gemm_nt_kernel<%d, %d, 64>", a->epilogue, mi_sel);
  TfTraceScope tr(nm, stream, fl);
  switch (mi_sel) {
    case 2: return launch_gemm_mi<2, 64>(a, stream);
    case 3: return launch_gemm_mi<3, 64>(a, stream);
    case 5: return launch_gemm_mi<5, 64>(a, stream);
    case 6: return launch_gemm_mi<6, 64>(a, stream);
    default: return launch_gemm_mi<4, 64>(a, stream);
  }
}

namespace {
// 1: 128x128 tile, 2: 256x128 tile.  Default: the caller that sizes the M-splits itself (the encoder runtime, whose wgrads
// share the chip with the backward chain on a side stream) gets the 256x128 kernel -- same-box whole-step A/B 5056 -> 5150
// samples/s; a wgrad that has the chip to itself needs more, smaller blocks to keep two workgroups on every CU (isolated:
// qkv 121 us with 128x128 vs 161 us with 256x128) and keeps the 128x128 kernel.
int wgrad_version(bool caller_sized) {
  static const int v = TF_ENV_INT("TF_WGRAD_V", 0);
  return v ? v : (caller_sized ? 2 : 1);
}
}  // namespace
// number of output tiles of the active wgrad kernel (the encoder runtime sizes its M-splits from it)
extern "C" int tf_wgrad_tiles(int N, int K, int caller_sized) {
  return wgrad_version(caller_sized != 0) == 2 ? ((N + 255) / 256) * ((K + 127) / 128) : ((N + 127) / 128) * ((K + 127) / 128);
}

extern "C" int tf_launch_wgrad_tn(const TfWgradArgs* a_in, hipStream_t stream) {
  TfWgradArgs a = *a_in;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return 0;
  const int ngroups = a.groups > 1 ? a.groups : 1;
  if (a.M % ngroups || a.group_rows[0] > 0) return -7;            // (ragged row ranges: tf_launch_wgrad_multi only)
  const int M_all = a.M;
  a.M = M_all / ngroups;                      // the sizing below is per group; the kernels get the full extent back
  if ((a.N % 8) || (a.K % 8) || (a.ldy % 8) || (a.ldx % 8) || a.zeros == nullptr) return -2;
  if (a.rgp < a.rg || a.cgp < a.cg || a.rg <= 0 || a.cg <= 0) return -3;
  const int v2 = wgrad_version(a.m_chunk > 0) == 2;
  const int tiles = tf_wgrad_tiles(a.N, a.K, a.m_chunk > 0);
  const int steps = (a.M + 31) / 32;
  // Every split adds one full fp32 |dW| of atomic traffic (chip-wide ~1.3 TB/s), so use the FEWEST splits that
  // still give one resident wave of blocks: 256 CUs x 2 blocks.
  static const int env_slots = TF_ENV_INT("TF_WGRAD_SLOTS", 0);   // experiment switches
  static const int env_splits = TF_ENV_INT("TF_WGRAD_SPLITS", 0);
  // Measured on MI355X (M = 22656, 128x128 tiles): the best block count grows with the tile count -- 36 tiles: ~6 splits
  // (216 blocks), 72: 5-6 (360-432), 108: 4 (432) -- i.e. about 200 + 2.15 * tiles blocks, never more than one resident wave.
  int target = env_slots > 0 ? env_slots : (v2 ? 288 : (int)(200 + 2.15 * tiles));
  if (target > 2 * num_cus()) target = 2 * num_cus();
  int splits = a.m_chunk > 0 ? (a.M + a.m_chunk - 1) / a.m_chunk : (target + tiles / 2) / tiles;
  while (a.m_chunk <= 0 && splits > 1 && splits * tiles > 2 * num_cus()) --splits;
  if (env_splits > 0 && a.m_chunk <= 0) splits = env_splits;
  if (splits > steps) splits = steps;
  if (splits < 1) splits = 1;
  if (a.m_chunk <= 0) a.m_chunk = ((steps + splits - 1) / splits) * 32;
  splits = (a.M + a.m_chunk - 1) / a.m_chunk;
  dim3 grid(tiles * splits, ngroups), block(256);
  const double flops = 2.0 * M_all * a.N * a.K;
  a.M = M_all;
  const bool split = a.dY_lo != nullptr;
  if (split && a.X_lo == nullptr) return -6;
  TfTraceScope tr(v2 ? (split ? "wgrad_tn2_kernel<x3>" : "wgrad_tn2_kernel") : (split ? "wgrad_tn_kernel<x3>" : "wgrad_tn_kernel"), stream,
                  flops);
  if (v2) {
    constexpr int LDS2 = 3 * (32 * 512 + 32 * 256);
    static const hipError_t once = hipFuncSetAttribute((const void*)wgrad_tn2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
    static const hipError_t once3 = hipFuncSetAttribute((const void*)wgrad_tn2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
    (void)once; (void)once3;
    static const int ring = TF_ENV_INT("TF_WGRAD_RING", 3);      // experiment switch (2: 48 KiB)
    if (split) hipLaunchKernelGGL(wgrad_tn2_kernel<true>, grid, block, LDS2, stream, a);
    else if (ring == 2) hipLaunchKernelGGL((wgrad_tn2_kernel<false, 2>), grid, block, LDS2 / 3 * 2, stream, a);
    else hipLaunchKernelGGL(wgrad_tn2_kernel<false>, grid, block, LDS2, stream, a);
  } else {
    if (split) hipLaunchKernelGGL(wgrad_tn_kernel<true>, grid, block, 4 * TILE_BYTES, stream, a);
    else hipLaunchKernelGGL(wgrad_tn_kernel<false>, grid, block, 4 * TILE_BYTES, stream, a);
  }
  return (int)hipGetLastError();
}
