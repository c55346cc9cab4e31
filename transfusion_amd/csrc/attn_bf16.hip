// Fused multi-head attention for the fusion block (K4 and its backward), bf16 MFMA on gfx950.
//
// Semantics (torch18_adapters.py:788-799 of the reference): P = softmax(q/sqrt(hd) . k^T + key_padding(-inf)),
// Pd = dropout(P), O = Pd . v.  No [S,S] tensor ever reaches HBM: forward keeps an online softmax
// and stores only LSE[b,h,s]; backward recomputes P from LSE.
//
// Orientation: the score tile is computed TRANSPOSED with v_mfma_f32_32x32x16_bf16, St[key][q] = K . Q^T,
// so the query index lives on the lane (lane & 31) and keys live in the 16 accumulator registers: the
// softmax row reduce is an in-register reduce plus ONE cross-half exchange, every per-row statistic
// (max, sum, alpha, LSE, delta) is a per-lane scalar, and the accumulator tile is already the B operand
// of the next product (O^T += V^T . Pt) with no LDS round trip (register 8s+j of the tile is row
// 16s + 8(j>>2) + 4(lane>>5) + (j&3) of k-step s; the V^T operand is fetched in that k order with
// ds_read_b64_tr_b16).  Backward uses the same trick three times (dQ^T += K^T . dSt ; dV^T += dO^T . Pd ;
// dK^T += Q^T . dS) in two kernels so that no gradient needs cross-workgroup atomics:
//   attn_bwd_dq  : query on the lane, loops over key tiles   -> dQ
//   attn_bwd_dkv : key   on the lane, loops over query tiles -> dK, dV
// LDS tiles are "dual use": row stride == 64 (mod 256) bytes and 16-B chunk ^= (row>>2)&3 make both the
// row reads (ds_read_b128) and the transposed reads (ds_read_b64_tr_b16) bank-conflict free.
#include "tf_common.h"
#include <cstdio>
#include <cstdlib>
#include "tf_kernels.h"
#include "attn_common.h"

// Ablation / variant macros (TF_ABL_FWD, TF_FWD_PRIO, TF_FWD_NO_DMA, TF_ABL_PAIR, TF_DKV16_QT) act in experiments builds only
// (-DTF_EXPERIMENTS: tools/build_variant.sh); without it they are forced off -- no flag changes what the shipped library computes.
#ifndef TF_EXPERIMENTS
#undef TF_FWD_WIDE_STORE
#endif
#ifndef TF_FWD_WIDE_STORE
#define TF_FWD_WIDE_STORE 1      // 16-byte output stores of the forward after a half-wave swap (round 6: 65.0 -> 61.2 us at the benchmark's packed rows)
#endif
#ifndef TF_EXPERIMENTS
#undef TF_DQ_WIDE_STORE
#undef TF_FWD_SKIP_IDLE
#endif
#ifndef TF_FWD_SKIP_IDLE
#define TF_FWD_SKIP_IDLE 1       // forward: waves that own no query of the sample skip their matrix / softmax work (round 6: 60.3 -> 57.2 us packed)
#endif
#ifndef TF_DQ_WIDE_STORE
#define TF_DQ_WIDE_STORE 1       // the same for attn_bwd_dq_ds_kernel's dQ rows, across 16-lane groups (round 6: 38.8 -> 34.9 us)
#endif
#ifndef TF_EXPERIMENTS
#undef TF_ABL_FWD
#undef TF_FWD_PRIO
#undef TF_FWD_NO_DMA
#undef TF_ABL_PAIR
#undef TF_DKV16_QT
#endif

// output-row stores of the forward, the dK / dV pair kernel and the dQ kernel.  TF_NT_ATTN (experiments builds): nontemporal
#if defined(TF_EXPERIMENTS) && defined(TF_NT_ATTN)
#define TF_ST_ROW(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define TF_ST_ROW(ptr, val) (*(ptr) = (val))
#endif

namespace {

// ================================================================================================
// forward
// ================================================================================================
template <int HDP, bool BLK>       // BLK: a block-bit matrix (TfAttnArgs.block_bits) is present
__global__ __launch_bounds__(256, (HDP <= 192 ? 2 : 1)) void attn_fwd_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kt = smem;
  unsigned char* vt = smem + 64 * G::TSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  // 1-D grid, XCD-aware: all query blocks of one (batch, head) -- which share K and V -- run on the same XCD / L2
  const int S = a.S;                                   // keys
  const int Sq = a.q != nullptr ? a.Sq : S;            // queries (their own tensor for cross attention, TfAttnArgs.q)
  const int nqb = (Sq + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nqb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int q0 = (logical % nqb) * 128 + wave * 32;
  // S / Sq stay the (maximum) lengths that index lse / the dropout rows and shape the grid; Sb / Sqb are THIS sample's row counts
  const SampleRows sr = sample_rows(a.cu_rows, b, S);
  const int Sb = sr.len, Sqb = a.q != nullptr ? Sq : Sb;
  if ((logical % nqb) * 128 >= Sqb) return;            // packed batches: query blocks past the sample's end (workgroup-uniform)
  const u16* __restrict__ qkv = (const u16*)a.qkv;
  const size_t ld = a.ld_qkv;
  const size_t ldq = a.q != nullptr ? (size_t)a.ld_q : ld;
  const size_t qrow0 = a.q != nullptr ? (size_t)b * Sq : sr.row0;      // first row of this sample's queries (out / dout rows follow it)
  const u16* qbase = a.q != nullptr ? (const u16*)a.q + qrow0 * ldq + (size_t)head * HDP
                                    : qkv + sr.row0 * ld + (size_t)(0 * a.H + head) * HDP;
  const u16* kbase = qkv + sr.row0 * ld + (size_t)(1 * a.H + head) * HDP;
  const u16* vbase = qkv + sr.row0 * ld + (size_t)(2 * a.H + head) * HDP;

  // Q^T B-operand fragments, resident in registers
  bf16x8 qf[G::KSTEPS];
  {
    const int qr = min(q0 + (lane & 31), Sqb - 1);
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) qf[ks] = as_bf16x8(*(const u32x4*)(qbase + (size_t)qr * ldq + ks * 16 + 8 * h));
  }
  f32x16 o[G::DBLK];
#pragma unroll
  for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  const float sc = a.scale * LOG2E;
  const int qrow = q0 + (lane & 31);
  const bool wave_live = TF_FWD_SKIP_IDLE ? (__builtin_amdgcn_readfirstlane(q0) < Sqb) : true;       // wave-uniform
  const int SW = (S + 63) / 64;
  const unsigned long long* drow = a.drop_thr ? (const unsigned long long*)a.drop_bits + ((size_t)bh * Sq + min(qrow, Sqb - 1)) * SW : nullptr;
  const unsigned long long* brow = BLK ? (const unsigned long long*)a.block_bits + (size_t)min(qrow, Sqb - 1) * SW : nullptr;

  const FragAddr<HDP> fk(lds_addr_of(kt), lane), fv(lds_addr_of(vt), lane);
  const int ntiles = (valid_key_limit(a.key_mask, b, Sb, lane) + 63) / 64;   // trailing all-padding key tiles are skipped
  // the key tiles this workgroup visits: all of them, minus -- with a block mask -- those blocked for every query of the block
  // (TfAttnArgs.block_skip_q: every probability there is exactly 0).  A scalar bit set; t = the tile in hand, tn = the next one.
  const int nts = __builtin_amdgcn_readfirstlane(ntiles);          // (scalar: the tile index is an LDS-DMA's scalar offset)
  unsigned long long act = nts >= 64 ? ~0ull : ((1ull << nts) - 1ull);
  if (BLK && a.block_skip_q != nullptr && nts <= 64) {
    const unsigned long long sk = ((const unsigned long long*)a.block_skip_q)[logical % nqb];
    act &= ~(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sk >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)sk));
  }
  const bool listed = BLK && nts <= 64;                   // (more than 64 tiles: S > 4096, no map; plain counting)
  int t = listed ? (act ? __builtin_ctzll(act) : -1) : (nts > 0 ? 0 : -1);
#ifndef TF_FWD_NO_DMA
  const TileDma<HDP, 4> dma(__builtin_amdgcn_readfirstlane(wave), lane, ld);
  const __amdgpu_buffer_rsrc_t krs = make_rsrc(kbase, ld, Sb, HDP), vrs = make_rsrc(vbase, ld, Sb, HDP);
  const unsigned tile_bytes = 64u * (unsigned)(ld * 2);
  unsigned long long dm_n = ~0ull, blk_n = 0ull;
  unsigned kmb_n = 0;
  auto fetch_words = [&](int tn) {          // raw loads only; they are consumed one tile later
    if (a.drop_thr) dm_n = drow[tn];
    if (BLK) blk_n = brow[tn];
    if (a.key_mask != nullptr) kmb_n = a.key_mask[(size_t)b * S + min(tn * 64 + lane, Sb - 1)];
  };
  if (t >= 0) { fetch_words(t); dma.issue(krs, (unsigned)t * tile_bytes, kt); }
#endif
  while (t >= 0) {
    int tn;
    if (listed) { act &= act - 1ull; tn = act ? __builtin_ctzll(act) : -1; }
    else tn = t + 1 < nts ? t + 1 : -1;
    const int kv0 = t * 64;
    // K / V staging is LDS-DMA (no staging registers: the kernel stays under 256 VGPRs, two workgroups per CU): V(t) travels
    // under S(t) + softmax, K(t+1) under PV(t); each buffer is rewritten only after the barrier that every wave reaches once it
    // has stopped reading it.  TF_FWD_NO_DMA keeps the register-staged form (global -> registers -> LDS between two barriers).
#if defined(TF_ABL_FWD) && (TF_ABL_FWD & 1)
    if (t == 0) {
      TileRegs<64, HDP> kr;
      kr.load(kbase, ld, kv0, Sb - 1, false, tid);
      __syncthreads();
      kr.store(kt, tid);
      kr.load(vbase, ld, kv0, Sb - 1, false, tid);
      kr.store(vt, tid);
      __syncthreads();
    }
    else { __syncthreads(); __syncthreads(); }
#elif !defined(TF_FWD_NO_DMA)
    dma_wait_barrier();                    // K(t) landed (it travelled under PV(t-1)) and visible; every wave has finished PV(t-1)
    dma.issue(vrs, t * tile_bytes, vt);    // V(t) travels under S(t) and the softmax
#else
    {
      TileRegs<64, HDP> kr;
      kr.load(kbase, ld, kv0, Sb - 1, false, tid);
      __syncthreads();                     // previous tile fully consumed
      kr.store(kt, tid);
      kr.load(vbase, ld, kv0, Sb - 1, false, tid);
      kr.store(vt, tid);
    }
    __syncthreads();
#endif
#if !defined(TF_FWD_NO_DMA) && !(defined(TF_ABL_FWD) && (TF_ABL_FWD & 1))
    // (the per-tile words were fetched one tile ahead, BEFORE the K transfer was issued: a load issued after a DMA can only be
    // waited for together with it)
    const unsigned long long dm = dm_n >> (4 * h), blk = blk_n;
    const unsigned long long vall = __ballot(kv0 + lane < Sb && kmb_n == 0);
#else
    const unsigned long long dm = a.drop_thr ? (drow[t] >> (4 * h)) : ~0ull;
    // keys this lane's query may attend: not padded / out of range (wave-uniform ballot) and, with a block mask, not
    // blocked for this query (per lane).  Without a block mask nothing per-lane is computed outside the rare masked tile.
    const unsigned long long vall = key_bits(a.key_mask, b, Sb, kv0, lane);
    const unsigned long long blk = BLK ? brow[t] : 0ull;
#endif
    const bool masked_tile = vall != ~0ull || (BLK && __any(blk != 0ull));
    const unsigned long long vbits = (vall & ~blk) >> (4 * h);

    unsigned pk[2][8];
    // A wave whose 32 queries all lie past the sample's end (packed batches: the last query block of most samples holds a few rows --
    // three of its four waves own none) takes part in the tile transfers and the barriers only: its matrix and vector slots go to the
    // wave of the co-resident workgroup on the same SIMD (TF_FWD_SKIP_IDLE=0: experiments builds keep them computing on clamped rows)
    if (wave_live) {
    // ---- St[key][q] = K . Q^T ----  (two accumulator chains interleaved, fragment reads PF ahead: attn_common.h)
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
    {
#ifdef TF_FWD_PRIO
      __builtin_amdgcn_s_setprio(TF_FWD_PRIO);
#endif
      constexpr int NF = 2 * G::KSTEPS, PF = 3;
      u32x4 fr[NF];
      static_for<PF>([&](auto I) { fr[I] = fk.template row<(I & 1) * 32, (I >> 1)>(); });
      static_for<NF>([&](auto I) {
        constexpr int i = I, nx = i + PF;
#if defined(TF_ABL_FWD) && (TF_ABL_FWD & 16)
        if constexpr (i >= PF) fr[i] = fr[i - PF];
#else
        if constexpr (nx < NF) fr[nx] = fk.template row<(nx & 1) * 32, (nx >> 1)>();
        lgkm_wait<(NF - 1 - i < PF ? NF - 1 - i : PF)>(fr[i]);
#endif
        st[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fr[i]), qf[i >> 1], st[i & 1], 0, 0, 0);
      });
    }
    // ---- online softmax (log2 domain; raw scores stay unscaled, the scale rides in the FMA) ----
#ifdef TF_FWD_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#if !(defined(TF_ABL_FWD) && (TF_ABL_FWD & 4))
    if (masked_tile) {                     // wave-uniform: only tiles that contain padded / out-of-range / blocked keys pay for the select
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!((vbits >> (kb * 32 + (r & 3) + 8 * (r >> 2))) & 1ull)) st[kb][r] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc;
    const bool need = mx - m_run > RESCALE_THR;   // per ROW: a row's arithmetic never depends on its wave-mates
    if (__any(need)) {                            // wave-uniform gate only; rows that keep their max multiply by exactly 1
      const float m_new = need ? fmaxf(m_run, mx) : m_run;
      const float alpha = need ? fast_exp2(m_run - m_new) : 1.0f;
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
    // probabilities two at a time: packed fp32 FMA / add, the keep bit as a sign-extended one-bit field ANDed into the value
    // (v_bfe_i32 + v_and instead of and / compare / select), one v_cvt_pk per pair.  pk[kb][i] = elements 2i, 2i+1 of sub-tile kb.
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 ps2 = {0.f, 0.f};
    const f32x2 sc2 = {sc, sc}, nm2 = {-m_run, -m_run};
    static_for<16>([&](auto I) {
      constexpr int kb = I / 8, r = 2 * (I % 8), b0 = (r & 3) + 8 * (r >> 2);
      const int dword = (int)(unsigned)(dm >> (32 * kb));
      const f32x2 e2 = f32x2{st[kb][r], st[kb][r + 1]} * sc2 + nm2;
      const float p0 = fast_exp2(e2[0]), p1 = fast_exp2(e2[1]);
      ps2 += f32x2{p0, p1};
      pk[kb][r >> 1] = cvt_pk_bf16(and_bit<b0>(p0, dword), and_bit<b0 + 1>(p1, dword));
    });
    l_run += ps2[0] + ps2[1];
#else
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) pk[kb][r >> 1] = cvt_pk_bf16(st[kb][r], st[kb][r + 1]);
#endif
    }   // wave_live
#ifndef TF_FWD_NO_DMA
    dma_wait_barrier();                    // V(t) landed and visible; every wave has finished S(t)
    if (tn >= 0) { fetch_words(tn); dma.issue(krs, (unsigned)tn * tile_bytes, kt); }     // K(next) travels under PV(t)
#endif
    // ---- O^T[d][q] += V^T . Pt ----  (transposed V fragments PF ahead of their MFMA)
    if (wave_live) {
      constexpr int NF = 4 * G::DBLK, PF = 2;
      u64 fa[NF], fb[NF];
      bf16x8 pf[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int kb = g >> 1, s4 = 4 * (g & 1);
        pf[g] = as_bf16x8(u32x4{pk[kb][s4], pk[kb][s4 + 1], pk[kb][s4 + 2], pk[kb][s4 + 3]});
      }
#ifdef TF_FWD_PRIO
      __builtin_amdgcn_s_setprio(TF_FWD_PRIO);
#endif
      static_for<PF>([&](auto I) { fv.template tr<(I / G::DBLK) * 16, (I % G::DBLK) * 32>(fa[I], fb[I]); });
      static_for<NF>([&](auto I) {
        constexpr int i = I, nx = i + PF;
#if defined(TF_ABL_FWD) && (TF_ABL_FWD & 8)
        if constexpr (i >= PF) { fa[i] = fa[i - PF]; fb[i] = fb[i - PF]; }
#else
        if constexpr (nx < NF) fv.template tr<(nx / G::DBLK) * 16, (nx % G::DBLK) * 32>(fa[nx], fb[nx]);
        lgkm_wait<2 * (NF - 1 - i < PF ? NF - 1 - i : PF)>(fa[i], fb[i]);
#endif
        o[i % G::DBLK] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join_tr64(fa[i], fb[i]), pf[i / G::DBLK], o[i % G::DBLK], 0, 0, 0);
      });
#ifdef TF_FWD_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    t = tn;
  }
  // ---- epilogue ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = (a.drop_thr ? a.drop_scale : 1.0f) / l_tot;
#if TF_FWD_WIDE_STORE
  // A lane holds 4 consecutive columns of every 8-column group, its partner (lane ^ 32, the same query) the other 4: as they stand
  // that is 24 8-byte stores per lane at head dim 192, each instruction writing 16 bytes of 32 rows.  The halves swap one group per pair
  // (v_permlane32_swap_b32: lanes 32-63 of the first register <-> lanes 0-31 of the second): the low half keeps the pair's even group,
  // the high half the odd one, whole -- 12 16-byte stores per lane.  (All 64 lanes take part in the swaps; rows past the end only skip
  // their stores.)
  {
    u16* orow = (u16*)a.out + (qrow0 + min(qrow, Sqb - 1)) * a.ld_out + (size_t)head * HDP;
#pragma unroll
    for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int ge = 2 * pr, go = 2 * pr + 1;
        unsigned e0 = pack2bf(o[d][4 * ge] * inv, o[d][4 * ge + 1] * inv), e1 = pack2bf(o[d][4 * ge + 2] * inv, o[d][4 * ge + 3] * inv);
        unsigned o0 = pack2bf(o[d][4 * go] * inv, o[d][4 * go + 1] * inv), o1 = pack2bf(o[d][4 * go + 2] * inv, o[d][4 * go + 3] * inv);
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
        // low half: (own even group, partner's even group); high half: (partner's odd group, own odd group) -- column order in both
        const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
        if (qrow < Sqb) TF_ST_ROW((u32x4*)(orow + d * 32 + 8 * (2 * pr + h)), v);
      }
    if (qrow < Sqb && h == 0 && a.lse != nullptr) a.lse[(size_t)bh * Sq + qrow] = m_run + log2f(l_tot);
  }
#else
  if (qrow < Sqb) {
    u16* orow = (u16*)a.out + (qrow0 + qrow) * a.ld_out + (size_t)head * HDP;
#pragma unroll
    for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        u32x2 v;
        v[0] = pack2bf(o[d][4 * g4] * inv, o[d][4 * g4 + 1] * inv);
        v[1] = pack2bf(o[d][4 * g4 + 2] * inv, o[d][4 * g4 + 3] * inv);
        TF_ST_ROW((u32x2*)(orow + d * 32 + 8 * g4 + 4 * h), v);
      }
    if (h == 0 && a.lse != nullptr) a.lse[(size_t)bh * Sq + qrow] = m_run + log2f(l_tot);
  }
#endif
}

// ================================================================================================
// backward, dQ: query on the lane, loop over key tiles
//   St = K.Q^T -> P = exp2(St*sc - LSE);  dPt = V.dO^T;  dSt = P * (keep/(1-p) * dPt - delta)
//   dQ^T[d][q] += K^T[d][key] . dSt[key][q];  dQ = scale * dQ^T^T
// ================================================================================================
template <int HDP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kt = smem;
  unsigned char* vt = smem + 64 * G::TSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  // 1-D grid, XCD-aware: all query blocks of one (batch, head) -- which share K and V -- run on the same XCD / L2
  const int S = a.S;                                   // keys
  const int Sq = a.q != nullptr ? a.Sq : S;            // queries
  const int nqb = (Sq + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nqb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int q0 = (logical % nqb) * 128 + wave * 32;
  const SampleRows sr = sample_rows(a.cu_rows, b, S);            // packed batches: see attn_fwd_kernel
  const int Sb = sr.len, Sqb = a.q != nullptr ? Sq : Sb;
  if ((logical % nqb) * 128 >= Sqb) return;
  const size_t ld = a.ld_qkv;
  const u16* qkv = (const u16*)a.qkv;
  const size_t ldq = a.q != nullptr ? (size_t)a.ld_q : ld;
  const size_t qrow0 = a.q != nullptr ? (size_t)b * Sq : sr.row0;
  const u16* qbase = a.q != nullptr ? (const u16*)a.q + qrow0 * ldq + (size_t)head * HDP
                                    : qkv + sr.row0 * ld + (size_t)(0 * a.H + head) * HDP;
  const u16* kbase = qkv + sr.row0 * ld + (size_t)(1 * a.H + head) * HDP;
  const u16* vbase = qkv + sr.row0 * ld + (size_t)(2 * a.H + head) * HDP;
  const u16* dobase = (const u16*)a.dout + qrow0 * a.ld_dout + (size_t)head * HDP;

  const int qrow = q0 + (lane & 31);
  const int qr = min(qrow, Sqb - 1);
  bf16x8 qf[G::KSTEPS], dof[G::KSTEPS];
#pragma unroll
  for (int ks = 0; ks < G::KSTEPS; ++ks) {
    qf[ks] = as_bf16x8(*(const u32x4*)(qbase + (size_t)qr * ldq + ks * 16 + 8 * h));
    dof[ks] = as_bf16x8(*(const u32x4*)(dobase + (size_t)qr * a.ld_dout + ks * 16 + 8 * h));
  }
  const float lse = a.lse[(size_t)bh * Sq + qr];
  // delta[q] = rowsum(dO . O): this wave already holds dO in B-operand layout (lane half h owns hd elements 16ks+8h..+7),
  // so read O the same way, reduce in registers + one cross-half exchange, and publish it for the dK/dV kernel
  float delta = 0.f;
  {
    const u16* orow = (const u16*)a.out + (qrow0 + qr) * a.ld_out + (size_t)head * HDP;
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) {
      float of[8], df[8];
      unpack8(*(const u32x4*)(orow + ks * 16 + 8 * h), of);
      unpack8(__builtin_bit_cast(u32x4, dof[ks]), df);
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
  const float dscale = a.drop_thr ? a.drop_scale : 1.0f;

  const int ntiles = (valid_key_limit(a.key_mask, b, Sb, lane) + 63) / 64;   // trailing all-padding key tiles are skipped
  // dQ needs Q, dO (B operands) and the dQ^T accumulator resident: > 256 registers, so this kernel runs one wave per
  // SIMD with the full 512-entry file and prefetches the next K/V tile into registers under the MFMA work instead.
  TileRegs<64, HDP> kr, vr;
  // The keep-bit word and the key-validity byte of a tile travel with the tile prefetch, one tile ahead in registers:
  // vmcnt retires in order, so loading them inside the body (after the 12 prefetch loads were issued) made their
  // s_waitcnt a vmcnt(0) that exposed the whole prefetch latency in every tile.
  // (RAW loaded values are carried: any arithmetic on them here would pull their wait up to this point)
  const uint8_t* kmrow = a.key_mask ? a.key_mask + (size_t)b * S : nullptr;
  const unsigned long long* brow = a.block_bits ? (const unsigned long long*)a.block_bits + (size_t)qr * SW : nullptr;
  unsigned long long dm_n = ~0ull, blk_n = 0ull;
  uint8_t km_n = 0;
  if (ntiles > 0) {
    kr.load(kbase, ld, 0, Sb - 1, false, tid);
    vr.load(vbase, ld, 0, Sb - 1, false, tid);
    if (a.drop_thr) dm_n = drow[0];
    if (brow) blk_n = brow[0];
    if (kmrow) km_n = kmrow[min(lane, Sb - 1)];
  }
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * 64;
    __syncthreads();
    kr.store(kt, tid);
    vr.store(vt, tid);
    const unsigned long long dm = dm_n >> (4 * h);
    const unsigned long long vlane = __ballot(kv0 + lane < Sb && km_n == 0) & ~blk_n;     // per query: valid and not blocked
    const bool all_valid = __all(vlane == ~0ull);
    const unsigned long long vbits = vlane >> (4 * h);
    __syncthreads();
    if (t + 1 < ntiles) {
      kr.load(kbase, ld, kv0 + 64, Sb - 1, false, tid);
      vr.load(vbase, ld, kv0 + 64, Sb - 1, false, tid);
      if (a.drop_thr) dm_n = drow[t + 1];
      if (brow) blk_n = brow[t + 1];
      if (kmrow) km_n = kmrow[min(kv0 + 64 + lane, Sb - 1)];
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
      // One wave per SIMD: nothing hides an LDS round trip but this wave's own instruction order, so the fragment reads
      // run AHEAD of the MFMAs that consume them (hipcc's default order was read -> lgkmcnt(0) -> MFMA, one read in
      // flight, i.e. ~110 cycles per 32-cycle MFMA).  A deeper software pipeline across the two halves (MFMAs of one
      // half under the softmax VALU of the other) was tried and spills at 512 registers -- see DESIGN.md.
      __builtin_amdgcn_sched_barrier(0);        // phases are separate scheduling regions: each group chain sees only its own reads / MFMAs
      bf16x8 kfr[G::KSTEPS], vfr[G::KSTEPS];
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks) kfr[ks] = row_frag<HDP>(kt, kb * 32, ks, lane);
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks) vfr[ks] = row_frag<HDP>(vt, kb * 32, ks, lane);
      // (hipcc issues the St chain before the dPt chain whatever the source order: the reads follow that order)
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks], qf[ks], st, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[ks], dof[ks], dp, 0, 0, 0);
      {
        constexpr int NR = 2 * G::KSTEPS, AHEAD = NR < 8 ? NR : (HDP <= 192 ? 8 : 4);   // hd > 192: fewer fragments in flight (512-register budget)
        __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
        for (int i = 0; i < NR - AHEAD; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // K^T fragments of the dQ product do not depend on the softmax: issue their reads before the VALU block
      constexpr bool KT_EARLY = HDP <= 192;       // hd > 192: the 2 * DBLK fragments held across the softmax would spill
      bf16x8 ktf[2][G::DBLK];
      if constexpr (KT_EARLY) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int d = 0; d < G::DBLK; ++d) ktf[s][d] = tr_frag<HDP>(kt, kb * 32 + 16 * s, d * 32, lane);
      }
      // dSt = P * (keep/(1-p) * dPt - delta); the key-validity select only exists on tiles that contain padded or
      // out-of-range keys (wave-uniform branch: with right padding that is the last tile of a sample)
      if (all_valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int bit = kb * 32 + (r & 3) + 8 * (r >> 2);
          const float p = fast_exp2(fmaf(st[r], sc, -lse));
          const float ks = ((dm >> bit) & 1ull) ? dscale : 0.f;
          st[r] = p * fmaf(dp[r], ks, -delta);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int bit = kb * 32 + (r & 3) + 8 * (r >> 2);
          const float p = ((vbits >> bit) & 1ull) ? fast_exp2(fmaf(st[r], sc, -lse)) : 0.f;
          const float ks = ((dm >> bit) & 1ull) ? dscale : 0.f;
          st[r] = p * fmaf(dp[r], ks, -delta);
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 dsf = acc_frag(st, s);
#pragma unroll
        for (int d = 0; d < G::DBLK; ++d) {
          if constexpr (!KT_EARLY) ktf[s][d] = tr_frag<HDP>(kt, kb * 32 + 16 * s, d * 32, lane);
          dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[s][d], dsf, dq[d], 0, 0, 0);
        }
      }
    }
  }
  if (qrow < Sqb) {
    u16* orow = a.q != nullptr ? (u16*)a.dq + (qrow0 + qrow) * a.ld_dq + (size_t)head * HDP
                               : (u16*)a.dqkv + (sr.row0 + qrow) * a.ld_dqkv + (size_t)(0 * a.H + head) * HDP;
#pragma unroll
    for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        u32x2 v;
        v[0] = pack2bf(dq[d][4 * g4] * a.scale, dq[d][4 * g4 + 1] * a.scale);
        v[1] = pack2bf(dq[d][4 * g4 + 2] * a.scale, dq[d][4 * g4 + 3] * a.scale);
        *(u32x2*)(orow + d * 32 + 8 * g4 + 4 * h) = v;
      }
  }
}

// ================================================================================================
// backward, dQ, two waves per SIMD: the same algorithm on v_mfma_f32_16x16x32_bf16 with 16 queries per wave
// (8 waves = 128 queries per workgroup).  Halving the rows a wave owns halves everything it keeps resident
// (dQ^T 48 + Q 24 + dO 24 registers at hd = 192), so the kernel fits 256 registers and a second wave per SIMD
// issues its MFMAs under this wave's softmax VALU and LDS round trips -- the overlap one wave per SIMD could only
// get from instruction order.  Accumulator layout: lane (g = lane>>4, n = lane&15) holds query n, rows 4g..4g+3
// of each 16-key block, which are exactly elements 0..3 / 4..7 of its B fragment for the 32-key k-step, so dSt feeds
// the dQ product from registers as before (k index 8g+e <-> key 16(e>>2) + 4g + (e&3); the K^T A-fragment is
// fetched in that order by two ds_read_b64_tr_b16 of 4 rows x 16 hd columns per 16-lane group).
// LDS tile: same 64 (mod 256)-byte row stride, 16-B chunk ^= {0,2,3,1}[(row>>2)&3]: conflict-free for the ds_read_b128
// lane groups of the 16-row operand AND for the transposed reads (the 32x32 swizzle is 2-way on the former).
// ================================================================================================
#ifndef TF_DKV16_QT
#define TF_DKV16_QT 64
#endif
__device__ __forceinline__ int swz16(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }
__device__ __forceinline__ int tile_off16(int row, int chunk, int tstr) { return row * tstr + ((chunk ^ swz16(row)) << 4); }

template <int ROWS, int HDP, int NT> struct TileRegs16 {
  static constexpr int TOTAL = ROWS * (HDP / 8);
  static constexpr int PER = (TOTAL + NT - 1) / NT;
  u32x4 v[PER];
  __device__ __forceinline__ void load(const u16* __restrict__ base, size_t ld, int row0, int row_max, int tid, bool zero_fill = false) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = i * NT + tid;
      if (TOTAL % NT == 0 || id < TOTAL) {
        const int r = id / (HDP / 8), c = id % (HDP / 8);
        v[i] = *(const u32x4*)((const unsigned char*)base + ((unsigned)min(row0 + r, row_max) * (unsigned)(ld * 2) + (unsigned)(c * 16)));   // (see TileRegs::load)
        if (zero_fill && row0 + r > row_max) v[i] = u32x4{0, 0, 0, 0};
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = i * NT + tid;
      if (TOTAL % NT == 0 || id < TOTAL) {
        const int r = id / (HDP / 8), c = id % (HDP / 8);
        *(u32x4*)(lds + tile_off16(r, c, Geo<HDP>::TSTR)) = v[i];
      }
    }
  }
};

template <int HDP, int NW>          // NW waves per workgroup, 16 queries each
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dq16_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  constexpr int NT = 64 * NW, QB = 16 * NW;
  constexpr int KS = HDP / 32;     // 32-deep k-steps over the head dim (St, dPt)
  constexpr int DB = HDP / 16;     // 16-row blocks of dQ^T
  constexpr int TSTR = G::TSTR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, n = lane & 15;
  const int S = a.S;
  const int nqb = (S + QB - 1) / QB;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nqb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int q0 = (logical % nqb) * QB + wave * 16;
  // S indexes lse / delta / the dropout rows and shapes the grid; Sb is THIS sample's row count (packed batches: TfAttnArgs.cu_rows)
  const SampleRows sr = sample_rows(a.cu_rows, b, S);
  const int Sb = sr.len;
  if ((logical % nqb) * QB >= Sb) return;              // query blocks past the sample's end (workgroup-uniform)
  const size_t ld = a.ld_qkv;
  const u16* qkv = (const u16*)a.qkv;
  const u16* qbase = qkv + sr.row0 * ld + (size_t)(0 * a.H + head) * HDP;
  const u16* kbase = qkv + sr.row0 * ld + (size_t)(1 * a.H + head) * HDP;
  const u16* vbase = qkv + sr.row0 * ld + (size_t)(2 * a.H + head) * HDP;
  const u16* dobase = (const u16*)a.dout + sr.row0 * a.ld_dout + (size_t)head * HDP;

  // the first K/V tile is requested before anything else: with one workgroup per CU nothing but this workgroup's own
  // instruction order overlaps the prologue's global round trips (tile, Q / dO / O rows, key-mask scan)
  TileRegs16<64, HDP, NT> kr, vr;
  kr.load(kbase, ld, 0, Sb - 1, tid);
  vr.load(vbase, ld, 0, Sb - 1, tid);
  const int qrow = q0 + n;
  const int qr = min(qrow, Sb - 1);
  bf16x8 qf[KS], dof[KS];          // B operands: query n, hd elements 32ks + 8g .. +7
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = as_bf16x8(*(const u32x4*)(qbase + (size_t)qr * ld + ks * 32 + 8 * g));
    dof[ks] = as_bf16x8(*(const u32x4*)(dobase + (size_t)qr * a.ld_dout + ks * 32 + 8 * g));
  }
  const float lse = a.lse[(size_t)bh * S + qr];
  float delta = 0.f;               // rowsum(dO . O): each lane group owns a quarter of the row
  {
    const u16* orow = (const u16*)a.out + (sr.row0 + qr) * a.ld_out + (size_t)head * HDP;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float of[8], df[8];
      unpack8(*(const u32x4*)(orow + ks * 32 + 8 * g), of);
      unpack8(__builtin_bit_cast(u32x4, dof[ks]), df);
#pragma unroll
      for (int e = 0; e < 8; ++e) delta = fmaf(of[e], df[e], delta);
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (g == 0 && qrow < Sb) a.delta[(size_t)bh * S + qrow] = delta;
  }
  f32x4 dq[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float sc = a.scale * LOG2E;
  const int SW = (S + 63) / 64;
  const unsigned long long* drow = a.drop_thr ? (const unsigned long long*)a.drop_bits + ((size_t)bh * S + qr) * SW : nullptr;
  const float dscale = a.drop_thr ? a.drop_scale : 1.0f;

  // per-lane LDS addresses: everything else is a compile-time offset
  const int rbase = n * TSTR + ((g ^ swz16(n)) << 4);                 // row read: + (32kb + 16j) * TSTR + 64 * ks
  const int q4 = n >> 2, p = n & 3, fz = swz16(4 * g);
  const int tbase = (4 * g + q4) * TSTR + 8 * (p & 1);                 // transposed read: + (32kb + 16t) * TSTR + 64 * (db>>1) + xe|xo
  const int xe = ((p >> 1) ^ fz) << 4, xo = ((2 + (p >> 1)) ^ fz) << 4;

  const int ntiles = (valid_key_limit(a.key_mask, b, Sb, lane) + 63) / 64;
  const uint8_t* kmrow = a.key_mask ? a.key_mask + (size_t)b * S : nullptr;
  const unsigned long long* brow = a.block_bits ? (const unsigned long long*)a.block_bits + (size_t)qr * SW : nullptr;
  unsigned long long dm_n = ~0ull, blk_n = 0ull;
  uint8_t km_n = 0;
  // LDS holds TWO K/V tile pairs (4 x 64 rows): tile t+1 is written into the other pair in the middle of tile t's work
  // (its global loads were issued a tile earlier), so a tile costs one barrier and no store phase of its own.
  constexpr int PAIR = 128 * TSTR;
  if (ntiles > 0) {
    if (a.drop_thr) dm_n = drow[0];
    if (brow) blk_n = brow[0];
    if (kmrow) km_n = kmrow[min(lane, Sb - 1)];
    kr.store(smem, tid);
    vr.store(smem + 64 * TSTR, tid);
    if (ntiles > 1) {
      kr.load(kbase, ld, 64, Sb - 1, tid);
      vr.load(vbase, ld, 64, Sb - 1, tid);
    }
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * 64;
    const unsigned char* kt = smem + (t & 1) * PAIR;
    const unsigned char* vt = kt + 64 * TSTR;
    const unsigned long long dm = dm_n >> (4 * g);
    const unsigned long long vlane = __ballot(kv0 + lane < Sb && km_n == 0) & ~blk_n;
    const bool all_valid = __all(vlane == ~0ull);
    const unsigned long long vbits = vlane >> (4 * g);
    if (t + 1 < ntiles) {
      if (a.drop_thr) dm_n = drow[t + 1];
      if (brow) blk_n = brow[t + 1];
      if (kmrow) km_n = kmrow[min(kv0 + 64 + lane, Sb - 1)];
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x4 st[2], dp[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) { st[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      // phase A: the 4*KS row fragments run AHEAD of the MFMAs that consume them (an LDS round trip is ~8 MFMA slots)
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 kfr[2 * KS], vfr[2 * KS];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kfr[j * KS + ks] = *(const bf16x8*)(kt + rbase + (32 * kb + 16 * j) * TSTR + 64 * ks);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vfr[j * KS + ks] = *(const bf16x8*)(vt + rbase + (32 * kb + 16 * j) * TSTR + 64 * ks);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[j * KS + ks], qf[ks], st[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[j * KS + ks], dof[ks], dp[j], 0, 0, 0);
      {
        constexpr int NR = 4 * KS, AHEAD = NR < 8 ? NR : (HDP >= 192 ? 6 : 8);
        __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
        for (int i = 0; i < NR - AHEAD; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // phase C operands (K^T) do not depend on the softmax: their reads are issued before the VALU block
      s16x4 tlo[DB], thi[DB];
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const unsigned char* tp = kt + tbase + (32 * kb) * TSTR + 64 * (d >> 1) + ((d & 1) ? xo : xe);
        tlo[d] = lds_read_tr16(tp);
        thi[d] = lds_read_tr16(tp + 16 * TSTR);
      }
      // phase B: dSt = P * (keep/(1-p) * dPt - delta)
      bf16x8 dsf;
      if (all_valid) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int bit = kb * 32 + 16 * j + i;
            const float pr = fast_exp2(fmaf(st[j][i], sc, -lse));
            const float ks = ((dm >> bit) & 1ull) ? dscale : 0.f;
            dsf[4 * j + i] = (__bf16)(pr * fmaf(dp[j][i], ks, -delta));
          }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int bit = kb * 32 + 16 * j + i;
            const float pr = ((vbits >> bit) & 1ull) ? fast_exp2(fmaf(st[j][i], sc, -lse)) : 0.f;
            const float ks = ((dm >> bit) & 1ull) ? dscale : 0.f;
            dsf[4 * j + i] = (__bf16)(pr * fmaf(dp[j][i], ks, -delta));
          }
      }
#pragma unroll
      for (int d = 0; d < DB; ++d) dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_tr(tlo[d], thi[d]), dsf, dq[d], 0, 0, 0);
      if (kb == 0 && t + 1 < ntiles) {
        // the other pair was last read in tile t-1, which every wave left through the barrier below
        unsigned char* nk = smem + ((t + 1) & 1) * PAIR;
        kr.store(nk, tid);
        vr.store(nk + 64 * TSTR, tid);
        if (t + 2 < ntiles) {
          kr.load(kbase, ld, kv0 + 128, Sb - 1, tid);
          vr.load(vbase, ld, kv0 + 128, Sb - 1, tid);
        }
      }
    }
    __syncthreads();
  }
  if (qrow < Sb) {
    u16* orow = (u16*)a.dqkv + (sr.row0 + qrow) * a.ld_dqkv + (size_t)(0 * a.H + head) * HDP;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      u32x2 v;
      v[0] = pack2bf(dq[d][0] * a.scale, dq[d][1] * a.scale);
      v[1] = pack2bf(dq[d][2] * a.scale, dq[d][3] * a.scale);
      *(u32x2*)(orow + d * 16 + 4 * g) = v;
    }
  }
}

// ================================================================================================
// backward, dK / dV: key on the lane, loop over query tiles of 32
//   S[q][key] = Q.K^T -> P ;  dP[q][key] = dO.V^T ;  Pd = P*keep/(1-p) ;  dS = P*(keep/(1-p)*dP - delta)
//   dV^T[d][key] += dO^T[d][q] . Pd[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key] ; dK *= scale
// ================================================================================================
template <int HDP, bool BLK>       // BLK: a block-bit matrix (TfAttnArgs.block_bits) is present
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* qt = smem;
  unsigned char* dot = smem + 32 * G::TSTR;
  float* lse_s = (float*)(smem + 64 * G::TSTR);
  float* del_s = lse_s + 32;
  unsigned* dw_s = (unsigned*)(del_s + 32);                       // [4 waves][32 query rows] keep-bit words of the wave's 32 keys
  unsigned* bw_s = dw_s + 128;                                    // [4 waves][32 query rows] block-bit words (BLK only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int S = a.S;                                   // keys
  const int Sq = a.q != nullptr ? a.Sq : S;            // queries
  const int nkb = (S + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);          // key blocks of one (batch, head) share Q and dO: same XCD
  const int bh = pair_of_group(logical / nkb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int key0 = (logical % nkb) * 128 + wave * 32;
  const SampleRows sr = sample_rows(a.cu_rows, b, S);            // packed batches: see attn_fwd_kernel
  const int Sb = sr.len, Sqb = a.q != nullptr ? Sq : Sb;
  if ((logical % nkb) * 128 >= Sb) return;             // key blocks past the sample's end (workgroup-uniform)
  const size_t ld = a.ld_qkv;
  const u16* qkv = (const u16*)a.qkv;
  const size_t ldq = a.q != nullptr ? (size_t)a.ld_q : ld;
  const size_t qrow0 = a.q != nullptr ? (size_t)b * Sq : sr.row0;
  const u16* qbase = a.q != nullptr ? (const u16*)a.q + qrow0 * ldq + (size_t)head * HDP
                                    : qkv + sr.row0 * ld + (size_t)(0 * a.H + head) * HDP;
  const u16* kbase = qkv + sr.row0 * ld + (size_t)(1 * a.H + head) * HDP;
  const u16* vbase = qkv + sr.row0 * ld + (size_t)(2 * a.H + head) * HDP;
  const u16* dobase = (const u16*)a.dout + qrow0 * a.ld_dout + (size_t)head * HDP;

  const int key = key0 + (lane & 31);
  const int kr_ = min(key, Sb - 1);
  bool key_ok = key < Sb;
  if (key_ok && a.key_mask != nullptr) key_ok = a.key_mask[(size_t)b * S + key] == 0;
  // K^T / V^T B-operand fragments (B[k = hd][col = key]) resident in registers
  bf16x8 kf[G::KSTEPS], vf[G::KSTEPS];
#pragma unroll
  for (int ks = 0; ks < G::KSTEPS; ++ks) {
    kf[ks] = as_bf16x8(*(const u32x4*)(kbase + (size_t)kr_ * ld + ks * 16 + 8 * h));
    vf[ks] = as_bf16x8(*(const u32x4*)(vbase + (size_t)kr_ * ld + ks * 16 + 8 * h));
  }
  f32x16 dk[G::DBLK], dv[G::DBLK];
#pragma unroll
  for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
  const float sc = a.scale * LOG2E;
  const float dscale = a.drop_thr ? a.drop_scale : 1.0f;

  // a key block that holds only padding receives exactly-zero dK / dV: skip its query loop
  const int ntiles = ((logical % nkb) * 128 >= valid_key_limit(a.key_mask, b, Sb, lane)) ? 0 : (Sqb + 31) / 32;
  TileRegs<32, HDP> qr, dr;
  // per-tile row scalars travel with the tile prefetch (one tile ahead, in registers): LSE / delta of row q0 + tid
  // (threads 0..31) and the keep-bit word of (row q0 + (lane & 31), this wave's 32 keys).  Loaded inside the loop body
  // they sat between the two barriers with their global latency fully exposed, once per tile.
  const int dw_ld = 2 * ((S + 63) / 64);
  // (a wave whose 32 keys all lie past S -- S mod 128 in 1 .. 64 -- has no word of its own in a row of dw_ld words: it reads the row's
  // last word and never uses it.  Unclamped, the last row of the last (batch, head) read 4 - 8 bytes past the end of drop_bits)
  const int wsel = min(key0 >> 5, dw_ld - 1);
  const unsigned* dbits = (const unsigned*)a.drop_bits + (size_t)bh * Sq * dw_ld + wsel;
  const unsigned* bbits = BLK ? (const unsigned*)a.block_bits + wsel : nullptr;
  float lse_n = 1.0e30f, del_n = 0.f;
  unsigned dw_n = 0xffffffffu, bw_n = 0u;
  // (RAW loaded values are carried, from clamped addresses; the row-validity selects happen when they are stored to
  // LDS one iteration later -- arithmetic on them here would pull their vmcnt wait up to this point)
  auto prefetch_rows = [&](int q0n) {
    if (tid < 32) {
      const int q = min(q0n + tid, Sqb - 1);
      lse_n = a.lse[(size_t)bh * Sq + q];
      del_n = a.delta[(size_t)bh * Sq + q];
    }
    if (a.drop_thr) dw_n = dbits[(size_t)min(q0n + (lane & 31), Sqb - 1) * dw_ld];
    if (BLK) bw_n = bbits[(size_t)min(q0n + (lane & 31), Sqb - 1) * dw_ld];
  };
  if (ntiles > 0) {
    qr.load(qbase, ldq, 0, Sqb - 1, false, tid);
    dr.load(dobase, a.ld_dout, 0, Sqb - 1, true, tid);       // rows >= Sqb contribute nothing
    prefetch_rows(0);
  }
  for (int t = 0; t < ntiles; ++t) {
    const int q0 = t * 32;
    __syncthreads();
    qr.store(qt, tid);
    dr.store(dot, tid);
    if (tid < 32) {
      const bool in = q0 + tid < Sqb;
      lse_s[tid] = in ? lse_n : 1.0e30f;                     // P = 0 for rows past the end
      del_s[tid] = in ? del_n : 0.f;
    }
    if (lane < 32) dw_s[wave * 32 + lane] = (!a.drop_thr || q0 + lane < Sqb) ? dw_n : 0u;
    if (BLK && lane < 32) bw_s[wave * 32 + lane] = bw_n;
    __syncthreads();
    if (t + 1 < ntiles) {
      qr.load(qbase, ldq, q0 + 32, Sqb - 1, false, tid);
      dr.load(dobase, a.ld_dout, q0 + 32, Sqb - 1, true, tid);
      prefetch_rows(q0 + 32);
    }
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
    // one wave per SIMD: fragment reads run ahead of their MFMAs (see the dQ kernel)
    {
      bf16x8 qfr[G::KSTEPS], dfr[G::KSTEPS];
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks) { qfr[ks] = row_frag<HDP>(qt, 0, ks, lane); dfr[ks] = row_frag<HDP>(dot, 0, ks, lane); }
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[ks], kf[ks], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr[ks], vf[ks], dp, 0, 0, 0);
      }
      constexpr int NR = 2 * G::KSTEPS, AHEAD = NR < 8 ? NR : (HDP <= 192 ? 8 : 4);   // hd > 192: fewer fragments in flight (512-register budget)
      __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
      for (int i = 0; i < NR - AHEAD; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
    }
    // registers 4g..4g+3 are query rows 8g + 4h + (0..3): one 16-B LDS read per group for LSE, delta and the keep words
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 l4 = *(const f32x4*)(lse_s + 8 * g4 + 4 * h);
      const f32x4 d4 = *(const f32x4*)(del_s + 8 * g4 + 4 * h);
      const u32x4 w4 = *(const u32x4*)(dw_s + wave * 32 + 8 * g4 + 4 * h);           // rows q0 + acc_row(4 g4 + i, h)
      u32x4 b4 = {0u, 0u, 0u, 0u};
      if (BLK) b4 = *(const u32x4*)(bw_s + wave * 32 + 8 * g4 + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g4 + i;
        const bool att = BLK ? (key_ok && !((b4[i] >> (lane & 31)) & 1u)) : key_ok;     // this (query row, key) pair is attended
        const float p = att ? fast_exp2(fmaf(st[r], sc, -l4[i])) : 0.f;   // (a wave-uniform all-keys-valid branch here measured slower)
        const float keep_scale = ((w4[i] >> (lane & 31)) & 1u) ? dscale : 0.f;
        st[r] = p * keep_scale;                           // Pd
        dp[r] = p * fmaf(dp[r], keep_scale, -d4[i]);      // dS
      }
    }
    {
      // dV^T += dO^T . Pd and dK^T += Q^T . dS: 4 DBLK transposed fragments (2 reads each), 4 fragments ahead of the MFMAs
      constexpr int NF = 4 * G::DBLK, AH = HDP <= 192 ? 4 : 2;
      const bf16x8 pf[2] = {acc_frag(st, 0), acc_frag(st, 1)}, dsf[2] = {acc_frag(dp, 0), acc_frag(dp, 1)};
      __builtin_amdgcn_sched_barrier(0);        // own scheduling region: the groups below then only see these reads / MFMAs
      bf16x8 tf[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int s2 = i / (2 * G::DBLK), d = (i >> 1) % G::DBLK;
        tf[i] = (i & 1) ? tr_frag<HDP>(qt, 16 * s2, d * 32, lane) : tr_frag<HDP>(dot, 16 * s2, d * 32, lane);
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int s2 = i / (2 * G::DBLK), d = (i >> 1) % G::DBLK;
        if (i & 1) dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i], dsf[s2], dk[d], 0, 0, 0);
        else dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i], pf[s2], dv[d], 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * AH, 0);
#pragma unroll
      for (int i = 0; i < NF - AH; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, AH, 0);
    }
  }
  if (key < Sb) {
    u16* krow = (u16*)a.dqkv + (sr.row0 + key) * a.ld_dqkv + (size_t)(1 * a.H + head) * HDP;
    u16* vrow = (u16*)a.dqkv + (sr.row0 + key) * a.ld_dqkv + (size_t)(2 * a.H + head) * HDP;
#pragma unroll
    for (int d = 0; d < G::DBLK; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        u32x2 v;
        v[0] = pack2bf(dk[d][4 * g4] * a.scale, dk[d][4 * g4 + 1] * a.scale);
        v[1] = pack2bf(dk[d][4 * g4 + 2] * a.scale, dk[d][4 * g4 + 3] * a.scale);
        *(u32x2*)(krow + d * 32 + 8 * g4 + 4 * h) = v;
        v[0] = pack2bf(dv[d][4 * g4], dv[d][4 * g4 + 1]);
        v[1] = pack2bf(dv[d][4 * g4 + 2], dv[d][4 * g4 + 3]);
        *(u32x2*)(vrow + d * 32 + 8 * g4 + 4 * h) = v;
      }
  }
}

// ================================================================================================
// backward, dK / dV, two waves per SIMD: 16 keys per wave on v_mfma_f32_16x16x32_bf16 (8 waves = 128 keys per
// workgroup), the same restructuring as attn_bwd_dq16_kernel.  Resident per wave: dK^T + dV^T (96 registers at
// hd = 192) + the K / V B-operand fragments (48).  Q / dO tiles of 32 query rows are double-buffered in LDS together
// with their row scalars (LSE, delta, the keep-bit / block-bit words of each wave's keys): one barrier per tile.
//   S[q][key] = Q.K^T -> P ;  dP = dO.V^T ;  Pd = P*keep/(1-p) ;  dS = P*(keep/(1-p)*dP - delta)
//   dV^T[d][key] += dO^T[d][q] . Pd[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key]
// ================================================================================================
// DS: the dS tile of every (16 keys of this wave) x (32-query sub-tile) is also written to TfAttnArgs.ds_work as one 1-KiB chunk, the lanes'
// 16-B fragments side by side (a fully coalesced store of registers the kernel holds anyway), for attn_bwd_dq_ds_kernel -- which then
// forms dQ = dS . K without recomputing S and dP.  Chunk (bh, kb16, qb32) lives at (((bh * nkb + kb16) * nqb + qb32) KiB; inside it the
// fragment of lane (g, n) -- key n, queries 16 j + 4 g + i in element 4 j + i -- sits at byte 64 n + 16 g with its two 8-B halves swapped
// when (n >> 2) is odd, which makes the consumer's transposed LDS reads bank-conflict free.
__host__ __device__ inline int ds_nkb(int S) { return 8 * ((S + 127) / 128); }     // 16-key blocks per (batch, head): whole 128-key workgroups
__host__ __device__ inline int ds_nqb(int S) { return 4 * ((S + 127) / 128); }     // 32-query blocks per (batch, head): whole 128-query workgroups

// WHICH: 2 = dK and dV in one pass (head dims <= 192).  At head dims 224 / 256 the two accumulator sets (2 x HDP / 4 registers) plus
// the K and V fragments overflow 256 registers, so the work is two passes of the same kernel, each inside the budget at two waves per
// SIMD: WHICH = 0 forms S -> P -> dV (K fragments and dV^T resident), WHICH = 1 forms S, dP -> dS -> dK (K, V fragments and dK^T
// resident) and emits the dS tiles.  One more S recompute, no spills (the one-pass 32-row kernel spilled 46 - 89 registers there).
template <int HDP, bool BLK, int QT, bool DS = false, int WHICH = 2>     // QT query rows per LDS tile (32 or 64)
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv16_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  constexpr bool DO_V = WHICH != 1, DO_K = WHICH != 0;
  static_assert(!DS || DO_K, "the dS tiles come from the pass that forms dS");
  constexpr int NT = 512;
  constexpr int KS = HDP / 32, DB = HDP / 16, TSTR = G::TSTR;
  constexpr int PAIR = 2 * QT * TSTR;                     // Q tile (QT rows) + dO tile (QT rows)
  constexpr int ROWS_BYTES = 8 * QT + 32 * QT + 32 * QT;  // lse[QT], delta[QT], keep words [8][QT], block words [8][QT]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, n = lane & 15;
  const int S = a.S;
  const int nkb = (S + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nkb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int key0 = (logical % nkb) * 128 + wave * 16;
  const SampleRows sr = sample_rows(a.cu_rows, b, S);            // packed batches: see attn_bwd_dq16_kernel
  const int Sb = sr.len;
  if ((logical % nkb) * 128 >= Sb) return;             // key blocks past the sample's end (workgroup-uniform)
  const size_t ld = a.ld_qkv;
  const u16* qkv = (const u16*)a.qkv;
  const u16* qbase = qkv + sr.row0 * ld + (size_t)(0 * a.H + head) * HDP;
  const u16* kbase = qkv + sr.row0 * ld + (size_t)(1 * a.H + head) * HDP;
  const u16* vbase = qkv + sr.row0 * ld + (size_t)(2 * a.H + head) * HDP;
  const u16* dobase = (const u16*)a.dout + sr.row0 * a.ld_dout + (size_t)head * HDP;

  TileRegs16<QT, HDP, NT> qr, dr;
  qr.load(qbase, ld, 0, Sb - 1, tid);
  dr.load(dobase, a.ld_dout, 0, Sb - 1, tid, true);       // rows >= Sb contribute nothing
  const int key = key0 + n;
  const int kr_ = min(key, Sb - 1);
  bool key_ok = key < Sb;
  if (key_ok && a.key_mask != nullptr) key_ok = a.key_mask[(size_t)b * S + key] == 0;
  bf16x8 kf[KS], vf[DO_K ? KS : 1];           // B operands: key n, hd elements 32ks + 8g .. +7 (V only where dP is formed)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = as_bf16x8(*(const u32x4*)(kbase + (size_t)kr_ * ld + ks * 32 + 8 * g));
    if constexpr (DO_K) vf[ks] = as_bf16x8(*(const u32x4*)(vbase + (size_t)kr_ * ld + ks * 32 + 8 * g));
  }
  f32x4 dk[DO_K ? DB : 1], dv[DO_V ? DB : 1];
#pragma unroll
  for (int d = 0; d < DB; ++d) {
    if constexpr (DO_K) dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (DO_V) dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float sc = a.scale * LOG2E;
  const float dscale = a.drop_thr ? a.drop_scale : 1.0f;
  const int kbit = (key0 & 16) + n;                       // this lane's key inside the 32-bit keep / block words
  // DS: this wave's row of chunks (one per 32-query sub-tile), already offset to the lane's 16-B slot
  unsigned char* ds_row = nullptr;
  if constexpr (DS) ds_row = (unsigned char*)a.ds_work + (((size_t)bh * ds_nkb(S) + (key0 >> 4)) * ds_nqb(S)) * 1024 + 64 * n + 16 * g;

  const int rbase = n * TSTR + ((g ^ swz16(n)) << 4);                 // row read: + 16j * TSTR + 64 * ks
  const int q4 = n >> 2, p = n & 3, fz = swz16(4 * g);
  const int tbase = (4 * g + q4) * TSTR + 8 * (p & 1);                 // transposed read: + 16t * TSTR + 64 * (db>>1) + xe|xo
  const int xe = ((p >> 1) ^ fz) << 4, xo = ((2 + (p >> 1)) ^ fz) << 4;

  const int ntiles = ((logical % nkb) * 128 >= valid_key_limit(a.key_mask, b, Sb, lane)) ? 0 : (Sb + QT - 1) / QT;
  const int dw_ld = 2 * ((S + 63) / 64);
  // (waves whose 16 keys all lie past S read the row's last word and never use it: see attn_bwd_dkv_kernel.  This unclamped index was
  // the round-5 abort of test_attention_fwd_bwd[False-2-64-1-96]: DESIGN.md, "Round 6")
  const int wsel = min(key0 >> 5, dw_ld - 1);
  const unsigned* dbits = (const unsigned*)a.drop_bits + (size_t)bh * S * dw_ld + wsel;
  const unsigned* bbits = BLK ? (const unsigned*)a.block_bits + wsel : nullptr;
  float lse_n = 1.0e30f, del_n = 0.f;
  unsigned dw_n = 0xffffffffu, bw_n = 0u;
  // (RAW loaded values are carried; the row-validity selects happen when they are stored)
  auto prefetch_rows = [&](int q0n) {
    if (tid < QT) {
      const int q = min(q0n + tid, Sb - 1);
      lse_n = a.lse[(size_t)bh * S + q];
      del_n = a.delta[(size_t)bh * S + q];
    }
    if (a.drop_thr) dw_n = dbits[(size_t)min(q0n + (lane & (QT - 1)), Sb - 1) * dw_ld];
    if (BLK) bw_n = bbits[(size_t)min(q0n + (lane & (QT - 1)), Sb - 1) * dw_ld];
  };
  auto store_tile = [&](int q0s, int buf) {
    unsigned char* qt_w = smem + buf * PAIR;
    qr.store(qt_w, tid);
    dr.store(qt_w + QT * TSTR, tid);
    float* lse_w = (float*)(smem + 2 * PAIR + buf * ROWS_BYTES);
    unsigned* dw_w = (unsigned*)(lse_w + 2 * QT);
    if (tid < QT) {
      const bool in = q0s + tid < Sb;
      lse_w[tid] = in ? lse_n : 1.0e30f;                   // P = 0 for rows past the end
      lse_w[QT + tid] = in ? del_n : 0.f;
    }
    if (lane < QT) dw_w[wave * QT + lane] = (!a.drop_thr || q0s + lane < Sb) ? dw_n : 0u;
    if (BLK && lane < QT) dw_w[8 * QT + wave * QT + lane] = bw_n;
  };
  if (ntiles > 0) {
    prefetch_rows(0);
    store_tile(0, 0);
    if (ntiles > 1) {
      qr.load(qbase, ld, QT, Sb - 1, tid);
      dr.load(dobase, a.ld_dout, QT, Sb - 1, tid, true);
      prefetch_rows(QT);
    }
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int q0 = t * QT;
#pragma unroll
    for (int u = 0; u < QT / 32; ++u) {
    const unsigned char* qt = smem + (t & 1) * PAIR + 32 * u * TSTR;
    const unsigned char* dot = qt + QT * TSTR;
    const float* lse_s = (const float*)(smem + 2 * PAIR + (t & 1) * ROWS_BYTES) + 32 * u;
    const unsigned* dw_s = (const unsigned*)(lse_s - 32 * u + 2 * QT) + wave * QT + 32 * u;
    f32x4 st[2], dp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { st[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // phase A: S and dP, the row fragments of Q / dO run ahead of the MFMAs
    __builtin_amdgcn_sched_barrier(0);
    {
      bf16x8 qfr[2 * KS], dfr[DO_K ? 2 * KS : 1];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qfr[j * KS + ks] = *(const bf16x8*)(qt + rbase + 16 * j * TSTR + 64 * ks);
      if constexpr (DO_K) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) dfr[j * KS + ks] = *(const bf16x8*)(dot + rbase + 16 * j * TSTR + 64 * ks);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr[j * KS + ks], kf[ks], st[j], 0, 0, 0);
      if constexpr (DO_K) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr[j * KS + ks], vf[ks], dp[j], 0, 0, 0);
      }
      constexpr int NR = (DO_K ? 4 : 2) * KS, AHEAD = NR < 6 ? NR : 6;
      __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
      for (int i = 0; i < NR - AHEAD; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next tile goes into the other buffer (last read in tile t-1, which every wave left through the barrier below)
    if (u == 0 && t + 1 < ntiles) {
      store_tile(q0 + QT, (t + 1) & 1);
      if (t + 2 < ntiles) {
        qr.load(qbase, ld, q0 + 2 * QT, Sb - 1, tid);
        dr.load(dobase, a.ld_dout, q0 + 2 * QT, Sb - 1, tid, true);
        prefetch_rows(q0 + 2 * QT);
      }
    }
    // phase B: registers i of block j are query rows q0 + 16j + 4g + i
    bf16x8 pf, dsf;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 l4 = *(const f32x4*)(lse_s + 16 * j + 4 * g);
      const f32x4 d4 = *(const f32x4*)(lse_s + QT + 16 * j + 4 * g);
      const u32x4 w4 = *(const u32x4*)(dw_s + 16 * j + 4 * g);
      u32x4 b4 = {0u, 0u, 0u, 0u};
      if (BLK) b4 = *(const u32x4*)(dw_s + 8 * QT + 16 * j + 4 * g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool att = BLK ? (key_ok && !((b4[i] >> kbit) & 1u)) : key_ok;
        const float pr = att ? fast_exp2(fmaf(st[j][i], sc, -l4[i])) : 0.f;
        const float keep_scale = ((w4[i] >> kbit) & 1u) ? dscale : 0.f;
        if constexpr (DO_V) pf[4 * j + i] = (__bf16)(pr * keep_scale);                             // Pd
        if constexpr (DO_K) dsf[4 * j + i] = (__bf16)(pr * fmaf(dp[j][i], keep_scale, -d4[i]));   // dS
      }
    }
    if constexpr (DS) {
      const u32x4 v = __builtin_bit_cast(u32x4, dsf);
      const bool swp = ((n >> 2) & 1) != 0;
      const u32x4 w = {swp ? v[2] : v[0], swp ? v[3] : v[1], swp ? v[0] : v[2], swp ? v[1] : v[3]};
      *(u32x4*)(ds_row + (size_t)(t * (QT / 32) + u) * 1024) = w;
    }
    // phase C: dV^T += dO^T . Pd and dK^T += Q^T . dS; 2 * DB transposed fragments (2 reads each), AH fragments ahead
    {
      constexpr int NF = (WHICH == 2 ? 2 : 1) * DB, AH = 4;
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 tf[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int d = WHICH == 2 ? i >> 1 : i;
        const bool from_q = WHICH == 2 ? (i & 1) != 0 : DO_K;       // dK^T += Q^T . dS reads the Q tile, dV^T += dO^T . Pd the dO tile
        const unsigned char* tp = (from_q ? qt : dot) + tbase + 64 * (d >> 1) + ((d & 1) ? xo : xe);
        tf[i] = join_tr(lds_read_tr16(tp), lds_read_tr16(tp + 16 * TSTR));
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int d = WHICH == 2 ? i >> 1 : i;
        const bool from_q = WHICH == 2 ? (i & 1) != 0 : DO_K;
        if constexpr (DO_K) { if (from_q) dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf[i], dsf, dk[d], 0, 0, 0); }
        if constexpr (DO_V) { if (!from_q) dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf[i], pf, dv[d], 0, 0, 0); }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * AH, 0);
#pragma unroll
      for (int i = 0; i < NF - AH; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, AH, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    __syncthreads();
  }
  if (key < Sb) {
    u16* krow = (u16*)a.dqkv + (sr.row0 + key) * a.ld_dqkv + (size_t)(1 * a.H + head) * HDP;
    u16* vrow = (u16*)a.dqkv + (sr.row0 + key) * a.ld_dqkv + (size_t)(2 * a.H + head) * HDP;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      u32x2 v;
      if constexpr (DO_K) {
        v[0] = pack2bf(dk[d][0] * a.scale, dk[d][1] * a.scale);
        v[1] = pack2bf(dk[d][2] * a.scale, dk[d][3] * a.scale);
        *(u32x2*)(krow + d * 16 + 4 * g) = v;
      }
      if constexpr (DO_V) {
        v[0] = pack2bf(dv[d][0], dv[d][1]);
        v[1] = pack2bf(dv[d][2], dv[d][3]);
        *(u32x2*)(vrow + d * 16 + 4 * g) = v;
      }
    }
  }
}

// ================================================================================================
// backward, dK / dV with ROLE-SPLIT wave pairs (head dims <= 192, with the dS workspace).
// What bounded attn_bwd_dkv16_kernel: every wave owned 16 keys and did everything for them -- S, dP, the softmax / dS arithmetic, dV
// and dK -- as one dependent chain per 32-query sub-tile (fragment reads -> 24 MFMAs -> ~150 VALU -> 24 MFMAs), reading the whole Q
// and dO tile twice (by rows and transposed) for its 16 keys: the matrix pipe was busy a quarter of the time (profiles/r03_v4), and at
// full rate the LDS would have had to deliver 250 B/clk.  Here waves w and w + 4 (the two waves of one SIMD) share 32 keys and split
// the PRODUCTS:
//   role A (waves 0-3): S = Q K^T -> P, Pd = dropout(P) ;  dV^T += dO^T Pd      reads Q by rows, dO transposed
//   role B (waves 4-7): dP = dO V^T ; dS = P (keep/(1-p) dP - delta) ; dK^T += Q^T dS ; stores the dS tiles      reads dO by rows, Q transposed
// so a fragment read from LDS feeds two 16-key blocks and each wave reads two of the four tile images: half the LDS bytes per MFMA, and
// the K / V fragments and one accumulator set per wave leave the registers to run one tile AHEAD: in iteration i a wave issues the 24
// MFMAs of S / dP for tile i + 1 and the 24 of dV / dK for tile i back to back, and the vector arithmetic of the two roles falls on
// OPPOSITE sides of the iteration's one barrier -- B forms dS(i) right after it, while A's 48 MFMAs run; A forms P(i + 1) right before
// the next one, under its own dV MFMAs and B's 48 -- so the SIMD's matrix pipe always has work queued.
// P crosses from A to B through LDS as fp32 in the accumulator layout (the two roles hold the same (query, key) element in the same
// lane and register: a lane-linear 16-B-per-lane image, written once and read once); dS leaves for the dQ kernel exactly as before.
// Q / dO tiles of 32 query rows, three buffers (tiles i and i + 1 in use, i + 2 being written), staged through registers.
// ================================================================================================
constexpr int TF_DKV_PAIR_QT = 32;
// bit `bit` (a per-lane register) of `word` as a sign-extended field: all ones or zero (v_bfe_i32)
__device__ __forceinline__ int bit_mask(int word, int bit) {
  int m;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "v"(bit));
  return m;
}
#ifndef TF_ABL_PAIR
#define TF_ABL_PAIR 0      // timing ablations (variant builds only; results are wrong): 1 no tile transfers, 2 no softmax / dS arithmetic,
#endif                     // 4 no S / dP MFMAs, 8 no dV / dK MFMAs, 16 no row scalars, 32 no tile loop, 64 no dK / dV stores, 128 no K / V loads
// (a __device__ function: the LDS-DMA builtin inside a non-generic lambda of a kernel TEMPLATE makes the host pass drop the
// instantiation silently -- the launch then fails to link)
__device__ __forceinline__ void dma_piece16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, TF_LDS_PTR(lds_dst), 16, voff, soff, 0, 0);
}
template <int HDP, bool BLK>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_pair_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  constexpr int QT = TF_DKV_PAIR_QT, NBUF = 3;
  constexpr int KS = HDP / 32, DB = HDP / 16, TSTR = G::TSTR;
  constexpr int PAIR = 2 * QT * TSTR;                     // Q tile + dO tile, back to back
  constexpr int NPC = (PAIR + 1023) / 1024, NI = (NPC + 7) / 8;       // 1-KiB LDS-DMA pieces of a tile pair; pieces per wave
  constexpr int ROWS_BYTES = 4 * QT * (2 + 4 + 4);        // lse[QT], delta[QT], keep words [4 pairs][QT], block words [4 pairs][QT]
  constexpr int XCH = 4 * 4096;                           // one exchange image: 4 pairs x 64 lanes x 64 B
  constexpr int BUFS = ((NBUF * PAIR + 1023) / 1024) * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const rows_base = smem + BUFS;
  unsigned char* const xch_base = rows_base + NBUF * ROWS_BYTES;
  unsigned char* const dummy = xch_base + 2 * XCH;        // where the surplus pieces of the 8 x NI transfers land (never read)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, n = lane & 15;
  const int pr = wave & 3;
  const bool role_b = wave >= 4;                          // wave-uniform
  const int S = a.S;
  const int nkb = (S + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nkb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int kblk = logical % nkb;
  const int key0 = kblk * 128 + pr * 32;
  const SampleRows sr = sample_rows(a.cu_rows, b, S);
  const int Sb = sr.len;
  if (kblk * 128 >= Sb) return;                           // key blocks past the sample's end (workgroup-uniform)
  const size_t ld = a.ld_qkv;
  const u16* qkv = (const u16*)a.qkv;
  const u16* qbase = qkv + sr.row0 * ld + (size_t)(0 * a.H + head) * HDP;
  const u16* kvbase = qkv + sr.row0 * ld + (size_t)((role_b ? 2 : 1) * a.H + head) * HDP;     // A holds K fragments, B holds V fragments
  const u16* dobase = (const u16*)a.dout + sr.row0 * a.ld_dout + (size_t)head * HDP;

  // ---- tile staging by LDS-DMA: piece k of a tile pair is bytes [1024 k, 1024 k + 1024) of the image Q tile | dO tile; wave w issues
  //      pieces NI w .. NI w + NI - 1.  The image is lane-linear, so the tile's swizzle sits on the SOURCE side; rows past the sample's
  //      end read zeros (buffer bounds).  No staging registers: NI loop-invariant offsets per lane.
  const __amdgpu_buffer_rsrc_t rq = make_rsrc(qbase, ld, Sb, HDP), rdo = make_rsrc(dobase, (size_t)a.ld_dout, Sb, HDP);
  int voff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = wave * NI + i;
    const int pos = (k * 1024 + lane * 16) % (QT * TSTR);              // byte inside the Q or the dO tile
    const int r = pos / TSTR, cp = (pos % TSTR) >> 4;
    const int col = cp < HDP / 8 ? ((cp ^ swz16(r)) << 4) : 0;          // (row padding re-fetches chunk 0: never read)
    const bool is_do = k * 1024 >= QT * TSTR;
    voff[i] = k < NPC ? r * (int)((is_do ? (size_t)a.ld_dout : ld) * 2) + col : 0;
  }
  // (i = position in the sequence of VISITED query tiles: picks the LDS buffer; T = the tile itself: picks the rows.  Without a block
  // mask the two are equal; with TfAttnArgs.block_skip_k the tiles whose 32 queries block all of this workgroup's keys are left out)
  auto dma = [&](int i, int T) {                           // tile pair T -> buffer i % NBUF
    unsigned char* img = smem + (i % NBUF) * PAIR;
    const int soq = T * QT * (int)(ld * 2), sod = T * QT * (int)(a.ld_dout * 2);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = wave * NI + i;                          // wave-uniform
      const bool is_do = k * 1024 >= QT * TSTR;
      unsigned char* dst = k < NPC ? img + k * 1024 : dummy + (k - NPC) * 1024;
      dma_piece16(is_do ? rdo : rq, dst, voff[i], is_do ? sod : soq);
    }
  };

  bool key_ok[2];
  int okm[2];                     // all ones where the lane's key of block kb may be attended
  bf16x8 bfrag[2][KS];            // B operands: key 16 kb + n, hd elements 32 ks + 8 g .. + 7 of K (role A) / V (role B)
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int key = key0 + 16 * kb + n;
    const int kr_ = min(key, Sb - 1);
    key_ok[kb] = key < Sb;
    if (key_ok[kb] && a.key_mask != nullptr) key_ok[kb] = a.key_mask[(size_t)b * S + key] == 0;
    okm[kb] = key_ok[kb] ? -1 : 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (TF_ABL_PAIR & 128) bfrag[kb][ks] = as_bf16x8(u32x4{(unsigned)kr_, 1u, 2u, 3u});
      else bfrag[kb][ks] = as_bf16x8(*(const u32x4*)(kvbase + (size_t)kr_ * ld + ks * 32 + 8 * g));
    }
  }
  f32x4 acc[2][DB];               // dV^T (role A) / dK^T (role B) of the wave pair's 32 keys
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int d = 0; d < DB; ++d) acc[kb][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 sacc[2][2];               // S (A) / dP (B) of the tile one ahead: [key block][16-query block]
  bf16x8 of[2];                   // Pd (A) / dS (B) of the current tile, as B operands of the accumulating product
  of[0] = of[1] = as_bf16x8(u32x4{0u, 0u, 0u, 0u});
  const float sc = a.scale * LOG2E;
  const float dscale = a.drop_thr ? a.drop_scale : 1.0f;
  // B: the pair's two rows of dS chunks (one chunk per 32-query tile), already offset to the lane's 16-B slot
  unsigned char* ds_row = (unsigned char*)a.ds_work + (((size_t)bh * ds_nkb(S) + (key0 >> 4)) * ds_nqb(S)) * 1024 + 64 * n + 16 * g;
  const size_t ds_kb_stride = (size_t)ds_nqb(S) * 1024;

  const int rbase = n * TSTR + ((g ^ swz16(n)) << 4);                 // row read: + 16j * TSTR + 64 * ks
  const int q4 = n >> 2, p = n & 3, fz = swz16(4 * g);
  const int tbase = (4 * g + q4) * TSTR + 8 * (p & 1);                 // transposed read: + 16t * TSTR + 64 * (db>>1) + xe|xo
  const int xe = ((p >> 1) ^ fz) << 4, xo = ((2 + (p >> 1)) ^ fz) << 4;
  const int row_sel = role_b ? QT * TSTR : 0;                          // tile read by rows: Q (A) / dO (B); transposed: the other one
  const int tr_sel = role_b ? 0 : QT * TSTR;

  const int ntiles = (kblk * 128 >= valid_key_limit(a.key_mask, b, Sb, lane)) ? 0 : (Sb + QT - 1) / QT;
  const int dw_ld = 2 * ((S + 63) / 64);
  // row scalars of tile t (threads 0 .. 4 QT - 1): lse / delta of row t & 31 (threads < QT); keep / block word of (row t & 31, pair t >> 5)
  const int wsel = min(4 * kblk + (tid >> 5), dw_ld - 1);
  const unsigned* dbits = (const unsigned*)a.drop_bits + (size_t)bh * S * dw_ld + wsel;
  const unsigned* bbits = BLK ? (const unsigned*)a.block_bits + wsel : nullptr;
  float lse_n = 1.0e30f, del_n = 0.f;
  unsigned dw_n = 0xffffffffu, bw_n = 0u;
  auto rows_load = [&](int t) {
    if (tid < 4 * QT) {
      const int q = min(t * QT + (tid & (QT - 1)), Sb - 1);
      if (tid < QT) { lse_n = a.lse[(size_t)bh * S + q]; del_n = a.delta[(size_t)bh * S + q]; }
      if (a.drop_thr) dw_n = dbits[(size_t)q * dw_ld];
      if (BLK) bw_n = bbits[(size_t)q * dw_ld];
    }
  };
  auto rows_store = [&](int i, int T) {
    if (tid < 4 * QT) {
      float* lse_w = (float*)(rows_base + (i % NBUF) * ROWS_BYTES);
      unsigned* dw_w = (unsigned*)(lse_w + 2 * QT);
      const bool in = T * QT + (tid & (QT - 1)) < Sb;
      if (tid < QT) {
        lse_w[tid] = in ? lse_n : 1.0e30f;                 // P = 0 for rows past the end
        lse_w[QT + tid] = in ? del_n : 0.f;
      }
      dw_w[tid] = (!a.drop_thr || in) ? dw_n : 0u;
      if (BLK) dw_w[4 * QT + tid] = bw_n;
    }
  };
  // S (A) / dP (B) of tile t: rows of Q / dO against the resident K / V fragments; a row fragment feeds both key blocks
  auto sx = [&](int t) {
    const unsigned char* rt = smem + (t % NBUF) * PAIR + row_sel;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 2; ++j) sacc[kb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 rf[2 * KS];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) rf[j * KS + ks] = *(const bf16x8*)(rt + rbase + 16 * j * TSTR + 64 * ks);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sacc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rf[j * KS + ks], bfrag[0][ks], sacc[0][j], 0, 0, 0);
        sacc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rf[j * KS + ks], bfrag[1][ks], sacc[1][j], 0, 0, 0);
      }
    constexpr int NR = 2 * KS, AHEAD = NR < 4 ? NR : 4;
    __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
    for (int i = 0; i < NR - AHEAD; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * AHEAD, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // dV^T += dO^T Pd (A) / dK^T += Q^T dS (B) of tile t: a transposed fragment feeds both key blocks.  The transposed reads come from
  // inline asm, AH fragments ahead of their MFMAs behind counted waits: as intrinsics hipcc puts s_waitcnt vmcnt(0) in front of the first
  // of them (it cannot tell the tile being read from the tile in flight) and the transfers of tile t + 2 drain in mid-iteration.
  const unsigned tr_lds = lds_addr_of(smem) + tr_sel + tbase;
  auto accum = [&](int t) {
    constexpr int AH = 4;
    const unsigned ae = tr_lds + (t % NBUF) * PAIR + xe, ao = tr_lds + (t % NBUF) * PAIR + xo;
    __builtin_amdgcn_sched_barrier(0);
    u64 fa[DB], fb[DB];
    static_for<(AH < DB ? AH : DB)>([&](auto I) {
      constexpr int d = decltype(I)::value;
      fa[d] = tr_read_asm<64 * (d >> 1)>((d & 1) ? ao : ae);
      fb[d] = tr_read_asm<64 * (d >> 1) + 16 * TSTR>((d & 1) ? ao : ae);
    });
    static_for<DB>([&](auto I) {
      constexpr int d = decltype(I)::value;
      constexpr int pend = (DB - 1 - d) < (AH - 1) ? (DB - 1 - d) : (AH - 1);          // fragments issued after fragment d
      lgkm_wait<2 * pend>(fa[d], fb[d]);
      const bf16x8 f = join_tr64(fa[d], fb[d]);
      acc[0][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, of[0], acc[0][d], 0, 0, 0);
      acc[1][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, of[1], acc[1][d], 0, 0, 0);
      if constexpr (d + AH < DB) {
        constexpr int e = d + AH;
        fa[e] = tr_read_asm<64 * (e >> 1)>((e & 1) ? ao : ae);
        fb[e] = tr_read_asm<64 * (e >> 1) + 16 * TSTR>((e & 1) ? ao : ae);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  };
  // The vector phases read their LDS inputs (row scalars; B: the P image) AHEAD, into registers, from a point where the wave has other
  // work to issue -- A before its dV MFMAs, B before the iteration's tile transfers -- instead of at the head of a dependent chain.
  f32x4 rs4[2]; u32x4 rw4[2], rb4[1]; f32x4 rp4[2][2];
  auto rows_a = [&](int t) {                               // role A: lse, keep words (block words) of tile t
    const float* lse_s = (const float*)(rows_base + (t % NBUF) * ROWS_BYTES);
    const unsigned* dw_s = (const unsigned*)(lse_s + 2 * QT) + pr * QT;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      rs4[j] = *(const f32x4*)(lse_s + 16 * j + 4 * g);
      rw4[j] = *(const u32x4*)(dw_s + 16 * j + 4 * g);
    }
  };
  auto rows_b = [&](int t) {                               // role B: delta, keep words and the P image of tile t
    const float* lse_s = (const float*)(rows_base + (t % NBUF) * ROWS_BYTES);
    const unsigned* dw_s = (const unsigned*)(lse_s + 2 * QT) + pr * QT;
    const unsigned char* xr = xch_base + (t & 1) * XCH + pr * 4096 + lane * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      rs4[j] = *(const f32x4*)(lse_s + QT + 16 * j + 4 * g);
      rw4[j] = *(const u32x4*)(dw_s + 16 * j + 4 * g);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) rp4[kb][j] = *(const f32x4*)(xr + (kb * 2 + j) * 1024);
    }
  };
  // role A: P and Pd of tile t from S; P goes to the exchange image, Pd stays as the next accumulating product's operand
  auto softmax_a = [&](int t) {
    unsigned char* xw = xch_base + (t & 1) * XCH + pr * 4096 + lane * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (BLK) rb4[0] = *(const u32x4*)(rows_base + (t % NBUF) * ROWS_BYTES + 4 * (2 * QT + pr * QT + 4 * QT + 16 * j + 4 * g));   // (in phase: registers)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int kbit = 16 * kb + n;
        f32x4 pv;
        float pd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // (masks ANDed into the value: a sign-extended one-bit field per key costs two VALU ops, a compare + select three and a VCC hazard)
          float prob = fast_exp2(fmaf(sacc[kb][j][i], sc, -rs4[j][i]));
          prob = __builtin_bit_cast(float, __builtin_bit_cast(int, prob) & okm[kb]);
          if (BLK) prob = __builtin_bit_cast(float, __builtin_bit_cast(int, prob) & ~bit_mask((int)rb4[0][i], kbit));
          pv[i] = prob;
          pd[i] = __builtin_bit_cast(float, __builtin_bit_cast(int, prob * dscale) & bit_mask((int)rw4[j][i], kbit));      // Pd
        }
        *(f32x4*)(xw + (kb * 2 + j) * 1024) = pv;
        u32x4 o = __builtin_bit_cast(u32x4, of[kb]);
        o[2 * j] = cvt_pk_bf16(pd[0], pd[1]); o[2 * j + 1] = cvt_pk_bf16(pd[2], pd[3]);
        of[kb] = as_bf16x8(o);
      }
    }
  };
  // role B: dS of tile t from P (exchange image) and dP; stored for the dQ kernel and kept as the next accumulating product's operand
  auto ds_b = [&](int T) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int kbit = 16 * kb + n;
        float ds[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float keep_scale = __builtin_bit_cast(float, __builtin_bit_cast(int, dscale) & bit_mask((int)rw4[j][i], kbit));
          ds[i] = rp4[kb][j][i] * fmaf(sacc[kb][j][i], keep_scale, -rs4[j][i]);                          // dS
        }
        u32x4 o = __builtin_bit_cast(u32x4, of[kb]);
        o[2 * j] = cvt_pk_bf16(ds[0], ds[1]); o[2 * j + 1] = cvt_pk_bf16(ds[2], ds[3]);
        of[kb] = as_bf16x8(o);
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const u32x4 v = __builtin_bit_cast(u32x4, of[kb]);
      const bool swp = ((n >> 2) & 1) != 0;
      const u32x4 w = {swp ? v[2] : v[0], swp ? v[3] : v[1], swp ? v[0] : v[2], swp ? v[1] : v[3]};
      *(u32x4*)(ds_row + kb * ds_kb_stride + (size_t)T * 1024) = w;
    }
  };
  // every transfer and LDS store of this wave has landed, then the workgroup meets (vmcnt counts the dS stores too)
  auto barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  // ---- the query tiles this workgroup visits (a scalar bit set; T0 / T1 / T2 = the tiles at sequence positions t, t + 1, t + 2) ----
  const int nts = __builtin_amdgcn_readfirstlane(ntiles);
  unsigned long long act = nts >= 64 ? ~0ull : ((1ull << nts) - 1ull);
  const bool listed = BLK && a.block_skip_k != nullptr && nts <= 64;
  if (listed) {
    const unsigned long long sk0 = ((const unsigned long long*)a.block_skip_k)[kblk];
    const unsigned long long sk = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sk0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)sk0);
    // the dQ kernel reads whole 64-key x 128-query regions of dS: a tile left out here is still part of regions it visits -- zeros
    if (role_b) {
      unsigned long long z = act & sk;
      while (z) {
        const int T = __builtin_ctzll(z);
        z &= z - 1ull;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) *(u32x4*)(ds_row + kb * ds_kb_stride + (size_t)T * 1024) = u32x4{0u, 0u, 0u, 0u};
      }
    }
    act &= ~sk;
  }
  const int nact = listed ? __builtin_popcountll(act) : nts;
  int seq = 0;                                             // tiles handed out so far (unlisted: the next tile)
  auto pop = [&]() -> int {
    if (!listed) return seq++;
    const int T = __builtin_ctzll(act);
    act &= act - 1ull;
    ++seq;
    return T;
  };
  int T0 = 0, T1 = 0, T2 = 0;
  if (nact > 0) {
    T0 = pop();
    dma(0, T0);
    rows_load(T0);
    rows_store(0, T0);
    if (nact > 1) { T1 = pop(); dma(1, T1); }
  }
  // (the builtin, not only the asm wait inside barrier(): hipcc must KNOW that the K / V fragment loads have landed, or it waits for them
  // -- vmcnt(0), and with them for the tile transfers in flight -- in front of the MFMAs of every iteration)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  barrier();
  if (nact > 0) {
    if (nact > 1) rows_load(T1);
    if (!role_b) rows_a(0);
    sx(0);
    if (!role_b) softmax_a(0);
    if (nact > 1) rows_store(1, T1);
  }
  barrier();
  for (int t = 0; t < ((TF_ABL_PAIR & 32) ? 0 : nact); ++t) {
    // tiles t, t + 1 are in LDS (tile t + 1's S / dP is done); buffer (t + 2) % 3 is free: everyone has left iteration t - 1
    if (t + 2 < nact) { T2 = pop(); if (!(TF_ABL_PAIR & 1)) dma(t + 2, T2); if (!(TF_ABL_PAIR & 16)) rows_load(T2); }
    if (role_b && !(TF_ABL_PAIR & 2)) { rows_b(t); ds_b(T0); }
    if (t + 1 < nact && !(TF_ABL_PAIR & 4)) sx(t + 1);
    if (!role_b && t + 1 < nact && !(TF_ABL_PAIR & 2)) rows_a(t + 1);
    if (!(TF_ABL_PAIR & 8)) accum(t);
    if (!role_b && t + 1 < nact && !(TF_ABL_PAIR & 2)) softmax_a(t + 1);
    if (t + 2 < nact && !(TF_ABL_PAIR & 16)) rows_store(t + 2, T2);
    barrier();
    T0 = T1; T1 = T2;
  }
  // ---- dV (A) / dK (B) rows.  A lane holds 4 consecutive head-dim elements per 16-element block (8 bytes): stored as they stand that is
  //      24 dwordx2 stores per lane, and the epilogue was store-ISSUE bound (21 of the kernel's 113 us).  Lanes g and g ^ 1 hold adjacent
  //      groups: they swap one group per PAIR of blocks (ds_swizzle, lane ^ 16), so that every lane stores 16 contiguous bytes -- the even
  //      lane of block 2m, the odd lane of block 2m + 1 -- and an instruction writes 64 contiguous bytes of each of its 16 rows.
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int key = key0 + 16 * kb + n;
    const bool ok = key < Sb && !((TF_ABL_PAIR & 64) && acc[kb][0][0] != 12345.f);
    u16* row = (u16*)a.dqkv + (sr.row0 + min(key, Sb - 1)) * a.ld_dqkv + (size_t)((role_b ? 1 : 2) * a.H + head) * HDP;      // B: dK (scaled), A: dV
    const float f = role_b ? a.scale : 1.0f;
    const bool odd = (g & 1) != 0;
#pragma unroll
    for (int m = 0; m < DB / 2; ++m) {
      const unsigned lo0 = cvt_pk_bf16(acc[kb][2 * m][0] * f, acc[kb][2 * m][1] * f), hi0 = cvt_pk_bf16(acc[kb][2 * m][2] * f, acc[kb][2 * m][3] * f);
      const unsigned lo1 = cvt_pk_bf16(acc[kb][2 * m + 1][0] * f, acc[kb][2 * m + 1][1] * f), hi1 = cvt_pk_bf16(acc[kb][2 * m + 1][2] * f, acc[kb][2 * m + 1][3] * f);
      // even lanes give away their group of block 2m + 1, odd lanes theirs of block 2m
      const unsigned rlo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(odd ? lo0 : lo1), 0x401F);      // lane ^ 16
      const unsigned rhi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(odd ? hi0 : hi1), 0x401F);
      const u32x4 v = odd ? u32x4{rlo, rhi, lo1, hi1} : u32x4{lo0, hi0, rlo, rhi};
      if (ok) TF_ST_ROW((u32x4*)(row + (2 * m + (odd ? 1 : 0)) * 16 + 4 * (g & 2)), v);
    }
  }
}

// ================================================================================================
// delta[b, h, q] = rowsum(dO . O) over the head's columns, one wave per token row (both tensors are read once, 16 B per lane)
// ================================================================================================
__global__ __launch_bounds__(256) void attn_delta_kernel(const TfAttnArgs a) {
  __shared__ float part[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = a.S, nrb = (S + 3) / 4;
  const int b = blockIdx.x / nrb, q = (blockIdx.x % nrb) * 4 + wave;
  const SampleRows sr = sample_rows(a.cu_rows, b, S);
  if (q >= sr.len) return;                                 // (no barrier below: waves are independent)
  const int cph = a.HDP / 8, nch = a.H * cph;             // 16-B chunks per head / per row (<= 256)
  const u16* orow = (const u16*)a.out + (sr.row0 + q) * a.ld_out;
  const u16* drow = (const u16*)a.dout + (sr.row0 + q) * a.ld_dout;
  for (int c = lane; c < nch; c += 64) {
    float of[8], df[8];
    unpack8(*(const u32x4*)(orow + c * 8), of);
    unpack8(*(const u32x4*)(drow + c * 8), df);
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
// backward, dQ from the dS tiles the dK / dV kernel wrote (attn_bwd_dkv16_kernel<.., DS = true>):
//   dQ^T[d][q] += K^T[d][key] . dS^T[key][q];  dQ = scale * dQ^T^T
// One third of the matrix work of attn_bwd_dq16_kernel and none of its softmax: S and dP are computed once per layer.  4 waves x 32
// queries per workgroup, two workgroups per CU; per 64-key tile the K tile is staged once for the workgroup (as in the dq16 kernel)
// and every wave copies the four 1-KiB chunks of ITS 32 queries into a wave-private LDS region, from which ds_read_b64_tr_b16 hands
// each lane (query n) the 8 keys of its B-operand fragment -- the transposition the chunk layout was built for.
// ================================================================================================
template <int HDP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_ds_kernel(const TfAttnArgs a) {
  using G = Geo<HDP>;
  constexpr int NT = 256, DB = HDP / 16, TSTR = G::TSTR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, n = lane & 15;
  unsigned char* kt = smem;
  unsigned char* dsw = smem + 64 * TSTR + wave * 4096;
  const int S = a.S;
  const int nqb = (S + 127) / 128;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = pair_of_group(logical / nqb, a.B * a.H), b = bh / a.H, head = bh % a.H;
  const int qblk = logical % nqb;
  const SampleRows sr = sample_rows(a.cu_rows, b, S);
  const int Sb = sr.len;
  if (qblk * 128 >= Sb) return;                          // query blocks past the sample's end (workgroup-uniform)
  const size_t ld = a.ld_qkv;
  const u16* kbase = (const u16*)a.qkv + sr.row0 * ld + (size_t)(1 * a.H + head) * HDP;
  const int ntiles = (valid_key_limit(a.key_mask, b, Sb, lane) + 63) / 64;
  const int qb32 = 4 * qblk + wave;
  // the dK / dV kernels walk the queries in tiles of 32 (pair kernel) or TF_DKV16_QT rows: sub-tiles at or beyond ceil(Sb / 32) * 32 hold no
  // query of this sample (and the pair kernel never wrote them), so a wave that owns one has nothing to compute (it still takes part in
  // the barriers)
  const bool active = qb32 * 32 < ((Sb + 31) / 32) * 32;
  const unsigned char* ds_col = (const unsigned char*)a.ds_work + (((size_t)bh * ds_nkb(S)) * ds_nqb(S) + qb32) * 1024 + 16 * lane;
  const size_t ds_kstride = (size_t)ds_nqb(S) * 1024;     // from one 16-key block to the next

  f32x4 dq[2][DB];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int d = 0; d < DB; ++d) dq[j][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane LDS addresses (as in attn_bwd_dq16_kernel): transposed K^T reads, and the chunk reads
  const int q4 = n >> 2, p = n & 3, fz = swz16(4 * g);
  const int tbase = (4 * g + q4) * TSTR + 8 * (p & 1);
  const int xe = ((p >> 1) ^ fz) << 4, xo = ((2 + (p >> 1)) ^ fz) << 4;
  const int cbase = 64 * (4 * g + q4) + 16 * p;          // chunk row of key 4g + q4, the 16-B slot of the producer lanes that held queries 4p..4p+3

  TileRegs16<64, HDP, NT> kr;
  u32x4 cr[4];
  // the key tiles this workgroup visits: with a block mask, not those blocked for every query of the block (TfAttnArgs.block_skip_q:
  // their dS is exactly 0 -- the dK / dV kernel writes zeros or nothing there).  t = the tile in hand, tn = the next one.
  const int nts = __builtin_amdgcn_readfirstlane(ntiles);
  unsigned long long act = nts >= 64 ? ~0ull : ((1ull << nts) - 1ull);
  // (gated on the block-bit matrix like the forward and the dK / dV pair kernel: a skip map in a TfAttnArgs WITHOUT block_bits -- a struct
  // reused across calls -- must not drop tiles; check() rejects that pairing before any launch)
  const bool listed = a.block_bits != nullptr && a.block_skip_q != nullptr && nts <= 64;
  if (listed) {
    const unsigned long long sk = ((const unsigned long long*)a.block_skip_q)[qblk];
    act &= ~(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sk >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)sk));
  }
  int t = listed ? (act ? __builtin_ctzll(act) : -1) : (nts > 0 ? 0 : -1);
  if (t >= 0) {
    kr.load(kbase, ld, t * 64, Sb - 1, tid);
    if (active) {
#pragma unroll
      for (int c = 0; c < 4; ++c) cr[c] = *(const u32x4*)(ds_col + (size_t)(4 * t + c) * ds_kstride);
    }
  }
  while (t >= 0) {
    int tn;
    if (listed) { act &= act - 1ull; tn = act ? __builtin_ctzll(act) : -1; }
    else tn = t + 1 < nts ? t + 1 : -1;
    __syncthreads();                                     // every wave is done with the previous K tile
    kr.store(kt, tid);
    if (active) {
#pragma unroll
      for (int c = 0; c < 4; ++c) *(u32x4*)(dsw + c * 1024 + 16 * lane) = cr[c];
    }
    __syncthreads();
    if (tn >= 0) {
      kr.load(kbase, ld, tn * 64, Sb - 1, tid);
      if (active) {
#pragma unroll
        for (int c = 0; c < 4; ++c) cr[c] = *(const u32x4*)(ds_col + (size_t)(4 * tn + c) * ds_kstride);
      }
    }
    if (active) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        bf16x8 bfr[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned char* ca = dsw + (2 * hh) * 1024 + cbase + 8 * (j ^ (g & 1));
          bfr[j] = join_tr(lds_read_tr16(ca), lds_read_tr16(ca + 1024));
        }
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const unsigned char* tp = kt + tbase + (32 * hh) * TSTR + 64 * (d >> 1) + ((d & 1) ? xo : xe);
          const bf16x8 af = join_tr(lds_read_tr16(tp), lds_read_tr16(tp + 16 * TSTR));
          dq[0][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[0], dq[0][d], 0, 0, 0);
          dq[1][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[1], dq[1][d], 0, 0, 0);
        }
      }
    }
    t = tn;
  }
#if TF_DQ_WIDE_STORE
  // lane group g holds quarter g (4 columns) of every 16-column block: neighbouring groups swap one block per PAIR of blocks
  // (v_permlane16_swap_b32: odd rows of the first register <-> even rows of the second), so that even groups keep block 2m, odd groups
  // block 2m + 1, two adjacent quarters each: 16-byte stores, half as many
  static_assert(DB % 2 == 0, "blocks are exchanged in pairs");
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qrow = qblk * 128 + wave * 32 + 16 * j + n;
    u16* orow = (u16*)a.dqkv + (sr.row0 + min(qrow, Sb - 1)) * a.ld_dqkv + (size_t)(0 * a.H + head) * HDP;
#pragma unroll
    for (int m = 0; m < DB / 2; ++m) {
      const unsigned e0 = pack2bf(dq[j][2 * m][0] * a.scale, dq[j][2 * m][1] * a.scale), e1 = pack2bf(dq[j][2 * m][2] * a.scale, dq[j][2 * m][3] * a.scale);
      const unsigned o0 = pack2bf(dq[j][2 * m + 1][0] * a.scale, dq[j][2 * m + 1][1] * a.scale), o1 = pack2bf(dq[j][2 * m + 1][2] * a.scale, dq[j][2 * m + 1][3] * a.scale);
      const u32x2 s0 = __builtin_amdgcn_permlane16_swap(e0, o0, false, false);
      const u32x2 s1 = __builtin_amdgcn_permlane16_swap(e1, o1, false, false);
      const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
      if (qrow < Sb) TF_ST_ROW((u32x4*)(orow + (2 * m + (g & 1)) * 16 + 8 * (g >> 1)), v);
    }
  }
#else
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qrow = qblk * 128 + wave * 32 + 16 * j + n;
    if (qrow < Sb) {
      u16* orow = (u16*)a.dqkv + (sr.row0 + qrow) * a.ld_dqkv + (size_t)(0 * a.H + head) * HDP;
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        u32x2 v;
        v[0] = pack2bf(dq[j][d][0] * a.scale, dq[j][d][1] * a.scale);
        v[1] = pack2bf(dq[j][d][2] * a.scale, dq[j][d][3] * a.scale);
        TF_ST_ROW((u32x2*)(orow + d * 16 + 4 * g), v);
      }
    }
  }
#endif
}

// ================================================================================================
// block-sparse tile maps of a block mask (TfAttnArgs.block_skip_q / block_skip_k), one workgroup per 128-row / 128-key block
// ================================================================================================
__global__ __launch_bounds__(256) void attn_block_skip_kernel(const unsigned long long* __restrict__ bits, int S, unsigned long long* __restrict__ skip_q,
                                                              unsigned long long* __restrict__ skip_k) {
  __shared__ unsigned long long word;
  const int SW = (S + 63) / 64, nb = (S + 127) / 128, nqt = (S + 31) / 32;
  const int blk = blockIdx.x % nb;
  const bool is_k = (int)blockIdx.x >= nb;
  if (threadIdx.x == 0) word = 0ull;
  __syncthreads();
  if (!is_k) {
    // bit t: all 64 keys of tile t blocked for every query row of this block (rows past S do not exist)
    if (SW <= 64) {
      for (int t = 0; t < SW; ++t) {
        bool all = true;
        for (int r = blk * 128 + (int)threadIdx.x; r < min(S, blk * 128 + 128); r += 256) all = all && bits[(size_t)r * SW + t] == ~0ull;
        if (__syncthreads_and(all ? 1 : 0) && threadIdx.x == 0) word |= 1ull << t;
      }
    }
    __syncthreads();
    // never every tile of a query block: a block whose rows attend nothing (softmax over the empty set) keeps tile 0, so the kernels
    // walk the same code as with element-wise masking and produce the same (undefined-in-the-reference, NaN there) rows bit for bit
    if (threadIdx.x == 0) skip_q[blk] = (SW <= 64 && word == (SW == 64 ? ~0ull : (1ull << SW) - 1ull)) ? (word & ~1ull) : word;
  } else {
    // bit j: the (up to) 128 keys of this block -- words 2 blk, 2 blk + 1 -- blocked for all 32 query rows of tile j
    if (nqt <= 64) {
      const int w0 = 2 * blk, w1 = 2 * blk + 1;
      for (int j0 = 0; j0 < nqt; j0 += 8) {                          // a half wave per query tile
        const int j = j0 + ((int)threadIdx.x >> 5), r = j * 32 + ((int)threadIdx.x & 31);
        bool ok = true;
        if (j < nqt && r < S) ok = bits[(size_t)r * SW + w0] == ~0ull && (w1 >= SW || bits[(size_t)r * SW + w1] == ~0ull);
        const unsigned long long bal = __ballot(ok);
        const unsigned half = (threadIdx.x & 32) ? (unsigned)(bal >> 32) : (unsigned)bal;
        if ((threadIdx.x & 31) == 0 && j < nqt && half == 0xffffffffu) atomicOr(&word, 1ull << j);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) skip_k[blk] = word;
  }
}
extern "C" int tf_launch_attn_block_skip(const void* block_bits, int S, void* skip_q, void* skip_k, hipStream_t st) {
  if (block_bits == nullptr || skip_q == nullptr || skip_k == nullptr || S <= 0) return -1;
  TfTraceScope tr("attn_block_skip_kernel", st);
  hipLaunchKernelGGL(attn_block_skip_kernel, dim3(2 * ((S + 127) / 128)), dim3(256), 0, st, (const unsigned long long*)block_bits, S,
                     (unsigned long long*)skip_q, (unsigned long long*)skip_k);
  return (int)hipGetLastError();
}

template <int HDP> int launch_fwd(const TfAttnArgs* a, hipStream_t st) {
  const size_t lds = 128 * Geo<HDP>::TSTR;
  static const hipError_t once = hipFuncSetAttribute((const void*)attn_fwd_kernel<HDP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static const hipError_t onceb = hipFuncSetAttribute((const void*)attn_fwd_kernel<HDP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)once; (void)onceb;
  char nm[56];
  snprintf(nm, sizeof(nm), "attn_fwd_kernel<%d>", HDP);
  TfTraceScope tr(nm, st, 4.0 * a->B * a->H * (double)(a->q != nullptr ? a->Sq : a->S) * a->S * HDP);
  const int Sq = a->q != nullptr ? a->Sq : a->S;
  const dim3 grid(((Sq + 127) / 128) * a->B * a->H);
  if (a->block_bits != nullptr) hipLaunchKernelGGL((attn_fwd_kernel<HDP, true>), grid, dim3(256), lds, st, *a);
  else hipLaunchKernelGGL((attn_fwd_kernel<HDP, false>), grid, dim3(256), lds, st, *a);
  return (int)hipGetLastError();
}
// part: 0 = dQ then dK / dV, 1 = dQ only (also fills `delta`), 2 = dK / dV only (after a part-1 launch on the same stream)
template <int HDP> int launch_bwd(const TfAttnArgs* a, hipStream_t st, int part) {
  const size_t lds_q = 128 * Geo<HDP>::TSTR, lds_kv = 64 * Geo<HDP>::TSTR + 256 + 1024;
  static const hipError_t once_q = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<HDP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
  static const hipError_t once_kv = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<HDP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
  static const hipError_t once_kvb = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<HDP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
  (void)once_q; (void)once_kv; (void)once_kvb;
  const bool cross = a->q != nullptr;                    // own query set: the 32-row kernels (the 16-row ones assume the packed layout)
  const int Sq = cross ? a->Sq : a->S;
  const dim3 grid_q(((Sq + 127) / 128) * a->B * a->H), grid(((a->S + 127) / 128) * a->B * a->H);
  // credited work: backward = 2x forward = four S x S x hd products; the recomputed St / dPt are not credited
  const double fl = 4.0 * a->B * a->H * (double)Sq * a->S * HDP;
  char nm[56];
  // The one-pass 16-row (two waves per SIMD) dQ / dK+dV kernels exist for head dims <= 192 only: at 224 / 256 their resident fragments
  // do not fit 256 registers.  With a dS workspace those widths still run at two waves per SIMD -- dV and dK as two passes of the
  // 16-row kernel (WHICH = 0 / 1) and the thin dQ kernel; without one they take the 32-row kernels (one wave per SIMD).
  constexpr bool HAS16 = HDP <= 192;
  static const int dq16 = TF_ENV_INT("TF_ATTN_DQ16", 1);
  static const int dkv16 = TF_ENV_INT("TF_ATTN_DKV16", 1);
  bool done_q = part == 2, done_kv = part == 1;
  {
    constexpr int QT = TF_DKV16_QT;
    static const int use_ds = TF_ENV_INT("TF_ATTN_DS", 1);     // A/B switch
    if (part == 0 && use_ds && dq16 && dkv16 && !cross && a->ds_work != nullptr) {
      // S and dP once: delta -> dK / dV (+ dS tiles) -> dQ = dS . K
      const size_t lds_kv16d = 4 * QT * Geo<HDP>::TSTR + 2 * (72 * QT), lds_qd = 64 * Geo<HDP>::TSTR + 4 * 4096;
      static const hipError_t o3 = hipFuncSetAttribute((const void*)attn_bwd_dq_ds_kernel<HDP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_qd);
      (void)o3;
      {
        TfTraceScope tr("attn_delta_kernel", st, 0.0, 4.0 * a->B * a->S * a->H * HDP);
        hipLaunchKernelGGL(attn_delta_kernel, dim3(a->B * ((a->S + 3) / 4)), dim3(256), 0, st, *a);
      }
      static const int use_pair = TF_ENV_INT("TF_ATTN_PAIR", 1);
      if constexpr (HAS16) {
        if (use_pair) {
          constexpr int PQ = TF_DKV_PAIR_QT;
          constexpr int PAIRB = 2 * PQ * Geo<HDP>::TSTR, NPCP = (PAIRB + 1023) / 1024;
          const size_t lds_pair = ((3 * PAIRB + 1023) / 1024) * 1024 + 3 * (4 * PQ * 10) + 2 * (4 * 4096) + (8 * ((NPCP + 7) / 8) - NPCP) * 1024;
          static const hipError_t p1 = hipFuncSetAttribute((const void*)attn_bwd_dkv_pair_kernel<HDP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair);
          static const hipError_t p2 = hipFuncSetAttribute((const void*)attn_bwd_dkv_pair_kernel<HDP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair);
          (void)p1; (void)p2;
          snprintf(nm, sizeof(nm), "attn_bwd_dkv_pair_kernel<%d>", HDP);
          TfTraceScope tr(nm, st, 1.5 * fl);               // credited: dP, dV, dK (the S recompute is not)
          if (a->block_bits != nullptr) hipLaunchKernelGGL((attn_bwd_dkv_pair_kernel<HDP, true>), grid, dim3(512), lds_pair, st, *a);
          else hipLaunchKernelGGL((attn_bwd_dkv_pair_kernel<HDP, false>), grid, dim3(512), lds_pair, st, *a);
        } else {
        static const hipError_t o1 = hipFuncSetAttribute((const void*)attn_bwd_dkv16_kernel<HDP, false, QT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv16d);
        static const hipError_t o2 = hipFuncSetAttribute((const void*)attn_bwd_dkv16_kernel<HDP, true, QT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv16d);
        (void)o1; (void)o2;
        snprintf(nm, sizeof(nm), "attn_bwd_dkv16_kernel<%d, dS>", HDP);
        TfTraceScope tr(nm, st, 1.5 * fl);                 // credited: dP, dV, dK (the S recompute is not)
        if (a->block_bits != nullptr) hipLaunchKernelGGL((attn_bwd_dkv16_kernel<HDP, true, QT, true>), grid, dim3(512), lds_kv16d, st, *a);
        else hipLaunchKernelGGL((attn_bwd_dkv16_kernel<HDP, false, QT, true>), grid, dim3(512), lds_kv16d, st, *a);
        }
      } else {
        static const hipError_t o1 = hipFuncSetAttribute((const void*)attn_bwd_dkv16_kernel<HDP, false, QT, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv16d);
        static const hipError_t o2 = hipFuncSetAttribute((const void*)attn_bwd_dkv16_kernel<HDP, true, QT, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv16d);
        static const hipError_t o4 = hipFuncSetAttribute((const void*)attn_bwd_dkv16_kernel<HDP, false, QT, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv16d);
        static const hipError_t o5 = hipFuncSetAttribute((const void*)attn_bwd_dkv16_kernel<HDP, true, QT, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv16d);
        (void)o1; (void)o2; (void)o4; (void)o5;
        {
          snprintf(nm, sizeof(nm), "attn_bwd_dkv16_kernel<%d, dV>", HDP);
          TfTraceScope tr(nm, st, 0.5 * fl);               // credited: dV
          if (a->block_bits != nullptr) hipLaunchKernelGGL((attn_bwd_dkv16_kernel<HDP, true, QT, false, 0>), grid, dim3(512), lds_kv16d, st, *a);
          else hipLaunchKernelGGL((attn_bwd_dkv16_kernel<HDP, false, QT, false, 0>), grid, dim3(512), lds_kv16d, st, *a);
        }
        {
          snprintf(nm, sizeof(nm), "attn_bwd_dkv16_kernel<%d, dK, dS>", HDP);
          TfTraceScope tr(nm, st, 1.0 * fl);               // credited: dP, dK
          if (a->block_bits != nullptr) hipLaunchKernelGGL((attn_bwd_dkv16_kernel<HDP, true, QT, true, 1>), grid, dim3(512), lds_kv16d, st, *a);
          else hipLaunchKernelGGL((attn_bwd_dkv16_kernel<HDP, false, QT, true, 1>), grid, dim3(512), lds_kv16d, st, *a);
        }
      }
      {
        snprintf(nm, sizeof(nm), "attn_bwd_dq_ds_kernel<%d>", HDP);
        TfTraceScope tr(nm, st, 0.5 * fl);
        hipLaunchKernelGGL(attn_bwd_dq_ds_kernel<HDP>, grid, dim3(256), lds_qd, st, *a);
      }
      return (int)hipGetLastError();
    }
  }
  if constexpr (HAS16) {
    constexpr int QT = TF_DKV16_QT;
    const size_t lds_q16 = 256 * Geo<HDP>::TSTR;        // two K/V tile pairs
    const size_t lds_kv16 = 4 * QT * Geo<HDP>::TSTR + 2 * (72 * QT);
    static const hipError_t once_q16 = hipFuncSetAttribute((const void*)attn_bwd_dq16_kernel<HDP, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q16);
    static const hipError_t once_kv16 = hipFuncSetAttribute((const void*)attn_bwd_dkv16_kernel<HDP, false, QT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv16);
    static const hipError_t once_kv16b = hipFuncSetAttribute((const void*)attn_bwd_dkv16_kernel<HDP, true, QT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv16);
    (void)once_q16; (void)once_kv16; (void)once_kv16b;
    if (!done_q && dq16 && !cross) {
      snprintf(nm, sizeof(nm), "attn_bwd_dq16_kernel<%d>", HDP);
      TfTraceScope tr(nm, st, fl);
      hipLaunchKernelGGL((attn_bwd_dq16_kernel<HDP, 8>), grid, dim3(512), lds_q16, st, *a);
      done_q = true;
    }
    if (part == 1) return (int)hipGetLastError();
    if (!done_kv && dkv16 && !cross) {
      if (!done_q) {                                       // TF_ATTN_DQ16=0 with the 16-row dK / dV kernel: dQ (and delta) first
        snprintf(nm, sizeof(nm), "attn_bwd_dq_kernel<%d>", HDP);
        TfTraceScope tr(nm, st, fl);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<HDP>, grid_q, dim3(256), lds_q, st, *a);
        done_q = true;
      }
      snprintf(nm, sizeof(nm), "attn_bwd_dkv16_kernel<%d>", HDP);
      TfTraceScope tr(nm, st, fl);
      if (a->block_bits != nullptr) hipLaunchKernelGGL((attn_bwd_dkv16_kernel<HDP, true, QT>), grid, dim3(512), lds_kv16, st, *a);
      else hipLaunchKernelGGL((attn_bwd_dkv16_kernel<HDP, false, QT>), grid, dim3(512), lds_kv16, st, *a);
      done_kv = true;
    }
  }
  if (!done_q) {
    snprintf(nm, sizeof(nm), "attn_bwd_dq_kernel<%d>", HDP);
    TfTraceScope tr(nm, st, fl);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<HDP>, grid_q, dim3(256), lds_q, st, *a);
  }
  if (part == 1) return (int)hipGetLastError();
  if (!done_kv) {
    snprintf(nm, sizeof(nm), "attn_bwd_dkv_kernel<%d>", HDP);
    TfTraceScope tr(nm, st, fl);
    if (a->block_bits != nullptr) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDP, true>), grid, dim3(256), lds_kv, st, *a);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDP, false>), grid, dim3(256), lds_kv, st, *a);
  }
  return (int)hipGetLastError();
}

}  // namespace
extern "C" size_t tf_attn_ds_bytes(int B, int H, int S) {
  if (B <= 0 || H <= 0 || S <= 0) return 0;
  return (size_t)B * H * ds_nkb(S) * ds_nqb(S) * 1024 + 4096;
}
namespace {

int check(const TfAttnArgs* a) {
  if (a->B <= 0 || a->S <= 0 || a->H <= 0) return 1;
  if ((a->ld_qkv % 8) || (a->ld_out % 8)) return -2;
  if ((long long)a->B * a->H * a->S * a->S >= (1ll << 32) && a->drop_thr) return -5;   // 32-bit dropout index space
  if (a->drop_thr && a->drop_bits == nullptr) return -6;
  if (a->q != nullptr && (a->Sq <= 0 || (a->ld_q % 8) || a->block_bits != nullptr)) return -7;   // cross attention: own query rows, no block mask
  if (a->cu_rows != nullptr && (a->key_mask != nullptr || a->q != nullptr)) return -8;           // packed batches hold real tokens only
  if ((a->block_skip_q != nullptr || a->block_skip_k != nullptr) && a->block_bits == nullptr) return -9;   // tile maps belong to a block mask
  return 0;
}

}  // namespace

#define TF_ATTN_DISPATCH(FN)                     \
  switch (a->HDP) {                              \
    case 32: return FN<32>(a, st);               \
    case 64: return FN<64>(a, st);               \
    case 96: return FN<96>(a, st);               \
    case 128: return FN<128>(a, st);             \
    case 160: return FN<160>(a, st);             \
    case 192: return FN<192>(a, st);             \
    case 224: return FN<224>(a, st);             \
    case 256: return FN<256>(a, st);             \
    default: return -3;                          \
  }

#define TF_ATTN_DISPATCH_PART(FN, P)                     \
  switch (a->HDP) {                              \
    case 32: return FN<32>(a, st, P);               \
    case 64: return FN<64>(a, st, P);               \
    case 96: return FN<96>(a, st, P);               \
    case 128: return FN<128>(a, st, P);             \
    case 160: return FN<160>(a, st, P);             \
    case 192: return FN<192>(a, st, P);             \
    case 224: return FN<224>(a, st, P);             \
    case 256: return FN<256>(a, st, P);             \
    default: return -3;                          \
  }


extern "C" int tf_launch_attn_fwd(const TfAttnArgs* a, hipStream_t st) {
  const int c = check(a);
  if (c) return c > 0 ? 0 : c;
  if (a->qkv_lo != nullptr) return tf_launch_attn_fwd_x3(a, st);
  TF_ATTN_DISPATCH(launch_fwd)
}
extern "C" int tf_launch_attn_bwd(const TfAttnArgs* a, hipStream_t st) {
  const int c = check(a);
  if (c) return c > 0 ? 0 : c;
  if ((a->ld_dout % 8) || (a->ld_dqkv % 8) || a->delta == nullptr || a->lse == nullptr) return -2;
  if (a->q != nullptr && (a->dq == nullptr || (a->ld_dq % 8))) return -7;
  if (a->qkv_lo != nullptr) return tf_launch_attn_bwd_x3(a, st);
  TF_ATTN_DISPATCH_PART(launch_bwd, 0)
}
// the two halves of tf_launch_attn_bwd as separate launches (the encoder runtime starts the Q rows of the in-proj weight gradient
// between them); the fp32-accuracy kernels run whole in part 1
extern "C" int tf_launch_attn_bwd_part(const TfAttnArgs* a, int part, hipStream_t st) {
  if (part != 1 && part != 2) return -2;
  const int c = check(a);
  if (c) return c > 0 ? 0 : c;
  if ((a->ld_dout % 8) || (a->ld_dqkv % 8) || a->delta == nullptr || a->lse == nullptr) return -2;
  if (a->q != nullptr && (a->dq == nullptr || (a->ld_dq % 8))) return -7;
  if (a->qkv_lo != nullptr) return part == 1 ? tf_launch_attn_bwd_x3(a, st) : 0;
  TF_ATTN_DISPATCH_PART(launch_bwd, part)
}
