// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the cross-fusion hot path.
// Wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef unsigned short u16;

#define TF_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define TF_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- XCD-aware workgroup order -------------------------------------------------------------------
// Bijective remap: blocks b, b+8, b+16.. share an XCD (round-robin dispatch); give each
// XCD a contiguous range of logical tiles so that tiles sharing an A panel hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// MI = 16-row fragments per wave along M: the tile is (32*MI) x 128, i.e. 128 / 160 / 192 rows.  All three run two
// workgroups per CU; the host picks MI per launch to minimise (rounds over 512 slots) x (tile height) -- at M = 22656
// a 128-row tiling of an N = 768 GEMM needs 3 rounds (1062 tiles) where 160-row tiles need 2 (852).

// ---- bf16 <-> f32 -----------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(u16 b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }  // RNE, NaN-safe (v_cvt_pk_bf16_f32)
// two floats -> packed bf16 pair in ONE v_cvt_pk_bf16_f32 (element-wise conversions of a vector compile to one instruction per
// element plus a v_perm per pair)
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// x where bit BIT of word is set, else +0: the bit as a sign-extended field ANDed into the value (two VALU ops; written as C the
// compiler turns it back into and / compare / select, three ops and a VCC hazard)
template <int BIT> __device__ __forceinline__ float and_bit(float x, int word) {
  int m;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "n"(BIT));
  return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & m);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }

// 8 bf16 packed in a u32x4 -> 8 floats
__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
    f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
  return v;
}

// ---- hi + lo bf16 plane pairs (fp32-accuracy mode): value = float(hi) + float(lo), 16 significant bits --------------
__device__ __forceinline__ void split8(const float (&f)[8], u32x4& hi, u32x4& lo) {
  float r[8];
  hi = pack8(f);
  unpack8(hi, r);
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = f[e] - r[e];      // exact in fp32
  lo = pack8(r);
}
__device__ __forceinline__ void join8(const u32x4& hi, const u32x4& lo, float (&f)[8]) {
  float l[8];
  unpack8(hi, f);
  unpack8(lo, l);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] += l[e];
}
// 8 values of a bf16 tensor with an optional lo plane at the same element offset
__device__ __forceinline__ void load8_split(const void* hi, const void* lo, size_t off, float (&f)[8]) {
  if (lo != nullptr) join8(*(const u32x4*)((const u16*)hi + off), *(const u32x4*)((const u16*)lo + off), f);
  else unpack8(*(const u32x4*)((const u16*)hi + off), f);
}
__device__ __forceinline__ void store8_split(void* hi, void* lo, size_t off, const float (&f)[8]) {
  if (lo != nullptr) {
    u32x4 h, l;
    split8(f, h, l);
    *(u32x4*)((u16*)hi + off) = h;
    *(u32x4*)((u16*)lo + off) = l;
  } else {
    *(u32x4*)((u16*)hi + off) = pack8(f);
  }
}

// ---- counter-based dropout RNG ----------------------------------------------------------------
// keep(idx) is a pure function of (key, element index): forward and backward regenerate the same mask from
// the element's logical index whatever the fragment layout, and tests can replay it (tf_dropout_mask).
// One 32-bit mixing hash serves a PAIR of consecutive indices (its two 16-bit halves); element idx is DROPPED
// when its half < thr16, thr16 = round(p * 65536).  (32-bit integer multiplies are quarter rate: hashing once
// per pair, and not at all inside the attention kernels -- they read a precomputed bitmask -- keeps dropout off
// the critical VALU path.)
__device__ __forceinline__ unsigned tf_hash32(unsigned x, unsigned key) {
  unsigned h = x + key;                       // "lowbias32" finaliser: two (quarter-rate) multiplies, full avalanche
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return h;
}
// The step clock (tf_clock_ptr / tf_clock_advance): a device word every dropout-drawing kernel folds into its key at entry.  It stays
// 0 unless the caller advances it, and mix(0) = 0, so eager callers -- who hand over a fresh key per call -- see no change; a step
// captured in a HIP graph bakes its keys into the graph, advances the clock as its first node, and so draws new masks on every
// replay.  One copy of the pointer per translation unit (no relocatable device code): tf_tu_set_clock, called once by tf_clock_ptr.
static __device__ const unsigned* g_tf_clock = nullptr;
__device__ __forceinline__ unsigned tf_salted(unsigned key) {
  const unsigned* c = g_tf_clock;
  unsigned h = c != nullptr ? *c : 0u;
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return (unsigned)__builtin_amdgcn_readfirstlane((int)(key ^ h));     // wave-uniform by construction: keep it in a scalar register
}
#define TF_TU_SET_CLOCK(name) extern "C" int name(const unsigned* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tf_clock), &p, sizeof(p)); }
__device__ __forceinline__ bool tf_keep(unsigned idx, unsigned key, unsigned thr16) {
  const unsigned h = tf_hash32(idx >> 1, key);
  return ((idx & 1u) ? (h >> 16) : (h & 0xffffu)) >= thr16;
}
// keep bits of 8 consecutive elements starting at an EVEN index (4 hashes); bit e = element base + e
__device__ __forceinline__ unsigned tf_keep8(unsigned base, unsigned key, unsigned thr16) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned h = tf_hash32((base >> 1) + i, key);
    m |= ((h & 0xffffu) >= thr16 ? 1u : 0u) << (2 * i);
    m |= ((h >> 16) >= thr16 ? 1u : 0u) << (2 * i + 1);
  }
  return m;
}

// ---- exact (erf) GELU and its derivative -------------------------------------------------------
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 output rounding of 2^-9): one v_rcp,
// one v_exp and five FMAs instead of libm's branchy erff; GELU and GELU' share the same exponential,
// exp(-(x/sqrt2)^2) = exp(-x^2/2).
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& ex) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);   // v_rcp_f32 (1 ulp); __frcp_rn expands to a 12-instruction IEEE divide
  ex = __expf(-ax * ax);                                   // = exp(-x^2 / 2)
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float erf_abs = 1.0f - poly * ex;
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float gelu_f(float x) { float c, e; gelu_parts(x, c, e); return x * c; }
__device__ __forceinline__ float gelu_grad_f(float x) {
  float c, e; gelu_parts(x, c, e);
  return c + x * 0.39894228040143268f * e;
}

// ---- wave reductions ----------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// transposed LDS read: lane i of each 16-lane group receives column i of a 4-row x 16-col block of
// 16-bit elements; lane 4q+p of the group supplies the address of row q, columns 4p..4p+3.
__device__ __forceinline__ s16x4 lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_addr);
}

// ---- transposed LDS reads issued through inline asm ---------------------------------------------------------------
// hipcc orders every LDS-DMA (global_load_lds) before any LATER ds_read_b64_tr_b16 *intrinsic* with s_waitcnt vmcnt(0)
// (it cannot prove the slots differ), which drains a multi-slot DMA ring every phase.  Reads issued from inline asm are
// invisible to that pass; the kernel then owns the ordering: tr_read_asm(...) x N, then tr_wait_asm<...>() naming every
// destination (so no consumer or register copy can be scheduled above the wait) + sched_barrier(0).
typedef unsigned long long u64;
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
template <int OFF> __device__ __forceinline__ u64 tr_read_asm(unsigned addr) {
  u64 d;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
  return d;
}
__device__ __forceinline__ bf16x8 join_tr64(u64 lo, u64 hi) {
  typedef __attribute__((ext_vector_type(2))) u64 u64x2;
  u64x2 v = {lo, hi};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 join_tr(const s16x4& a, const s16x4& b) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}
