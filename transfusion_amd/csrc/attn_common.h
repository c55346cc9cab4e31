// Helpers shared by the attention kernels (attn_bf16.hip: bf16 compute; attn_x3.hip: fp32-accuracy mode on hi + lo planes):
// dual-use LDS tile geometry, operand fragment reads, key-validity scans, the (batch, head) -> XCD mapping.
#pragma once
#include "tf_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_THR = 4.0f;   // log2 domain: O/l are rescaled only when some row's max grew by > 2^4 (T13, exact math)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // raw v_exp_f32 (x <= ~4 here)

template <int HDP> struct Geo {
  static constexpr int CPR = HDP / 8;                                   // 16-B chunks per row
  static constexpr int TSTR = ((2 * HDP - 64 + 255) / 256) * 256 + 64;  // row stride in bytes
  static constexpr int KSTEPS = HDP / 16;
  static constexpr int DBLK = HDP / 32;
};

__device__ __forceinline__ int tile_off(int row, int chunk, int tstr) { return row * tstr + ((chunk ^ ((row >> 2) & 3)) << 4); }

// cooperative global -> registers -> LDS tile copy (ROWS x HDP bf16), 256 threads
template <int ROWS, int HDP> struct TileRegs {
  static constexpr int TOTAL = ROWS * (HDP / 8);
  static constexpr int PER = (TOTAL + 255) / 256;
  u32x4 v[PER];
  // row r of the tile comes from global row min(row0 + r, row_max) (clamped) or zeros when zero_fill && row0 + r > row_max
  __device__ __forceinline__ void load(const u16* __restrict__ base, size_t ld, int row0, int row_max, bool zero_fill, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = i * 256 + tid;
      if (TOTAL % 256 == 0 || id < TOTAL) {
        const int r = id / (HDP / 8), c = id % (HDP / 8);
        const int gr = row0 + r;
        // wave-uniform base + a 32-bit per-lane byte offset (a sample's rows span far less than 4 GiB): one offset register per
        // piece.  As 64-bit per-lane pointers the loop-invariant parts were hoisted out of the tile loop -- 2 x 6 pointer pairs at head
        // dim 192 -- and spilled there (11-15 VGPRs in attn_fwd_kernel<192>).
        // (branch-free: the clamped row is always fetched and a select zeroes it -- an exec-masked branch around every piece kept one
        // loop-invariant offset register per piece alive and put a scratch reload + vmcnt(0) into the tile loop of the dK / dV kernels)
        v[i] = *(const u32x4*)((const unsigned char*)base + ((unsigned)min(gr, row_max) * (unsigned)(ld * 2) + (unsigned)(c * 16)));
        if (zero_fill && gr > row_max) v[i] = u32x4{0, 0, 0, 0};
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = i * 256 + tid;
      if (TOTAL % 256 == 0 || id < TOTAL) {
        const int r = id / (HDP / 8), c = id % (HDP / 8);
        *(u32x4*)(lds + tile_off(r, c, Geo<HDP>::TSTR)) = v[i];
      }
    }
  }
};

// A-operand row fragment (32 rows x 16 k) of a dual-use tile: row = row0 + (lane&31), chunk 2*ks + (lane>>5)
template <int HDP> __device__ __forceinline__ bf16x8 row_frag(const unsigned char* tile, int row0, int ks, int lane) {
  const int r = row0 + (lane & 31);
  return *(const bf16x8*)(tile + tile_off(r, 2 * ks + (lane >> 5), Geo<HDP>::TSTR));
}
// A-operand TRANSPOSED fragment: A[i = column col0 + (lane&31)][k], where element j of lane-half h is tile row
// krow0 + 8*(j>>2) + 4*h + (j&3)   (the k order of an accumulator tile used as B operand)
template <int HDP> __device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int krow0, int col0, int lane) {
  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p = li & 3;
  const int h = g >> 1, cb = g & 1;
  const int r0 = krow0 + 4 * h + q4, r1 = r0 + 8;
  const int ch = (col0 + cb * 16 + 4 * p) >> 3, o8 = (p & 1) * 8;   // 4p elements -> byte 8p -> chunk (p>>1), +8*(p&1)
  const s16x4 a = lds_read_tr16(tile + tile_off(r0, ch, Geo<HDP>::TSTR) + o8);
  const s16x4 b = lds_read_tr16(tile + tile_off(r1, ch, Geo<HDP>::TSTR) + o8);
  return join_tr(a, b);
}
// registers 8s..8s+7 of a 32x32 accumulator -> bf16 B-operand fragment of k-step s
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (__bf16)x[8 * s + j];
  return f;
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Rows of sample b in the token-major tensors (qkv / out / dout / dqkv): the dense layout gives every sample S rows starting at
// b * S; with packed batches (TfAttnArgs.cu_rows) sample b owns rows cu[b] .. cu[b+1]-1 and all of them are real tokens.
struct SampleRows { size_t row0; int len; };
__device__ __forceinline__ SampleRows sample_rows(const int* __restrict__ cu, int b, int S) {
  if (cu == nullptr) return SampleRows{(size_t)b * S, S};
  const int r0 = cu[b];
  return SampleRows{(size_t)r0, cu[b + 1] - r0};
}

// 64-bit validity mask of keys kv0 .. kv0+63 (bit = key may be attended)
__device__ __forceinline__ unsigned long long key_bits(const uint8_t* __restrict__ km, int b, int S, int kv0, int lane) {
  const int key = kv0 + lane;
  bool ok = key < S;
  if (ok && km != nullptr) ok = km[(size_t)b * S + key] == 0;
  return __ballot(ok);
}

// One past the last key of sample b that may be attended (S when there is no mask).  Key tiles at or beyond it hold
// only padding: every probability there is exactly 0, so skipping them leaves all results bit-identical.
__device__ __forceinline__ int valid_key_limit(const uint8_t* __restrict__ km, int b, int S, int lane) {
  if (km == nullptr) return S;
  // lane-strided scan with independent loads, then one wave max (a ballot per 64 keys made every load wait for the
  // previous one: ~S/64 serial global round trips in the prologue of every workgroup)
  const uint8_t* __restrict__ row = km + (size_t)b * S;
  int limit = 0;
#pragma unroll 4
  for (int k = lane; k < S; k += 64) limit = row[k] == 0 ? k + 1 : limit;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) limit = max(limit, __shfl_xor(limit, o, 64));
  return limit;
}

// (batch, head) pair of a workgroup.  xcd_remap hands each XCD a CONTIGUOUS range of logical ids (so the blocks of one
// pair share an L2); the pairs themselves are dealt to the XCDs with stride 8, so that every XCD serves many different
// batch samples and ragged padding lengths do not unbalance the chiplets.
__device__ __forceinline__ int pair_of_group(int g, int npairs) {
  return (npairs & 7) == 0 ? (g % (npairs >> 3)) * 8 + g / (npairs >> 3) : g;
}

}  // namespace
