// Helpers shared by the attention kernels (attn_bf16.hip: bf16 compute; attn_x3.hip: fp32-accuracy mode on hi + lo planes):
// dual-use LDS tile geometry, operand fragment reads, key-validity scans, the (batch, head) -> XCD mapping.
#pragma once
#include "tf_common.h"
#include <utility>

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

// LDS-DMA staging of a 64-row dual-use tile by NW waves (global_load_lds_dwordx4: a wave-instruction fills 1 KiB of CONSECUTIVE
// LDS, 16 B per lane).  The tile image is lane-linear, so the layout lives on the SOURCE side: the lane whose slot is chunk position
// cp of tile row r fetches global chunk cp ^ ((r >> 2) & 3) of that row; lanes whose slot is row padding re-fetch chunk 0 (never read).
// No staging registers: per lane NI loop-invariant 32-bit source offsets; a tile advances the wave-uniform base.
template <int HDP, int NW> struct TileDma {
  static constexpr int T = Geo<HDP>::TSTR, NI = 64 * T / 1024 / NW;
  static_assert(64 * T % (1024 * NW) == 0, "tile bytes divide into whole wave-instructions");
  unsigned off[NI];                   // byte offset of this lane's source chunk inside a 64-row block, instruction i
  int wave, lane;
  __device__ __forceinline__ TileDma(int wave_, int lane_, size_t ld) : wave(wave_), lane(lane_) {
#pragma unroll
    for (int i = 0; i < NI; ++i) off[i] = slot_row(i) * (unsigned)(ld * 2) + slot_col(i);
  }
  __device__ __forceinline__ unsigned slot_row(int i) const { return (unsigned)(((wave * NI + i) * 1024 + lane * 16) / T); }
  __device__ __forceinline__ unsigned slot_col(int i) const {
    const int pos = (wave * NI + i) * 1024 + lane * 16, r = pos / T, cp = (pos % T) >> 4;
    return cp < HDP / 8 ? ((cp ^ ((r >> 2) & 3)) << 4) : 0;
  }
  // 64 rows starting soff bytes into the buffer (a sample's rows of one head: make_rsrc); bytes past the buffer's end arrive as zeros,
  // so a ragged last tile needs no clamping and its surplus V rows are finite
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rsrc, unsigned soff_, unsigned char* tile) const {
    const int soff = __builtin_amdgcn_readfirstlane((int)soff_);      // (the tile counter descends from a wave reduction)
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, TF_LDS_PTR(tile + (wave * NI + i) * 1024), 16, (int)off[i], soff, 0, 0);
  }
};
// buffer over rows 0 .. rows-1 of one head's HDP columns in a [*, ld] bf16 matrix (base = row 0, the head's first column)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const u16* base, size_t ld, int rows, int hdp) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((size_t)(rows - 1) * ld * 2 + hdp * 2), 0x00020000);
}
__device__ __forceinline__ void dma_wait_barrier() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }

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
// ---- software-pipelined operand reads (inline asm) ----------------------------------------------------------------
// hipcc schedules "ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma" with ONE fragment buffer when the reads are intrinsics
// (every MFMA then waits a full LDS round trip: the forward kernel's matrix pipe was busy a third of the time).  Issued
// from inline asm the reads run PF fragments ahead of the MFMA that consumes them; the wait asm names the fragment it
// releases, so its consumer cannot be scheduled above it.  LDS returns in order, so lgkmcnt(n) with n = reads issued
// after the wanted one is exact, and any other outstanding LDS / scalar op only makes it conservative.
template <int N, class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <int OFF> __device__ __forceinline__ u32x4 rd128_asm(unsigned addr) {
  u32x4 d;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
  return d;
}
template <int N> __device__ __forceinline__ void lgkm_wait(u32x4& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
template <int N> __device__ __forceinline__ void lgkm_wait(u64& a, u64& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

// Per-lane LDS byte addresses of the two fragment kinds of a dual-use tile (tile_off with the per-k-step / per-column parts as
// immediates): row fragments need two addresses (even / odd k-step), transposed fragments two (rows r0 and r0 + 8).
template <int HDP> struct FragAddr {
  unsigned row_e, row_o, tr_a, tr_b;
  __device__ __forceinline__ FragAddr(unsigned tile_lds, int lane) {
    constexpr int T = Geo<HDP>::TSTR;
    const int r = lane & 31, x = (r >> 2) & 3, hh = lane >> 5;
    row_e = tile_lds + r * T + ((hh ^ x) << 4);
    row_o = tile_lds + r * T + (((2 + hh) ^ x) << 4);
    const int li = lane & 15, q4 = li >> 2, p = li & 3, cb = (lane >> 4) & 1, low2 = 2 * cb + (p >> 1);
    tr_a = tile_lds + (4 * hh + q4) * T + ((low2 ^ hh) << 4) + (p & 1) * 8;
    tr_b = tile_lds + (4 * hh + q4 + 8) * T + ((low2 ^ (hh + 2)) << 4) + (p & 1) * 8;
  }
  // row_frag(tile, ROW0, KS): ROW0 a multiple of 32
  template <int ROW0, int KS> __device__ __forceinline__ u32x4 row() const {
    return rd128_asm<ROW0 * Geo<HDP>::TSTR + 64 * (KS >> 1)>((KS & 1) ? row_o : row_e);
  }
  // tr_frag(tile, KROW0, COL0): KROW0 a multiple of 16, COL0 a multiple of 32
  template <int KROW0, int COL0> __device__ __forceinline__ void tr(u64& a, u64& b) const {
    a = tr_read_asm<KROW0 * Geo<HDP>::TSTR + 2 * COL0>(tr_a);
    b = tr_read_asm<KROW0 * Geo<HDP>::TSTR + 2 * COL0>(tr_b);
  }
};

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
