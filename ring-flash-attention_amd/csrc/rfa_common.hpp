// rfa_common.hpp — device-side building blocks shared by the gfx950 attention kernels.
//
// Conventions used by every kernel in this directory (CDNA4, wave64):
//   * MFMA shape is v_mfma_f32_32x32x16_{bf16,f16}: D(32x32) += A(32x16) * B(16x32).
//       A operand: lane l holds row  m = l&31, k = 8*(l>>5) .. +7   (8 x 16-bit = 4 VGPRs)
//       B operand: lane l holds col  n = l&31, k = 8*(l>>5) .. +7
//       C/D      : lane l, reg r holds col n = l&31, row m = (r&3) + 8*(r>>2) + 4*(l>>5)
//     so after an MFMA a lane owns ONE column and 16 rows.  All kernels are arranged so that
//     the softmax row (forward / dQ) or the key (dK/dV) is that column: per-row statistics
//     are lane-local scalars and P never has to move between lanes — the k index of the
//     second GEMM is simply *defined* as "whatever rows this lane got out of the first one":
//        kmap(ks, g, e) = 16*ks + 8*(e>>2) + 4*g + (e&3)      (ks = 16-wide k step, g = l>>5)
//     and the other operand is fetched from LDS with ds_read_b64_tr_b16 to match.
//   * LDS tiles are [rows][128] 16-bit, 256 B per row, 16 chunks of 16 B, XOR-swizzled with
//       phys_chunk = chunk ^ swz(row),  swz(row) = ((row&3)<<2) | ((row>>2)&3)
//     which is conflict-free both for row-per-lane ds_read_b128 (16-lane groups see 16
//     distinct 16-B slots) and for the 4-row x 64-B footprint of a transpose read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rfa {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

template <typename T> using vec8 = T __attribute__((ext_vector_type(8)));
template <typename T> using vec4 = T __attribute__((ext_vector_type(4)));
template <typename T> using vec2 = T __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

typedef __attribute__((address_space(3))) char lds_t;

constexpr int kHeadDim = 128;        // compiled head dim; runtime D <= 128, D % 8 == 0 (zero padded)
constexpr int kRowBytes = kHeadDim * 2;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ f32x16 mfma(vec8<bf16_t> a, vec8<bf16_t> b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma(vec8<f16_t> a, vec8<f16_t> b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// byte offset of 16-B chunk `chunk` of row `row` inside a swizzled [rows][128] tile
__device__ __forceinline__ int tile_off(int row, int chunk) {
  return row * kRowBytes + ((chunk ^ swz(row)) << 4);
}

// Absolute LDS addresses as plain 32-bit integers: per-lane bases are computed ONCE (dynamic-LDS base
// included) and the hot loops only add instruction immediates / XOR swizzle bits, instead of re-adding
// the (link-time) base symbol at every access.
__device__ __forceinline__ int lds_addr(lds_t* p) { return (int)(unsigned)(__UINTPTR_TYPE__)p; }
// keeps the compiler from splitting a finished per-lane address back into (base symbol) + (offset) and
// re-adding the two inside the loop
__device__ __forceinline__ void pin_vgpr(int& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ lds_t* lds_ptr(int addr) { return (lds_t*)(__UINTPTR_TYPE__)(unsigned)addr; }

template <typename T>
__device__ __forceinline__ vec8<T> lds_read128(lds_t* p) {
  return *(__attribute__((address_space(3))) vec8<T>*)p;
}
template <typename T>
__device__ __forceinline__ void lds_write128(lds_t* p, vec8<T> v) {
  *(__attribute__((address_space(3))) vec8<T>*)p = v;
}

// Transposed 4x16 read: within each 16-lane group lane i supplies the address of 4
// contiguous 16-bit elements = row (i>>2), cols 4*(i&3)..+3 of a 4x16 block; it receives
// column i of that block (4 elements, one per row).
template <typename T>
__device__ __forceinline__ vec4<T> lds_read_tr(lds_t* p) {
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(vec4<T>, r);
}

// ---- head-dim geometry -------------------------------------------------------------------------------
// The attention kernels are compiled for kD = 128, 96 and 64 (runtime D <= kD, zero padded; 96 = the 128-wide LDS layout
// with three instead of four 32-column blocks of MFMA work: 6 k-steps over d, 3 accumulator tiles).  LDS tiles always
// use 256-byte PHYSICAL rows so that the one swizzle above serves both: with kD = 64 a physical row holds two
// logical rows (logical row r, 16-byte chunk c in 0..7 -> physical row r >> 1, chunk ((r & 1) << 3) | c).  Both
// conflict properties carry over: a 16-lane ds_read_b128 group (16 distinct rows, one chunk) still sees 16
// distinct 16-byte slots, and the 4-row x 64-byte footprint of a transpose read still covers 256 distinct bytes.
template <int kD>
struct HeadGeo {
  static_assert(kD == 64 || kD == 96 || kD == 128, "compiled head dims");
  static constexpr int kLay = kD == 64 ? 64 : 128;       // width of the LDS layout (a 96-wide row lives in a 128-wide one)
  static constexpr int kRowBytes = kLay * 2;             // logical row (in LDS)
  static constexpr int kKSteps = kD / 16;                // 16-wide k-steps of a contraction over d
  static constexpr int kDBlocks = kD / 32;               // 32-wide blocks of d (accumulator tiles)
  static constexpr int kSwzRows = kLay == 128 ? 16 : 32; // logical rows after which the swizzle repeats:
                                                         // tile_off_d(r + kSwzRows, c) = tile_off_d(r, c) + kSwzRows * kRowBytes
};
template <int kD>
__device__ __forceinline__ int tile_off_d(int row, int chunk) {
  if (kD != 64) return row * 256 + ((chunk ^ swz(row)) << 4);
  const int prow = row >> 1, c = ((row & 1) << 3) | chunk;
  return prow * 256 + ((c ^ swz(prow)) << 4);
}
// (logical row, chunk) that lane `lane` of the LDS-DMA instruction filling physical rows 4 piece .. 4 piece + 3
// (1 KiB, lane-linear destination) has to fetch so that the tile ends up swizzled
template <int kD>
__device__ __forceinline__ void dma_lane_src(int piece, int lane, int& row, int& chunk) {
  const int prow = 4 * piece + (lane >> 4);
  const int c = (lane & 15) ^ swz(prow);
  if (kD != 64) {
    row = prow;
    chunk = c;
  } else {
    row = 2 * prow + (c >> 3);
    chunk = c & 7;
  }
}
// Transpose-read address (byte offset inside a tile) with which lane `lane` takes part in reading rows
// rb .. rb+3 (rb % 4 == 0) of d block dblk: MFMA A-operand lane (m = 32 dblk + lane & 31) receives
// tile[rb + j][m], j = 0..3.  Valid for rb + any multiple of kSwzRows by adding rows * kRowBytes.
template <int kD>
__device__ __forceinline__ int tr_off_d(int lane, int dblk, int rb) {
  const int i = lane & 15, sub = (lane >> 4) & 1;
  const int chunk = 4 * dblk + 2 * sub + ((i & 3) >> 1);
  return tile_off_d<kD>(rb + (i >> 2), chunk) + ((i & 1) << 3);
}

template <typename T>
__device__ __forceinline__ vec8<T> concat(vec4<T> a, vec4<T> b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Byte offset (before the row term) that lane `lane` must use so that a transpose read at
// rows rb..rb+3 (rb % 4 == 0, given through rsel = (rb>>2)&3) returns, for MFMA A-operand
// lane (m = lane&31 + 32*dblk), the 4 rows rb+0..3 of column m:
//     result[j] = tile[rb + j][32*dblk + (lane&31)]
// Full LDS address = tile_base + (rb + (i>>2)) * 256 + tr_lane_off(lane, dblk, rsel).
__device__ __forceinline__ int tr_lane_off(int lane, int dblk, int rsel) {
  const int i = lane & 15;
  const int sub = (lane >> 4) & 1;              // which 16-column half of the 32-wide block
  const int chunk = 4 * dblk + 2 * sub + ((i & 3) >> 1);
  const int x = ((i >> 2) << 2) | rsel;         // swz(rb + (i>>2))
  return ((chunk ^ x) << 4) + ((i & 1) << 3);
}

// pack 8 fp32 (two groups of 4 consecutive C rows) into an MFMA 16-bit operand
template <typename T>
__device__ __forceinline__ vec8<T> pack8(const f32x16& s, int base) {
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  f32x8 x;
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = s[base + e];
  return __builtin_convertvector(x, vec8<T>);
}

template <typename T>
__device__ __forceinline__ vec8<T> zero8() {
  vec8<T> z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (T)0.0f;
  return z;
}

// key / query index of C-layout register r for lane group g inside a 32-row block
__device__ __forceinline__ int crow(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// (block, batch) of a launch's remaining blockIdx digits.  Workgroups are dispatched in blockIdx order and the kernels number
// a sequence's blocks heaviest first (causal).  Round 6, dK/dV kernels: the BATCH index runs faster than the block index.
// With the batch as the slowest digit a multi-batch launch dealt all key blocks of sequence 0 — heavy to light — before
// the heavy blocks of sequence 1: with more workgroups than CUs the late sequences' heavy blocks started last and the
// launch ended in a long tail (B 4 x S 4096, unshared dK/dV plan: backward 1.95 -> 1.60 ms, B 8 x S 2048 1.05 -> 0.88).
// Batch-fastest deals the heaviest blocks of ALL sequences first — the order the launch plans' makespan estimate
// (rfa_api.cpp) assumes.  The query-block kernels (forward, dQ) have 4 x the workgroups, whose tail is short anyway, and
// lose 2 - 6 % of their K/V tile reuse in the L2 when co-resident workgroups belong to different sequences: they keep the
// batch as the slowest digit (profiles/r06_batch_order.md).
#ifndef RFA_BATCH_FAST_KV
#define RFA_BATCH_FAST_KV 1
#endif
#ifndef RFA_BATCH_FAST_Q
#define RFA_BATCH_FAST_Q 0
#endif
template <bool kBatchFast>
__device__ __forceinline__ void split_block_batch(int idx, int nblk, int nbatch, int& blk, int& b) {
  if (kBatchFast) {
    b = idx % nbatch;
    blk = idx / nbatch;
  } else {
    blk = idx % nblk;
    b = idx / nblk;
  }
}

struct Strides {
  int64_t batch, row, head;
};

// Resolved addressing of one (sequence, half) of a dense or packed tensor.
struct SeqSpan {
  int64_t row0;   // absolute first row (already includes batch offset for dense, as rows)
  int len;        // number of rows
};

// dense: rows are [0,S) of batch b (batch offset applied through Strides::batch separately)
// varlen: rows are [cu[b], cu[b+1])
__device__ __forceinline__ SeqSpan resolve_span(const int32_t* cu, int b, int S, int half) {
  int start = 0, len = S;
  if (cu) {
    start = cu[b];
    len = cu[b + 1] - start;
  }
  if (half == 1) {
    len = len / 2;                 // front half: [start, (start+end)/2)
  } else if (half == 2) {
    int mid = len / 2;             // back half:  [(start+end)/2, end)
    start += mid;
    len -= mid;
  }
  SeqSpan s;
  s.row0 = start;
  s.len = len;
  return s;
}

// s_waitcnt vmcnt(0) in the form the compiler's waitcnt scoreboard sees (gfx9 encoding:
// vmcnt = simm16[3:0] | simm16[15:14], expcnt = [6:4], lgkmcnt = [11:8]).  Used before a
// software-pipelined loop so that no prologue load is still "pending" at the loop header —
// otherwise hipcc's in-loop waits for it also drain the loop's own prefetch loads.
__device__ __forceinline__ void wait_all_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// Epilogue store of a wave's C-layout tile as 16-bit rows.  acc[dblk][4*jj + e] of lane (row = l&31,
// g = l>>5) is column 32*dblk + 8*jj + 4*g + e: the two half-waves of a row each hold 4 of every 8
// consecutive columns.  One v_permlane32_swap per dword exchanges "upper half's group jj" with "lower
// half's group jj+1", after which each lane owns 8 consecutive columns = ONE 16-byte store (lower
// half: group jj, upper half: group jj+1) instead of two 8-byte stores.  All 64 lanes must call it.
template <typename T, bool kFullD, int kNB = 4>
__device__ __forceinline__ void store_rows16(T* row_ptr, const f32x16 (&acc)[kNB], float scale, int g, int D,
                                             bool row_ok) {
#pragma unroll
  for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x4 x0, x1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x0[e] = acc[dblk][8 * m + e] * scale;         // group jj = 2m
        x1[e] = acc[dblk][8 * m + 4 + e] * scale;     // group jj = 2m + 1
      }
      const vec4<T> h0 = __builtin_convertvector(x0, vec4<T>);
      const vec4<T> h1 = __builtin_convertvector(x1, vec4<T>);
      i32x2 a = __builtin_bit_cast(i32x2, h0), b = __builtin_bit_cast(i32x2, h1);
      auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
      i32x4 w;
      w[0] = r0[0]; w[1] = r1[0]; w[2] = r0[1]; w[3] = r1[1];
      const int d0 = 32 * dblk + 16 * m + 8 * g;
      if (row_ok && (kFullD || d0 < D)) *(i32x4*)(row_ptr + d0) = w;
    }
}

// Raw buffer access (MUBUF): a 128-bit descriptor in SGPRs {base, num_records bytes, flags}, a 32-bit
// per-lane byte offset; lanes whose offset falls outside num_records read 0.
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const void* base, int num_bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, num_bytes, 0x00020000);
}
template <typename T>
__device__ __forceinline__ vec8<T> buffer_load128(buf_rsrc_t r, int voffset) {
  return __builtin_bit_cast(vec8<T>, __builtin_amdgcn_raw_buffer_load_b128(r, voffset, 0, 0));
}
// global -> LDS without staging registers (buffer_load_dwordx4 ... lds): lane L's 16 bytes land at
// lds_wave_base + 16 L; out-of-range lanes write zeros.  Issued through inline asm ON PURPOSE: hipcc's
// waitcnt pass cannot tell which LDS stage a DMA fills and protects every later ds_read with vmcnt(0),
// which serialises a ring deeper than two stages.  Hidden from it, the kernels place their own counted
// `s_waitcnt vmcnt(n)` (wait_vmem<n>) in front of the barrier that publishes a stage.  Compiler-issued
// vector memory operations stay correct: vmcnt retires in order, extra outstanding DMAs can only make
// the compiler's own waits longer, never shorter.
struct dma_rsrc_t {
  i32x4 w;   // {base[31:0], base[47:32], num_records (bytes), flags}
};
__device__ __forceinline__ dma_rsrc_t make_dma_rsrc(const void* base, int num_bytes) {
  const unsigned long long a = (unsigned long long)base;
  dma_rsrc_t r;
  // readfirstlane: the descriptor must sit in SGPRs even where the compiler's uniformity analysis gives
  // up (e.g. values defined under a wave-uniform role branch); free when the value already is scalar
  r.w[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.w[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
  r.w[2] = __builtin_amdgcn_readfirstlane(num_bytes);
  r.w[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void dma_load128(dma_rsrc_t r, int lds_wave_base, int voffset) {
  asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
               :
               : "v"(voffset), "s"(r.w), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base))
               : "m0");
}
// s_waitcnt vmcnt(n), n < 64, lgkmcnt / expcnt untouched (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4).
// Round 5: until now only the low four bits were encoded — correct for every n < 16, but the 128-row form of dq_ds_kernel
// asks for 16 (two tiles of 8 DMA instructions may stay in flight), which encoded as vmcnt(0): that form drained its whole
// ring at every tile and ran latency-bound (a llama3 head group: 1.07 GB of dS at 3.9 TB/s instead of the stream's 5.5+).
template <int n>
__device__ __forceinline__ void wait_vmem() {
  static_assert(n >= 0 && n < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (n & 15) | ((n >> 4) << 14));
}
// 4-byte buffer load hidden from the compiler's waitcnt bookkeeping (like the LDS DMA above): the caller
// counts it in its own wait_vmem<n>() and must then pass the result through load_landed() before using it.
// Needed where the compiler would otherwise protect the value with vmcnt(0) and drain younger stores.
__device__ __forceinline__ float buffer_load32_async(dma_rsrc_t r, int voffset) {
  float v;
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voffset), "s"(r.w));
  return v;
}
__device__ __forceinline__ void load_landed(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ float buffer_load32(buf_rsrc_t r, int voffset) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voffset, 0, 0));
}

// ---- dropout: counter-based keep mask --------------------------------------------------------------------
// The mask is a pure function of (seed, head, query position, key position) — NOT of the tiling, the kernel or the
// rank that evaluates it — so the forward, both backward kernels, the CPU oracle (oracle/flash_attn_ref.py:
// dropout_keep) and every rank of a sharded call see the same bits:
//     word(i, jq) = fmix32( (i * 0x9E3779B1) ^ (jq * 0x85EBCA77) ^ head_key ),   jq = j >> 2
//     element (i, j) is KEPT iff byte (j & 3) of word(i, j >> 2) < keep            (keep = round((1-p) * 256))
// with fmix32 the 32-bit finalizer of MurmurHash3 and head_key = drop_head_key(seed, batch, head) below.
// A word serves 4 consecutive keys of one query: one hash per 4 elements where a lane owns a query row
// (forward, dQ), one per element where it owns a key (dK/dV).
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t drop_head_key(uint64_t seed, uint32_t batch, uint32_t head) {
  return fmix32((uint32_t)seed ^ fmix32((uint32_t)(seed >> 32) ^ fmix32(head * 0x27D4EB2Fu ^ fmix32(batch + 0x165667B1u))));
}
__device__ __forceinline__ uint32_t drop_word(uint32_t head_key, uint32_t i, uint32_t jq) {
  return fmix32((i * 0x9E3779B1u) ^ (jq * 0x85EBCA77u) ^ head_key);
}
__device__ __forceinline__ bool drop_keep(uint32_t word, int byte, uint32_t keep) {
  return ((word >> (8 * byte)) & 0xffu) < keep;
}

__device__ __forceinline__ float shfl_xor32(float v) { return __shfl_xor(v, 32, 64); }
// max / sum of a value over the two half-waves (lane l and lane l ^ 32) WITHOUT the LDS round trip of a
// ds_bpermute: v_permlane32_swap of a register with a copy of itself leaves {lower half, lower half} in one and
// {upper half, upper half} in the other, so one swap + one max (add) gives every lane the combined value.
__device__ __forceinline__ float max_xor32(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  // (one v_max_f32 through asm: fmaxf() on values the compiler cannot see through first canonicalises both operands
  //  with a v_max x, x each — three instructions for one)
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"((unsigned)r[0]), "v"((unsigned)r[1]));
  return d;
}
__device__ __forceinline__ float sum_xor32(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

}  // namespace rfa
