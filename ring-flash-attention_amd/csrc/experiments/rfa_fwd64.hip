// rfa_fwd64.hip — flash-attention forward for gfx950, head dim 128: 4 waves x 64 query rows, ONE wave per SIMD with
// the whole 512-entry register file, software-pipelined over the key tiles.
//
// Same operation, arguments and epilogues (plain out/lse or fused merge into fp32 accumulators) as rfa_fwd.hip's
// fwd_kernel<T, 128, true, false>; what changes is the mapping onto the CU:
//   * a wave owns TWO 32-row query blocks: every K fragment (ds_read_b128) and every V^T fragment
//     (ds_read_b64_tr_b16 pair) read from LDS feeds two MFMAs instead of one — 0.5 KiB of LDS operand traffic
//     per MFMA instead of 1 KiB, which is what bounded the 8-wave form;
//   * O (2 x 4 x 16 = 128 registers) and the Q fragments (64 registers) live in the ACCUMULATOR half of the
//     register file for the whole kernel (a[0:127], a[128:191]) and are only ever touched by inline-asm MFMAs /
//     accvgpr moves with literal register names: the compiler never sees them, so it cannot shuttle them through
//     arch VGPRs (a plain-HIP 64-row kernel drowns in v_accvgpr copies: 1389 of them around 128 MFMAs);
//   * one wave per SIMD has nobody to hide its softmax behind, so the loop is pipelined across tiles: phase A of
//     iteration j runs S(j+1) = K(j+1) Q^T on the matrix pipe while the VALU works through the softmax of S(j)
//     (two S buffers in arch VGPRs), phase B runs O += V(j)^T P(j) while the rest of the exponentials are issued;
//     the MFMAs are volatile asm statements in program order, the compiler fills the gaps between them;
//   * K tiles are fetched two iterations ahead, V tiles one, by LDS-DMA into 2-deep rings; one barrier per tile.
// hipcc pads no hazards around inline asm (cdna_hip_programming.md section 5.7): the places where an MFMA result
// meets a VALU / accvgpr reader are marked below and carry their own wait states.
#include <type_traits>
#include <utility>

#include "rfa_common.hpp"
#include "rfa_kernels.hpp"

#ifndef RFA_F64_EXP_IN_QK
#define RFA_F64_EXP_IN_QK 12
#endif
// measurement-only switches (results are wrong when one is 0): loop cost without the exponentials / the running max /
// the LDS fragment reads / the per-tile wait + barrier / the tile DMA
#ifndef RFA_F64_X_EXP
#define RFA_F64_X_EXP 1
#endif
#ifndef RFA_F64_X_MAX
#define RFA_F64_X_MAX 1
#endif
#ifndef RFA_F64_X_LDS
#define RFA_F64_X_LDS 1
#endif
#ifndef RFA_F64_X_SYNC
#define RFA_F64_X_SYNC 1
#endif
#ifndef RFA_F64_X_DMA
#define RFA_F64_X_DMA 1
#endif
#ifndef RFA_F64_DEFER
#define RFA_F64_DEFER 8      // deferred rescale threshold in log2 units (as RFA_FWD_DEFER)
#endif

namespace rfa {

constexpr int kF64Waves = 4;
constexpr int kF64Threads = kF64Waves * 64;
constexpr int kF64QRows = kF64Waves * 64;           // 256 query rows per workgroup (as the 8-wave form)
constexpr int kF64KV = 64;
constexpr int kF64TileBytes = kF64KV * 256;         // 16 KiB
constexpr int kF64Smem = 4 * kF64TileBytes;         // K[2] + V[2]
// accumulator-file map (asm-owned, the TOP 192 registers): O of query block qb, d block dblk: a[64 + 64 qb + 16 dblk .. +15];
// Q fragment of query block qb, k-step kk: a[192 + 32 qb + 4 kk .. +3].  a0 .. a63 stay the compiler's: hipcc uses free
// accumulator registers as spill space for arch VGPRs and hands them out from a0 upwards.  Every asm statement of
// this file lists a64 .. a255 as clobbered (nothing of the compiler's may live there across them), and
// tests/test_abi.py audits the generated code: no compiler-issued v_accvgpr_* names a register above a63.
constexpr int kAO = 64, kAQ = 192;

#define RFA_A10(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define RFA_ACC_CLOBBERS                                                                                              \
  "a64", "a65", "a66", "a67", "a68", "a69", RFA_A10(7), RFA_A10(8), RFA_A10(9), RFA_A10(10), RFA_A10(11), RFA_A10(12), \
      RFA_A10(13), RFA_A10(14), RFA_A10(15), RFA_A10(16), RFA_A10(17), RFA_A10(18), RFA_A10(19), RFA_A10(20),         \
      RFA_A10(21), RFA_A10(22), RFA_A10(23), RFA_A10(24), "a250", "a251", "a252", "a253", "a254", "a255"

// ---- asm-owned accumulator file -------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void acc_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"i"(N) : RFA_ACC_CLOBBERS); }
template <int N>
__device__ __forceinline__ void acc_write(int v) { asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "i"(N) : RFA_ACC_CLOBBERS); }
template <int N>
__device__ __forceinline__ float acc_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(N) : RFA_ACC_CLOBBERS);
  return v;
}
template <int N>
__device__ __forceinline__ void acc_scale(float alpha) {
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c2, %0"
               : "=&v"(t) : "v"(alpha), "i"(N) : RFA_ACC_CLOBBERS);
}
template <int B0, int... I>
__device__ __forceinline__ void acc_zero_range(std::integer_sequence<int, I...>) { (acc_zero<B0 + I>(), ...); }
template <int B0, int... I>
__device__ __forceinline__ void acc_scale_range(float alpha, std::integer_sequence<int, I...>) { (acc_scale<B0 + I>(alpha), ...); }
template <int B0, int... I>
__device__ __forceinline__ void acc_read16(f32x16& x, std::integer_sequence<int, I...>) { ((x[I] = acc_read<B0 + I>()), ...); }

// S[t][qb] (arch VGPRs) (+)= K fragment (arch VGPR, from LDS) x Q fragment (accumulator file)
template <typename T, int QA, bool kFirst>
__device__ __forceinline__ void mfma_s(f32x16& s, vec8<T> k) {
  if constexpr (std::is_same<T, bf16_t>::value) {
    if constexpr (kFirst) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(s) : "v"(k), "i"(QA), "i"(QA + 3) : RFA_ACC_CLOBBERS);
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(k), "i"(QA), "i"(QA + 3) : RFA_ACC_CLOBBERS);
  } else {
    if constexpr (kFirst) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], 0" : "=&v"(s) : "v"(k), "i"(QA), "i"(QA + 3) : RFA_ACC_CLOBBERS);
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(k), "i"(QA), "i"(QA + 3) : RFA_ACC_CLOBBERS);
  }
}
// O[qb][dblk] (accumulator file) += V^T fragment x P fragment (both arch VGPRs).  kFresh: P was written by VALU
// conversions that may sit right in front of this statement (VALU write -> MFMA operand read: 2 wait states)
template <typename T, int OA, bool kFresh>
__device__ __forceinline__ void mfma_o(vec8<T> v, vec8<T> pfrag) {
  if constexpr (std::is_same<T, bf16_t>::value) {
    if constexpr (kFresh) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pfrag), "i"(OA), "i"(OA + 15) : RFA_ACC_CLOBBERS);
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pfrag), "i"(OA), "i"(OA + 15) : RFA_ACC_CLOBBERS);
  } else {
    if constexpr (kFresh) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pfrag), "i"(OA), "i"(OA + 15) : RFA_ACC_CLOBBERS);
    else asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pfrag), "i"(OA), "i"(OA + 15) : RFA_ACC_CLOBBERS);
  }
}

template <typename T>
__global__ __launch_bounds__(kF64Threads, 1) void fwd64_kernel(const FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  // a64 .. a255 are this kernel's (see the map above)
  asm volatile("" ::: RFA_ACC_CLOBBERS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5;
  const int l31 = lane & 31;

  int idx = blockIdx.x;
  const int G = p.H / p.Hk;
  const int hk = idx % p.Hk;
  idx /= p.Hk;
  const int gq = idx % G;
  idx /= G;
  const int qblk = p.nqblk - 1 - (idx % p.nqblk);   // heavy (late) causal blocks first
  const int b = idx / p.nqblk;
  const int h = hk * G + gq;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int qwg0 = qblk * kF64QRows;
  if (qwg0 >= lq) return;
  const int off = lk - lq;                 // bottom-right causal alignment
  const int qw0 = qwg0 + wave * 64;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;
  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + ks.row0 * p.v_st.row + (int64_t)hk * p.v_st.head;

  // ---- O = 0, Q fragments -> accumulator file
  acc_zero_range<kAO>(std::make_integer_sequence<int, 128>{});
  {
    auto put = [&](auto qbc) {
      constexpr int qb = decltype(qbc)::value;
      int qrow = qw0 + 32 * qb + l31;
      qrow = qrow < lq ? qrow : lq - 1;
      const T* qbase = (const T*)p.q + qbatch * p.q_st.batch + (qs.row0 + qrow) * p.q_st.row + (int64_t)h * p.q_st.head;
      i32x4 f[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) f[kk] = *(const i32x4*)(qbase + 16 * kk + 8 * g);
      auto put_kk = [&](auto kkc) {
        constexpr int kk = decltype(kkc)::value;
        acc_write<kAQ + 32 * qb + 4 * kk + 0>(f[kk][0]);
        acc_write<kAQ + 32 * qb + 4 * kk + 1>(f[kk][1]);
        acc_write<kAQ + 32 * qb + 4 * kk + 2>(f[kk][2]);
        acc_write<kAQ + 32 * qb + 4 * kk + 3>(f[kk][3]);
      };
      put_kk(std::integral_constant<int, 0>{}); put_kk(std::integral_constant<int, 1>{});
      put_kk(std::integral_constant<int, 2>{}); put_kk(std::integral_constant<int, 3>{});
      put_kk(std::integral_constant<int, 4>{}); put_kk(std::integral_constant<int, 5>{});
      put_kk(std::integral_constant<int, 6>{}); put_kk(std::integral_constant<int, 7>{});
    };
    put(std::integral_constant<int, 0>{});
    put(std::integral_constant<int, 1>{});
  }

  // ---- KV range of this workgroup
  const int qend = (qwg0 + kF64QRows < lq) ? qwg0 + kF64QRows : lq;
  const bool hi = p.causal != 0;
  int kmax = lk;
  if (hi && qend + off < kmax) kmax = qend + off;
  const int ntiles = kmax > 0 ? (kmax + kF64KV - 1) / kF64KV : 0;
  // tiles this WAVE computes: a causal wave stops at its own diagonal (its rows see nothing beyond) and only keeps
  // issuing its share of the tile DMA and taking part in the barriers for the tiles the later waves still need
  int ntiles_w = ntiles;
  if (hi) {
    const int last_key = qw0 + 63 + off;
    const int n = last_key >= 0 ? last_key / kF64KV + 1 : 0;
    ntiles_w = n < ntiles ? n : ntiles;
  }
  if (qw0 >= lq) ntiles_w = 0;

  // ---- tile staging by LDS-DMA: 16 pieces of 1 KiB per tile, 4 per wave; lane L of piece c lands in row 4c + L/16,
  // physical chunk L%16 and fetches the logical chunk the swizzle puts there (rfa_common.hpp: dma_lane_src)
  // (pieces wave, wave + 4, wave + 8, wave + 12 of a tile are 16 rows apart and share their swizzle: ONE per-lane
  //  byte offset per tensor, the piece advance goes through the scalar offset of the buffer instruction)
  int voff_k0, voff_v0;
  {
    int row, chunk;
    dma_lane_src<128>(wave, lane, row, chunk);
    voff_k0 = (row * (int)p.k_st.row + chunk * 8) * 2;
    voff_v0 = (row * (int)p.v_st.row + chunk * 8) * 2;
  }
  const int kstep = __builtin_amdgcn_readfirstlane(16 * (int)p.k_st.row * 2), vstep = __builtin_amdgcn_readfirstlane(16 * (int)p.v_st.row * 2);
  auto dma = [&](dma_rsrc_t r, int lds_wave_base, int voffset, int soffset) {
    asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                 :
                 : "v"(voffset), "s"(r.w), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)), "s"(soffset)
                 : "m0", RFA_ACC_CLOBBERS);
  };
  auto load_k = [&](int j, auto stage) {             // K tile j -> K stage
    constexpr int kStage = decltype(stage)::value;
    int rows = lk - j * kF64KV;
    rows = rows < kF64KV ? rows : kF64KV;
    const int nk = rows > 0 ? ((rows - 1) * (int)p.k_st.row + p.D) * 2 : 0;
    const dma_rsrc_t rk = make_dma_rsrc(kbase + (int64_t)j * kF64KV * p.k_st.row, nk);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(rk, lds_addr(smem) + kStage * kF64TileBytes + (wave + kF64Waves * i) * 1024, voff_k0, i * kstep);
  };
  auto load_v = [&](int j, auto stage) {             // V tile j -> V stage
    constexpr int kStage = decltype(stage)::value;
    int rows = lk - j * kF64KV;
    rows = rows < kF64KV ? rows : kF64KV;
    const int nv = rows > 0 ? ((rows - 1) * (int)p.v_st.row + p.D) * 2 : 0;
    const dma_rsrc_t rv = make_dma_rsrc(vbase + (int64_t)j * kF64KV * p.v_st.row, nv);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dma(rv, lds_addr(smem) + (2 + kStage) * kF64TileBytes + (wave + kF64Waves * i) * 1024, voff_v0, i * vstep);
  };

  // ---- per-lane LDS addresses.  The swizzle makes chunk selection an XOR on address bits 4..7, so a fragment's
  // address is (one base register) ^ (kk << 5) resp. ^ (dblk << 6) plus an immediate — 3 address registers instead of
  // 16 (the dynamic LDS block starts at 0: these kernels have no static LDS; checked below)
  if (lds_addr(smem) & 0xffff) __builtin_trap();
  int kaddr = lds_addr(smem) + tile_off_d<128>(l31, g);                       // K fragment kk: kaddr ^ (kk << 5)
  int vaddr[2];                                                               // V^T fragment (dblk, hh): vaddr[hh] ^ (dblk << 6)
  vaddr[0] = lds_addr(smem) + tr_off_d<128>(lane, 0, 4 * g);
  vaddr[1] = lds_addr(smem) + tr_off_d<128>(lane, 0, 8 + 4 * g);
  pin_vgpr(kaddr); pin_vgpr(vaddr[0]); pin_vgpr(vaddr[1]);

  const float c = p.scale * kLog2e;
  float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};

  typedef std::integral_constant<int, 0> st0;
  typedef std::integral_constant<int, 1> st1;

  // ===== the software pipeline =====================================================================================
  // iteration j (cur = S(j) -> P(j) in place, nxt = S(j+1)):
  //   [DMA K(j+2), V(j+1)]
  //   QK phase, 16 steps: K(j+1) fragment i (read 3 ahead) -> 2 MFMAs into nxt   ||  exp unit i of cur (4 scores per
  //                       query block: fma, exp2, row-sum add) against the row max fixed at the end of iteration j-1
  //   PV phase, 16 steps: V(j)^T fragment (read one group ahead), P fragments packed per 16-key group -> 2 MFMAs into O
  //                       ||  mask + running max of 4 scores of nxt per query block
  //   finalize: row max of tile j+1 -> (deferred) rescale of O / row sums — after the last MFMA of P(j), before the
  //             first exponential of tile j+1, so everything at the old scale is scaled exactly once — [barrier]
  // One wave per SIMD issues in order: an MFMA occupies the matrix pipe for 32 cycles while the VALU / LDS
  // instructions placed behind it issue; `__builtin_amdgcn_sched_barrier(0)` pins that placement (the compiler
  // would otherwise gather the VALU work into one block in front of the MFMAs it feeds).
  constexpr int kAhead = 2;
  constexpr int kExpInQK = RFA_F64_EXP_IN_QK;   // exp units (of 16) issued beside the QK MFMAs; the rest beside the first PV MFMAs
  constexpr int kVAhead = 2;                 // V^T fragments read ahead of their MFMAs
  auto kfrag = [&](int i, int kbo) {
    if (!RFA_F64_X_LDS) return __builtin_bit_cast(vec8<T>, i32x4{kaddr, vaddr[0], vaddr[1], i});
    return lds_read128<T>(lds_ptr(kaddr ^ ((i % 8) << 5)) + kbo + (i / 8) * 32 * 256);
  };
  auto vfrag = [&](int ks, int dblk, int vbo) {
    const int imm = vbo + 16 * ks * 256;
    if (!RFA_F64_X_LDS) return __builtin_bit_cast(vec8<T>, i32x4{kaddr, vaddr[0], vaddr[1], imm});
    return concat<T>(lds_read_tr<T>(lds_ptr(vaddr[0] ^ (dblk << 6)) + imm), lds_read_tr<T>(lds_ptr(vaddr[1] ^ (dblk << 6)) + imm));
  };
  float mc[2] = {0.f, 0.f};              // (row max) * c the exponentials of the current tile use
  float psum[2] = {0.f, 0.f};            // row sums of the current tile
  float mx[2];                           // running max of the next tile's scores

  // exp unit u (0..15) of `s`, query block qb: the 2 scores r0 = 2 (u % 8), r0 + 1 of sub-tile t = u / 8
  // (16 units x 2 query blocks x 2 scores = the 64 scores a lane holds per tile)
  auto exp_unit = [&](f32x16 (&s)[2][2], auto uc, auto qbc) {
    constexpr int u = decltype(uc)::value, qb = decltype(qbc)::value;
    constexpr int t = u / 8, r0 = 2 * (u % 8);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float pv_ = fast_exp2(__builtin_fmaf(s[t][qb][r0 + e], c, -mc[qb]));
      s[t][qb][r0 + e] = pv_;
      psum[qb] += pv_;
    }
    // anchor: the results exist HERE, between the volatile MFMA statements around this call (LLVM would otherwise sink
    // the exponentials down to their first use, the P pack of the PV phase, across any scheduling barrier)
    asm volatile("" : "+v"(s[t][qb]), "+v"(psum[qb]) : : RFA_ACC_CLOBBERS);
  };
  // max unit u (0..15) of `s` of tile j, query block qb: mask + running max of the same 2 scores
  auto max_unit = [&](f32x16 (&s)[2][2], int j, auto mask_c, auto uc, auto qbc) {
    constexpr int u = decltype(uc)::value, qb = decltype(qbc)::value;
    constexpr bool need_mask = decltype(mask_c)::value;
    constexpr int t = u / 8, r0 = 2 * (u % 8);
    if (need_mask) {
      const int qrow = qw0 + 32 * qb + l31;
      const int lim = hi ? ((qrow + off < lk - 1) ? qrow + off : lk - 1) : lk - 1;
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (j * kF64KV + 32 * t + crow(r0 + e, g) > lim) s[t][qb][r0 + e] = -INFINITY;
    }
    mx[qb] = fmaxf(mx[qb], fmaxf(s[t][qb][r0], s[t][qb][r0 + 1]));
    if (need_mask) asm volatile("" : "+v"(s[t][qb]), "+v"(mx[qb]) : : RFA_ACC_CLOBBERS);
    else asm volatile("" : "+v"(mx[qb]) : : RFA_ACC_CLOBBERS);
  };
  auto tile_needs_mask = [&](int j) {
    const int kt0 = j * kF64KV;
    return (kt0 + kF64KV > lk) || (hi && kt0 + kF64KV - 1 > qw0 + off);
  };
  // row max of the tile whose running max is in mx[] -> m / mc, with the deferred rescale of O and the row sums.
  // after_mfma: O's last MFMA may have just been issued (12 wait states before an accvgpr read)
  auto finalize = [&]() {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      lsum[qb] += psum[qb];
      psum[qb] = 0.f;
      const float mloc = fmaxf(mx[qb], shfl_xor32(mx[qb]));
      const float mnew = fmaxf(m[qb], mloc);
      bool rescale = true;
      if (RFA_F64_DEFER > 0) rescale = !__all((mnew - m[qb]) * c <= (float)RFA_F64_DEFER);
      if (rescale) {
        const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = fast_exp2(m[qb] * c - msafe * c);
        m[qb] = mnew;
        lsum[qb] *= alpha;
        asm volatile("s_nop 11" ::: RFA_ACC_CLOBBERS);  // MFMA D (accumulator file) -> accvgpr read
        if (qb == 0) acc_scale_range<kAO>(alpha, std::make_integer_sequence<int, 64>{});
        else acc_scale_range<kAO + 64>(alpha, std::make_integer_sequence<int, 64>{});
        asm volatile("s_nop 3" ::: RFA_ACC_CLOBBERS);   // accvgpr write -> MFMA C read
      }
      mc[qb] = ((m[qb] == -INFINITY) ? 0.f : m[qb]) * c;
    }
  };

#define RFA_F64_SEQ16(F)                                                                                          \
  F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)

  // QK phase: nxt = K(stage) Q^T, interleaved with the exponentials of cur (kExp) — or bare (prologue)
  auto qk_phase = [&](f32x16 (&nxt)[2][2], f32x16 (&cur)[2][2], auto stage, auto with_exp) {
    constexpr int kbo = decltype(stage)::value * kF64TileBytes;
    constexpr bool kExp = decltype(with_exp)::value;
    vec8<T> a[16];
#pragma unroll
    for (int i = 0; i < kAhead; ++i) a[i] = kfrag(i, kbo);
#define RFA_F64_QK(i)                                                                                             \
    {                                                                                                              \
      if (i + kAhead < 16) a[i + kAhead] = kfrag(i + kAhead, kbo);                                                \
      mfma_s<T, kAQ + 4 * (i % 8), (i % 8) == 0>(nxt[i / 8][0], a[i]);                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      if (RFA_F64_X_EXP && kExp && i < kExpInQK) exp_unit(cur, std::integral_constant<int, i>{}, st0{});          \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      mfma_s<T, kAQ + 32 + 4 * (i % 8), (i % 8) == 0>(nxt[i / 8][1], a[i]);                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      if (RFA_F64_X_EXP && kExp && i < kExpInQK) exp_unit(cur, std::integral_constant<int, i>{}, st1{});          \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
    RFA_F64_SEQ16(RFA_F64_QK)
#undef RFA_F64_QK
  };
  // PV phase: O += V(stage)^T P, P = cur; interleaved with the running max of nxt (kMax)
  auto pv_phase = [&](f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], int jn, auto need_mask, auto stage, auto with_max) {
    constexpr int vbo = (2 + decltype(stage)::value) * kF64TileBytes;
    constexpr bool kMax = decltype(with_max)::value;
    vec8<T> vf[16];
    vec8<T> pb0, pb1;
#pragma unroll
    for (int i = 0; i < kVAhead; ++i) vf[i] = vfrag(0, i, vbo);
#define RFA_F64_PV(i)                                                                                             \
    {                                                                                                              \
      constexpr int ks = i / 4, dblk = i % 4;                                                                      \
      if (dblk == 0) {                                                                                             \
        pb0 = pack8<T>(cur[ks / 2][0], 8 * (ks % 2));                                                              \
        pb1 = pack8<T>(cur[ks / 2][1], 8 * (ks % 2));                                                              \
      }                                                                                                            \
      if (i + kVAhead < 16) vf[i + kVAhead] = vfrag((i + kVAhead) / 4, (i + kVAhead) % 4, vbo);                    \
      mfma_o<T, kAO + 16 * dblk, dblk == 0>(vf[i], pb0);                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      if (RFA_F64_X_EXP && kMax && kExpInQK + i < 16) exp_unit(cur, std::integral_constant<int, (kExpInQK + i) % 16>{}, st0{}); \
      if (RFA_F64_X_MAX && kMax) max_unit(nxt, jn, need_mask, std::integral_constant<int, i>{}, st0{});            \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      mfma_o<T, kAO + 64 + 16 * dblk, dblk == 0>(vf[i], pb1);   /* (the compiler may convert pb1 right here) */     \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      if (RFA_F64_X_EXP && kMax && kExpInQK + i < 16) exp_unit(cur, std::integral_constant<int, (kExpInQK + i) % 16>{}, st1{}); \
      if (RFA_F64_X_MAX && kMax) max_unit(nxt, jn, need_mask, std::integral_constant<int, i>{}, st1{});            \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
    RFA_F64_SEQ16(RFA_F64_PV)
#undef RFA_F64_PV
  };
  typedef std::integral_constant<bool, true> yes_t;
  typedef std::integral_constant<bool, false> no_t;

  // ---- prologue: K(0), K(1), V(0) in flight; S(0), its row max
  load_k(0, st0{});
  load_v(0, st0{});
  if (ntiles > 1) load_k(1, st1{});
  wait_all_vmem();
  __syncthreads();
  f32x16 sa[2][2], sb[2][2];
  if (ntiles_w > 0) {
    qk_phase(sa, sb, st0{}, no_t{});
    asm volatile("s_nop 11" : "+v"(sa[0][0]), "+v"(sa[0][1]), "+v"(sa[1][0]), "+v"(sa[1][1]) : : RFA_ACC_CLOBBERS);   // MFMA D -> VALU reader
    mx[0] = mx[1] = -INFINITY;
#define RFA_F64_MX(i) max_unit(sa, 0, yes_t{}, std::integral_constant<int, i>{}, st0{}); max_unit(sa, 0, yes_t{}, std::integral_constant<int, i>{}, st1{});
    RFA_F64_SEQ16(RFA_F64_MX)
#undef RFA_F64_MX
    finalize();
  }
  __syncthreads();                                     // every wave is done with K(0) before K(2) overwrites it

  // tile DMA of iteration j (K two tiles ahead, V one) and the end-of-iteration hand-over
  auto dma_for = [&](int j, auto par) {
    constexpr int kPar = decltype(par)::value;                 // j & 1
    if (RFA_F64_X_DMA && j + 2 < ntiles) load_k(j + 2, std::integral_constant<int, kPar>{});       // K stage j&1 held K(j): read in the previous iteration
    if (RFA_F64_X_DMA && j + 1 < ntiles) load_v(j + 1, std::integral_constant<int, kPar ^ 1>{});   // V stage (j+1)&1 held V(j-1)
  };
  auto sync = [&]() {
    if (RFA_F64_X_SYNC) {
      wait_all_vmem();
      __syncthreads();
    }
    // (every path through an iteration crosses a statement that owns a64 .. a255)
    asm volatile("" ::: RFA_ACC_CLOBBERS);
  };
  // One iteration = one tile of the WORKGROUP: the tile DMA, this wave's work on it, the barrier.  Wave-uniform modes:
  //   j + 1 < ntiles_w   the tile has a successor: everything overlapped (cur = S(j) -> P(j), nxt = S(j+1))
  //   j + 1 == ntiles_w  this wave's last tile: nothing to overlap with
  //   j >= ntiles_w      only the later waves still compute (or j is the padding slot of an odd tile count)
  // The loop always runs whole PAIRS of iterations with no branch between them: the S buffers swap roles inside the
  // pair and are back in place at the loop edge (a conditional second half costs 128 register moves per trip).
  auto iter = [&](int j, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], auto par) {
    constexpr int kPar = decltype(par)::value;
    typedef std::integral_constant<int, kPar> same_t;
    typedef std::integral_constant<int, kPar ^ 1> other_t;
    dma_for(j, par);
    if (j + 1 < ntiles_w) {
      qk_phase(nxt, cur, other_t{}, yes_t{});
      mx[0] = mx[1] = -INFINITY;
      if (tile_needs_mask(j + 1)) pv_phase(cur, nxt, j + 1, yes_t{}, same_t{}, yes_t{});     // (diagonal / tail tiles)
      else pv_phase(cur, nxt, j + 1, no_t{}, same_t{}, yes_t{});
      finalize();
    } else if (j + 1 == ntiles_w) {
#define RFA_F64_EX(i) exp_unit(cur, std::integral_constant<int, i>{}, st0{}); exp_unit(cur, std::integral_constant<int, i>{}, st1{});
      RFA_F64_SEQ16(RFA_F64_EX)
#undef RFA_F64_EX
      pv_phase(cur, nxt, 0, no_t{}, same_t{}, no_t{});
      lsum[0] += psum[0];
      lsum[1] += psum[1];
      asm volatile("" : "+v"(lsum[0]), "+v"(lsum[1]) : : RFA_ACC_CLOBBERS);
    }
    sync();
  };
  for (int j = 0; j < ntiles; j += 2) {
    iter(j, sa, sb, st0{});
    iter(j + 1, sb, sa, st1{});
  }
#undef RFA_F64_SEQ16

  // ---------------- epilogue (per query block; as rfa_fwd.hip) ----------------
  asm volatile("s_nop 11" ::: RFA_ACC_CLOBBERS);         // last MFMAs -> accvgpr reads
  auto finish = [&](auto qbc) {
    constexpr int qb = decltype(qbc)::value;
    const int qrow = qw0 + 32 * qb + l31;
    f32x16 o[4];
    acc_read16<kAO + 64 * qb + 0>(o[0], std::make_integer_sequence<int, 16>{});
    acc_read16<kAO + 64 * qb + 16>(o[1], std::make_integer_sequence<int, 16>{});
    acc_read16<kAO + 64 * qb + 32>(o[2], std::make_integer_sequence<int, 16>{});
    acc_read16<kAO + 64 * qb + 48>(o[3], std::make_integer_sequence<int, 16>{});
    const float lsum_h = lsum[qb];
    const float l = lsum_h + shfl_xor32(lsum_h);           // (all 64 lanes take part)
    if (qrow >= lq) return;
    const bool has = l > 0.f;
    const float inv = has ? 1.f / l : 0.f;
    const float blse = has ? m[qb] * p.scale + __logf(l) : INFINITY;   // natural log
    const int64_t orow = qs.row0 + qrow;
    if (p.out_acc == nullptr) {
      T* ob = (T*)p.out + qbatch * p.out_st.batch + orow * p.out_st.row + (int64_t)h * p.out_st.head;
      store_rows16<T, true, 4>(ob, o, inv, g, p.D, true);
      if (g == 0) p.lse[qbatch * p.lse_batch + (int64_t)h * p.lse_head + orow] = blse;
    } else {
      float* ab = p.out_acc + qbatch * p.out_acc_st.batch + orow * p.out_acc_st.row + (int64_t)h * p.out_acc_st.head;
      float* lp = p.lse_acc + qbatch * p.lse_acc_batch + (int64_t)h * p.lse_acc_head + orow;
      if (p.acc_init) {
#pragma unroll
        for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = o[dblk][4 * jj + e] * inv;
            *(f32x4*)(ab + 32 * dblk + 8 * jj + 4 * g) = x;
          }
        if (g == 0) *lp = has ? blse : -INFINITY;
      } else if (has) {
        const float lold = *lp;
        const float mx = fmaxf(lold, blse);
        const float eo = __expf(lold - mx);
        const float eb = __expf(blse - mx);
        const float den = eo + eb;
        const float wo = eo / den;
        const float wb = eb / den * inv;
        const float lnew = mx + __logf(den);
#pragma unroll
        for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            float* ap = ab + 32 * dblk + 8 * jj + 4 * g;
            f32x4 x = *(f32x4*)ap;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = x[e] * wo + o[dblk][4 * jj + e] * wb;
            *(f32x4*)ap = x;
          }
        if (g == 0) *lp = lnew;
      }
    }
  };
  finish(st0{});
  finish(st1{});
}

template <typename T>
static int launch_fwd64_t(const FwdParams& p, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  if (int rc = opt_in_dynamic_lds((const void*)fwd64_kernel<T>, kF64Smem, attr_done)) return rc;
  const int64_t nblocks = (int64_t)p.nqblk * p.H * p.B;
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL((fwd64_kernel<T>), dim3((unsigned)nblocks), dim3(kF64Threads), kF64Smem, stream, p);
  return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchFailed;
}

int launch_fwd64(const FwdParams& p, int dtype, hipStream_t stream) {
  return dtype == 0 ? launch_fwd64_t<bf16_t>(p, stream) : launch_fwd64_t<f16_t>(p, stream);
}

}  // namespace rfa
