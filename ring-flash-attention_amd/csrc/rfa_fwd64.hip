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
// accumulator-file map (asm-owned): O of query block qb, d block dblk: a[64 qb + 16 dblk .. +15];
// Q fragment of query block qb, k-step kk: a[128 + 32 qb + 4 kk .. +3]
constexpr int kAO = 0, kAQ = 128, kANum = 192;

#define RFA_A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define RFA_ACC_CLOBBERS                                                                                             \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", RFA_A8(1), RFA_A8(2), RFA_A8(3), RFA_A8(4), RFA_A8(5), \
      RFA_A8(6), RFA_A8(7), RFA_A8(8), RFA_A8(9), RFA_A8(10), RFA_A8(11), RFA_A8(12), RFA_A8(13), RFA_A8(14),        \
      RFA_A8(15), RFA_A8(16), RFA_A8(17), RFA_A8(18), "a190", "a191"

// ---- asm-owned accumulator file -------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void acc_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"i"(N)); }
template <int N>
__device__ __forceinline__ void acc_write(int v) { asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "i"(N)); }
template <int N>
__device__ __forceinline__ float acc_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(N));
  return v;
}
template <int N>
__device__ __forceinline__ void acc_scale(float alpha) {
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c2, %0"
               : "=&v"(t) : "v"(alpha), "i"(N));
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
    if constexpr (kFirst) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(s) : "v"(k), "i"(QA), "i"(QA + 3));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(k), "i"(QA), "i"(QA + 3));
  } else {
    if constexpr (kFirst) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], 0" : "=&v"(s) : "v"(k), "i"(QA), "i"(QA + 3));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(k), "i"(QA), "i"(QA + 3));
  }
}
// O[qb][dblk] (accumulator file) += V^T fragment x P fragment (both arch VGPRs).  kFresh: P was written by VALU
// conversions that may sit right in front of this statement (VALU write -> MFMA operand read: 2 wait states)
template <typename T, int OA, bool kFresh>
__device__ __forceinline__ void mfma_o(vec8<T> v, vec8<T> pfrag) {
  if constexpr (std::is_same<T, bf16_t>::value) {
    if constexpr (kFresh) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pfrag), "i"(OA), "i"(OA + 15));
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pfrag), "i"(OA), "i"(OA + 15));
  } else {
    if constexpr (kFresh) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pfrag), "i"(OA), "i"(OA + 15));
    else asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pfrag), "i"(OA), "i"(OA + 15));
  }
}

template <typename T>
__global__ __launch_bounds__(kF64Threads, 1) void fwd64_kernel(const FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  // reserve the accumulator registers this kernel owns (the kernel descriptor allocates what is clobbered)
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

  // ---- tile staging by LDS-DMA: 16 pieces of 1 KiB per tile, 4 per wave; lane L of piece c lands in row 4c + L/16,
  // physical chunk L%16 and fetches the logical chunk the swizzle puts there (rfa_common.hpp: dma_lane_src)
  int voff_k[4], voff_v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row, chunk;
    dma_lane_src<128>(wave + kF64Waves * i, lane, row, chunk);
    voff_k[i] = (row * (int)p.k_st.row + chunk * 8) * 2;
    voff_v[i] = (row * (int)p.v_st.row + chunk * 8) * 2;
  }
  auto load_k = [&](int j, auto stage) {             // K tile j -> K stage
    constexpr int kStage = decltype(stage)::value;
    int rows = lk - j * kF64KV;
    rows = rows < kF64KV ? rows : kF64KV;
    const int nk = rows > 0 ? ((rows - 1) * (int)p.k_st.row + p.D) * 2 : 0;
    const dma_rsrc_t rk = make_dma_rsrc(kbase + (int64_t)j * kF64KV * p.k_st.row, nk);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_load128(rk, lds_addr(smem) + kStage * kF64TileBytes + (wave + kF64Waves * i) * 1024, voff_k[i]);
  };
  auto load_v = [&](int j, auto stage) {             // V tile j -> V stage
    constexpr int kStage = decltype(stage)::value;
    int rows = lk - j * kF64KV;
    rows = rows < kF64KV ? rows : kF64KV;
    const int nv = rows > 0 ? ((rows - 1) * (int)p.v_st.row + p.D) * 2 : 0;
    const dma_rsrc_t rv = make_dma_rsrc(vbase + (int64_t)j * kF64KV * p.v_st.row, nv);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dma_load128(rv, lds_addr(smem) + (2 + kStage) * kF64TileBytes + (wave + kF64Waves * i) * 1024, voff_v[i]);
  };

  // ---- per-lane LDS addresses (absolute; the loop only adds immediates)
  int koff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    koff[kk] = lds_addr(smem) + tile_off_d<128>(l31, 2 * kk + g);
    pin_vgpr(koff[kk]);
  }
  int voff[4][2];
#pragma unroll
  for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      voff[dblk][hh] = lds_addr(smem) + tr_off_d<128>(lane, dblk, 8 * hh + 4 * g);
      pin_vgpr(voff[dblk][hh]);
    }

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
  constexpr int kAhead = 3;
  auto kfrag = [&](int i, int kbo) { return lds_read128<T>(lds_ptr(koff[i % 8]) + kbo + (i / 8) * 32 * 256); };
  auto vfrag = [&](int ks, int dblk, int vbo) {
    const int imm = vbo + 16 * ks * 256;
    return concat<T>(lds_read_tr<T>(lds_ptr(voff[dblk][0]) + imm), lds_read_tr<T>(lds_ptr(voff[dblk][1]) + imm));
  };
  float mc[2] = {0.f, 0.f};              // (row max) * c the exponentials of the current tile use
  float psum[2] = {0.f, 0.f};            // row sums of the current tile
  float mx[2];                           // running max of the next tile's scores

  // exp unit u (0..15) of `s`: scores r = 4 (u & 3) .. +3 of sub-tile t = u >> 2 (wait: 16 regs per (t, qb))
  auto exp_unit = [&](f32x16 (&s)[2][2], auto uc) {
    constexpr int u = decltype(uc)::value;
    constexpr int t = u / 8, r0 = 2 * (u % 8);           // 2 scores per query block and (t, r0) -> 16 units x 4 scores
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float pv_ = fast_exp2(__builtin_fmaf(s[t][qb][r0 + e], c, -mc[qb]));
        s[t][qb][r0 + e] = pv_;
        psum[qb] += pv_;
      }
    // anchor: the results exist HERE, between the volatile MFMA statements around this call (LLVM would otherwise sink
    // the exponentials down to their first use, the P pack of the PV phase, across any scheduling barrier)
    asm volatile("" : "+v"(s[t][0]), "+v"(s[t][1]), "+v"(psum[0]), "+v"(psum[1]));
  };
  // max unit u (0..15) of `s` of tile j: mask + running max of the same 4 scores
  auto max_unit = [&](f32x16 (&s)[2][2], int j, auto mask_c, auto uc) {
    constexpr int u = decltype(uc)::value;
    constexpr bool need_mask = decltype(mask_c)::value;
    constexpr int t = u / 8, r0 = 2 * (u % 8);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if (need_mask) {
        const int qrow = qw0 + 32 * qb + l31;
        const int lim = hi ? ((qrow + off < lk - 1) ? qrow + off : lk - 1) : lk - 1;
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if (j * kF64KV + 32 * t + crow(r0 + e, g) > lim) s[t][qb][r0 + e] = -INFINITY;
      }
      mx[qb] = fmaxf(mx[qb], fmaxf(s[t][qb][r0], s[t][qb][r0 + 1]));
    }
    if (need_mask) asm volatile("" : "+v"(s[t][0]), "+v"(s[t][1]), "+v"(mx[0]), "+v"(mx[1]));
    else asm volatile("" : "+v"(mx[0]), "+v"(mx[1]));
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
        asm volatile("s_nop 11");                       // MFMA D (accumulator file) -> accvgpr read
        if (qb == 0) acc_scale_range<kAO>(alpha, std::make_integer_sequence<int, 64>{});
        else acc_scale_range<kAO + 64>(alpha, std::make_integer_sequence<int, 64>{});
        asm volatile("s_nop 3");                        // accvgpr write -> MFMA C read
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
      mfma_s<T, kAQ + 32 + 4 * (i % 8), (i % 8) == 0>(nxt[i / 8][1], a[i]);                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      if (kExp) exp_unit(cur, std::integral_constant<int, i>{});                                                   \
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
    for (int i = 0; i < 4; ++i) vf[i] = vfrag(0, i, vbo);
#define RFA_F64_PV(i)                                                                                             \
    {                                                                                                              \
      constexpr int ks = i / 4, dblk = i % 4;                                                                      \
      if (dblk == 0) {                                                                                             \
        pb0 = pack8<T>(cur[ks / 2][0], 8 * (ks % 2));                                                              \
        pb1 = pack8<T>(cur[ks / 2][1], 8 * (ks % 2));                                                              \
      }                                                                                                            \
      if (i + 4 < 16) vf[i + 4] = vfrag((i + 4) / 4, (i + 4) % 4, vbo);                                            \
      mfma_o<T, kAO + 16 * dblk, dblk == 0>(vf[i], pb0);                                                           \
      mfma_o<T, kAO + 64 + 16 * dblk, false>(vf[i], pb1);                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      if (kMax) max_unit(nxt, jn, need_mask, std::integral_constant<int, i>{});                                    \
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
  if (ntiles > 0) {
    qk_phase(sa, sb, st0{}, no_t{});
    asm volatile("s_nop 11" : "+v"(sa[0][0]), "+v"(sa[0][1]), "+v"(sa[1][0]), "+v"(sa[1][1]));   // MFMA D -> VALU reader
    mx[0] = mx[1] = -INFINITY;
#define RFA_F64_MX(i) max_unit(sa, 0, yes_t{}, std::integral_constant<int, i>{});
    RFA_F64_SEQ16(RFA_F64_MX)
#undef RFA_F64_MX
    finalize();
  }
  __syncthreads();                                     // every wave is done with K(0) before K(2) overwrites it

  auto iter = [&](int j, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], auto par) {
    constexpr int kPar = decltype(par)::value;                 // j & 1
    typedef std::integral_constant<int, kPar> same_t;
    typedef std::integral_constant<int, kPar ^ 1> other_t;
    if (j + 2 < ntiles) load_k(j + 2, same_t{});               // K stage j&1 held K(j): read in the previous iteration
    if (j + 1 < ntiles) load_v(j + 1, other_t{});              // V stage (j+1)&1 held V(j-1): read in the previous iteration
    if (j + 1 < ntiles) {
      qk_phase(nxt, cur, other_t{}, yes_t{});
      mx[0] = mx[1] = -INFINITY;
      if (tile_needs_mask(j + 1)) pv_phase(cur, nxt, j + 1, yes_t{}, same_t{}, yes_t{});     // (diagonal / tail tiles)
      else pv_phase(cur, nxt, j + 1, no_t{}, same_t{}, yes_t{});
      finalize();
    } else {                                                   // last tile: nothing to overlap with
#define RFA_F64_EX(i) exp_unit(cur, std::integral_constant<int, i>{});
      RFA_F64_SEQ16(RFA_F64_EX)
#undef RFA_F64_EX
      pv_phase(cur, nxt, 0, no_t{}, same_t{}, no_t{});
      lsum[0] += psum[0];
      lsum[1] += psum[1];
    }
    wait_all_vmem();
    __syncthreads();
  };
  for (int j = 0; j < ntiles; j += 2) {
    iter(j, sa, sb, st0{});
    if (j + 1 < ntiles) iter(j + 1, sb, sa, st1{});
  }
#undef RFA_F64_SEQ16

  // ---------------- epilogue (per query block; as rfa_fwd.hip) ----------------
  asm volatile("s_nop 11");                              // last MFMAs -> accvgpr reads
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
