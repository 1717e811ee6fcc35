// rfa_bwd1.hip — dK/dV kernel, one wave per SIMD (gfx950).  EXPERIMENT, NOT ON THE PRODUCT PATH: rfa_bwd.hip
// only routes to it when built with -DRFA_DKDV1=1.  Measured on MI355X (S = 8192, H = 32/8, D = 128, causal,
// dS spill): 1.157 ms against 1.135 ms for the 8-wave dkdv_kernel, and one ragged case (B=2, S=700, GQA) of
// tests/test_gpu_kernels.py still fails with it.  Why it loses, by switching parts of the loop off
// (RFA_W1_X_* below, bench.py kernels_ms.bwd_dkdv):
//     MFMAs only (64 per tile and head)                      0.54 ms   (the practical ceiling of this loop)
//     + lse / delta reads, packing, the 4 spill stores       0.67
//     + the 96 LDS fragment reads (b128 or transposed: same) 0.89
//     + exp / multiply                                       0.92-0.98
//     + per-tile wait + barrier                              +0.04
//     + the 8 LDS-DMA loads per wave and tile                +0.18   -> 1.16
// i.e. with a single wave on a SIMD every LDS / VMEM instruction costs its own issue time on top of the MFMA
// stream (a 1 KiB buffer / LDS-DMA instruction is 65-80 cycles, two MFMA slots; an LDS fragment read about 9):
// only plain VALU work fits the MFMA shadows.  Two waves per SIMD hide exactly this, which is why the 8-wave
// kernel with its worse register budget (V rows in LDS, partial accumulators) is still the faster one.  Also
// learned here: consecutive MFMAs on one accumulator with fillers between them cost +43 cycles each (alternate
// two chains: 1.19 -> 1.16 ms); an asm v_mul behind a v_exp hides the trans -> VALU wait state from hipcc's
// hazard recognizer (wrong results, not a crash); an "=v" MFMA destination needs the early-clobber.
//
// Same math, same workgroup ownership (128 keys of one K/V head, all G query heads of the group, Q/dO tiles
// walked from the last one down) and same optional dS spill as dkdv_kernel in rfa_bwd.hip, but organised for
// a 512-register wave:
//   * 4 waves per workgroup, one per SIMD, wave w owns key block w (32 keys) for BOTH 32-row sub-tiles of a
//     64-row Q/dO tile: no parity split, no partial accumulators, no exchange through LDS at the end;
//   * K_w AND V_w stay in registers as MFMA B operands (64 registers) next to the dK^T / dV^T accumulators
//     (128): nothing of K/V lives in LDS, the V_w fragment re-reads of the 8-wave form are gone;
//   * with a single in-order wave per SIMD nothing hides a dependent MFMA -> VALU -> MFMA chain, so the two
//     sub-tiles are software pipelined in the source: S/dP of sub-tile 1 run beside the exp / multiply work of
//     sub-tile 0, dV/dK of sub-tile 0 beside that of sub-tile 1, and the spill stores sit in MFMA shadows.
#include <type_traits>

#include "rfa_common.hpp"
#include "rfa_kernels.hpp"

#ifndef RFA_W1_AHEAD
#define RFA_W1_AHEAD 4        // LDS operand fragments read this many MFMAs ahead
#endif
#ifndef RFA_W1_SPILL_AUX
#define RFA_W1_SPILL_AUX 2    // nt
#endif
// measurement-only switches (results are wrong when one is 0): what does the tile loop cost without its staging
// loads / its per-tile wait + barrier / its exp work?
#ifndef RFA_W1_X_LOAD
#define RFA_W1_X_LOAD 1
#endif
#ifndef RFA_W1_X_SYNC
#define RFA_W1_X_SYNC 1
#endif
#ifndef RFA_W1_X_TR
#define RFA_W1_X_TR 1
#endif
#ifndef RFA_W1_X_LDS
#define RFA_W1_X_LDS 1
#endif
#ifndef RFA_W1_X_MISC
#define RFA_W1_X_MISC 1
#endif
#ifndef RFA_W1_X_CHAIN4
#define RFA_W1_X_CHAIN4 0
#endif
#ifndef RFA_W1_X_VALU
#define RFA_W1_X_VALU 1
#endif

namespace rfa {

constexpr int kW1Waves = 4;
constexpr int kW1Threads = kW1Waves * 64;
constexpr int kW1Keys = 128;
constexpr int kW1Q = 64;
constexpr int kW1TileBytes = kW1Q * kRowBytes;      // 16 KiB
constexpr int kW1StatBytes = 2 * kW1Q * 4;          // lse[64] + delta[64] per stage
constexpr int kW1Smem = 4 * kW1TileBytes + 2 * kW1StatBytes;   // Q[2] dO[2] stats[2] = 65 KiB

// keep a value in the accumulation half of the register file (MFMA-only data) / in the arch half (VALU data)
template <typename V> __device__ __forceinline__ void pin_acc(V& v) { asm volatile("" : "+a"(v)); }
template <typename V> __device__ __forceinline__ void pin_arch(V& v) { asm volatile("" : "+v"(v)); }

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// MFMAs with explicit register classes (see the slot schedule in the kernel).  acc in arch VGPRs, B in AGPRs:
template <typename T> __device__ __forceinline__ void mfma_arch_acc(f32x16& acc, const vec8<T>& a, const vec8<T>& b);
template <> __device__ __forceinline__ void mfma_arch_acc<bf16_t>(f32x16& acc, const vec8<bf16_t>& a, const vec8<bf16_t>& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
template <> __device__ __forceinline__ void mfma_arch_acc<f16_t>(f32x16& acc, const vec8<f16_t>& a, const vec8<f16_t>& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
// first MFMA of a chain: C = 0 (no zero-fill of the accumulator registers)
template <typename T> __device__ __forceinline__ void mfma_arch_first(f32x16& acc, const vec8<T>& a, const vec8<T>& b);
template <> __device__ __forceinline__ void mfma_arch_first<bf16_t>(f32x16& acc, const vec8<bf16_t>& a, const vec8<bf16_t>& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
}
template <> __device__ __forceinline__ void mfma_arch_first<f16_t>(f32x16& acc, const vec8<f16_t>& a, const vec8<f16_t>& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
}
template <typename T> __device__ __forceinline__ void mfma_acc_first(f32x16& acc, const vec8<T>& a, const vec8<T>& b);
template <> __device__ __forceinline__ void mfma_acc_first<bf16_t>(f32x16& acc, const vec8<bf16_t>& a, const vec8<bf16_t>& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma_acc_first<f16_t>(f32x16& acc, const vec8<f16_t>& a, const vec8<f16_t>& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}
// acc in AGPRs, A and B in arch VGPRs
template <typename T> __device__ __forceinline__ void mfma_acc_acc(f32x16& acc, const vec8<T>& a, const vec8<T>& b);
template <> __device__ __forceinline__ void mfma_acc_acc<bf16_t>(f32x16& acc, const vec8<bf16_t>& a, const vec8<bf16_t>& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma_acc_acc<f16_t>(f32x16& acc, const vec8<f16_t>& a, const vec8<f16_t>& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <typename T, bool kSpill>
__global__ __launch_bounds__(kW1Threads, 1) void dkdv1_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  constexpr int kOffDo = 2 * kW1TileBytes;            // dO = Q + 32K (immediate)
  constexpr int kOffStat = 4 * kW1TileBytes;          // 64K
  if (lds_addr(smem) & 0xffff) __builtin_trap();      // the XOR stage toggles need a 64 KiB-aligned block

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = key block of this wave
  const int g = lane >> 5;
  const int l31 = lane & 31;

  int idx = blockIdx.x;
  const int G = p.H / p.Hk;
  const int hk = idx % p.Hk;
  idx /= p.Hk;
  const int kblk = idx % p.nkblk;
  const int b = idx / p.nkblk;
  const int h0 = hk * G;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int kwg0 = kblk * kW1Keys;
  if (kwg0 >= lk) return;
  const int off = lk - lq;
  const int kw0 = kwg0 + wave * 32;
  const int krow = kw0 + l31;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;

  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + ks.row0 * p.v_st.row + (int64_t)hk * p.v_st.head;
  const T* qbase0 = (const T*)p.q + qbatch * p.q_st.batch + qs.row0 * p.q_st.row + (int64_t)h0 * p.q_st.head;
  const T* dobase0 = (const T*)p.dout + qbatch * p.dout_st.batch + qs.row0 * p.dout_st.row +
                     (int64_t)h0 * p.dout_st.head;
  const float* lsebase0 = p.lse + qbatch * p.lse_batch + (int64_t)h0 * p.lse_head + qs.row0;
  const float* dltbase0 = p.delta + qbatch * p.delta_batch + (int64_t)h0 * p.delta_head + qs.row0;

  const bool causal = p.causal != 0;
  int qfirst = 0;
  if (causal) {
    qfirst = kwg0 - off;
    if (qfirst < 0) qfirst = 0;
  }
  const int jt0 = qfirst / kW1Q;
  int jt1 = (lq + kW1Q - 1) / kW1Q;
  if (jt1 < jt0) jt1 = jt0;

  // ---- this wave's K and V rows as MFMA B operands (lane = key, k index = d): 64 registers, accumulation half
  vec8<T> kw[8], vw[8];
  {
    const int kr = krow < lk ? krow : lk - 1;
    const T* kp = kbase + (int64_t)kr * p.k_st.row;
    const T* vp = vbase + (int64_t)kr * p.v_st.row;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      kw[kk] = *(const vec8<T>*)(kp + (2 * kk + g) * 8);
      vw[kk] = *(const vec8<T>*)(vp + (2 * kk + g) * 8);
    }
  }

  // ---- Q/dO tile staging by LDS-DMA: 16 pieces of 1 KiB per tensor, 4 per wave
  int voff_q[4], voff_do[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row, chunk;
    dma_lane_src<128>(wave + kW1Waves * i, lane, row, chunk);
    voff_q[i] = (row * (int)p.q_st.row + chunk * 8) * 2;
    voff_do[i] = (row * (int)p.dout_st.row + chunk * 8) * 2;
  }
  float statreg = 0.f;
  const float stat_scale = (tid & kW1Q) ? -1.f : -kLog2e;     // wave 0: lse * -log2e, wave 1: -delta
  int dma_stage = 0;
  int ld_g = 0, ld_j = jt1 > 0 ? jt1 - 1 : 0;
  auto load_tile = [&]() {
    const int j = ld_j;
    const T* qb = qbase0 + (int64_t)ld_g * p.q_st.head;
    const T* dob = dobase0 + (int64_t)ld_g * p.dout_st.head;
    const float* lsb = lsebase0 + (int64_t)ld_g * p.lse_head;
    const float* dlb = dltbase0 + (int64_t)ld_g * p.delta_head;
    if (++ld_g >= G) {
      ld_g = 0;
      --ld_j;
    }
    int rows = lq - j * kW1Q;
    rows = rows < kW1Q ? rows : kW1Q;
    const int nq = rows > 0 ? ((rows - 1) * (int)p.q_st.row + p.D) * 2 : 0;
    const int ndo = rows > 0 ? ((rows - 1) * (int)p.dout_st.row + p.D) * 2 : 0;
    const dma_rsrc_t dq_ = make_dma_rsrc(qb + (int64_t)j * kW1Q * p.q_st.row, nq);
    const dma_rsrc_t ddo = make_dma_rsrc(dob + (int64_t)j * kW1Q * p.dout_st.row, ndo);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int dst = lds_addr(smem) + dma_stage + (wave + kW1Waves * i) * 1024;
      dma_load128(dq_, dst, voff_q[i]);
      dma_load128(ddo, dst + kOffDo, voff_do[i]);
    }
    if (wave < 2) {
      const dma_rsrc_t rs = make_dma_rsrc((wave ? dlb : lsb) + j * kW1Q, rows > 0 ? rows * 4 : 0);
      statreg = buffer_load32_async(rs, lane * 4);
    }
    dma_stage ^= kW1TileBytes;
  };
  int ws = lds_addr(smem) + kOffStat + tid * 4;        // (threads 0..127)
  auto write_stats = [&]() {
    if (tid < 2 * kW1Q) *(__attribute__((address_space(3))) float*)lds_ptr(ws) = statreg * stat_scale;
  };

  // ---- per-lane LDS addresses (stage toggled by XOR once per tile)
  int aq = lds_addr(smem) + tile_off(l31, g);                                   // row l31 of sub-tile 0, chunk g
  int tq[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
    tq[hh] = lds_addr(smem) + (8 * hh + 4 * g + ((lane & 15) >> 2)) * kRowBytes + tr_lane_off(lane, 0, (2 * hh + g) & 3);
  int sa = lds_addr(smem) + kOffStat + 4 * g * 4;
  pin_vgpr(aq); pin_vgpr(tq[0]); pin_vgpr(tq[1]); pin_vgpr(sa); pin_vgpr(ws);

  const int ds_lane = 16 * (16 * (l31 >> 2) + 4 * g + (l31 & 3));
  const int ds_nkb = ds_blocks(p.Sk, p.k_half);  // extents of the longest (half) sequence (= lk, lq when dense): rfa_dqs.hip
  const int64_t ds_head_bytes = (int64_t)ds_blocks(p.Sq, p.q_half) * ds_nkb * kDsBlockBytes;
  const char* ds_b = kSpill ? (const char*)p.ds + (int64_t)b * p.H * ds_head_bytes : nullptr;
  const int ds_kb = __builtin_amdgcn_readfirstlane(kblk * 4 + wave);

  const float c = p.scale * kLog2e;
  // dK^T / dV^T accumulators.  They are born in AGPRs (a zero MFMA with an "=a" destination) so that the loop-carried
  // values have the AGPR register class: every later definition is an "+a" MFMA (initialised from VGPR zeros hipcc
  // keeps them in arch VGPRs and copies 16 registers in and out around every MFMA)
  f32x16 dk[4], dv[4];
  {
    vec8<T> z = zero8<T>();
    // hipcc does not see an MFMA in the asm below: the VALU -> MFMA operand wait states are ours to provide
    asm volatile("s_nop 7" : "+v"(z));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma_acc_first<T>(dk[i], z, z);
      mfma_acc_first<T>(dv[i], z, z);
    }
  }

  load_tile();
  wait_all_vmem();
  load_landed(statreg);
  write_stats();
  ws ^= kW1StatBytes;
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) { pin_acc(kw[kk]); pin_acc(vw[kk]); }

  // fragment fetchers (stage is in the XOR-toggled base registers)
  // GEMM 1 alternates its two accumulator chains (even slot: dP += dO V_w^T, odd slot: S += Q K_w^T, k-step i >> 1):
  // two MFMAs on the SAME accumulator only pipeline when nothing is issued between them (+43 cycles per gap
  // otherwise, measured), and this schedule puts fillers into every gap
  auto fa = [&](int t, int i) {      // fragment of slot i of sub-tile t
    return lds_read128<T>(lds_ptr(aq ^ ((i >> 1) << 5)) + ((i & 1) ? 0 : kOffDo) + t * 32 * kRowBytes);
  };
  auto ftr = [&](int t, int i) {     // i: [ks2][which: 0 = dO^T (dV), 1 = Q^T (dK)][dblk]
    const int ks2 = i >> 3, which = (i >> 2) & 1, dblk = i & 3;
    const int imm = (which ? 0 : kOffDo) + (32 * t + 16 * ks2) * kRowBytes;
    if (!RFA_W1_X_TR) return lds_read128<T>(lds_ptr(aq ^ (dblk << 6)) + imm);
    vec4<T> lo = lds_read_tr<T>(lds_ptr(tq[0] ^ (dblk << 6)) + imm);
    vec4<T> hi = lds_read_tr<T>(lds_ptr(tq[1] ^ (dblk << 6)) + imm);
    return concat<T>(lo, hi);
  };

  vec8<T> kw_dummy = zero8<T>();     // (RFA_W1_X_LDS=0 measurement only)
  pin_arch(kw_dummy);
  const int ntile = jt1 > jt0 ? (jt1 - jt0) * G : 0;
  int j = jt1 - 1, cg = 0;

  // One (tile, head) = 64 MFMA slots, hand placed (a single in-order wave per SIMD: what is not put into the gap
  // behind an MFMA is not hidden).  Every slot is [one MFMA][its fillers], pinned by sched_barrier(0):
  //   A  slots  0-15  dP0 / S0 alternating       fillers: fragment reads 4 slots ahead, -delta / lse of both sub-tiles
  //   B  slots 16-31  dP1, S1                    + exp / multiply of sub-tile 0 (from slot 18: two MFMAs behind the
  //                                                end of the S0 chain), first packing of sub-tile 0
  //   C  slots 32-47  dV += dO0^T P0, dK += Q0^T dS0   + exp / multiply of sub-tile 1, packing, spill stores of 0
  //   D  slots 48-63  the same for sub-tile 1    + packing, spill stores of 1
  // The MFMAs are inline asm so that the register classes are the ones this schedule needs: S / dP chains
  // accumulate in arch VGPRs (the VALU reads them in place; hipcc would keep every MFMA result of a 512-register
  // kernel in AGPRs and copy), dK / dV accumulate in AGPRs, K_w / V_w are AGPR B operands.  The compiler still
  // tracks the LDS reads feeding them (it places the lgkmcnt waits); the MFMA -> VALU and VALU -> MFMA wait
  // states are guaranteed by the slot distances noted above (>= 2 MFMAs = 64 cycles; 12 are required).
  auto tile_body = [&](auto masked) {
    constexpr bool kMask = decltype(masked)::value;
    constexpr int kAhead = RFA_W1_AHEAD;
    const int qt0 = j * kW1Q;
    f32x16 s[2], dp[2];
    f32x4 l2v[2][4];
    vec8<T> pb[2][2], dsb[2][2];
    vec8<T> a1[2][16], a2[2][16];

    auto rd_delta = [&](int t, int jj) {          // dP accumulator starts at -delta[q] (4 LDS reads per sub-tile)
      const f32x4 nd = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + (kW1Q + 32 * t + 8 * jj) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) dp[t][4 * jj + e] = nd[e];
    };
    auto rd_lse = [&](int t, int jj) {
      l2v[t][jj] = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + (32 * t + 8 * jj) * 4);
    };
    auto elem = [&](int t, int r) {               // P and dS of one accumulator element
      float pv = fast_exp2(__builtin_fmaf(s[t][r], c, l2v[t][r >> 2][r & 3]));
      if (kMask) {
        const int q = qt0 + 32 * t + crow(r, g);
        const bool ok = (q < lq) && (krow < lk) && (!causal || krow <= q + off);
        pv = ok ? pv : 0.f;
      }
      s[t][r] = pv;
      // the empty asm keeps hipcc from SLP-packing neighbouring products into v_pk_mul_f32 (slow in an MFMA shadow);
      // the multiply itself stays a compiler instruction (an asm v_mul hides the v_exp -> VALU wait state from the
      // hazard recognizer: measured wrong results)
      float m = dp[t][r] * pv;
      asm("" : "+v"(m));
      dp[t][r] = m;
    };
    auto pack = [&](int t, int half) {
      pb[t][half] = pack8<T>(s[t], 8 * half);
      dsb[t][half] = pack8<T>(dp[t], 8 * half);
    };
    auto spill = [&](int t, int half) {
      if (!kSpill) return;
      const char* blk = ds_b + (int64_t)(h0 + cg) * ds_head_bytes +
                        ((int64_t)(2 * j + t) * ds_nkb + ds_kb) * kDsBlockBytes;
      const buf_rsrc_t rb = make_rsrc(blk, kDsBlockBytes);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dsb[t][half]), rb, ds_lane + 128 * half, 0,
                                             RFA_W1_SPILL_AUX);
    };

    // prologue of the tile: first fragments, -delta of sub-tile 0
#pragma unroll
    for (int i = 0; i < kAhead; ++i) a1[0][i] = fa(0, i);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) rd_delta(0, jj);
    if (!RFA_W1_X_MISC) {
      pb[0][0] = pb[0][1] = pb[1][0] = pb[1][1] = kw_dummy;
      dsb[0][0] = dsb[0][1] = dsb[1][0] = dsb[1][1] = kw_dummy;
    }
    __builtin_amdgcn_sched_barrier(0);

    static_for<0, 64>([&](auto ic) {
      constexpr int n = decltype(ic)::value;
      constexpr int ph = n >> 4, i = n & 15;
      // ---- the MFMA of this slot
      if constexpr (RFA_W1_X_CHAIN4 && ph < 2) {      // (measurement only: four chains in rotation)
        constexpr int nn = n & 31, t = (nn >> 1) & 1, ii = (nn & 1) | ((nn >> 2) << 1);
        if constexpr ((ii & 1) == 0) mfma_arch_acc<T>(dp[t], a1[t][ii], vw[ii >> 1]);
        else if constexpr (ii == 1) mfma_arch_first<T>(s[t], a1[t][ii], kw[0]);
        else mfma_arch_acc<T>(s[t], a1[t][ii], kw[ii >> 1]);
      } else if constexpr (ph < 2) {
        constexpr int t = ph;
        if constexpr ((i & 1) == 0) mfma_arch_acc<T>(dp[t], a1[t][i], vw[i >> 1]);
        else if constexpr (i == 1) mfma_arch_first<T>(s[t], a1[t][i], kw[0]);
        else mfma_arch_acc<T>(s[t], a1[t][i], kw[i >> 1]);
      } else {
        constexpr int t = ph - 2;
        constexpr int ks2 = i >> 3, which = (i >> 2) & 1, dblk = i & 3;
        if constexpr (which == 0) mfma_acc_acc<T>(dv[dblk], a2[t][i], pb[t][ks2]);
        else mfma_acc_acc<T>(dk[dblk], a2[t][i], dsb[t][ks2]);
      }
      // ---- fragment reads, kAhead slots ahead (across the phase boundaries)
      constexpr int m = n + kAhead;
      if constexpr (!RFA_W1_X_LDS) {
        if constexpr (m < 32) a1[m >> 4][m & 15] = kw_dummy;
        else if constexpr (m < 64) a2[(m >> 4) - 2][m & 15] = kw_dummy;
      } else if constexpr (m < 32) a1[m >> 4][m & 15] = fa(m >> 4, m & 15);
      else if constexpr (m < 64) a2[(m >> 4) - 2][m & 15] = ftr((m >> 4) - 2, m & 15);
      // ---- row statistics
      if constexpr (RFA_W1_X_MISC && n >= 2 && n < 6) rd_lse(0, n - 2);
      if constexpr (RFA_W1_X_MISC && n >= 8 && n < 12) rd_delta(1, n - 8);
      if constexpr (RFA_W1_X_MISC && n >= 18 && n < 22) rd_lse(1, n - 18);
      // ---- exp / multiply of sub-tile 0 in slots 18..31, of sub-tile 1 in slots 34..47 (16 elements over 14 slots)
      if constexpr (RFA_W1_X_VALU && n >= 18 && n < 32) {
        constexpr int k = n - 18;
        elem(0, k);
        if constexpr (k < 2) elem(0, 14 + k);
      }
      if constexpr (RFA_W1_X_VALU && n >= 34 && n < 48) {
        constexpr int k = n - 34;
        elem(1, k);
        if constexpr (k < 2) elem(1, 14 + k);
      }
      // ---- packing (elements 0-7 of a sub-tile are final 8 slots into its exp phase, 8-15 at its end)
      if constexpr (RFA_W1_X_MISC && n == 27) pack(0, 0);
      if constexpr (RFA_W1_X_MISC && n == 31) pack(0, 1);        // consumed from slot 40
      if constexpr (RFA_W1_X_MISC && n == 43) pack(1, 0);        // consumed from slot 48
      if constexpr (RFA_W1_X_MISC && n == 47) pack(1, 1);        // consumed from slot 56
      // ---- dS spill stores, one per slot, in MFMA shadows
      if constexpr (RFA_W1_X_MISC && n == 34) spill(0, 0);
      if constexpr (RFA_W1_X_MISC && n == 37) spill(0, 1);
      if constexpr (RFA_W1_X_MISC && n == 50) spill(1, 0);
      if constexpr (RFA_W1_X_MISC && n == 53) spill(1, 1);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // Tiles are walked from the last one down: [tail tiles that need masking] [interior tiles] [tiles on the causal
  // diagonal of this workgroup's 128 keys].  Three loops with ONE body instance each: with both instances in one
  // loop hipcc gives the loop-carried accumulators and the asm operands different AGPR tuples and copies 128
  // registers per tile.  The masked instance also covers key blocks / sub-tiles that are entirely invisible (their P
  // is zero), so there is no per-wave skip inside the loops.
  auto needs_mask = [&](int jj) {
    const int qt0 = jj * kW1Q;
    return (qt0 + kW1Q > lq) || (kwg0 + kW1Keys > lk) || (causal && qt0 + off < kwg0 + kW1Keys - 1);
  };
  int nA = 0, nB = 0, jj = jt1 - 1;
  while (jj >= jt0 && needs_mask(jj)) { ++nA; --jj; }
  while (jj >= jt0 && !needs_mask(jj)) { ++nB; --jj; }
  const int nC = jj - jt0 + 1;
  int f = 0;
  auto run = [&](auto masked, int count) {
    for (int n = 0; n < count; ++n, ++f) {
      if (RFA_W1_X_LOAD && f + 1 < ntile) load_tile();
      tile_body(masked);
      // tile f+1 (and its statistics) must have landed; the 4 dS spill stores of this tile (the youngest
      // operations) may stay in flight
      if (RFA_W1_X_SYNC) {
        if (kSpill) wait_vmem<4>();
        else wait_all_vmem();
      }
      load_landed(statreg);
      if (f + 1 < ntile) write_stats();
      if (++cg >= G) {
        cg = 0;
        --j;
      }
      aq ^= kW1TileBytes;
      tq[0] ^= kW1TileBytes;
      tq[1] ^= kW1TileBytes;
      sa ^= kW1StatBytes;
      ws ^= kW1StatBytes;
      if (RFA_W1_X_SYNC) __syncthreads();
    }
  };
  run(std::true_type{}, nA * G);
  run(std::false_type{}, nB * G);
  run(std::true_type{}, nC * G);

  // the dK / dV chains were written by asm MFMAs: give the last one its 12+ wait states before v_accvgpr_read
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  if (krow >= lk) return;
  const int64_t orow = ks.row0 + krow;
  if (p.kv_f32) {
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const f32x16(&fin)[4] = which ? dv : dk;
      const float sc_ = which ? 1.f : p.scale;
      const Strides st = which ? p.dv_st : p.dk_st;
      float* ob = (float*)(which ? p.dv : p.dk) + kbatch * st.batch + orow * st.row + (int64_t)hk * st.head;
#pragma unroll
      for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          f32x4 x;
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fin[dblk][4 * jj + e] * sc_;
          *(f32x4*)(ob + 32 * dblk + 8 * jj + 4 * g) = x;
        }
    }
    return;
  }
  T* dkb = (T*)p.dk + kbatch * p.dk_st.batch + orow * p.dk_st.row + (int64_t)hk * p.dk_st.head;
  store_rows16<T, true>(dkb, dk, p.scale, g, p.D, true);
  T* dvb = (T*)p.dv + kbatch * p.dv_st.batch + orow * p.dv_st.row + (int64_t)hk * p.dv_st.head;
  store_rows16<T, true>(dvb, dv, 1.f, g, p.D, true);
}

template <typename T, bool kSpill>
static int launch_dkdv1_t(const BwdParams& p, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  if (int rc = opt_in_dynamic_lds((const void*)dkdv1_kernel<T, kSpill>, kW1Smem, attr_done)) return rc;
  const int64_t nblocks = (int64_t)p.nkblk * p.Hk * p.B;
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL((dkdv1_kernel<T, kSpill>), dim3((unsigned)nblocks), dim3(kW1Threads), kW1Smem, stream, p);
  return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchFailed;
}

// head dim 128 exactly, no window: the caller (launch_bwd_dkdv) checks
int launch_bwd_dkdv1(const BwdParams& p, int dtype, hipStream_t stream) {
  if (p.ds != nullptr) return dtype == 0 ? launch_dkdv1_t<bf16_t, true>(p, stream) : launch_dkdv1_t<f16_t, true>(p, stream);
  return dtype == 0 ? launch_dkdv1_t<bf16_t, false>(p, stream) : launch_dkdv1_t<f16_t, false>(p, stream);
}

}  // namespace rfa
