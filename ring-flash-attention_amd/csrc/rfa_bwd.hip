// rfa_bwd.hip — flash-attention backward for gfx950 (MI355X): dQ kernel + dK/dV kernel.
//
// Replaces flash_attn._flash_attn_backward / _flash_attn_varlen_backward as called from
// /root/reference/ring_flash_attn/zigzag_ring_flash_attn.py:156, ring_flash_attn.py:131,
// ring_flash_attn_varlen.py:169, zigzag_ring_flash_attn_varlen.py:275,
// llama3_flash_attn_varlen.py:282.  Math (per head, P uses the GLOBAL lse of the row):
//     P  = exp(scale·QKᵀ − lse)          dP = dO·Vᵀ          dS = P ∘ (dP − Δ),  Δ = rowsum(dO∘O)
//     dQ = scale·dS·K        dK = scale·dSᵀ·Q        dV = Pᵀ·dO
// Deterministic by construction: no atomics.  The work is split by OUTPUT ownership:
//   * dq_kernel    : a workgroup owns 256 query rows, streams K/V tiles through LDS
//                    (S,dP recomputed per tile), accumulates dQ in registers, and either
//                    stores it or adds it into a caller fp32 accumulator (ring steps).
//   * dkdv_kernel  : a workgroup owns 128 keys of ONE query head, streams Q/dO tiles through
//                    LDS, accumulates dK,dV in registers (4 waves x 32 keys, 512-register
//                    budget).  GQA group reduction is a separate HBM-bound kernel
//                    (rfa_aux.hip: reduce_kernel), like flash_attn's dk_expanded + sum.
// Lane ownership mirrors the forward kernel (see rfa_common.hpp): after the first GEMM a
// lane owns one query row (dQ kernel) or one key (dK/dV kernel), and the probabilities go
// straight from the accumulator registers into the B operand of the second GEMM.
#include <type_traits>

#include "rfa_common.hpp"
#include "rfa_kernels.hpp"

// ---- tuning knobs (A/B'd on hardware with tools/ab_variants.py; defaults = best measured) ----
#ifndef RFA_KV_PIPE
#define RFA_KV_PIPE 0        // 1: software-pipeline the two 32-row sub-tiles of a Q tile in dkdv_kernel
#endif
#ifndef RFA_KV_AHEAD1
#define RFA_KV_AHEAD1 4      // dkdv: A fragments read this many MFMAs ahead in the S/dP GEMMs
#endif
#ifndef RFA_KV_AHEAD2
#define RFA_KV_AHEAD2 2      // dkdv: transpose-read fragment pairs ahead in the dV/dK GEMMs
#endif
#ifndef RFA_DQ_PIN
#define RFA_DQ_PIN 0          // >0: pin the dQ transpose-read/MFMA pipeline with this read-ahead depth
#endif
#ifndef RFA_DQ_AHEAD1
#define RFA_DQ_AHEAD1 3
#endif
#ifndef RFA_KV_SPECIALIZED
#define RFA_KV_SPECIALIZED 0  // 0: dkdv_kernel (4 waves, 512 regs) = 1.15 ms; 1: dkdv2_kernel (8 waves, role-split) = 1.26 ms
#endif
#ifndef RFA_KV_DMA
#define RFA_KV_DMA 0         // 1: dkdv_kernel (D == 128) stages Q/dO with global_load_lds (47 fewer VGPRs; measured neutral: 1.183 vs 1.190 ms)
#endif
#ifndef RFA_KV_ILV
#define RFA_KV_ILV 0         // 1: alternate the S / dP (and dV / dK) accumulator chains MFMA by MFMA
#endif

namespace rfa {

// =====================================================================================
// dQ kernel
// =====================================================================================
constexpr int kDqWaves = 8;
constexpr int kDqThreads = kDqWaves * 64;
constexpr int kDqRows = kDqWaves * 32;          // 256 query rows / workgroup
constexpr int kDqKV = 64;
constexpr int kDqTileBytes = kDqKV * kRowBytes; // 16 KiB
constexpr int kDqSmem = 4 * kDqTileBytes;       // K[2] V[2]

template <typename T, bool kFullD>
__global__ __launch_bounds__(kDqThreads, 2) void dq_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;

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
  const int qblk = p.nqblk - 1 - (idx % p.nqblk);
  const int b = idx / p.nqblk;
  const int h = hk * G + gq;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int qwg0 = qblk * kDqRows;
  if (qwg0 >= lq) return;
  const int off = lk - lq;
  const int qw0 = qwg0 + wave * 32;
  const int qrow = qw0 + l31;
  const int qrow_c = qrow < lq ? qrow : lq - 1;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;
  const int64_t arow = qs.row0 + qrow_c;

  const T* qbase = (const T*)p.q + qbatch * p.q_st.batch + arow * p.q_st.row + (int64_t)h * p.q_st.head;
  const T* dobase = (const T*)p.dout + qbatch * p.dout_st.batch + arow * p.dout_st.row +
                    (int64_t)h * p.dout_st.head;
  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + ks.row0 * p.v_st.row + (int64_t)hk * p.v_st.head;

  vec8<T> qf[8], dof[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int d0 = 16 * kk + 8 * g;
    qf[kk] = (kFullD || d0 < p.D) ? *(const vec8<T>*)(qbase + d0) : zero8<T>();
    dof[kk] = (kFullD || d0 < p.D) ? *(const vec8<T>*)(dobase + d0) : zero8<T>();
  }
  const float L2 = p.lse[qbatch * p.lse_batch + (int64_t)h * p.lse_head + arow] * kLog2e;
  const float dlt = p.delta[qbatch * p.delta_batch + (int64_t)h * p.delta_head + arow];

  const int qend = (qwg0 + kDqRows < lq) ? qwg0 + kDqRows : lq;
  int kmax = lk;
  if (p.causal && qend + off < kmax) kmax = qend + off;
  const int ntiles = kmax > 0 ? (kmax + kDqKV - 1) / kDqKV : 0;

  const int sc = tid & 15;
  const int sr = tid >> 4;
  const bool sd_ok = kFullD || sc * 8 < p.D;
  vec8<T> kreg[2], vreg[2];
  auto load_tile = [&](int j) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int kr = j * kDqKV + sr + 32 * i;
      kr = kr < lk ? kr : lk - 1;
      kr = kr < 0 ? 0 : kr;
      if (sd_ok) {
        kreg[i] = *(const vec8<T>*)(kbase + (int64_t)kr * p.k_st.row + sc * 8);
        vreg[i] = *(const vec8<T>*)(vbase + (int64_t)kr * p.v_st.row + sc * 8);
      } else {
        kreg[i] = zero8<T>();
        vreg[i] = zero8<T>();
      }
    }
  };
  auto write_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = tile_off(sr + 32 * i, sc);
      lds_write128<T>(smem + buf * kDqTileBytes + o, kreg[i]);
      lds_write128<T>(smem + (2 + buf) * kDqTileBytes + o, vreg[i]);
    }
  };

  int koff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) koff[kk] = tile_off(l31, 2 * kk + g);
  int toff[4][2];
#pragma unroll
  for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
      toff[dblk][hh] = (8 * hh + 4 * g + ((lane & 15) >> 2)) * kRowBytes +
                       tr_lane_off(lane, dblk, (2 * hh + g) & 3);

  const float c = p.scale * kLog2e;
  f32x16 dq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

  load_tile(0);            // unconditional (rows clamped): one path into the loop, see rfa_fwd.hip
  write_tile(0);
  wait_all_vmem();
  __syncthreads();

  for (int j = 0; j < ntiles; ++j) {
    const int buf = j & 1;
    lds_t* kb = smem + buf * kDqTileBytes;
    lds_t* vb = smem + (2 + buf) * kDqTileBytes;
    if (j + 1 < ntiles) load_tile(j + 1);
    const int kt0 = j * kDqKV;
    const bool active = (qw0 < lq) && !(p.causal && kt0 > qw0 + 31 + off);
    if (active) {
      const bool need_mask = (kt0 + kDqKV > lk) || (p.causal && kt0 + kDqKV - 1 > qw0 + off);
      const int lim = p.causal ? ((qrow + off < lk - 1) ? qrow + off : lk - 1) : lk - 1;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        {
          // S^T = K Q^T and dP^T = V dO^T as one 16-step pipeline, A fragments read kAhead ahead
          constexpr int kAhead = RFA_DQ_AHEAD1;
          vec8<T> a[16];
#pragma unroll
          for (int i = 0; i < kAhead; ++i)
            a[i] = lds_read128<T>((i < 8 ? kb : vb) + t * 32 * kRowBytes + koff[i & 7]);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (i + kAhead < 16)
              a[i + kAhead] = lds_read128<T>(((i + kAhead) < 8 ? kb : vb) + t * 32 * kRowBytes + koff[(i + kAhead) & 7]);
            if (i < 8) s = mfma(a[i], qf[i], s);
            else dp = mfma(a[i], dof[i - 8], dp);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
          for (int i = 0; i < 16 - kAhead; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(__builtin_fmaf(s[r], c, -L2));
        if (need_mask) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt0 + 32 * t + crow(r, g);
            s[r] = key > lim ? 0.f : s[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = s[r] * (dp[r] - dlt);
#if RFA_DQ_PIN
        {
          const vec8<T> dsb0 = pack8<T>(s, 0), dsb1 = pack8<T>(s, 8);
          constexpr int kAhead = RFA_DQ_PIN;
          vec8<T> a[8];
          auto frag = [&](int i) {                        // i: [ks2][dblk]
            lds_t* kt = kb + (32 * t + 16 * (i >> 2)) * kRowBytes;
            vec4<T> lo = lds_read_tr<T>(kt + toff[i & 3][0]);
            vec4<T> hi = lds_read_tr<T>(kt + toff[i & 3][1]);
            return concat<T>(lo, hi);
          };
#pragma unroll
          for (int i = 0; i < kAhead; ++i) a[i] = frag(i);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (i + kAhead < 8) a[i + kAhead] = frag(i + kAhead);
            dq[i & 3] = mfma(a[i], (i >> 2) ? dsb1 : dsb0, dq[i & 3]);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * kAhead, 1);
#pragma unroll
          for (int i = 0; i < 8 - kAhead; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 1);
        }
#else
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          const vec8<T> dsb = pack8<T>(s, 8 * ks2);
          lds_t* kt = kb + (32 * t + 16 * ks2) * kRowBytes;
#pragma unroll
          for (int dblk = 0; dblk < 4; ++dblk) {
            vec4<T> lo = lds_read_tr<T>(kt + toff[dblk][0]);
            vec4<T> hi = lds_read_tr<T>(kt + toff[dblk][1]);
            dq[dblk] = mfma(concat<T>(lo, hi), dsb, dq[dblk]);
          }
        }
#endif
      }
    }
    if (j + 1 < ntiles) write_tile(buf ^ 1);
    __syncthreads();
  }

  if (qrow >= lq) return;
  const int64_t orow = qs.row0 + qrow;
  if (p.dq_acc == nullptr) {
    T* ob = (T*)p.dq + qbatch * p.dq_st.batch + orow * p.dq_st.row + (int64_t)h * p.dq_st.head;
    store_rows16<T, kFullD>(ob, dq, p.scale, g, p.D, true);
  } else {
    float* ab = p.dq_acc + qbatch * p.dq_acc_st.batch + orow * p.dq_acc_st.row +
                (int64_t)h * p.dq_acc_st.head;
#pragma unroll
    for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int d0 = 32 * dblk + 8 * jj + 4 * g;
        if (kFullD || d0 < p.D) {
          f32x4 x;
          if (p.acc_init) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = 0.f;
          } else {
            x = *(f32x4*)(ab + d0);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] += dq[dblk][4 * jj + e] * p.scale;
          *(f32x4*)(ab + d0) = x;
        }
      }
  }
}

// =====================================================================================
// dK/dV kernel
// =====================================================================================
constexpr int kKvWaves = 4;
constexpr int kKvThreads = kKvWaves * 64;
constexpr int kKvKeys = kKvWaves * 32;           // 128 keys / workgroup
constexpr int kKvQ = 64;                          // query rows per tile
constexpr int kKvTileBytes = kKvQ * kRowBytes;    // 16 KiB
constexpr int kKvStatBytes = 2 * kKvQ * 4;        // lse2[64] + delta[64] per stage
constexpr int kKvSmem = 4 * kKvTileBytes + 2 * kKvStatBytes;   // Q[2] dO[2] stats[2]

template <typename T, bool kFullD>
__global__ __launch_bounds__(kKvThreads) void dkdv_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  lds_t* stat_base = smem + 4 * kKvTileBytes;

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
  const int kblk = idx % p.nkblk;                 // early keys see most queries: heavy first
  const int b = idx / p.nkblk;
  const int h = hk * G + gq;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int kwg0 = kblk * kKvKeys;
  if (kwg0 >= lk) return;
  const int off = lk - lq;
  const int kw0 = kwg0 + wave * 32;
  const int krow = kw0 + l31;
  const int krow_c = krow < lk ? krow : lk - 1;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;

  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + (ks.row0 + krow_c) * p.k_st.row +
                   (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + (ks.row0 + krow_c) * p.v_st.row +
                   (int64_t)hk * p.v_st.head;
  const T* qbase = (const T*)p.q + qbatch * p.q_st.batch + qs.row0 * p.q_st.row + (int64_t)h * p.q_st.head;
  const T* dobase = (const T*)p.dout + qbatch * p.dout_st.batch + qs.row0 * p.dout_st.row +
                    (int64_t)h * p.dout_st.head;
  const float* lsebase = p.lse + qbatch * p.lse_batch + (int64_t)h * p.lse_head + qs.row0;
  const float* dltbase = p.delta + qbatch * p.delta_batch + (int64_t)h * p.delta_head + qs.row0;

  // K_w / V_w fragments: B operands (lane key = l31, d = 16kk + 8g .. +7)
  vec8<T> kf[8], vf[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int d0 = 16 * kk + 8 * g;
    kf[kk] = (kFullD || d0 < p.D) ? *(const vec8<T>*)(kbase + d0) : zero8<T>();
    vf[kk] = (kFullD || d0 < p.D) ? *(const vec8<T>*)(vbase + d0) : zero8<T>();
  }

  // query tile range: causal => only rows q with q + off >= first key of the block
  int qfirst = 0;
  if (p.causal) {
    qfirst = kwg0 - off;
    if (qfirst < 0) qfirst = 0;
  }
  const int jt0 = qfirst / kKvQ;
  const int jt1 = (lq + kKvQ - 1) / kKvQ;     // exclusive

  // staging: thread -> chunk sc of rows sr + 16 i (i = 0..3), for Q and dO
  const int sc = tid & 15;
  const int sr = tid >> 4;                    // 0..15
  const bool sd_ok = kFullD || sc * 8 < p.D;
  vec8<T> qreg[4], doreg[4];
  float statreg = 0.f;
  auto load_tile = [&](int j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int qr = j * kKvQ + sr + 16 * i;
      qr = qr < lq ? qr : lq - 1;
      qr = qr < 0 ? 0 : qr;
      if (sd_ok) {
        qreg[i] = *(const vec8<T>*)(qbase + (int64_t)qr * p.q_st.row + sc * 8);
        doreg[i] = *(const vec8<T>*)(dobase + (int64_t)qr * p.dout_st.row + sc * 8);
      } else {
        qreg[i] = zero8<T>();
        doreg[i] = zero8<T>();
      }
    }
    {
      // every thread issues this 4-byte load (no branch, no use of the value here: a use would
      // make hipcc wait vmcnt(0) right behind the tile prefetch); threads >= 128 just discard it
      int qr = j * kKvQ + (tid & (kKvQ - 1));
      qr = qr < lq ? qr : lq - 1;
      qr = qr < 0 ? 0 : qr;
      const float* sp = (tid & kKvQ) ? dltbase : lsebase;
      statreg = sp[qr];
    }
  };
  auto write_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = tile_off(sr + 16 * i, sc);
      lds_write128<T>(smem + buf * kKvTileBytes + o, qreg[i]);
      lds_write128<T>(smem + (2 + buf) * kKvTileBytes + o, doreg[i]);
    }
    if (tid < 2 * kKvQ)
      *(__attribute__((address_space(3))) float*)(stat_base + buf * kKvStatBytes + tid * 4) = statreg;
  };
  // DMA staging (D == 128 only): global_load_lds writes LDS at wave-uniform base + 16*lane, i.e. one
  // instruction fills 4 consecutive tile rows in PHYSICAL chunk order; the XOR swizzle is therefore
  // applied to the SOURCE chunk each lane fetches.  Rows 4*(4i+wave) .. +3; swz(row) of lane l is
  // ((l>>4)<<2) | wave for every i, so the logical chunk is a per-lane constant.
  constexpr bool kDma = RFA_KV_DMA && kFullD;
  const int dma_chunk = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | wave);
  auto dma_tile = [&](int j, int buf) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cidx = 4 * i + wave;
      int qr = j * kKvQ + 4 * cidx + (lane >> 4);
      qr = qr < lq ? qr : lq - 1;
      qr = qr < 0 ? 0 : qr;
      __builtin_amdgcn_global_load_lds((gptr_t)(qbase + (int64_t)qr * p.q_st.row + dma_chunk * 8),
                                       (lptr_t)(smem + buf * kKvTileBytes + cidx * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(dobase + (int64_t)qr * p.dout_st.row + dma_chunk * 8),
                                       (lptr_t)(smem + (2 + buf) * kKvTileBytes + cidx * 1024), 16, 0, 0);
    }
    if (wave < 2) {                                   // wave 0: lse[64], wave 1: delta[64]  (raw values)
      int qr = j * kKvQ + lane;
      qr = qr < lq ? qr : lq - 1;
      qr = qr < 0 ? 0 : qr;
      const float* sp = wave ? dltbase : lsebase;
      __builtin_amdgcn_global_load_lds((gptr_t)(sp + qr), (lptr_t)(stat_base + buf * kKvStatBytes + wave * 256), 4, 0, 0);
    }
  };

  int aoff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) aoff[kk] = tile_off(l31, 2 * kk + g);
  int toff[4][2];
#pragma unroll
  for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
      toff[dblk][hh] = (8 * hh + 4 * g + ((lane & 15) >> 2)) * kRowBytes +
                       tr_lane_off(lane, dblk, (2 * hh + g) & 3);

  const float c = p.scale * kLog2e;
  f32x16 dk[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  if (kDma) {
    dma_tile(jt0, 0);
  } else {
    load_tile(jt0);        // unconditional (rows clamped): one path into the loop, see rfa_fwd.hip
    write_tile(0);
  }
  wait_all_vmem();
  __syncthreads();

  for (int j = jt0; j < jt1; ++j) {
    const int buf = (j - jt0) & 1;
    lds_t* qb = smem + buf * kKvTileBytes;
    lds_t* dob = smem + (2 + buf) * kKvTileBytes;
    lds_t* st = stat_base + buf * kKvStatBytes;
    if (j + 1 < jt1) {
      if (kDma) dma_tile(j + 1, buf ^ 1);   // lands in the idle buffer while this tile is computed
      else load_tile(j + 1);
    }
    const int qt0 = j * kKvQ;

    // ---- building blocks of one 32-row sub-tile t ------------------------------------------
    auto read_stats = [&](int t, f32x4 (&l2v)[4], f32x4 (&dlv)[4]) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int rq = 32 * t + 8 * jj + 4 * g;      // rows rq..rq+3 of the tile
        l2v[jj] = *(__attribute__((address_space(3))) f32x4*)(st + rq * 4);
        dlv[jj] = *(__attribute__((address_space(3))) f32x4*)(st + (kKvQ + rq) * 4);
      }
    };
    // S = Q K_w^T and dP = dO V_w^T: two independent accumulator chains, interleaved, A
    // fragments read kAhead ahead of their MFMA (even i: Q/K chain, odd i: dO/V chain)
    // `lead` (compile-time): sub-tile whose 8 stat reads open this pinned pipeline, or -1
    auto gemm1 = [&](int t, f32x16& s, f32x16& dp, f32x4 (&l2v)[4], f32x4 (&dlv)[4], auto lead_c, auto sync) {
      constexpr int sync_id = decltype(sync)::value;   // sched_group_barrier wants a literal
      constexpr int lead = decltype(lead_c)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      if (lead >= 0) read_stats(lead, l2v, dlv);       // 8 LDS reads, first group of the pipeline
      constexpr int kAhead = RFA_KV_AHEAD1;
      vec8<T> a[16];
      // step i -> (chain, k-step).  Back-to-back MFMAs on ONE accumulator measured faster than
      // alternating the two chains (dkdv 1.15 vs 1.33 ms), hence RFA_KV_ILV = 0 by default.
      auto chain = [](int i) { return RFA_KV_ILV ? (i & 1) : (i >> 3); };
      auto kstep = [](int i) { return RFA_KV_ILV ? (i >> 1) : (i & 7); };
      auto frag = [&](int i) { return lds_read128<T>((chain(i) ? dob : qb) + t * 32 * kRowBytes + aoff[kstep(i)]); };
#pragma unroll
      for (int i = 0; i < kAhead; ++i) a[i] = frag(i);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i + kAhead < 16) a[i + kAhead] = frag(i + kAhead);
        if (chain(i)) dp = mfma(a[i], vf[kstep(i)], dp);
        else s = mfma(a[i], kf[kstep(i)], s);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, (lead >= 0 ? 8 : 0) + kAhead, sync_id);
#pragma unroll
      for (int i = 0; i < 16 - kAhead; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, sync_id);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, sync_id);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, kAhead, sync_id);
    };
    // P = exp2(S*c - lse2), dS = P * (dP - delta)   (in place: s <- P, dp <- dS)
    auto softmax_ds = [&](int t, f32x16& s, f32x16& dp, const f32x4 (&l2v)[4], const f32x4 (&dlv)[4], bool masked) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[4 * jj + e] = fast_exp2(__builtin_fmaf(s[4 * jj + e], c, -kLog2e * l2v[jj][e]));
      if (masked) {
        const int qs0 = qt0 + 32 * t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = qs0 + crow(r, g);
          const bool ok = (q < lq) && (!p.causal || krow <= q + off);
          s[r] = ok ? s[r] : 0.f;
        }
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[4 * jj + e] = s[4 * jj + e] * (dp[4 * jj + e] - dlv[jj][e]);
    };
    // dV^T += dO^T P  and  dK^T += Q^T dS : 16 MFMAs on 8 independent accumulators, each fed by
    // two transpose reads issued kAhead MFMAs ahead
    auto gemm2 = [&](int t, const f32x16& s, const f32x16& dp, f32x4 (&l2v)[4], f32x4 (&dlv)[4], auto lead_c, auto sync) {
      constexpr int sync_id = decltype(sync)::value;
      constexpr int lead = decltype(lead_c)::value;
      if (lead >= 0) read_stats(lead, l2v, dlv);
      const vec8<T> pb0 = pack8<T>(s, 0), pb1 = pack8<T>(s, 8);
      const vec8<T> ds0 = pack8<T>(dp, 0), ds1 = pack8<T>(dp, 8);
      constexpr int kAhead = RFA_KV_AHEAD2;
      vec8<T> a[16];
      auto frag = [&](int i) {
        // i: [ks2][which: 0 = dO^T (dV), 1 = Q^T (dK)][dblk]   (ILV: [ks2][dblk][which])
        const int ks2 = i >> 3, dblk = RFA_KV_ILV ? (i >> 1) & 3 : i & 3, which = RFA_KV_ILV ? i & 1 : (i >> 2) & 1;
        lds_t* base = (which ? qb : dob) + (32 * t + 16 * ks2) * kRowBytes;
        vec4<T> lo = lds_read_tr<T>(base + toff[dblk][0]);
        vec4<T> hi = lds_read_tr<T>(base + toff[dblk][1]);
        return concat<T>(lo, hi);
      };
#pragma unroll
      for (int i = 0; i < kAhead; ++i) a[i] = frag(i);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i + kAhead < 16) a[i + kAhead] = frag(i + kAhead);
        const int ks2 = i >> 3, dblk = RFA_KV_ILV ? (i >> 1) & 3 : i & 3, which = RFA_KV_ILV ? i & 1 : (i >> 2) & 1;
        if (which == 0) dv[dblk] = mfma(a[i], ks2 ? pb1 : pb0, dv[dblk]);
        else dk[dblk] = mfma(a[i], ks2 ? ds1 : ds0, dk[dblk]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, (lead >= 0 ? 8 : 0) + 2 * kAhead, sync_id);
#pragma unroll
      for (int i = 0; i < 16 - kAhead; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, sync_id);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, sync_id);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, kAhead, sync_id);
    };

    // wave-uniform classification of the two sub-tiles
    bool act[2], msk[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qs0 = qt0 + 32 * t;
      // inactive: entirely above the diagonal for this wave's keys, or entirely past lq
      act[t] = (kw0 < lk) && (qs0 < lq) && !(p.causal && qs0 + 31 + off < kw0);
      msk[t] = (qs0 + 32 > lq) || (p.causal && qs0 + off < kw0 + 31);
    }
    if (RFA_KV_PIPE && act[0] && act[1] && !msk[0] && !msk[1]) {
      // ---- steady state (all but the diagonal / tail tiles): software pipeline across the two
      // sub-tiles so that the exp2 / dS VALU work of one runs under the MFMAs of the other
      f32x16 s0, dp0, s1, dp1;
      f32x4 l2a[4], dla[4], l2b[4], dlb[4];
      using I = std::integral_constant<int, -1>;
      gemm1(0, s0, dp0, l2a, dla, I{}, std::integral_constant<int, 0>{});
      gemm1(1, s1, dp1, l2a, dla, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      softmax_ds(0, s0, dp0, l2a, dla, false);      // overlaps gemm1(1)
      gemm2(0, s0, dp0, l2b, dlb, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
      softmax_ds(1, s1, dp1, l2b, dlb, false);      // overlaps gemm2(0)
      gemm2(1, s1, dp1, l2b, dlb, I{}, std::integral_constant<int, 3>{});
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (act[t]) {
          f32x16 s, dp;
          f32x4 l2v[4], dlv[4];
          if (t == 0) gemm1(0, s, dp, l2v, dlv, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
          else gemm1(1, s, dp, l2v, dlv, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
          softmax_ds(t, s, dp, l2v, dlv, msk[t]);
          gemm2(t, s, dp, l2v, dlv, std::integral_constant<int, -1>{}, std::integral_constant<int, 5>{});
        }
      }
    }
    if (kDma) {
      wait_all_vmem();                       // this wave's DMA pieces landed; the barrier publishes all
    } else if (j + 1 < jt1) {
      write_tile(buf ^ 1);
    }
    __syncthreads();
  }

  if (krow >= lk) return;
  const int64_t orow = ks.row0 + krow;
  T* dkb = (T*)p.dk + kbatch * p.dk_st.batch + orow * p.dk_st.row + (int64_t)h * p.dk_st.head;
  T* dvb = (T*)p.dv + kbatch * p.dv_st.batch + orow * p.dv_st.row + (int64_t)h * p.dv_st.head;
  store_rows16<T, kFullD>(dkb, dk, p.scale, g, p.D, true);
  store_rows16<T, kFullD>(dvb, dv, 1.f, g, p.D, true);
}

// =====================================================================================
// dK/dV kernel, wave-specialised form (alternative, RFA_KV_SPECIALIZED=1): 8 waves = 4 key blocks x 2 roles
// =====================================================================================
// Wave (kb, role): kb = 32-key block of the workgroup's 128 keys; the two roles of a key block
// sit on the same SIMD (waves w and w+4) and split the five-GEMM chain WITHOUT duplicating work:
//     role A:  S = Q·K_wᵀ  ->  P = exp2(S·c − lse2)  ->  dVᵀ += dOᵀ·P        (16 MFMA / sub-tile)
//     role B:  dP = dO·V_wᵀ ............ dS = P∘(dP − Δ) -> dKᵀ += Qᵀ·dS      (16 MFMA / sub-tile)
// P crosses from A to B as packed 16-bit values through a 2 KiB LDS slot per key block (double
// buffered), published by the one workgroup barrier per 32-row sub-tile.  Each wave needs < 200
// registers, so two waves share a SIMD and one wave's exp2 / dS VALU work runs under the other's
// MFMAs — the overlap the single-wave-per-SIMD form (dkdv_kernel above) cannot get.
constexpr int kKv2Waves = 8;
constexpr int kKv2Threads = kKv2Waves * 64;
constexpr int kKv2PBytes = 4 * 2048;                               // one P slot set: 4 key blocks x 2 KiB
constexpr int kKv2Smem = 4 * kKvTileBytes + 2 * kKvStatBytes + 2 * kKv2PBytes;

template <typename T, bool kFullD>
__global__ __launch_bounds__(kKv2Threads, 2) void dkdv2_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  lds_t* stat_base = smem + 4 * kKvTileBytes;
  lds_t* pbase = stat_base + 2 * kKvStatBytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kbw = wave & 3;           // key block inside the workgroup
  const int role = wave >> 2;         // 0 = A (S, P, dV)   1 = B (dP, dS, dK)
  const int g = lane >> 5;
  const int l31 = lane & 31;

  int idx = blockIdx.x;
  const int G = p.H / p.Hk;
  const int hk = idx % p.Hk;
  idx /= p.Hk;
  const int gq = idx % G;
  idx /= G;
  const int kblk = idx % p.nkblk;
  const int b = idx / p.nkblk;
  const int h = hk * G + gq;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int kwg0 = kblk * kKvKeys;
  if (kwg0 >= lk) return;
  const int off = lk - lq;
  const int kw0 = kwg0 + kbw * 32;
  const int krow = kw0 + l31;
  const int krow_c = krow < lk ? krow : lk - 1;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;

  // role A keeps K_w, role B keeps V_w (B operands: lane key = l31, d = 16kk + 8g .. +7)
  const T* wbase = (role == 0 ? (const T*)p.k + kbatch * p.k_st.batch + (ks.row0 + krow_c) * p.k_st.row +
                                    (int64_t)hk * p.k_st.head
                              : (const T*)p.v + kbatch * p.v_st.batch + (ks.row0 + krow_c) * p.v_st.row +
                                    (int64_t)hk * p.v_st.head);
  const T* qbase = (const T*)p.q + qbatch * p.q_st.batch + qs.row0 * p.q_st.row + (int64_t)h * p.q_st.head;
  const T* dobase = (const T*)p.dout + qbatch * p.dout_st.batch + qs.row0 * p.dout_st.row +
                    (int64_t)h * p.dout_st.head;
  const float* lsebase = p.lse + qbatch * p.lse_batch + (int64_t)h * p.lse_head + qs.row0;
  const float* dltbase = p.delta + qbatch * p.delta_batch + (int64_t)h * p.delta_head + qs.row0;

  vec8<T> wf[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int d0 = 16 * kk + 8 * g;
    wf[kk] = (kFullD || d0 < p.D) ? *(const vec8<T>*)(wbase + d0) : zero8<T>();
  }

  int qfirst = 0;
  if (p.causal) {
    qfirst = kwg0 - off;
    if (qfirst < 0) qfirst = 0;
  }
  const int jt0 = qfirst / kKvQ;
  const int jt1 = (lq + kKvQ - 1) / kKvQ;     // exclusive

  // staging: 512 threads -> chunk sc of rows sr + 32 i (i = 0,1), for Q and dO
  const int sc = tid & 15;
  const int sr = tid >> 4;                    // 0..31
  const bool sd_ok = kFullD || sc * 8 < p.D;
  vec8<T> qreg[2], doreg[2];
  float statreg = 0.f;
  auto load_tile = [&](int j) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int qr = j * kKvQ + sr + 32 * i;
      qr = qr < lq ? qr : lq - 1;
      qr = qr < 0 ? 0 : qr;
      if (sd_ok) {
        qreg[i] = *(const vec8<T>*)(qbase + (int64_t)qr * p.q_st.row + sc * 8);
        doreg[i] = *(const vec8<T>*)(dobase + (int64_t)qr * p.dout_st.row + sc * 8);
      } else {
        qreg[i] = zero8<T>();
        doreg[i] = zero8<T>();
      }
    }
    {
      int qr = j * kKvQ + (tid & (kKvQ - 1));
      qr = qr < lq ? qr : lq - 1;
      qr = qr < 0 ? 0 : qr;
      const float* sp = (tid & kKvQ) ? dltbase : lsebase;
      statreg = sp[qr];                        // no use here (see dkdv_kernel)
    }
  };
  auto write_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = tile_off(sr + 32 * i, sc);
      lds_write128<T>(smem + buf * kKvTileBytes + o, qreg[i]);
      lds_write128<T>(smem + (2 + buf) * kKvTileBytes + o, doreg[i]);
    }
    if (tid < 2 * kKvQ)
      *(__attribute__((address_space(3))) float*)(stat_base + buf * kKvStatBytes + tid * 4) =
          tid < kKvQ ? statreg * kLog2e : statreg;
  };

  int aoff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) aoff[kk] = tile_off(l31, 2 * kk + g);
  int toff[4][2];
#pragma unroll
  for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
      toff[dblk][hh] = (8 * hh + 4 * g + ((lane & 15) >> 2)) * kRowBytes +
                       tr_lane_off(lane, dblk, (2 * hh + g) & 3);
  lds_t* pslot = pbase + kbw * 2048 + lane * 16;      // two lane-contiguous 1 KiB planes; + sub * kKv2PBytes

  const float c = p.scale * kLog2e;
  f32x16 acc[4];                                      // role A: dV^T, role B: dK^T
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  load_tile(jt0);
  write_tile(0);
  wait_all_vmem();
  __syncthreads();

  // Skewed ("ping-pong") schedule, one workgroup barrier per 32-row sub-tile u:
  //   period u, role A:  dV(u-1) MFMAs | S(u) MFMAs | exp2/mask/pack(u) VALU | P(u) -> LDS
  //   period u, role B:  P(u-1) <- LDS, dS(u-1) VALU | dK(u-1) MFMAs | dP(u) MFMAs
  // so on each SIMD one wave's VALU stretch runs under the other wave's MFMA stretch.
  vec8<T> pk0 = zero8<T>(), pk1 = zero8<T>();   // role A: packed P of the previous sub-tile
  f32x16 xprev;                                  // role B: dP of the previous sub-tile
#pragma unroll
  for (int r = 0; r < 16; ++r) xprev[r] = 0.f;
  bool prev_active = false;
  lds_t* prev_q = smem;
  lds_t* prev_do = smem;
  lds_t* prev_st = stat_base;
  int prev_t = 0;

  // second half of sub-tile (prev): 8 MFMAs into acc (role A: dV^T += dO^T P, role B: dK^T += Q^T dS)
  auto finish_prev = [&]() {
    if (!prev_active) return;
    vec8<T> b0, b1;
    lds_t* trsrc;
    if (role == 0) {
      b0 = pk0;
      b1 = pk1;
      trsrc = prev_do;
    } else {
      lds_t* pr = pslot + prev_t * kKv2PBytes;
      const vec8<T> p0 = lds_read128<T>(pr);
      const vec8<T> p1 = lds_read128<T>(pr + 1024);
      f32x16 ds;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x4 dlv = *(__attribute__((address_space(3))) f32x4*)(prev_st + (kKvQ + 32 * prev_t + 8 * jj + 4 * g) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * jj + e;
          const float pv = (float)(r < 8 ? p0[r] : p1[r - 8]);
          ds[r] = pv * (xprev[r] - dlv[e]);          // dS = P (dP - delta)
        }
      }
      b0 = pack8<T>(ds, 0);
      b1 = pack8<T>(ds, 8);
      trsrc = prev_q;
    }
    constexpr int kAhead = 2;
    vec8<T> a[8];
    auto frag = [&](int i) {                          // i: [ks2][dblk]
      lds_t* base = trsrc + (32 * prev_t + 16 * (i >> 2)) * kRowBytes;
      vec4<T> lo = lds_read_tr<T>(base + toff[i & 3][0]);
      vec4<T> hi = lds_read_tr<T>(base + toff[i & 3][1]);
      return concat<T>(lo, hi);
    };
#pragma unroll
    for (int i = 0; i < kAhead; ++i) a[i] = frag(i);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i + kAhead < 8) a[i + kAhead] = frag(i + kAhead);
      acc[i & 3] = mfma(a[i], (i >> 2) ? b1 : b0, acc[i & 3]);
    }
  };

  for (int j = jt0; j < jt1; ++j) {
    const int buf = (j - jt0) & 1;
    lds_t* qb = smem + buf * kKvTileBytes;
    lds_t* dob = smem + (2 + buf) * kKvTileBytes;
    lds_t* st = stat_base + buf * kKvStatBytes;
    if (j + 1 < jt1) load_tile(j + 1);
    const int qt0 = j * kKvQ;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qs0 = qt0 + 32 * t;
      const bool active = (kw0 < lk) && (qs0 < lq) && !(p.causal && qs0 + 31 + off < kw0);
      const bool masked = (qs0 + 32 > lq) || (p.causal && qs0 + off < kw0 + 31);

      finish_prev();

      f32x16 x;                                       // A: S -> P     B: dP
      if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = 0.f;
        lds_t* src = (role == 0 ? qb : dob) + t * 32 * kRowBytes;
        {
          constexpr int kAhead = 3;
          vec8<T> a[8];
#pragma unroll
          for (int i = 0; i < kAhead; ++i) a[i] = lds_read128<T>(src + aoff[i]);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (i + kAhead < 8) a[i + kAhead] = lds_read128<T>(src + aoff[i + kAhead]);
            x = mfma(a[i], wf[i], x);
          }
        }
        if (role == 0) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x4 l2v = *(__attribute__((address_space(3))) f32x4*)(st + (32 * t + 8 * jj + 4 * g) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * jj + e] = fast_exp2(__builtin_fmaf(x[4 * jj + e], c, -l2v[e]));
          }
          if (masked) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int q = qs0 + crow(r, g);
              const bool ok = (q < lq) && (!p.causal || krow <= q + off);
              x[r] = ok ? x[r] : 0.f;
            }
          }
          pk0 = pack8<T>(x, 0);
          pk1 = pack8<T>(x, 8);
          lds_t* pw = pslot + t * kKv2PBytes;
          lds_write128<T>(pw, pk0);
          lds_write128<T>(pw + 1024, pk1);
        } else {
          xprev = x;
        }
      }
      prev_active = active;
      prev_q = qb;
      prev_do = dob;
      prev_st = st;
      prev_t = t;
      // tile j+1 goes to LDS under the second sub-tile's barrier (its buffer is idle during tile j
      // and during the first period of tile j+1, which still finishes sub-tile (j, 1))
      if (t == 1 && j + 1 < jt1) write_tile(buf ^ 1);
      __syncthreads();
    }
  }
  finish_prev();                                      // drain: second half of the last sub-tile

  if (krow >= lk) return;
  const int64_t orow = ks.row0 + krow;
  T* ob = role == 0 ? (T*)p.dv + kbatch * p.dv_st.batch + orow * p.dv_st.row + (int64_t)h * p.dv_st.head
                    : (T*)p.dk + kbatch * p.dk_st.batch + orow * p.dk_st.row + (int64_t)h * p.dk_st.head;
  const float oscale = role == 0 ? 1.f : p.scale;
  store_rows16<T, kFullD>(ob, acc, oscale, g, p.D, true);
}

template <typename T, bool kFullD>
static int launch_dq_t(const BwdParams& p, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)dq_kernel<T, kFullD>, hipFuncAttributeMaxDynamicSharedMemorySize, kDqSmem);
    attr_done = true;
  }
  const int64_t nblocks = (int64_t)p.nqblk * p.H * p.B;
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL((dq_kernel<T, kFullD>), dim3((unsigned)nblocks), dim3(kDqThreads), kDqSmem, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename T, bool kFullD>
static int launch_dkdv_t(const BwdParams& p, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)dkdv_kernel<T, kFullD>, hipFuncAttributeMaxDynamicSharedMemorySize, kKvSmem);
    attr_done = true;
  }
  const int64_t nblocks = (int64_t)p.nkblk * p.H * p.B;
  if (nblocks <= 0) return 0;
#if RFA_KV_SPECIALIZED
  static bool attr2_done = false;
  if (!attr2_done) {
    (void)hipFuncSetAttribute((const void*)dkdv2_kernel<T, kFullD>, hipFuncAttributeMaxDynamicSharedMemorySize, kKv2Smem);
    attr2_done = true;
  }
  hipLaunchKernelGGL((dkdv2_kernel<T, kFullD>), dim3((unsigned)nblocks), dim3(kKv2Threads), kKv2Smem, stream, p);
#else
  hipLaunchKernelGGL((dkdv_kernel<T, kFullD>), dim3((unsigned)nblocks), dim3(kKvThreads), kKvSmem, stream, p);
#endif
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_bwd_dq(const BwdParams& p, int dtype, hipStream_t stream) {
  const bool full = p.D == kHeadDim;
  if (dtype == 0) return full ? launch_dq_t<bf16_t, true>(p, stream) : launch_dq_t<bf16_t, false>(p, stream);
  return full ? launch_dq_t<f16_t, true>(p, stream) : launch_dq_t<f16_t, false>(p, stream);
}
int launch_bwd_dkdv(const BwdParams& p, int dtype, hipStream_t stream) {
  const bool full = p.D == kHeadDim;
  if (dtype == 0) return full ? launch_dkdv_t<bf16_t, true>(p, stream) : launch_dkdv_t<bf16_t, false>(p, stream);
  return full ? launch_dkdv_t<f16_t, true>(p, stream) : launch_dkdv_t<f16_t, false>(p, stream);
}
int bwd_dq_rows_per_block() { return kDqRows; }
int bwd_dkdv_keys_per_block() { return kKvKeys; }

}  // namespace rfa
