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
//   * dkdv_kernel  : a workgroup owns 128 keys of ONE query head (K, V rows resident in LDS),
//                    streams Q/dO tiles through LDS, accumulates dK,dV in registers (8 waves =
//                    4 key blocks x 2 sub-tile parities).  GQA group reduction is a separate
//                    HBM-bound kernel (rfa_aux.hip: reduce_kernel), like flash_attn's
//                    dk_expanded + sum.
// Lane ownership mirrors the forward kernel (see rfa_common.hpp): after the first GEMM a
// lane owns one query row (dQ kernel) or one key (dK/dV kernel), and the probabilities go
// straight from the accumulator registers into the B operand of the second GEMM.
#include <type_traits>

#include "rfa_common.hpp"
#include "rfa_kernels.hpp"

// ---- tuning knobs (A/B'd on hardware with tools/ab_variants.py; defaults = best measured) ----
#ifndef RFA_DQ_PIN
#define RFA_DQ_PIN 0          // >0: pin the dQ transpose-read/MFMA pipeline with this read-ahead depth
#endif
#ifndef RFA_DQ_AHEAD1
#define RFA_DQ_AHEAD1 3
#endif
#ifndef RFA_KV_AHEAD
#define RFA_KV_AHEAD 2       // dkdv: fragment pairs read this many MFMAs ahead in the S/dP GEMMs (3, 4: +1.5 %)
#endif
#ifndef RFA_KV_AHEAD2
#define RFA_KV_AHEAD2 2      // dkdv: transpose-read fragment pairs ahead in the dV/dK GEMMs
#endif
#ifndef RFA_KV_PRIO
#define RFA_KV_PRIO 0        // 1: the two waves of a SIMD get different priorities (measured neutral)
#endif
#ifndef RFA_KV_PIN
#define RFA_KV_PIN 1         // pin the LDS-read / MFMA interleave with sched_group_barrier (0: +5 %)
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
// A workgroup owns 128 keys of ONE query head: 8 waves = 4 key blocks (32 keys) x 2 parities.  Wave
// (kb, par) processes the 32-row sub-tile t = par of every 64-row Q/dO tile for key block kb and
// accumulates its own partial dK^T / dV^T (128 accumulator registers).  Two waves share a SIMD
// (<= 256 registers each), so the K_w / V_w B operands are NOT register resident: the workgroup's 128
// K and V rows are staged once into swizzled LDS tiles and re-read as fragments every sub-tile
// (1.5 KiB of LDS reads per MFMA instead of 1.0 — LDS reads stay below their 256 B/clk roof, see
// DESIGN.md §4).  The two parities' partials are combined through LDS at the end: parity 1 hands over
// dK, parity 0 hands over dV, each then finishes and stores one of the two tensors.
// (Round-1 history: a 4-wave / 512-register form with K_w, V_w in registers measured 4.5 % slower,
//  a role-split producer/consumer form 18 % slower — DESIGN.md §7.)
constexpr int kKvWaves = 8;
constexpr int kKvThreads = kKvWaves * 64;
constexpr int kKvKeys = 128;                       // keys / workgroup
constexpr int kKvQ = 64;                           // query rows per tile (2 sub-tiles of 32)
constexpr int kKvTileBytes = kKvQ * kRowBytes;     // 16 KiB
constexpr int kKvStatBytes = 2 * kKvQ * 4;         // lse[64] + delta[64] per stage
constexpr int kKvKvBytes = kKvKeys * kRowBytes;    // 32 KiB per K / V tile
constexpr int kKvSmem = 2 * kKvKvBytes + 4 * kKvTileBytes + 2 * kKvStatBytes;   // 129 KiB

template <typename T, bool kFullD>
__global__ __launch_bounds__(kKvThreads, 2) void dkdv_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  lds_t* ktile = smem;                                // [128 keys][128] swizzled
  lds_t* vtile = smem + kKvKvBytes;
  lds_t* qd = smem + 2 * kKvKvBytes;                 // Q[2] then dO[2], 16 KiB each
  lds_t* stat_base = qd + 4 * kKvTileBytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kbw = wave & 3;
  const int par = wave >> 2;
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
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;

  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + ks.row0 * p.v_st.row + (int64_t)hk * p.v_st.head;
  const T* qbase = (const T*)p.q + qbatch * p.q_st.batch + qs.row0 * p.q_st.row + (int64_t)h * p.q_st.head;
  const T* dobase = (const T*)p.dout + qbatch * p.dout_st.batch + qs.row0 * p.dout_st.row +
                    (int64_t)h * p.dout_st.head;
  const float* lsebase = p.lse + qbatch * p.lse_batch + (int64_t)h * p.lse_head + qs.row0;
  const float* dltbase = p.delta + qbatch * p.delta_batch + (int64_t)h * p.delta_head + qs.row0;

  int qfirst = 0;
  if (p.causal) {
    qfirst = kwg0 - off;
    if (qfirst < 0) qfirst = 0;
  }
  const int jt0 = qfirst / kKvQ;
  const int jt1 = (lq + kKvQ - 1) / kKvQ;     // exclusive

  const int sc = tid & 15;
  const int sr = tid >> 4;                    // 0..31
  const bool sd_ok = kFullD || sc * 8 < p.D;

  // ---- stage the workgroup's K / V rows once: 128 rows x 16 chunks x 2 tensors = 8 chunks / thread
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = sr + 32 * i;
    int kr = kwg0 + row;
    kr = kr < lk ? kr : lk - 1;
    vec8<T> kc = zero8<T>(), vc = zero8<T>();
    if (sd_ok) {
      kc = *(const vec8<T>*)(kbase + (int64_t)kr * p.k_st.row + sc * 8);
      vc = *(const vec8<T>*)(vbase + (int64_t)kr * p.v_st.row + sc * 8);
    }
    lds_write128<T>(ktile + tile_off(row, sc), kc);
    lds_write128<T>(vtile + tile_off(row, sc), vc);
  }

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
      statreg = sp[qr];                        // raw value: any arithmetic here would wait on the load and drain the prefetch
    }
  };
  auto write_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = tile_off(sr + 32 * i, sc);
      lds_write128<T>(qd + buf * kKvTileBytes + o, qreg[i]);
      lds_write128<T>(qd + (2 + buf) * kKvTileBytes + o, doreg[i]);
    }
    if (tid < 2 * kKvQ)
      *(__attribute__((address_space(3))) float*)(stat_base + buf * kKvStatBytes + tid * 4) = statreg;
  };

  // Fragment offsets are kept as ONE base each and derived with an XOR at the point of use (the swizzle
  // makes chunk selection an XOR on address bits 4..7): 16 fewer live registers than offset tables,
  // which is what lets this kernel fit the 256-register budget of two waves per SIMD.
  int aoff0 = tile_off(l31, g);                       // aoff(kk) = aoff0 ^ (kk << 5)
  int toff0[2];                                        // toff(dblk, hh) = toff0[hh] ^ (dblk << 6)
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
    toff0[hh] = (8 * hh + 4 * g + ((lane & 15) >> 2)) * kRowBytes + tr_lane_off(lane, 0, (2 * hh + g) & 3);
  lds_t* kw = ktile + kbw * 32 * kRowBytes;          // this wave's 32 key rows (same lane map as aoff)
  lds_t* vw = vtile + kbw * 32 * kRowBytes;

  const float c = p.scale * kLog2e;
  f32x16 dk[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  load_tile(jt0);
  write_tile(0);
  wait_all_vmem();
  __syncthreads();

  const int t = par;                                  // this wave's sub-tile of every Q tile
#if RFA_KV_PRIO
  if (par == 0) __builtin_amdgcn_s_setprio(2);
#endif
  for (int j = jt0; j < jt1; ++j) {
    const int buf = (j - jt0) & 1;
    asm volatile("" : "+v"(aoff0), "+v"(toff0[0]), "+v"(toff0[1]));   // keep the XORs inside the loop
    lds_t* qb = qd + buf * kKvTileBytes;
    lds_t* dob = qd + (2 + buf) * kKvTileBytes;
    lds_t* st = stat_base + buf * kKvStatBytes;
    if (j + 1 < jt1) load_tile(j + 1);
    const int qs0 = j * kKvQ + 32 * t;
    const bool active = (kw0 < lk) && (qs0 < lq) && !(p.causal && qs0 + 31 + off < kw0);
    if (active) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      {
        // S = Q K_w^T then dP = dO V_w^T: 16 MFMAs, BOTH operands from LDS, read kAhead steps ahead
        constexpr int kAhead = RFA_KV_AHEAD;
        vec8<T> a[16], w[16];
        auto fa = [&](int i) { return lds_read128<T>((i < 8 ? qb : dob) + t * 32 * kRowBytes + (aoff0 ^ ((i & 7) << 5))); };
        auto fw = [&](int i) { return lds_read128<T>((i < 8 ? kw : vw) + (aoff0 ^ ((i & 7) << 5))); };
#pragma unroll
        for (int i = 0; i < kAhead; ++i) { a[i] = fa(i); w[i] = fw(i); }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i + kAhead < 16) { a[i + kAhead] = fa(i + kAhead); w[i + kAhead] = fw(i + kAhead); }
          if (i < 8) s = mfma(a[i], w[i], s);
          else dp = mfma(a[i], w[i], dp);
        }
#if RFA_KV_PIN
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * kAhead, 0);
#pragma unroll
        for (int i = 0; i < 16 - kAhead; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
#endif
      }
      // row statistics are read only now: holding them across GEMM 1 would cost 32 registers
      f32x4 l2v[4], dlv[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int rq = 32 * t + 8 * jj + 4 * g;
        l2v[jj] = *(__attribute__((address_space(3))) f32x4*)(st + rq * 4);
        dlv[jj] = *(__attribute__((address_space(3))) f32x4*)(st + (kKvQ + rq) * 4);
      }
      const bool need_mask = (qs0 + 32 > lq) || (p.causal && qs0 + off < kw0 + 31);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          s[4 * jj + e] = fast_exp2(__builtin_fmaf(s[4 * jj + e], c, -kLog2e * l2v[jj][e]));
      if (need_mask) {
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
      {
        const vec8<T> pb0 = pack8<T>(s, 0), pb1 = pack8<T>(s, 8);
        const vec8<T> ds0 = pack8<T>(dp, 0), ds1 = pack8<T>(dp, 8);
        constexpr int kAhead = RFA_KV_AHEAD2;
        vec8<T> a[16];
        auto frag = [&](int i) {                       // i: [ks2][which: 0 = dO^T (dV), 1 = Q^T (dK)][dblk]
          const int ks2 = i >> 3, which = (i >> 2) & 1, dblk = i & 3;
          lds_t* base = (which ? qb : dob) + (32 * t + 16 * ks2) * kRowBytes;
          vec4<T> lo = lds_read_tr<T>(base + (toff0[0] ^ (dblk << 6)));
          vec4<T> hi = lds_read_tr<T>(base + (toff0[1] ^ (dblk << 6)));
          return concat<T>(lo, hi);
        };
#pragma unroll
        for (int i = 0; i < kAhead; ++i) a[i] = frag(i);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i + kAhead < 16) a[i + kAhead] = frag(i + kAhead);
          const int ks2 = i >> 3, which = (i >> 2) & 1, dblk = i & 3;
          if (which == 0) dv[dblk] = mfma(a[i], ks2 ? pb1 : pb0, dv[dblk]);
          else dk[dblk] = mfma(a[i], ks2 ? ds1 : ds0, dk[dblk]);
        }
#if RFA_KV_PIN
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * kAhead, 1);
#pragma unroll
        for (int i = 0; i < 16 - kAhead; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 1);
#endif
      }
    }
    if (j + 1 < jt1) write_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- combine the two parities: parity 1 hands over its dK^T partial, parity 0 its dV^T partial
  // (fp32, [kb][dblk][r][lane] so that the partner lane reads exactly what its twin wrote); the K/V
  // tiles and the Q/dO buffers (128 KiB, contiguous) are dead by now and take the 2 x 64 KiB of partials.
  {
    float* xbuf = (float*)smem_raw;                    // generic pointer into LDS
    const int slot = (par == 1 ? 0 : 16384) + kbw * 4096;  // floats: 4 kb x 4096 per tensor (2 x 64 KiB)
    const f32x16(&mine)[4] = (par == 1) ? dk : dv;
#pragma unroll
    for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
      for (int r = 0; r < 16; ++r) xbuf[slot + (dblk * 16 + r) * 64 + lane] = mine[dblk][r];
    __syncthreads();
    const int rslot = (par == 0 ? 0 : 16384) + kbw * 4096;  // parity 0 finishes dK, parity 1 finishes dV
    f32x16(&fin)[4] = (par == 0) ? dk : dv;
#pragma unroll
    for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
      for (int r = 0; r < 16; ++r) fin[dblk][r] += xbuf[rslot + (dblk * 16 + r) * 64 + lane];
  }

  if (krow >= lk) return;
  const int64_t orow = ks.row0 + krow;
  if (par == 0) {
    T* dkb = (T*)p.dk + kbatch * p.dk_st.batch + orow * p.dk_st.row + (int64_t)h * p.dk_st.head;
    store_rows16<T, kFullD>(dkb, dk, p.scale, g, p.D, true);
  } else {
    T* dvb = (T*)p.dv + kbatch * p.dv_st.batch + orow * p.dv_st.row + (int64_t)h * p.dv_st.head;
    store_rows16<T, kFullD>(dvb, dv, 1.f, g, p.D, true);
  }
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
  hipLaunchKernelGGL((dkdv_kernel<T, kFullD>), dim3((unsigned)nblocks), dim3(kKvThreads), kKvSmem, stream, p);
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
