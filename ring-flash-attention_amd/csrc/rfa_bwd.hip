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
//                    stores it or adds it into a caller fp32 accumulator (ring steps).  The 7-GEMM
//                    form: head dim != 128, bounded left windows, spill switched off / over its
//                    memory limit.  Otherwise dQ comes from rfa_dqs.hip (dq_ds_kernel), which streams
//                    the dS blocks the dK/dV kernel spilled (kSpill) instead of recomputing S and dP.
//   * dkdv_kernel  : a workgroup owns 128 (256: kWide) keys of ONE K/V head (its V rows resident in
//                    LDS, each wave's K rows in registers), streams the Q/dO tiles of ALL query heads
//                    of that K/V group through LDS and accumulates dK,dV of the whole group in
//                    registers — no per-head partials and no group-reduction pass.  8 waves = 4 key
//                    blocks x 2 sub-tile parities, or (kWide, the headline launch) 8 key blocks whose
//                    waves run both sub-tiles; the kWide form may share a key block's query range
//                    between workgroups (rfa_api.cpp: bwd_dkdv_plan), whose fp32 partials
//                    rfa_aux.hip's reduce_kernel sums (one rounding to the io dtype, like an unsplit launch).  reduce_kernel also serves accumulate /
//                    two-phase calls (workspace partials -> fp32 accumulators).
// Lane ownership mirrors the forward kernel (see rfa_common.hpp): after the first GEMM a
// lane owns one query row (dQ kernel) or one key (dK/dV kernel), and the probabilities go
// straight from the accumulator registers into the B operand of the second GEMM.
#include <type_traits>

#include "rfa_common.hpp"
#include "rfa_kernels.hpp"

// ---- tuning knobs (A/B'd on hardware with tools/ab_variants.py; defaults = best measured) ----
#ifndef RFA_DQ_AHEAD1
#define RFA_DQ_AHEAD1 3
#endif
#ifndef RFA_KV_AHEAD
#define RFA_KV_AHEAD 3       // dkdv: fragment pairs read this many MFMAs ahead in the S/dP GEMMs (rounds 1-3, one-body loop: 3, 4 were
                             // 1.5 % slower than 2; with the unrolled sub-tile bodies 1 -> 1.117 ms, 2 -> 1.098, 3 -> 1.081,
                             // 4 -> 1.090 against 1.080; not kept from the same runs: the first transposed fragments of the dV / dK pair
                             // fetched before the exponentials (255 registers, equal), the fragment bases aq ^ (kk << 5) / tq ^ (dblk << 6)
                             // formed once per tile for both sub-tiles (249 - 255 registers, 0 - 1 % slower), RFA_KV_AHEAD2 3 (slower))
#endif
#ifndef RFA_KV_WIDE_UNROLL
#define RFA_KV_WIDE_UNROLL 1 // dkdv kWide: 1 = the two sub-tile bodies unrolled, 0 = a runtime loop over one body.  Round 4, after the mask
                             // rewrite had freed 15 registers (244 registers, no scratch): the headline launch 1.092 -> 1.063 ms, the whole
                             // step 2.033 -> 2.000 ms (profiles/history/r04_dkdv_variants.txt, three passes).  The unrolled bodies address the
                             // second sub-tile with instruction immediates (+ 32 rows) instead of toggling every fragment base
#endif
#ifndef RFA_KV_AHEAD2
#define RFA_KV_AHEAD2 2      // dkdv: transpose-read fragment pairs ahead in the dV/dK GEMMs
#endif
// measurement-only switches for dkdv_kernel (results are wrong when one is 0): loop cost without the staging loads /
// the LDS fragment reads / the exp-mask-multiply work / the per-tile wait + barrier
#ifndef RFA_KV_X_LOAD
#define RFA_KV_X_LOAD 1
#endif
#ifndef RFA_KV_X_LDS
#define RFA_KV_X_LDS 1       // 0: no fragment reads; 2 (round 6): only sub-tile 0 of the 256-key form reads its fragments — HALF the
#endif                       // reads per MFMA, the upper bound of what a 64-keys-per-wave / one-wave-per-SIMD form could save
#ifndef RFA_KV_X_VALU
#define RFA_KV_X_VALU 1
#endif
#ifndef RFA_KV_X_SYNC
#define RFA_KV_X_SYNC 1
#endif
#ifndef RFA_KV_PRIO
#define RFA_KV_PRIO 0        // 1: the two waves of a SIMD get different priorities (measured neutral); 2: waves 4-7 at s_setprio 1
                             // (neutral); s_setprio 1 around 3: the dP / S GEMM pair, 4: the dV / dK GEMM pair, 5: both.  Round 4 A/B of
                             // the headline launch (profiles/history/r04_dkdv_variants.txt): with the one-body loop 0 -> 1.0870 ms, 3 -> 1.0836,
                             // 4 -> 1.0788, 5 -> 1.0965; with the two sub-tile bodies unrolled (the default now) 4 and 0 are equal
                             // (whole step 2.001 vs 1.994 ms over three passes), so the default stays 0
#endif
#ifndef RFA_SPILL_AUX
#define RFA_SPILL_AUX 2      // cache policy bits of the dS spill stores: 2 = nt (streamed once; 0: dkdv +3 %)
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
template <int kD> constexpr int dq_smem() { return 4 * kDqKV * HeadGeo<kD>::kRowBytes; }   // K[2] V[2]

// kD: compiled head dim (128 / 64); kFullD: D == kD (LDS-DMA staging), else zero padded (register staging)
// kDrop: dropout (the forward's mask, rfa_common.hpp: drop_word) applied to dP; instances without a window only
template <typename T, int kD, bool kFullD, bool kWin, bool kDrop = false>
__global__ __launch_bounds__(kDqThreads, 2) void dq_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  typedef HeadGeo<kD> Geo;
  constexpr int kRowBytes = Geo::kRowBytes;                  // (shadows the 128-wide namespace constant)
  constexpr int kNK = Geo::kKSteps, kNB = Geo::kDBlocks;
  constexpr int kDqTileBytes = kDqKV * kRowBytes;            // 16 KiB (8 KiB at kD = 64)
  constexpr int kShare = kDqTileBytes / 1024 / kDqWaves;     // 1 KiB DMA pieces per wave and tile
  constexpr int kChunks = Geo::kLay / 8;
  constexpr int kRowsPerPass = kDqThreads / kChunks;

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
  int qblk_i, b;
  split_block_batch<RFA_BATCH_FAST_Q>(idx, p.nqblk, p.B, qblk_i, b);
  const int qblk = p.nqblk - 1 - qblk_i;
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

  vec8<T> qf[kNK], dof[kNK];
#pragma unroll
  for (int kk = 0; kk < kNK; ++kk) {
    const int d0 = 16 * kk + 8 * g;
    qf[kk] = (kFullD || d0 < p.D) ? *(const vec8<T>*)(qbase + d0) : zero8<T>();
    dof[kk] = (kFullD || d0 < p.D) ? *(const vec8<T>*)(dobase + d0) : zero8<T>();
  }
  const float L2 = p.lse[qbatch * p.lse_batch + (int64_t)h * p.lse_head + arow] * kLog2e;
  const float dlt = p.delta[qbatch * p.delta_batch + (int64_t)h * p.delta_head + arow];

  const int qend = (qwg0 + kDqRows < lq) ? qwg0 + kDqRows : lq;
  // Attention band: query row i sees keys [i + off - wl, i + off + wr], each bound only if set.  Windowed calls
  // (a left bound, or a right bound without `causal`) run the kWin instances; the default instances keep the plain
  // causal logic (hi = causal, wr = 0, no left bound) so that their hot loops stay free of the extra predicates.
  const bool hi = kWin ? p.wr >= 0 : p.causal != 0;
  const bool lo = kWin && p.wl >= 0;
  const int wr = kWin ? p.wr : 0, wl = kWin ? p.wl : 0;
  int kmax = lk;
  if (hi && qend + off + wr < kmax) kmax = qend + off + wr;
  const int ntiles = kmax > 0 ? (kmax + kDqKV - 1) / kDqKV : 0;
  int kmin = lo ? qwg0 + off - wl : 0;
  kmin = kmin > 0 ? kmin : 0;
  const int jt0 = (kmin / kDqKV) & ~1;                  // first tile (even: LDS stage = j & 1)

  const int sc = tid % kChunks;
  const int sr = tid / kChunks;
  const bool sd_ok = (kFullD && kD == Geo::kLay) || sc * 8 < p.D;   // (kD = 96: the layout's chunks 12..15 are not part of the head)
  // K/V tiles are fetched with raw buffer loads (fixed per-thread byte offsets, the tile advance lives in
  // the scalar descriptor, rows past the end of the sequence read as zero).  With D == 128 they go
  // global -> LDS directly (buffer_load ... lds): a wave-instruction fills 64 consecutive 16-byte slots =
  // 4 tile rows, so lane L of the DMA for row group c = wave + 8 i lands in row 4c + L/16, physical
  // chunk L%16 and must FETCH the logical chunk the swizzle puts there.  D < 128 needs the chunks
  // beyond D zeroed and takes the register path.
  constexpr bool kDma = kFullD;
  vec8<T> kreg[kShare], vreg[kShare];
  int voff_k[kShare], voff_v[kShare];
#pragma unroll
  for (int i = 0; i < kShare; ++i) {
    int row = sr + kRowsPerPass * i, chunk = sc;
    if (kDma) dma_lane_src<kD>(wave + kDqWaves * i, lane, row, chunk);
    voff_k[i] = (row * (int)p.k_st.row + chunk * 8) * 2;
    voff_v[i] = (row * (int)p.v_st.row + chunk * 8) * 2;
  }
  auto load_tile = [&](int j, auto stage) {
    constexpr int kStage = decltype(stage)::value;
    int rows = lk - j * kDqKV;
    rows = rows < kDqKV ? rows : kDqKV;
    const int nk = rows > 0 ? ((rows - 1) * (int)p.k_st.row + p.D) * 2 : 0;
    const int nv = rows > 0 ? ((rows - 1) * (int)p.v_st.row + p.D) * 2 : 0;
    const buf_rsrc_t rk = make_rsrc(kbase + (int64_t)j * kDqKV * p.k_st.row, nk);
    const buf_rsrc_t rv = make_rsrc(vbase + (int64_t)j * kDqKV * p.v_st.row, nv);
    const dma_rsrc_t dk = make_dma_rsrc(kbase + (int64_t)j * kDqKV * p.k_st.row, nk);
    const dma_rsrc_t dv = make_dma_rsrc(vbase + (int64_t)j * kDqKV * p.v_st.row, nv);
#pragma unroll
    for (int i = 0; i < kShare; ++i) {
      if (kDma) {
        const int dst = lds_addr(smem) + kStage * kDqTileBytes + (wave + kDqWaves * i) * 1024;
        dma_load128(dk, dst, voff_k[i]);
        dma_load128(dv, dst + 2 * kDqTileBytes, voff_v[i]);
      } else {
        kreg[i] = buffer_load128<T>(rk, voff_k[i]);
        vreg[i] = buffer_load128<T>(rv, voff_v[i]);
        if (!sd_ok) {
          kreg[i] = zero8<T>();
          vreg[i] = zero8<T>();
        }
      }
    }
  };
  auto write_tile = [&](auto stage) {
    constexpr int kStage = decltype(stage)::value;
    if (!kDma) {
#pragma unroll
      for (int i = 0; i < kShare; ++i) {
        const int o = tile_off_d<kD>(sr + kRowsPerPass * i, sc);
        lds_write128<T>(smem + kStage * kDqTileBytes + o, kreg[i]);
        lds_write128<T>(smem + (2 + kStage) * kDqTileBytes + o, vreg[i]);
      }
    }
  };

  int koff[kNK];
#pragma unroll
  for (int kk = 0; kk < kNK; ++kk) {
    koff[kk] = lds_addr(smem) + tile_off_d<kD>(l31, 2 * kk + g);
    pin_vgpr(koff[kk]);
  }
  // K^T transpose reads at rows 32 t + 16 ks + 8 hh + 4 g: row bits below the swizzle period are in the address
  constexpr int kTK = Geo::kSwzRows / 16;
  int toff[kNB][kTK][2];
#pragma unroll
  for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
    for (int kv = 0; kv < kTK; ++kv)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        toff[dblk][kv][hh] = lds_addr(smem) + tr_off_d<kD>(lane, dblk, 16 * kv + 8 * hh + 4 * g);
        pin_vgpr(toff[dblk][kv][hh]);
      }

  const uint32_t drop_key = kDrop ? drop_head_key(p.drop_seed, p.cu_q ? 0u : (uint32_t)b, p.head0 + (uint32_t)h) : 0u;
  const uint32_t drop_i = kDrop ? p.q_pos0 + (uint32_t)(p.cu_q ? qs.row0 : 0) + (uint32_t)qrow : 0u;
  const uint32_t drop_j0 = kDrop ? p.k_pos0 + (uint32_t)(p.cu_k ? ks.row0 : 0) : 0u;
  const float c = p.scale * kLog2e;
  f32x16 dq[kNB];
#pragma unroll
  for (int i = 0; i < kNB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

  typedef std::integral_constant<int, 0> stage0_t;
  typedef std::integral_constant<int, 1> stage1_t;
  load_tile(jt0, stage0_t{});   // unconditional (rows past the end read as zero): one path into the loop
  write_tile(stage0_t{});
  wait_all_vmem();
  __syncthreads();

  // One KV tile.  The LDS stage is a compile-time constant (the tile loop is unrolled by two), so every
  // LDS address in here is a per-lane table entry plus an instruction immediate: no address arithmetic.
  auto tile_step = [&](int j, auto stage) {
    constexpr int kStage = decltype(stage)::value;
    constexpr int kbo = kStage * kDqTileBytes;            // K stage
    constexpr int vbo = (2 + kStage) * kDqTileBytes;      // V stage
    typedef std::integral_constant<int, kStage ^ 1> next_t;
    if (j + 1 < ntiles) load_tile(j + 1, next_t{});
    const int kt0 = j * kDqKV;
    const bool active = (qw0 < lq) && !(hi && kt0 > qw0 + 31 + off + wr) &&
                        !(lo && kt0 + kDqKV - 1 < qw0 + off - wl);
    if (active) {
      const bool need_mask = (kt0 + kDqKV > lk) || (hi && kt0 + kDqKV - 1 > qw0 + off + wr) ||
                             (lo && kt0 < qw0 + 31 + off - wl);
      const int lim = hi ? ((qrow + off + wr < lk - 1) ? qrow + off + wr : lk - 1) : lk - 1;
      const int lim_lo = lo ? qrow + off - wl : -0x40000000;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        {
          // S^T = K Q^T and dP^T = V dO^T as one 2 kNK-step pipeline, A fragments read kAhead ahead
          constexpr int kAhead = RFA_DQ_AHEAD1;
          constexpr int kN = 2 * kNK;
          vec8<T> a[kN];
          auto fa = [&](int i) {
            return lds_read128<T>(lds_ptr(koff[i % kNK]) + (i < kNK ? kbo : vbo) + t * 32 * kRowBytes);
          };
#pragma unroll
          for (int i = 0; i < kAhead; ++i) a[i] = fa(i);
#pragma unroll
          for (int i = 0; i < kN; ++i) {
            if (i + kAhead < kN) a[i + kAhead] = fa(i + kAhead);
            if (i < kNK) s = mfma(a[i], qf[i], s);
            else dp = mfma(a[i], dof[i - kNK], dp);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
          for (int i = 0; i < kN - kAhead; ++i) {
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
            s[r] = (key > lim || (kWin && key < lim_lo)) ? 0.f : s[r];
          }
        }
        if (kDrop) {
          // dP = mask ∘ (dO V^T) / (1 - p): the gradient of the DROPPED probabilities; dS = P ∘ (dP - delta) keeps the
          // undropped P (a dropped element still contributes -P delta)
          const int mis = __builtin_amdgcn_readfirstlane((int)(drop_j0 & 3u));
#pragma unroll
          for (int mm = 0; mm < 4; ++mm) {
            const uint32_t jg = drop_j0 + (uint32_t)(kt0 + 32 * t + 8 * mm + 4 * g);
            uint32_t w = drop_word(drop_key, drop_i, jg >> 2);
            if (mis) w = __builtin_amdgcn_alignbyte(drop_word(drop_key, drop_i, (jg >> 2) + 1), w, (uint32_t)mis);
#pragma unroll
            for (int e = 0; e < 4; ++e) dp[4 * mm + e] = drop_keep(w, e, p.drop_keep) ? dp[4 * mm + e] * p.drop_scale : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = s[r] * (dp[r] - dlt);
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          const vec8<T> dsb = pack8<T>(s, 8 * ks2);
          constexpr int kPer = Geo::kSwzRows;
          const int rows = 32 * t + 16 * ks2;
          const int imm = kbo + (rows / kPer) * kPer * kRowBytes;
          const int kv = (rows % kPer) / 16;
#pragma unroll
          for (int dblk = 0; dblk < kNB; ++dblk) {
            vec4<T> lo = lds_read_tr<T>(lds_ptr(toff[dblk][kv][0]) + imm);
            vec4<T> hi = lds_read_tr<T>(lds_ptr(toff[dblk][kv][1]) + imm);
            dq[dblk] = mfma(concat<T>(lo, hi), dsb, dq[dblk]);
          }
        }
      }
    }
    if (j + 1 < ntiles) write_tile(next_t{});
    if (kDma) wait_all_vmem();                           // the DMA of tile j+1 must have landed before the barrier
    __syncthreads();
  };
  for (int j = jt0; j < ntiles; j += 2) {
    tile_step(j, stage0_t{});
    if (j + 1 < ntiles) tile_step(j + 1, stage1_t{});
  }

  if (qrow >= lq) return;
  const int64_t orow = qs.row0 + qrow;
  if (p.dq_acc == nullptr) {
    T* ob = (T*)p.dq + qbatch * p.dq_st.batch + orow * p.dq_st.row + (int64_t)h * p.dq_st.head;
    store_rows16<T, kFullD, kNB>(ob, dq, p.scale, g, p.D, true);
  } else {
    float* ab = p.dq_acc + qbatch * p.dq_acc_st.batch + orow * p.dq_acc_st.row +
                (int64_t)h * p.dq_acc_st.head;
#pragma unroll
    for (int dblk = 0; dblk < kNB; ++dblk)
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
constexpr int kKvStatBytes = 2 * kKvQ * 4;         // lse[64] + delta[64] per stage
template <int kD> constexpr int kv_smem() {        // 129 KiB (65 KiB at kD = 64)
  return 2 * kKvKeys * HeadGeo<kD>::kRowBytes + 4 * kKvQ * HeadGeo<kD>::kRowBytes + 2 * kKvStatBytes;
}

// kD: compiled head dim (128 / 64); kFullD: D == kD (LDS-DMA staging) else zero padded (register staging);
// kSpill: store dS for rfa_dqs.hip (kD = 128 only)
// kWide (head dim 128, and — round 5 — 64): the workgroup owns 256 keys = 8 key blocks, wave w runs BOTH sub-tiles of every Q/dO tile for key
// block w.  Same registers per wave as the parity form (one sub-tile is live at a time), same LDS (the V rows
// take the space the final parity exchange used), but every staged Q/dO tile now feeds twice the MFMAs: the
// LDS-DMA pieces per MFMA are halved (measured: dropping half of the pieces is worth 15 % of the kernel),
// one barrier per 128 MFMAs per SIMD instead of 64, no exchange at the end.  256-key workgroups are too few for
// a causal launch at Hk = 8, so the tile range of a key block can be split over p.nsplit workgroups whose
// partials (fp32, workspace) are summed by reduce_kernel (rfa_api.cpp).
// kDrop: dropout — dV takes the dropped, rescaled probabilities, dS the masked dP (128-key form without spill / window)
// kBal (round 6, kWide only): the BALANCED causal schedule — see "balanced schedule" below the work decode
template <typename T, int kD, bool kFullD, bool kSpill, bool kWin, bool kWide, bool kDrop = false, bool kBal = false>
__global__ __launch_bounds__(kKvThreads, 2) void dkdv_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  static_assert(!kSpill || (kD == 128 && !kWin), "the dS spill path: head dim 128, no window");
  static_assert(!kWide || ((kD == 128 || kD == 64) && kFullD && !kWin), "the 256-key form: head dim 128 / 64 exactly, no window");
  static_assert(!kDrop || (!kSpill && !kWin && !kWide), "dropout: the plain 128-key instances");
  static_assert(!kBal || kWide, "the balanced schedule: the 256-key form");
  constexpr int kKeys = kWide ? 2 * kKvKeys : kKvKeys;       // keys per workgroup
  typedef HeadGeo<kD> Geo;
  constexpr int kRowBytes = Geo::kRowBytes;                  // (shadows the 128-wide namespace constant)
  constexpr int kNK = Geo::kKSteps, kNB = Geo::kDBlocks;
  constexpr int kKvTileBytes = kKvQ * kRowBytes;             // 16 KiB (8 KiB at kD = 64)
  constexpr int kKvKvBytes = kKvKeys * kRowBytes;            // 32 KiB (16 KiB) per 128 keys
  constexpr int kChunks = Geo::kLay / 8;
  constexpr int kRowsPerPass = kKvThreads / kChunks;         // register staging: one chunk per thread and pass
  constexpr int kPasses = kKvQ / kRowsPerPass;               // passes (= 1 KiB DMA pieces per wave) per Q / dO tile
  constexpr int kTK = Geo::kSwzRows / 16;                    // k-steps of a transpose read that need their own address
  // LDS map (bytes): Q[2] 0 / 16K, dO[2] 32K / 48K, V 64K, stats[2] 128K.  Every address used in
  // the main loop is ONE per-lane VGPR (toggled between the two stages with an XOR once per tile) plus a
  // compile-time immediate — no vector address arithmetic beside the swizzle XORs.
  constexpr int kOffDo = 2 * kKvTileBytes;            // dO = Q + 32K  (immediate)
  constexpr int kOffV = 4 * kKvTileBytes;             // 64K
  constexpr int kOffStat = kOffV + 2 * kKvKvBytes;    // 128K (96K..128K: the final parity exchange / V rows 128..255 of kWide)
  lds_t* vtile = smem + kOffV;                        // [128 keys][128] swizzled

  if (lds_addr(smem) & 0xffff) __builtin_trap();     // address XOR tricks below need a 64 KiB-aligned block
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kbw = kWide ? wave : (wave & 3);
  const int par = kWide ? 0 : (wave >> 2);
  const int g = lane >> 5;
  const int l31 = lane & 31;

  int idx = blockIdx.x;
  const int G = p.H / p.Hk;
  const int hk = idx % p.Hk;
  idx /= p.Hk;
  const int nsplit = (kWide && !kBal) ? p.nsplit : 1;
  const int qsplit = idx % nsplit;                    // which part of the key block's tile range
  idx /= nsplit;
  int wblk, b;                                        // the key block (kBal: the workgroup's role, below) and the batch
  if (kBal) {
    wblk = idx % p.nkblk;
    b = idx / p.nkblk;
  } else {
    split_block_batch<RFA_BATCH_FAST_KV>(idx, p.nkblk, p.B, wblk, b);
  }
  const int h0 = hk * G;                              // the G query heads h0 .. h0+G-1 share this K/V head
  // ---- balanced schedule (kBal; rfa_api.cpp: dense causal self-attention blocks, lq == lk a multiple of 512 rows) ----
  // A causal launch's key blocks differ 1 : nkb in length.  Sharing every block's tile range between ns workgroups
  // balances the launch but writes ns fp32 partials per key and needs reduce_kernel (134 MB written + read per launch
  // whatever the sequence length: at B 8 x S 1024 that costs more than the imbalance).  Here the nkb workgroups of a
  // (batch, K/V head) each take EXACTLY T/2 + 2 of the T = 4 nkb Q/dO tiles' worth of work, and only the lower half of
  // the key blocks is shared — by exactly two workgroups, which add their partials between themselves:
  //   A_j (j < nkb/2): key block j — its top 4 + 4j tiles walking DOWN from the last tile, then its first tiles walking
  //                    UP from its diagonal (tiles 4j .. T/2 - 3)
  //   B_j            : first ALL of key block nkb-1-j (its 4 + 4j tiles, down from the last tile; stored directly), then
  //                    the middle of key block j (tiles T-5-4j down to T/2-2)
  // At any moment the workgroups of a K/V head (one XCD's L2) are on at most two different tiles: the one the downward
  // walkers share and the one the upward walkers share.  Key block j's two partial sums meet in its LAST arriver: the first
  // to finish publishes its accumulators (fp32, write-through) and a flag, the other adds them to its own registers
  // (a + b == b + a: the result does not depend on who came first) and stores the block — no reduce pass.
  const int bal_half = p.nkblk >> 1;
  const bool bal_b = kBal && wblk < bal_half;
  const int bal_j = kBal ? (bal_b ? wblk : wblk - bal_half) : 0;
  const int nseg = bal_b ? 2 : 1;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int off = lk - lq;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;
  // one pass per key block of this workgroup (two for a type-B workgroup of the balanced schedule, else one)
#pragma unroll 1
  for (int seg = 0; seg < nseg; ++seg) {
  // (kBal: every lane-derived value below is formed again per pass from an opaque copy of the thread index — hoisted
  //  out of this loop by the compiler they would stay live across the epilogue and push its 128 accumulators to scratch)
  int tid_pass = threadIdx.x;
  if (kBal) asm volatile("" : "+v"(tid_pass));
  const int tid = tid_pass;
  const int lane = tid & 63;
  const int g = lane >> 5;
  const int l31 = lane & 31;
  const int kblk = kBal ? ((bal_b && seg == 0) ? p.nkblk - 1 - bal_j : bal_j) : wblk;
  const int kwg0 = kblk * kKeys;
  if (kwg0 >= lk) return;
  const int kw0 = kwg0 + kbw * 32;
  const int krow = kw0 + l31;

  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + ks.row0 * p.v_st.row + (int64_t)hk * p.v_st.head;
  const T* qbase0 = (const T*)p.q + qbatch * p.q_st.batch + qs.row0 * p.q_st.row + (int64_t)h0 * p.q_st.head;
  const T* dobase0 = (const T*)p.dout + qbatch * p.dout_st.batch + qs.row0 * p.dout_st.row +
                     (int64_t)h0 * p.dout_st.head;
  const float* lsebase0 = p.lse + qbatch * p.lse_batch + (int64_t)h0 * p.lse_head + qs.row0;
  const float* dltbase0 = p.delta + qbatch * p.delta_batch + (int64_t)h0 * p.delta_head + qs.row0;

  // queries that see key j: [j - off - wr, j - off + wl], each bound only if set (kWin: see dq_kernel)
  const bool hi = kWin ? p.wr >= 0 : p.causal != 0;
  const bool lo = kWin && p.wl >= 0;
  const int wr = kWin ? p.wr : 0, wl = kWin ? p.wl : 0;
  int qfirst = 0;
  if (hi) {
    qfirst = kwg0 - off - wr;
    if (qfirst < 0) qfirst = 0;
  }
  int qlast = lq;                              // exclusive
  if (lo && kwg0 + kKeys - off + wl < qlast) qlast = kwg0 + kKeys - off + wl;
  const int jt0 = qfirst / kKvQ;
  int jt1 = (qlast + kKvQ - 1) / kKvQ;         // exclusive
  if (jt1 <= jt0) jt1 = jt0;                   // nothing visible: no tiles (the unconditional prologue fetch below
                                               // then reads tile jt0 - 1 >= -1: clamped to 0 there)
  // This workgroup's share of the tiles jt0 .. jt1-1: every nsplit-th one, counted from the top (jtop, jtop -
  // nsplit, ...; may be none: it then stores zeros).  Interleaved rather than contiguous ranges so that all
  // workgroups of a launch — whatever their key block and split — walk down the SAME tiles at about the same
  // time and the L2 serves a Q/dO tile to all of them (contiguous halves: 2.4x the HBM fetch, measured).
  int jtop = jt1 - 1 - qsplit;
  int ntile_q = jtop >= jt0 ? (jtop - jt0) / nsplit + 1 : 0;
  // kBal: w_n1 tiles from jtop downward, then (type A) upward from w_j2
  int w_n1 = 0x7fffffff, w_j2 = 0;
  if (kBal) {
    const int nt = p.nkblk * (kKeys / kKvQ);           // T: the 64-row Q/dO tiles of the sequence
    const int n_top = 4 + 4 * bal_j, n_low = nt / 2 - 2 - 4 * bal_j;
    if (!bal_b) {
      jtop = nt - 1; w_n1 = n_top; w_j2 = 4 * bal_j; ntile_q = n_top + n_low;
    } else if (seg == 0) {
      jtop = nt - 1; ntile_q = n_top;
    } else {
      jtop = nt - 5 - 4 * bal_j; ntile_q = n_low;
    }
  }

  const int sc = tid % kChunks;
  const int sr = tid / kChunks;               // 0 .. kRowsPerPass-1
  const bool sd_ok = (kFullD && kD == Geo::kLay) || sc * 8 < p.D;   // (kD = 96: the layout's chunks 12..15 are not part of the head)

  // ---- V rows of the workgroup go to LDS once (128 rows x kChunks chunks); this wave's
  // K rows stay in registers as the B operand of the S GEMM (4 kNK registers)
#pragma unroll
  for (int i = 0; i < kKeys / kRowsPerPass; ++i) {
    const int row = sr + kRowsPerPass * i;
    int kr = kwg0 + row;
    kr = kr < lk ? kr : lk - 1;
    vec8<T> vc = zero8<T>();
    if (sd_ok) vc = *(const vec8<T>*)(vbase + (int64_t)kr * p.v_st.row + sc * 8);
    lds_write128<T>(vtile + tile_off_d<kD>(row, sc), vc);
  }
  vec8<T> kwr[kNK];
  {
    const int kr = krow < lk ? krow : lk - 1;
    const T* kp = kbase + (int64_t)kr * p.k_st.row;
#pragma unroll
    for (int kk = 0; kk < kNK; ++kk) {
      const int chunk = 2 * kk + g;
      kwr[kk] = (kFullD || chunk * 8 < p.D) ? *(const vec8<T>*)(kp + chunk * 8) : zero8<T>();
    }
  }

  vec8<T> qreg[kPasses], doreg[kPasses];
  float statreg = 0.f;
  // Row statistics are stored pre-multiplied (one multiply per staged value, done at LDS-write time when
  // the load has long landed): lse by -log2(e), so that P = exp2(S*c + stat) is a single FMA per element,
  // and delta by -1, so that it can be loaded straight into the dP accumulator (dP - delta for free).
  const float stat_scale = (tid & kKvQ) ? -1.f : -kLog2e;
  // Q/dO tiles are fetched with raw buffer loads: the per-thread byte offsets are fixed for the whole
  // kernel and the tile advance lives in the (scalar) buffer descriptor, so a tile costs no vector
  // address arithmetic; rows past the end of the sequence are out of the descriptor's range and read
  // as zero (no clamping).  The descriptor covers exactly the tile's valid rows of this head.
  // With D == 128 the tile goes global -> LDS directly (buffer_load ... lds, no staging registers): a
  // wave-instruction fills 64 consecutive 16-byte slots = 4 tile rows, so lane L of the DMA for row
  // group c = wave + 8 i lands in row 4c + L/16, physical chunk L%16 and must FETCH the logical chunk
  // that the swizzle puts there.  Otherwise (D < 128) the chunks beyond D have to be zeroed, which
  // needs the register path.
#ifndef RFA_KV_DMA
#define RFA_KV_DMA 1
#endif
  constexpr bool kDma = kFullD && RFA_KV_DMA;
  int voff_q[kPasses], voff_do[kPasses];
#pragma unroll
  for (int i = 0; i < kPasses; ++i) {
    int row = sr + kRowsPerPass * i, chunk = sc;
    if (kDma) dma_lane_src<kD>(wave + kKvWaves * i, lane, row, chunk);
    voff_q[i] = (row * (int)p.q_st.row + chunk * 8) * 2;
    voff_do[i] = (row * (int)p.dout_st.row + chunk * 8) * 2;
  }
  // The workgroup streams the Q/dO tiles jt0 .. jt1-1 of ALL G query heads of its K/V head through the
  // same two LDS stages: dK/dV of the group are summed in the fp32 accumulators and written once — no
  // per-head partials, no group-reduction pass.
  int dma_stage = 0;                                  // LDS stage the next load_tile() fills (scalar)
  // Walk order: tiles from the LAST one down to jt0, and for every tile the G heads of the group.  Every
  // workgroup of a launch, whatever its causal start, then reads the same (tile, head) at about the same
  // time, so the (kv head = XCD) L2 serves a tile to all co-resident key blocks.  (Walking upward from
  // jt0, head after head, spreads them over the whole sequence and every tile is re-fetched from HBM per
  // workgroup: 3.5x the fabric traffic, measured.)
  int ld_g = 0, ld_j = jtop > 0 ? jtop : 0;           // (head in group, tile) the next load_tile() fetches
  int ld_c = 0;                                       // kBal: tiles the loader has finished
  // Round 6 (second session): the loader's addresses are RUNNING 64-bit scalars — the next head of the tile is one add, a
  // new tile one multiply — instead of four base + head * stride + tile * 64 * stride products per tile-step.  The loop
  // carried 250 scalar instructions per 64 MFMAs (two thirds of them this address arithmetic and the dS block address
  // below), all in front of the tile's first fragment read, in BOTH waves of a SIMD at once behind the barrier.
  const int64_t q_tile_e = (int64_t)kKvQ * p.q_st.row, do_tile_e = (int64_t)kKvQ * p.dout_st.row;
  const float* st_base0 = wave ? dltbase0 : lsebase0;  // (wave 0 stages lse, wave 1 delta)
  const int64_t st_head_e = wave ? p.delta_head : p.lse_head;
  const T* ld_q = qbase0 + ld_j * q_tile_e;
  const T* ld_do = dobase0 + ld_j * do_tile_e;
  const float* ld_st = st_base0 + ld_j * kKvQ;
  auto load_tile = [&]() {
    const int j = RFA_KV_X_LOAD == 2 ? 0 : ld_j;       // (2: measurement, every load hits the same hot tile)
    const T* qtile = RFA_KV_X_LOAD == 2 ? qbase0 : ld_q;
    const T* dotile = RFA_KV_X_LOAD == 2 ? dobase0 : ld_do;
    const float* sttile = ld_st;
    if (++ld_g >= G) {
      ld_g = 0;
      if (kBal) {
        ++ld_c;
        ld_j = ld_c == w_n1 ? w_j2 : (ld_c < w_n1 ? ld_j - 1 : ld_j + 1);
      } else {
        ld_j -= nsplit;
      }
      ld_q = qbase0 + ld_j * q_tile_e;
      ld_do = dobase0 + ld_j * do_tile_e;
      ld_st = st_base0 + ld_j * kKvQ;
    } else {
      ld_q += p.q_st.head;
      ld_do += p.dout_st.head;
      ld_st += st_head_e;
    }
    int rows = lq - j * kKvQ;
    rows = rows < kKvQ ? rows : kKvQ;
    const int nq = rows > 0 ? ((rows - 1) * (int)p.q_st.row + p.D) * 2 : 0;
    const int ndo = rows > 0 ? ((rows - 1) * (int)p.dout_st.row + p.D) * 2 : 0;
    const buf_rsrc_t rq = make_rsrc(qtile, nq);
    const buf_rsrc_t rdo = make_rsrc(dotile, ndo);
    const dma_rsrc_t dq_ = make_dma_rsrc(qtile, nq);
    const dma_rsrc_t ddo = make_dma_rsrc(dotile, ndo);
#pragma unroll
    for (int i = 0; i < kPasses; ++i) {
      if (kDma) {
        const int dst = lds_addr(smem) + dma_stage + (wave + kKvWaves * i) * 1024;
        dma_load128(dq_, dst, voff_q[i]);
        if (RFA_KV_X_LOAD != 3) dma_load128(ddo, dst + kOffDo, voff_do[i]);      // (3: measurement, Q only)
      } else {
        qreg[i] = buffer_load128<T>(rq, voff_q[i]);
        doreg[i] = buffer_load128<T>(rdo, voff_do[i]);
        if (!sd_ok) {                          // chunk beyond D but inside the row: not covered by the range check
          qreg[i] = zero8<T>();
          doreg[i] = zero8<T>();
        }
      }
    }
    if (wave < 2) {                            // wave 0 stages lse, wave 1 delta (wave-uniform descriptor)
      // asynchronous like the tile DMA (counted by the wait in front of write_tile): a compiler-tracked load
      // would be protected with vmcnt(0) at its use and drain the dS spill stores issued after it every tile
      const dma_rsrc_t rs = make_dma_rsrc(sttile, rows > 0 ? rows * 4 : 0);
      statreg = buffer_load32_async(rs, lane * 4);
    }
    dma_stage ^= kKvTileBytes;
  };
  int wq = lds_addr(smem) + tile_off_d<kD>(sr, sc);                    // staging write address, stage 0 (further passes: immediate)
  int ws = lds_addr(smem) + kOffStat + tid * 4;
  auto write_tile = [&]() {
    if (!kDma) {
#pragma unroll
      for (int i = 0; i < kPasses; ++i) {
        lds_write128<T>(lds_ptr(wq) + i * kRowsPerPass * kRowBytes, qreg[i]);
        lds_write128<T>(lds_ptr(wq) + i * kRowsPerPass * kRowBytes + kOffDo, doreg[i]);
      }
    }
    if (tid < 2 * kKvQ) *(__attribute__((address_space(3))) float*)lds_ptr(ws) = statreg * stat_scale;
  };

  // Fragment addresses: the swizzle makes chunk selection an XOR on address bits 4..7, so a fragment's
  // address is (one base register) ^ (kk << 5) resp. ^ (dblk << 6) — far fewer live registers than
  // offset tables, which is what lets this kernel fit the 256-register budget of two waves per SIMD.
  // (XOR toggling / swizzling on absolute addresses: the dynamic LDS block starts 64 KiB-aligned — at 0 —
  //  because these kernels have no static LDS; checked at the top of the kernel)
  int aq = lds_addr(smem) + tile_off_d<kD>(l31, g) + par * 32 * kRowBytes;                 // Q / dO sub-tile rows (stage toggled)
  const int av = lds_addr(smem) + tile_off_d<kD>(l31, g) + kOffV + kbw * 32 * kRowBytes;   // this wave's V rows
  // transpose-read bases (stage toggled), rows 32 par + 16 ks + 8 hh + 4 g: the row bits below the swizzle period
  // (hh, g at kD = 128; ks, hh, g at kD = 64) are part of the address
  int tq[kTK][2];
#pragma unroll
  for (int kv = 0; kv < kTK; ++kv)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      tq[kv][hh] = lds_addr(smem) + par * 32 * kRowBytes + tr_off_d<kD>(lane, 0, 16 * kv + 8 * hh + 4 * g);
      pin_vgpr(tq[kv][hh]);
    }
  int mask_g4 = 4 * g, mask_kg = krow - 4 * g;        // mask of (query sq + 4 g, key krow): see the exp / mask section
  pin_vgpr(mask_g4); pin_vgpr(mask_kg);
  int sa = lds_addr(smem) + kOffStat + (32 * par + 4 * g) * 4;                       // row statistics (stage toggled)
  int avp = av;
  pin_vgpr(aq); pin_vgpr(avp); pin_vgpr(sa); pin_vgpr(wq); pin_vgpr(ws);

  // dS spill (rfa_dqs.hip): the two packed dS operands of every active (32 query x 32 key) block go to the
  // scratch block (b, h, qt = query row / 32, kb = key / 32) in the slot order that kernel reads back:
  // slot 16 (key>>2) + 8 i + 4 g + (key&3), i = 0 / 1 for the rows 0-15 / 16-31 of the sub-tile
  const int ds_lane = 16 * (16 * (l31 >> 2) + 4 * g + (l31 & 3));
  const int ds_nkb = ds_blocks(p.Sk, p.k_half);  // extents of the longest (half) sequence (= lk, lq when dense): rfa_dqs.hip
  // rows of the scratch: rectangular (p.ds_c >= ds_nkb) or packed triangular (dense causal), rfa_kernels.hpp
  const int64_t ds_head_bytes = p.ds_head_blocks * kDsBlockBytes;   // (< 4 GiB: rfa_api.cpp bwd_spill_eligible — the block of a
                                                                    //  store is addressed by a 32-bit scalar offset from its head's base)
  const char* ds_b = kSpill ? (const char*)p.ds + ds_base_blocks(p, b) * kDsBlockBytes + (int64_t)h0 * ds_head_bytes : nullptr;
  const int64_t ds_g0 = (qs.row0 >> 5) + b;            // packed layout: global index of this (half) sequence's first row
  const int ds_kb = __builtin_amdgcn_readfirstlane(kblk * (kKeys / 32) + kbw);

  // dropout: this lane's key position; the mask word of (query i, key j) is word(i, j >> 2), byte j & 3
  const uint32_t drop_j = kDrop ? p.k_pos0 + (uint32_t)(p.cu_k ? ks.row0 : 0) + (uint32_t)krow : 0u;
  const uint32_t drop_i0 = kDrop ? p.q_pos0 + (uint32_t)(p.cu_q ? qs.row0 : 0) : 0u;
  const float c = p.scale * kLog2e;
  f32x16 dk[kNB], dv[kNB];
#pragma unroll
  for (int i = 0; i < kNB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  load_tile();                                         // (jt0 >= jt1: zero rows, nothing is read)
  wait_all_vmem();
  load_landed(statreg);
  write_tile();
  wq ^= kKvTileBytes;
  ws ^= kKvStatBytes;
  __syncthreads();

#if RFA_KV_PRIO == 1
  if (par == 0) __builtin_amdgcn_s_setprio(2);
#elif RFA_KV_PRIO == 2
  if (wave >= kKvWaves / 2) __builtin_amdgcn_s_setprio(1);   // the half dispatched second (loses every VALU arbitration)
#endif
  const int ntile = ntile_q * G;
  int j = jtop, cg = 0, jc = 0;
  // dS scratch rows of tile j's two sub-tiles (blocks from the head's base: formed when j changes, not per tile-step) and
  // the base of the current head's share of the scratch
  int ds_roff0 = kSpill ? (int)ds_rowpart(p, 2 * j, ds_g0, ds_nkb) : 0;
  int ds_rlen0 = kSpill ? ds_rowlen(p, 2 * j, ds_nkb) : 0;
  const char* ds_hp = ds_b;
  for (int f = 0; f < ntile; ++f) {
    if (RFA_KV_X_LOAD && f + 1 < ntile) load_tile();
    int nact = 0;                                      // sub-tiles this wave computed (= pairs of spill stores issued)
    bool active = false;
    // parity form: the one sub-tile t = par; kWide: both, one after the other (the sub-tile lives in address
    // bit 13 of the Q/dO fragment bases and bit 7 of the statistics base: toggled, not re-computed)
#if RFA_KV_WIDE_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int t0 = 0; t0 < (kWide ? 2 : 1); ++t0) {
    const int t = kWide ? t0 : par;
    // unrolled kWide bodies: the sub-tile is an IMMEDIATE on every LDS access (the bases have address bit 13 — rows 32..63
    // of a tile — and bit 7 — the statistics of those rows — clear, so adding equals toggling)
    constexpr bool kImm = kWide && RFA_KV_WIDE_UNROLL;
    const int toff = kImm ? t0 * 32 * kRowBytes : 0, soff_t = kImm ? t0 * 32 * 4 : 0;
    const int qs0 = j * kKvQ + 32 * t;
    active = (kw0 < lk) && (qs0 < lq) && !(hi && qs0 + 31 + off + wr < kw0) &&
             !(lo && qs0 + off - wl > kw0 + 31);
    if (active) {
      ++nact;
      f32x16 s, dp;
      if (RFA_KV_PRIO == 3 || RFA_KV_PRIO == 5) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {               // dp starts at -delta[q]: 4 LDS reads, no VALU
        const f32x4 nd = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + (kKvQ + 8 * jj) * 4 + soff_t);
#pragma unroll
        for (int e = 0; e < 4; ++e) { dp[4 * jj + e] = nd[e]; s[4 * jj + e] = 0.f; }
      }
      {
        // dP - delta = dO V_w^T (+ init; V_w fragments from LDS) then S = Q K_w^T (K_w in registers):
        // 2 kNK MFMAs, LDS operands read kAhead steps ahead
        constexpr int kAhead = RFA_KV_AHEAD;
        constexpr int kN = 2 * kNK;
        vec8<T> a[kN], w[kNK];
        auto fa = [&](int i) {
          if (!RFA_KV_X_LDS || (RFA_KV_X_LDS == 2 && t0 == 1)) return kwr[i % kNK];
          return lds_read128<T>(lds_ptr(aq ^ ((i % kNK) << 5)) + (i < kNK ? kOffDo : 0) + toff);
        };
        auto fw = [&](int i) {
          if (!RFA_KV_X_LDS || (RFA_KV_X_LDS == 2 && t0 == 1)) return kwr[i];
          return lds_read128<T>(lds_ptr(avp ^ (i << 5)));
        };
#pragma unroll
        for (int i = 0; i < kAhead; ++i) { a[i] = fa(i); w[i] = fw(i); }
#pragma unroll
        for (int i = 0; i < kN; ++i) {
          if (i + kAhead < kN) a[i + kAhead] = fa(i + kAhead);
          if (i + kAhead < kNK) w[i + kAhead] = fw(i + kAhead);
          if (i < kNK) dp = mfma(a[i], w[i], dp);
          else s = mfma(a[i], kwr[i - kNK], s);
        }
#if RFA_KV_PIN
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * kAhead, 0);
#pragma unroll
        for (int i = 0; i < kNK - kAhead; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < kNK; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
#endif
      }
      if (RFA_KV_PRIO == 3 || RFA_KV_PRIO == 5) __builtin_amdgcn_s_setprio(0);
      // lse is read only now: holding it across GEMM 1 would cost 16 registers
      f32x4 l2v[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) l2v[jj] = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + 8 * jj * 4 + soff_t);
      const bool need_mask = (qs0 + 32 > lq) || (hi && qs0 + off + wr < kw0 + 31) ||
                             (lo && qs0 + 31 + off - wl > kw0);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (RFA_KV_X_VALU) s[4 * jj + e] = fast_exp2(__builtin_fmaf(s[4 * jj + e], c, l2v[jj][e]));
      if (RFA_KV_X_VALU && need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // query row q = qs0 + crow(r, g) = sq + 4 g with sq wave-uniform: every comparison is (a loop-invariant lane
          // value) against (a scalar) — no per-element vector arithmetic, and nothing for the compiler to hoist into
          // 16 registers that stay live across the whole tile loop (255 -> 240 registers in the headline instance)
          const int sq = qs0 + crow(r, 0);
          const bool ok = (mask_g4 < lq - sq) & (!hi | (mask_kg <= sq + off + wr)) & (!lo | (mask_kg >= sq + off - wl));
          s[r] = ok ? s[r] : 0.f;
        }
      }
      if (kDrop) {
        // dp holds dO V^T - delta (it was initialised with -delta).  With dropout
        //     dP = keep ? (dO V^T) / (1 - p) : 0,   dS = P (dP - delta),   dV takes keep ? P / (1 - p) : 0
        const uint32_t hkey = drop_head_key(p.drop_seed, p.cu_q ? 0u : (uint32_t)b, p.head0 + (uint32_t)(h0 + cg));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x4 nd = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + (kKvQ + 8 * jj) * 4);   // -delta
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * jj + e;
            const uint32_t w = drop_word(hkey, drop_i0 + (uint32_t)(qs0 + crow(r, g)), drop_j >> 2);
            const bool keep = drop_keep(w, (int)(drop_j & 3u), p.drop_keep);
            const float dpd = keep ? (dp[r] - nd[e]) * p.drop_scale + nd[e] : nd[e];
            dp[r] = dpd * s[r];
            s[r] = keep ? s[r] * p.drop_scale : 0.f;
          }
        }
      } else {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (RFA_KV_X_VALU) dp[4 * jj + e] *= s[4 * jj + e];
      }
      {
        const vec8<T> pb0 = pack8<T>(s, 0), pb1 = pack8<T>(s, 8);
        const vec8<T> ds0 = pack8<T>(dp, 0), ds1 = pack8<T>(dp, 8);
#ifndef RFA_SPILL_PROBE
#define RFA_SPILL_PROBE 0     // timing probes (results invalid unless 0): 1 no stores, 2 stores after the dV/dK GEMMs,
#endif                        // 3 one store per sub-tile, 4 every store of a wave hits one L2-resident block
        auto spill = [&]() {
          if (!kSpill || RFA_SPILL_PROBE == 1) return;
          // one descriptor per head (its whole share of the scratch), the block as the store's SCALAR offset
          unsigned boff = (unsigned)(ds_roff0 + (t ? ds_rlen0 : 0) + ds_kb) * (unsigned)kDsBlockBytes;
          if (RFA_SPILL_PROBE == 4) boff = (unsigned)(blockIdx.x * 8 + wave) * (unsigned)kDsBlockBytes;
          const buf_rsrc_t rb = make_rsrc(RFA_SPILL_PROBE == 4 ? (const char*)p.ds : ds_hp, -1);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ds0), rb, ds_lane, (int)boff, RFA_SPILL_AUX);
          if (RFA_SPILL_PROBE != 3)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ds1), rb, ds_lane + 128, (int)boff, RFA_SPILL_AUX);
        };
        if (RFA_SPILL_PROBE != 2) spill();
        if (RFA_KV_PRIO == 4 || RFA_KV_PRIO == 5) __builtin_amdgcn_s_setprio(1);
        constexpr int kAhead = RFA_KV_AHEAD2;
        constexpr int kN2 = 4 * kNB;                   // [ks2][which: 0 = dO^T (dV), 1 = Q^T (dK)][dblk]
        vec8<T> a[kN2];
        auto frag = [&](int i) {
          const int ks2 = i / (2 * kNB), which = (i / kNB) & 1, dblk = i % kNB;
          constexpr int kPer = Geo::kSwzRows;
          const int kv = (16 * ks2 % kPer) / 16;
          const int imm = (which ? 0 : kOffDo) + (16 * ks2 / kPer) * kPer * kRowBytes + toff;
          if (!RFA_KV_X_LDS || (RFA_KV_X_LDS == 2 && t0 == 1)) return kwr[i % kNK];
          vec4<T> lo = lds_read_tr<T>(lds_ptr(tq[kv][0] ^ (dblk << 6)) + imm);
          vec4<T> hi = lds_read_tr<T>(lds_ptr(tq[kv][1] ^ (dblk << 6)) + imm);
          return concat<T>(lo, hi);
        };
#pragma unroll
        for (int i = 0; i < kAhead; ++i) a[i] = frag(i);
#pragma unroll
        for (int i = 0; i < kN2; ++i) {
          if (i + kAhead < kN2) a[i + kAhead] = frag(i + kAhead);
          const int ks2 = i / (2 * kNB), which = (i / kNB) & 1, dblk = i % kNB;
          if (which == 0) dv[dblk] = mfma(a[i], ks2 ? pb1 : pb0, dv[dblk]);
          else dk[dblk] = mfma(a[i], ks2 ? ds1 : ds0, dk[dblk]);
        }
#if RFA_KV_PIN
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * kAhead, 1);
#pragma unroll
        for (int i = 0; i < kN2 - kAhead; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 1);
#endif
        if (RFA_KV_PRIO == 4 || RFA_KV_PRIO == 5) __builtin_amdgcn_s_setprio(0);
        if (RFA_SPILL_PROBE == 2) spill();
      }
    }
    if (kWide && !kImm) {
      aq ^= 32 * kRowBytes;
#pragma unroll
      for (int kv = 0; kv < kTK; ++kv) {
        tq[kv][0] ^= 32 * kRowBytes;
        tq[kv][1] ^= 32 * kRowBytes;
      }
      sa ^= 32 * 4;
    }
    }   // sub-tile loop
    // The DMA (and the row statistics) of tile j+1 must have landed before they are published.  vmcnt counts
    // stores too and retires in issue order: the two dS spill stores of this sub-tile are the youngest
    // operations and stay in flight (draining them costs a store round trip per tile: 0.93 -> 1.2 ms)
    if (!RFA_KV_X_SYNC) {
    } else if (kWide && kSpill && nact == 2) wait_vmem<4>();
    else if (kWide && kSpill && nact == 1) wait_vmem<2>();
    else if (kWide) wait_all_vmem();
    else if (kSpill && active && RFA_SPILL_PROBE != 1 && RFA_SPILL_PROBE != 3) wait_vmem<2>();
    else if (kSpill && active && RFA_SPILL_PROBE == 3) wait_vmem<1>();
    else wait_all_vmem();
    load_landed(statreg);
    if (f + 1 < ntile) write_tile();
    if (++cg >= G) {
      cg = 0;
      if (kBal) {
        ++jc;
        j = jc == w_n1 ? w_j2 : (jc < w_n1 ? j - 1 : j + 1);
      } else {
        j -= nsplit;
      }
      if (kSpill) {
        ds_roff0 = (int)ds_rowpart(p, 2 * j, ds_g0, ds_nkb);
        ds_rlen0 = ds_rowlen(p, 2 * j, ds_nkb);
        ds_hp = ds_b;
      }
    } else if (kSpill) {
      ds_hp += ds_head_bytes;
    }
    aq ^= kKvTileBytes;                                // flip every stage-dependent address
#pragma unroll
    for (int kv = 0; kv < kTK; ++kv) {
      tq[kv][0] ^= kKvTileBytes;
      tq[kv][1] ^= kKvTileBytes;
    }
    sa ^= kKvStatBytes;
    wq ^= kKvTileBytes;
    ws ^= kKvStatBytes;
    if (RFA_KV_X_SYNC) __syncthreads();
  }

  // ---- combine the two parities: parity 1 hands over its dK^T partial, parity 0 its dV^T partial
  // (fp32, [kb][dblk][r][lane] so that the partner lane reads exactly what its twin wrote); the K/V
  // tiles and the Q/dO buffers (128 KiB, contiguous) are dead by now and take the 2 x 64 KiB of partials.
  if (!kWide) {
    float* xbuf = (float*)smem_raw;                    // generic pointer into LDS
    constexpr int kXW = kNB * 1024;                    // floats per key block and tensor (4 kb per tensor: 2 x 64 KiB at kD = 128)
    const int slot = (par == 1 ? 0 : 4 * kXW) + kbw * kXW;
    const f32x16(&mine)[kNB] = (par == 1) ? dk : dv;
#pragma unroll
    for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
      for (int r = 0; r < 16; ++r) xbuf[slot + (dblk * 16 + r) * 64 + lane] = mine[dblk][r];
    __syncthreads();
    const int rslot = (par == 0 ? 0 : 4 * kXW) + kbw * kXW;  // parity 0 finishes dK, parity 1 finishes dV
    f32x16(&fin)[kNB] = (par == 0) ? dk : dv;
#pragma unroll
    for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
      for (int r = 0; r < 16; ++r) fin[dblk][r] += xbuf[rslot + (dblk * 16 + r) * 64 + lane];
  }

  // ---- kBal: key block j's two partial sums meet in whichever of A_j / B_j arrives last
  bool do_final = true, add_partner = false;
  const __attribute__((address_space(1))) char* pair_src = nullptr;     // this lane's 16 bytes of the partner's slot
  if (kBal && !(bal_b && seg == 0)) {
    typedef __attribute__((address_space(1))) unsigned gu32;
    const int64_t pid = ((int64_t)b * p.Hk + hk) * bal_half + bal_j;
    gu32* flag = (gu32*)(p.pair_flags + pid);
    __attribute__((address_space(3))) unsigned* role_w = (__attribute__((address_space(3))) unsigned*)(smem + kOffStat);
    if (tid == 0) {
      unsigned seen = 0u;                                // 0 -> 1: this workgroup is first and will publish
      __hip_atomic_compare_exchange_strong(flag, &seen, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *role_w = seen;
    }
    __syncthreads();
    const unsigned role = __builtin_amdgcn_readfirstlane(*role_w);
    // this wave's slice of the pair's slot: [tensor][dblk][jj][lane] x 16 bytes — the partner's wave of the same key
    // block index reads lane for lane what its twin wrote (whole 1 KiB wave-instructions)
    constexpr int kSlotWave = 2 * kNB * 4 * 1024;
    char* slot = (char*)p.pair_ws + (pid * kKvWaves + wave) * (int64_t)kSlotWave;
    pair_src = (const __attribute__((address_space(1))) char*)slot + lane * 16;
    if (role == 0u) {
      // publish (the guide's R1 recipe): payload write-through (sc1) so that no release fence is needed, every storing
      // wave drains its stores, ONE lane then stores the flag at agent scope
      const buf_rsrc_t rs = make_rsrc(slot, kSlotWave);
#pragma unroll
      for (int which = 0; which < 2; ++which)
#pragma unroll
        for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = which ? dv[dblk][4 * jj + e] : dk[dblk][4 * jj + e];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), rs,
                                                   ((which * kNB + dblk) * 4 + jj) * 1024 + lane * 16, 0, 16);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      do_final = false;
    } else {
      if (wave == 0) {                                   // ONE wave polls ONE word, relaxed; ONE acquire after the match
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 2u) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1u << 24)) __builtin_trap();    // (the publisher is resident and inside its epilogue: microseconds)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      add_partner = true;                                // (added where the block is stored, below: the accumulators stay read-only)
    }
  }

  if (!kBal && krow >= lk) return;
  const int64_t orow = ks.row0 + krow;
  // parity form: parity 0 stores dK, parity 1 dV; kWide: this wave stores both (which = 0: dK, 1: dV)
  const int64_t soff = (int64_t)qsplit * p.kv_split_stride;      // this split's partial (elements; 0 without a split)
  if (do_final && krow < lk)
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    if (!kWide && which != par) continue;
    const f32x16(&fin)[kNB] = which ? dv : dk;
    const float sc_ = which ? 1.f : p.scale;
    const Strides st = which ? p.dv_st : p.dk_st;
    const int64_t eoff = kbatch * st.batch + orow * st.row + (int64_t)hk * st.head + soff;
    // kBal, last arriver of a shared key block: the partner's fp32 partial of this tensor (one tensor's 16 loads in flight:
    // 64 plain registers; the MFMA accumulator tuples are only read — adding into them under the role branch made hipcc
    // shuffle whole tuples through scratch at the merge)
    f32x4 py[kBal ? kNB : 1][4];
    if (kBal) {
#pragma unroll
      for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int e = 0; e < 4; ++e) py[dblk][jj][e] = 0.f;
      if (add_partner) {
#pragma unroll
        for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            py[dblk][jj] = *(const __attribute__((address_space(1))) f32x4*)(pair_src + ((which * kNB + dblk) * 4 + jj) * 1024);
      }
    }
    if (p.kv_f32) {
      // fp32 store straight into the caller's accumulator slot (overwrite): lane (key, g) holds, per
      // (dblk, jj), the 4 consecutive columns 32 dblk + 8 jj + 4 g .. +3
      float* ob = (float*)(which ? p.dv : p.dk) + eoff;
#pragma unroll
      for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int d0 = 32 * dblk + 8 * jj + 4 * g;
          if (kFullD || d0 < p.D) {
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = (kBal ? fin[dblk][4 * jj + e] + py[dblk][jj][e] : fin[dblk][4 * jj + e]) * sc_;
            if (p.kv_accum) {                      // (a later query-head fraction of a chunked launch sequence)
              const f32x4 old = *(f32x4*)(ob + d0);
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] += old[e];
            }
            *(f32x4*)(ob + d0) = x;
          }
        }
    } else if (kBal) {
      // store_rows16 with the partner's values added (same half-wave exchange, same 16-byte stores)
      T* row_ptr = (T*)(which ? p.dv : p.dk) + eoff;
#pragma unroll
      for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          f32x4 x0, x1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x0[e] = (fin[dblk][8 * m + e] + py[dblk][2 * m][e]) * sc_;
            x1[e] = (fin[dblk][8 * m + 4 + e] + py[dblk][2 * m + 1][e]) * sc_;
          }
          const vec4<T> h0 = __builtin_convertvector(x0, vec4<T>);
          const vec4<T> h1 = __builtin_convertvector(x1, vec4<T>);
          i32x2 a_ = __builtin_bit_cast(i32x2, h0), b_ = __builtin_bit_cast(i32x2, h1);
          auto r0 = __builtin_amdgcn_permlane32_swap(a_[0], b_[0], false, false);
          auto r1 = __builtin_amdgcn_permlane32_swap(a_[1], b_[1], false, false);
          i32x4 w;
          w[0] = r0[0]; w[1] = r1[0]; w[2] = r0[1]; w[3] = r1[1];
          *(i32x4*)(row_ptr + 32 * dblk + 16 * m + 8 * g) = w;
        }
    } else {
      store_rows16<T, kFullD, kNB>((T*)(which ? p.dv : p.dk) + eoff, fin, sc_, g, p.D, true);
    }
  }
  }   // segment loop
}

template <typename T, int kD, bool kFullD, bool kWin, bool kDrop = false>
static int launch_dq_t(const BwdParams& p, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  if (int rc = opt_in_dynamic_lds((const void*)dq_kernel<T, kD, kFullD, kWin, kDrop>, dq_smem<kD>(), attr_done)) return rc;
  const int64_t nblocks = (int64_t)p.nqblk * p.H * p.B;
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL((dq_kernel<T, kD, kFullD, kWin, kDrop>), dim3((unsigned)nblocks), dim3(kDqThreads), dq_smem<kD>(), stream, p);
  return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchFailed;
}

__global__ void zero_words_kernel(unsigned* w, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) w[i] = 0u;
}

template <typename T, int kD, bool kFullD, bool kSpill, bool kWin, bool kWide = false, bool kDrop = false, bool kBal = false>
static int launch_dkdv_t(const BwdParams& p, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  if (int rc = opt_in_dynamic_lds((const void*)dkdv_kernel<T, kD, kFullD, kSpill, kWin, kWide, kDrop, kBal>, kv_smem<kD>(), attr_done)) return rc;
  // one workgroup per (key block, K/V head) [x tile-range split of the 256-key form]
  const int64_t nblocks = (int64_t)p.nkblk * p.Hk * p.B * ((kWide && !kBal) ? p.nsplit : 1);
  if (nblocks <= 0) return 0;
  if (kBal) {
    // the pair flags are polled words: zeroed before EVERY launch, by a kernel of our own in front of the dK/dV kernel —
    // NOT hipMemsetAsync: captured into a HIP graph (torch.cuda.graph) its memset node left the flags as they were on replay
    // and every last arriver spun into the trap (tests/test_gpu_kernels.py: test_step_under_hip_graph_capture[dense_balanced])
    const int nflags = p.B * p.Hk * (p.nkblk / 2);
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((nflags + 255) / 256)), dim3(256), 0, stream, p.pair_flags, nflags);
    if (hipGetLastError() != hipSuccess) return kLaunchFailed;
  }
  hipLaunchKernelGGL((dkdv_kernel<T, kD, kFullD, kSpill, kWin, kWide, kDrop, kBal>), dim3((unsigned)nblocks), dim3(kKvThreads), kv_smem<kD>(), stream, p);
  return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchFailed;
}

template <typename T, bool kWin, bool kDrop>
static int launch_dq_d(const BwdParams& p, hipStream_t stream) {
  if (p.D == 128) return launch_dq_t<T, 128, true, kWin, kDrop>(p, stream);
  if constexpr (!kWin && !kDrop) {          // 64 < D <= 96: the 128-wide layout with three quarters of the MFMA work
    if (p.D == 96) return launch_dq_t<T, 96, true, kWin, kDrop>(p, stream);
    if (p.D > 64 && p.D < 96) return launch_dq_t<T, 96, false, kWin, kDrop>(p, stream);
  }
  if (p.D > 64) return launch_dq_t<T, 128, false, kWin, kDrop>(p, stream);
  if (p.D == 64) return launch_dq_t<T, 64, true, kWin, kDrop>(p, stream);
  return launch_dq_t<T, 64, false, kWin, kDrop>(p, stream);
}
int launch_bwd_dq(const BwdParams& p, int dtype, hipStream_t stream) {
  if (p.D > 128) return launch_bwd_dq_big(p, dtype, stream);            // rfa_bigd.hip
  if (p.drop_keep < 256) return dtype == 0 ? launch_dq_d<bf16_t, false, true>(p, stream) : launch_dq_d<f16_t, false, true>(p, stream);
  if (windowed(p.causal, p.wl, p.wr)) return dtype == 0 ? launch_dq_d<bf16_t, true, false>(p, stream) : launch_dq_d<f16_t, true, false>(p, stream);
  return dtype == 0 ? launch_dq_d<bf16_t, false, false>(p, stream) : launch_dq_d<f16_t, false, false>(p, stream);
}
template <typename T, bool kWin, bool kDrop>
static int launch_dkdv_d(const BwdParams& p, hipStream_t stream) {
  if (p.D == 128) return launch_dkdv_t<T, 128, true, false, kWin, false, kDrop>(p, stream);
  if constexpr (!kWin && !kDrop) {          // 64 < D <= 96: the 128-wide layout with three quarters of the MFMA work
    if (p.D == 96) return launch_dkdv_t<T, 96, true, false, kWin, false, kDrop>(p, stream);
    if (p.D > 64 && p.D < 96) return launch_dkdv_t<T, 96, false, false, kWin, false, kDrop>(p, stream);
  }
  if (p.D > 64) return launch_dkdv_t<T, 128, false, false, kWin, false, kDrop>(p, stream);
  if (p.D == 64) return launch_dkdv_t<T, 64, true, false, kWin, false, kDrop>(p, stream);
  return launch_dkdv_t<T, 64, false, false, kWin, false, kDrop>(p, stream);
}
int launch_bwd_dkdv(const BwdParams& p, int dtype, hipStream_t stream) {
  if (p.D > 128) return launch_bwd_dkdv_big(p, dtype, stream);          // rfa_bigd.hip (rfa_api.cpp: no spill, no 256-key form)
  const bool win = windowed(p.causal, p.wl, p.wr);
  if (p.drop_keep < 256)                          // rfa_api.cpp: dropout calls run the 128-key form without spill / window
    return dtype == 0 ? launch_dkdv_d<bf16_t, false, true>(p, stream) : launch_dkdv_d<f16_t, false, true>(p, stream);
  if (p.wide && p.D == 64 && p.bal)               // (round 6: its balanced causal schedule)
    return dtype == 0 ? launch_dkdv_t<bf16_t, 64, true, false, false, true, false, true>(p, stream)
                      : launch_dkdv_t<f16_t, 64, true, false, false, true, false, true>(p, stream);
  if (p.wide && p.D == 64)                        // round 5: the 256-key form for head dim 64 (7-GEMM backward: no dS hand-off there)
    return dtype == 0 ? launch_dkdv_t<bf16_t, 64, true, false, false, true>(p, stream)
                      : launch_dkdv_t<f16_t, 64, true, false, false, true>(p, stream);
  if (p.wide && p.bal) {                          // round 6: the balanced causal schedule (rfa_api.cpp: head dims 128 and 64)
    if (p.ds != nullptr)
      return dtype == 0 ? launch_dkdv_t<bf16_t, 128, true, true, false, true, false, true>(p, stream)
                        : launch_dkdv_t<f16_t, 128, true, true, false, true, false, true>(p, stream);
    return dtype == 0 ? launch_dkdv_t<bf16_t, 128, true, false, false, true, false, true>(p, stream)
                      : launch_dkdv_t<f16_t, 128, true, false, false, true, false, true>(p, stream);
  }
  if (p.wide) {                                   // rfa_api.cpp: only for head dim 128 / 64 without a window
    if (p.ds != nullptr)
      return dtype == 0 ? launch_dkdv_t<bf16_t, 128, true, true, false, true>(p, stream)
                        : launch_dkdv_t<f16_t, 128, true, true, false, true>(p, stream);
    return dtype == 0 ? launch_dkdv_t<bf16_t, 128, true, false, false, true>(p, stream)
                      : launch_dkdv_t<f16_t, 128, true, false, false, true>(p, stream);
  }
  if (p.ds != nullptr && p.D == 128 && !win)      // dS spill instance (rfa_api.cpp only passes ds for eligible calls)
    return dtype == 0 ? launch_dkdv_t<bf16_t, 128, true, true, false>(p, stream)
                      : launch_dkdv_t<f16_t, 128, true, true, false>(p, stream);
  if (win) return dtype == 0 ? launch_dkdv_d<bf16_t, true, false>(p, stream) : launch_dkdv_d<f16_t, true, false>(p, stream);
  return dtype == 0 ? launch_dkdv_d<bf16_t, false, false>(p, stream) : launch_dkdv_d<f16_t, false, false>(p, stream);
}
int bwd_dq_rows_per_block() { return kDqRows; }
int bwd_dkdv_keys_per_block(bool wide) { return wide ? 2 * kKvKeys : kKvKeys; }

}  // namespace rfa
