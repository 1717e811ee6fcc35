// rfa_bigd.hip — attention forward and backward for head dims 129 … 256 on gfx950 (MI355X).
//
// flash_attn accepts head dims up to 256 and the reference only asks for d % 8 == 0
// (/root/reference/test/test_zigzag_ring_flash_attn_func.py:35); the tuned kernels of rfa_fwd.hip / rfa_bwd.hip are
// built around one 128-column LDS tile and 256 registers per wave and stop at 128.  The three kernels here cover
// the rest with the same parameter blocks, masks, merge / accumulate epilogues and dropout mask, so that every
// schedule runs unchanged (rfa_api.cpp sends D > 128 here):
//   * a head is handled as TWO 128-column chunks: an LDS tile [rows][256] is two of the swizzled [rows][128] chunk
//     tiles of rfa_common.hpp back to back, so the fragment reads (row-per-lane ds_read_b128, ds_read_b64_tr_b16)
//     and the LDS-DMA lane mapping are the ones of the 128-wide kernels plus a chunk offset; D < 256 is zero padded
//     by pointing the DMA lanes of the missing 16-byte chunks past the buffer range (they then write zeros);
//   * 4 waves per workgroup = one wave per SIMD with the whole 512-entry register file: the Q (and dO) fragments of a
//     256-wide row are 64 (128) registers, the O / dQ accumulators 128, dK + dV 256;
//   * forward / dQ: a wave owns 32 query rows, 64-key K/V tiles, two LDS stages (128 KiB);
//     dK/dV: a wave owns 32 keys with K_w, V_w as register B operands, 32-row Q/dO tiles of all query heads of the
//     K/V group through a 4-stage LDS-DMA ring (130 KiB), dK/dV of the group summed in registers; one launch per
//     tensor, the tile range of a key block shared by up to 4 workgroups (fp32 partials, reduce_kernel);
//   * D == 256: the dK launch stores its dS blocks and rfa_dqs.hip computes dQ from them (one launch per 128-column
//     chunk) — the 5-GEMM form of rfa_bwd.hip; other wide dims recompute S and dP in dq_big_kernel.
// These are coverage kernels, written for clarity: no hand-placed schedules beyond reading LDS operands ahead of their
// MFMAs and issuing the next tile's DMA pieces in MFMA shadows.  Measured rates are in DESIGN.md §3.2.
#include <type_traits>

#include "rfa_common.hpp"
#include "rfa_kernels.hpp"

namespace rfa {

constexpr int kBgWaves = 4;
constexpr int kBgThreads = kBgWaves * 64;
constexpr int kBgRows = kBgWaves * 32;        // query rows (forward, dQ) / keys (dK/dV) per workgroup
// (per kernel instance: kNK = 4 kQ 16-wide k-steps of a contraction, kNB = 2 kQ 32-wide column blocks of an accumulator
//  row, kQ = the 64-column quarters of the 256-wide row that hold data: 3 for head dims <= 192, else 4)
constexpr int kBgOob = 0x7ffffff0;            // byte offset past every descriptor range: the lane reads / DMAs zeros

// Per-lane global byte offsets of the LDS-DMA pieces one wave issues for a wide tile of R rows.  Piece P (1 KiB =
// 4 rows of a chunk tile) belongs to chunk c = P / (R / 4); waves take the pieces wave, wave + 4, ...
template <int R>
__device__ __forceinline__ void big_dma_offsets(int wave, int lane, int row_stride, int D, int (&voff)[R / 8]) {
#pragma unroll
  for (int i = 0; i < R / 8; ++i) {
    const int P = wave + kBgWaves * i;
    const int c = P / (R / 4), pc = P % (R / 4);
    int row, ch;
    dma_lane_src<128>(pc, lane, row, ch);
    const int col = 128 * c + 8 * ch;
    voff[i] = col < D ? (row * row_stride + col) * 2 : kBgOob;
  }
}
template <int R>
__device__ __forceinline__ void big_dma_piece(dma_rsrc_t r, int lds_base, int wave, const int (&voff)[R / 8], int i) {
  const int P = wave + kBgWaves * i;
  const int c = P / (R / 4), pc = P % (R / 4);
  dma_load128(r, lds_base + c * R * 256 + pc * 1024, voff[i]);
}
template <int R>
__device__ __forceinline__ void big_dma_tile(dma_rsrc_t r, int lds_base, int wave, const int (&voff)[R / 8]) {
#pragma unroll
  for (int i = 0; i < R / 8; ++i) {
    const int P = wave + kBgWaves * i;
    const int c = P / (R / 4), pc = P % (R / 4);
    dma_load128(r, lds_base + c * R * 256 + pc * 1024, voff[i]);
  }
}

// kN MFMAs whose A operands come from LDS: operand i is fetched by fa(i) — kReads LDS instructions — kAhead MFMAs
// before mm(i, a) consumes it.  One wave per SIMD hides nothing by itself: without the read-ahead every MFMA waits
// out the full LDS latency (measured: 5x).  The sched_group_barriers pin the issue order the loop spells out.
#ifndef RFA_BG_AHEAD
#define RFA_BG_AHEAD 4
#endif
// measurement-only switches for dkdv_big_kernel (results are wrong when one is 0): the loop without its DMA pieces /
// its per-tile wait + barrier / the exp-mask-multiply work / the LDS fragment reads
#ifndef RFA_BG_FUSED_KV
#define RFA_BG_FUSED_KV 0    // 1: head dims <= 192 compute dK and dV in ONE launch (see dkdv_big_kernel: spills with hipcc 7.2)
#endif
#ifndef RFA_BG_X_DMA
#define RFA_BG_X_DMA 1
#endif
#ifndef RFA_BG_X_SYNC
#define RFA_BG_X_SYNC 1
#endif
#ifndef RFA_BG_X_VALU
#define RFA_BG_X_VALU 1
#endif
#ifndef RFA_BG_X_LDS
#define RFA_BG_X_LDS 1
#endif
struct big_no_side {
  __device__ __forceinline__ void operator()(int) const {}
};
// side(i): issued behind MFMA i — the LDS-DMA pieces of the next tile go here, so that their issue time falls into the
// MFMA's shadow instead of in front of the GEMM (one in-order wave per SIMD overlaps nothing by itself)
template <typename T, int kN, int kReads, typename FA, typename MM, typename SD = big_no_side>
__device__ __forceinline__ void big_gemm(FA fa, MM mm, SD side = SD()) {
  constexpr int kAhead = RFA_BG_AHEAD < kN ? RFA_BG_AHEAD : kN;
  vec8<T> a[kN];
#pragma unroll
  for (int i = 0; i < kAhead; ++i) a[i] = fa(i);
#pragma unroll
  for (int i = 0; i < kN; ++i) {
    if (i + kAhead < kN) a[i + kAhead] = fa(i + kAhead);
    mm(i, a[i]);
    side(i);
  }
  __builtin_amdgcn_sched_group_barrier(0x100, kReads * kAhead, 0);
#pragma unroll
  for (int i = 0; i < kN - kAhead; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
}

// =====================================================================================
// forward
// =====================================================================================
constexpr int kBgKV = 64;                          // keys per K/V tile (forward, dQ)
constexpr int kBgChunkTile = kBgKV * 256;          // [64][128] chunk tile
constexpr int kBgTile = 2 * kBgChunkTile;          // [64][256]
constexpr int kBgFwdSmem = 4 * kBgTile;            // K[2] V[2]: 128 KiB

template <typename T, int kQ>
__global__ __launch_bounds__(kBgThreads) void fwd_big_kernel(const FwdParams p) {
  constexpr int kNK = 4 * kQ, kNB = 2 * kQ;         // k-steps / column blocks that hold data (kQ quarters of 64 columns)
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
  int qblk_i, b;
  split_block_batch<RFA_BATCH_FAST_Q>(idx, p.nqblk, p.B, qblk_i, b);
  const int qblk = p.nqblk - 1 - qblk_i;
  const int h = hk * G + gq;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int qwg0 = qblk * kBgRows;
  if (qwg0 >= lq) return;
  const int off = lk - lq;
  const int qw0 = qwg0 + wave * 32;
  const int qrow = qw0 + l31;
  const int qrow_c = qrow < lq ? qrow : lq - 1;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;

  const T* qbase = (const T*)p.q + qbatch * p.q_st.batch + (qs.row0 + qrow_c) * p.q_st.row + (int64_t)h * p.q_st.head;
  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + ks.row0 * p.v_st.row + (int64_t)hk * p.v_st.head;

  vec8<T> qf[kNK];
#pragma unroll
  for (int kk = 0; kk < kNK; ++kk) {
    const int d0 = 16 * kk + 8 * g;
    qf[kk] = d0 < p.D ? *(const vec8<T>*)(qbase + d0) : zero8<T>();
  }

  const int qend = (qwg0 + kBgRows < lq) ? qwg0 + kBgRows : lq;
  const bool win = p.wl >= 0 || (p.wr >= 0 && !p.causal);       // (rfa_kernels.hpp: windowed)
  const bool hi = win ? p.wr >= 0 : p.causal != 0;
  const bool lo = win && p.wl >= 0;
  const int wr = win ? p.wr : 0, wl = win ? p.wl : 0;
  int kmax = lk;
  if (hi && qend + off + wr < kmax) kmax = qend + off + wr;
  const int ntiles = kmax > 0 ? (kmax + kBgKV - 1) / kBgKV : 0;
  int kmin = lo ? qwg0 + off - wl : 0;
  kmin = kmin > 0 ? kmin : 0;
  const int jt0 = kmin / kBgKV;

  int voff_k[kBgKV / 8], voff_v[kBgKV / 8];
  big_dma_offsets<kBgKV>(wave, lane, (int)p.k_st.row, p.D, voff_k);
  big_dma_offsets<kBgKV>(wave, lane, (int)p.v_st.row, p.D, voff_v);
  struct TileLoad {
    dma_rsrc_t rk, rv;
    int kb, vb;
  };
  auto prep_tile = [&](int j, int stage) {               // descriptors and LDS targets of a K/V tile (scalar work)
    int rows = lk - j * kBgKV;
    rows = rows < kBgKV ? rows : kBgKV;
    const int nk = rows > 0 ? ((rows - 1) * (int)p.k_st.row + p.D) * 2 : 0;
    const int nv = rows > 0 ? ((rows - 1) * (int)p.v_st.row + p.D) * 2 : 0;
    TileLoad t;
    t.rk = make_dma_rsrc(kbase + (int64_t)j * kBgKV * p.k_st.row, nk);
    t.rv = make_dma_rsrc(vbase + (int64_t)j * kBgKV * p.v_st.row, nv);
    t.kb = lds_addr(smem) + stage * kBgTile;
    t.vb = lds_addr(smem) + (2 + stage) * kBgTile;
    return t;
  };
  constexpr int kPieces = 2 * (kBgKV / 8);               // DMA pieces per wave and tile (K, then V)
  auto issue_piece = [&](const TileLoad& t, int i) {
    if (i < kBgKV / 8) big_dma_piece<kBgKV>(t.rk, t.kb, wave, voff_k, i);
    else big_dma_piece<kBgKV>(t.rv, t.vb, wave, voff_v, i - kBgKV / 8);
  };
  auto load_tile = [&](int j, int stage) {
    const TileLoad t = prep_tile(j, stage);
#pragma unroll
    for (int i = 0; i < kPieces; ++i) issue_piece(t, i);
  };

  // fragment addresses inside a chunk tile: K rows (A operand of S^T = K Q^T), V^T by transpose reads
  int koff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) koff[kk] = lds_addr(smem) + tile_off(l31, 2 * kk + g);
  int voff[4][2];
#pragma unroll
  for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) voff[dblk][hh] = lds_addr(smem) + tr_off_d<128>(lane, dblk, 8 * hh + 4 * g);

  const bool drop = p.drop_keep < 256;
  const uint32_t drop_key = drop ? drop_head_key(p.drop_seed, p.cu_q ? 0u : (uint32_t)b, p.head0 + (uint32_t)h) : 0u;
  const uint32_t drop_i = drop ? p.q_pos0 + (uint32_t)(p.cu_q ? qs.row0 : 0) + (uint32_t)qrow : 0u;
  const uint32_t drop_j0 = drop ? p.k_pos0 + (uint32_t)(p.cu_k ? ks.row0 : 0) : 0u;
  const float c = p.scale * kLog2e;
  float m = -INFINITY;
  float lsum = 0.f;
  f32x16 o[kNB];
#pragma unroll
  for (int i = 0; i < kNB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

  load_tile(jt0, 0);                     // unconditional (rows past the end read as zero)
  wait_all_vmem();
  __syncthreads();

  for (int j = jt0; j < ntiles; ++j) {
    const int stage = (j - jt0) & 1;
    const bool more = j + 1 < ntiles;
    const TileLoad nxt = prep_tile(more ? j + 1 : j, stage ^ 1);
    const int kbo = stage * kBgTile, vbo = (2 + stage) * kBgTile;
    const int kt0 = j * kBgKV;
    const bool active = (qw0 < lq) && !(hi && kt0 > qw0 + 31 + off + wr) && !(lo && kt0 + kBgKV - 1 < qw0 + off - wl);
    if (!active && more) {
#pragma unroll
      for (int i = 0; i < kPieces; ++i) issue_piece(nxt, i);
    }
    if (active) {
      f32x16 s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
      big_gemm<T, 2 * kNK, 1>(
          [&](int i) {                                  // i = t * kNK + kk
            const int t = i / kNK, kk = i % kNK;
            return lds_read128<T>(lds_ptr(koff[kk & 7]) + kbo + (kk >> 3) * kBgChunkTile + t * 32 * 256);
          },
          [&](int i, vec8<T> a) { s[i / kNK] = mfma(a, qf[i % kNK], s[i / kNK]); },
          [&](int i) {                                  // the next tile's DMA pieces in the shadows of the first MFMAs
            if (i < kPieces && more) issue_piece(nxt, i);
          });
      const bool need_mask = (kt0 + kBgKV > lk) || (hi && kt0 + kBgKV - 1 > qw0 + off + wr) || (lo && kt0 < qw0 + 31 + off - wl);
      if (need_mask) {
        const int lim = hi ? ((qrow + off + wr < lk - 1) ? qrow + off + wr : lk - 1) : lk - 1;
        const int lim_lo = lo ? qrow + off - wl : -0x40000000;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt0 + 32 * t + crow(r, g);
            if (key > lim || key < lim_lo) s[t][r] = -INFINITY;
          }
      }
      float mloc = s[0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[1][r]);
      mloc = max_xor32(mloc);               // v_permlane32_swap, no LDS round trip (rfa_common.hpp)
      const float mnew = fmaxf(m, mloc);
      // deferred rescale as in rfa_fwd.hip: while no row of the wave grew its max by more than 8 log2 units the stale
      // max stays (P <= 2^8, exact in fp32) and the 128-register rescale of O — accumulator registers read, scaled
      // and written back — is skipped
      if (!__all((mnew - m) * c <= 8.f)) {
        const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = fast_exp2(m * c - msafe * c);
        m = mnew;
        lsum *= alpha;
#pragma unroll
        for (int i = 0; i < kNB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      }
      const float mc = ((m == -INFINITY) ? 0.f : m) * c;
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(__builtin_fmaf(s[t][r], c, -mc));
          s[t][r] = pv;
          psum += pv;
        }
      lsum += psum;
      if (drop) {
        // the forward kernel's mask (rfa_fwd.hip): row sum and lse stay those of the undropped softmax
        const int mis = __builtin_amdgcn_readfirstlane((int)(drop_j0 & 3u));
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int mm = 0; mm < 4; ++mm) {
            const uint32_t jg = drop_j0 + (uint32_t)(kt0 + 32 * t + 8 * mm + 4 * g);
            uint32_t w = drop_word(drop_key, drop_i, jg >> 2);
            if (mis) w = __builtin_amdgcn_alignbyte(drop_word(drop_key, drop_i, (jg >> 2) + 1), w, (uint32_t)mis);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (!drop_keep(w, e, p.drop_keep)) s[t][4 * mm + e] = 0.f;
          }
      }
      {
        vec8<T> pb[4];                                  // [t][ks2]
#pragma unroll
        for (int x = 0; x < 4; ++x) pb[x] = pack8<T>(s[x >> 1], 8 * (x & 1));
        big_gemm<T, 4 * kNB, 2>(
            [&](int i) {                                // i = (t, ks2) * kNB + dblk: rows 32 t + 16 ks2 of the V tile
              const int dblk = i % kNB, cimm = vbo + (i / kNB) * 16 * 256 + (dblk >> 2) * kBgChunkTile;
              return concat<T>(lds_read_tr<T>(lds_ptr(voff[dblk & 3][0]) + cimm), lds_read_tr<T>(lds_ptr(voff[dblk & 3][1]) + cimm));
            },
            [&](int i, vec8<T> a) { o[i % kNB] = mfma(a, pb[i / kNB], o[i % kNB]); });
      }
    }
    wait_all_vmem();           // tile j+1 has landed
    __syncthreads();
  }

  if (qrow >= lq) return;
  const float l = sum_xor32(lsum);
  const bool has = l > 0.f;
  const float inv = has ? (drop ? p.drop_scale : 1.f) / l : 0.f;
  const float blse = has ? m * p.scale + __logf(l) : INFINITY;
  const int64_t orow = qs.row0 + qrow;
  if (p.out_acc == nullptr) {
    T* ob = (T*)p.out + qbatch * p.out_st.batch + orow * p.out_st.row + (int64_t)h * p.out_st.head;
    store_rows16<T, false, kNB>(ob, o, inv, g, p.D, true);
    if (g == 0) p.lse[qbatch * p.lse_batch + (int64_t)h * p.lse_head + orow] = blse;
  } else {
    // fused online merge into the caller's fp32 accumulators: the epilogue of rfa_fwd.hip
    float* ab = p.out_acc + qbatch * p.out_acc_st.batch + orow * p.out_acc_st.row + (int64_t)h * p.out_acc_st.head;
    float* lp = p.lse_acc + qbatch * p.lse_acc_batch + (int64_t)h * p.lse_acc_head + orow;
    if (p.acc_init) {
#pragma unroll
      for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int d0 = 32 * dblk + 8 * jj + 4 * g;
          if (d0 < p.D) {
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = o[dblk][4 * jj + e] * inv;
            *(f32x4*)(ab + d0) = x;
          }
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
      for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int d0 = 32 * dblk + 8 * jj + 4 * g;
          if (d0 < p.D) {
            f32x4 x = *(f32x4*)(ab + d0);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = x[e] * wo + o[dblk][4 * jj + e] * wb;
            *(f32x4*)(ab + d0) = x;
          }
        }
      if (g == 0) *lp = lnew;
    }
  }
}

// =====================================================================================
// dQ (7-GEMM form: S and dP recomputed)
// =====================================================================================
template <typename T, int kQ>
__global__ __launch_bounds__(kBgThreads) void dq_big_kernel(const BwdParams p) {
  constexpr int kNK = 4 * kQ, kNB = 2 * kQ;
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
  int qblk_i, b;
  split_block_batch<RFA_BATCH_FAST_Q>(idx, p.nqblk, p.B, qblk_i, b);
  const int qblk = p.nqblk - 1 - qblk_i;
  const int h = hk * G + gq;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int qwg0 = qblk * kBgRows;
  if (qwg0 >= lq) return;
  const int off = lk - lq;
  const int qw0 = qwg0 + wave * 32;
  const int qrow = qw0 + l31;
  const int qrow_c = qrow < lq ? qrow : lq - 1;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;
  const int64_t arow = qs.row0 + qrow_c;

  const T* qbase = (const T*)p.q + qbatch * p.q_st.batch + arow * p.q_st.row + (int64_t)h * p.q_st.head;
  const T* dobase = (const T*)p.dout + qbatch * p.dout_st.batch + arow * p.dout_st.row + (int64_t)h * p.dout_st.head;
  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + ks.row0 * p.v_st.row + (int64_t)hk * p.v_st.head;

  vec8<T> qf[kNK], dof[kNK];
#pragma unroll
  for (int kk = 0; kk < kNK; ++kk) {
    const int d0 = 16 * kk + 8 * g;
    qf[kk] = d0 < p.D ? *(const vec8<T>*)(qbase + d0) : zero8<T>();
    dof[kk] = d0 < p.D ? *(const vec8<T>*)(dobase + d0) : zero8<T>();
  }
  const float L2 = p.lse[qbatch * p.lse_batch + (int64_t)h * p.lse_head + arow] * kLog2e;
  const float dlt = p.delta[qbatch * p.delta_batch + (int64_t)h * p.delta_head + arow];

  const int qend = (qwg0 + kBgRows < lq) ? qwg0 + kBgRows : lq;
  const bool win = p.wl >= 0 || (p.wr >= 0 && !p.causal);       // (rfa_kernels.hpp: windowed)
  const bool hi = win ? p.wr >= 0 : p.causal != 0;
  const bool lo = win && p.wl >= 0;
  const int wr = win ? p.wr : 0, wl = win ? p.wl : 0;
  int kmax = lk;
  if (hi && qend + off + wr < kmax) kmax = qend + off + wr;
  const int ntiles = kmax > 0 ? (kmax + kBgKV - 1) / kBgKV : 0;
  int kmin = lo ? qwg0 + off - wl : 0;
  kmin = kmin > 0 ? kmin : 0;
  const int jt0 = kmin / kBgKV;

  int voff_k[kBgKV / 8], voff_v[kBgKV / 8];
  big_dma_offsets<kBgKV>(wave, lane, (int)p.k_st.row, p.D, voff_k);
  big_dma_offsets<kBgKV>(wave, lane, (int)p.v_st.row, p.D, voff_v);
  auto load_tile = [&](int j, int stage) {
    int rows = lk - j * kBgKV;
    rows = rows < kBgKV ? rows : kBgKV;
    const int nk = rows > 0 ? ((rows - 1) * (int)p.k_st.row + p.D) * 2 : 0;
    const int nv = rows > 0 ? ((rows - 1) * (int)p.v_st.row + p.D) * 2 : 0;
    const dma_rsrc_t rk = make_dma_rsrc(kbase + (int64_t)j * kBgKV * p.k_st.row, nk);
    const dma_rsrc_t rv = make_dma_rsrc(vbase + (int64_t)j * kBgKV * p.v_st.row, nv);
    big_dma_tile<kBgKV>(rk, lds_addr(smem) + stage * kBgTile, wave, voff_k);
    big_dma_tile<kBgKV>(rv, lds_addr(smem) + (2 + stage) * kBgTile, wave, voff_v);
  };
  int koff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) koff[kk] = lds_addr(smem) + tile_off(l31, 2 * kk + g);
  int toff[4][2];
#pragma unroll
  for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) toff[dblk][hh] = lds_addr(smem) + tr_off_d<128>(lane, dblk, 8 * hh + 4 * g);

  const bool drop = p.drop_keep < 256;
  const uint32_t drop_key = drop ? drop_head_key(p.drop_seed, p.cu_q ? 0u : (uint32_t)b, p.head0 + (uint32_t)h) : 0u;
  const uint32_t drop_i = drop ? p.q_pos0 + (uint32_t)(p.cu_q ? qs.row0 : 0) + (uint32_t)qrow : 0u;
  const uint32_t drop_j0 = drop ? p.k_pos0 + (uint32_t)(p.cu_k ? ks.row0 : 0) : 0u;
  const float c = p.scale * kLog2e;
  f32x16 dq[kNB];
#pragma unroll
  for (int i = 0; i < kNB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

  load_tile(jt0, 0);
  wait_all_vmem();
  __syncthreads();

  for (int j = jt0; j < ntiles; ++j) {
    const int stage = (j - jt0) & 1;
    // (the next tile's DMA is issued in front of this one: interleaving the pieces with sub-tile 0's MFMAs, which gains
    //  3 - 7 % in the forward and dK/dV kernels, measured 1.01 -> 1.21 ms here)
    if (j + 1 < ntiles) load_tile(j + 1, stage ^ 1);
    const int kbo = stage * kBgTile, vbo = (2 + stage) * kBgTile;
    const int kt0 = j * kBgKV;
    const bool active = (qw0 < lq) && !(hi && kt0 > qw0 + 31 + off + wr) && !(lo && kt0 + kBgKV - 1 < qw0 + off - wl);
    if (active) {
      const bool need_mask = (kt0 + kBgKV > lk) || (hi && kt0 + kBgKV - 1 > qw0 + off + wr) || (lo && kt0 < qw0 + 31 + off - wl);
      const int lim = hi ? ((qrow + off + wr < lk - 1) ? qrow + off + wr : lk - 1) : lk - 1;
      const int lim_lo = lo ? qrow + off - wl : -0x40000000;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        big_gemm<T, 2 * kNK, 1>(
            [&](int i) {                                // i = (0: K tile for S, 1: V tile for dP) * kNK + kk
              const int kk = i % kNK;
              return lds_read128<T>(lds_ptr(koff[kk & 7]) + ((i / kNK) ? vbo : kbo) + (kk >> 3) * kBgChunkTile + t * 32 * 256);
            },
            [&](int i, vec8<T> a) {
              if (i < kNK) s = mfma(a, qf[i % kNK], s);
              else dp = mfma(a, dof[i % kNK], dp);
            });
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(__builtin_fmaf(s[r], c, -L2));
        if (need_mask) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt0 + 32 * t + crow(r, g);
            s[r] = (key > lim || key < lim_lo) ? 0.f : s[r];
          }
        }
        if (drop) {
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
        {
          const vec8<T> dsb[2] = {pack8<T>(s, 0), pack8<T>(s, 8)};
          big_gemm<T, 2 * kNB, 2>(
              [&](int i) {                              // i = ks2 * kNB + dblk
                const int dblk = i % kNB, cimm = kbo + (32 * t + 16 * (i / kNB)) * 256 + (dblk >> 2) * kBgChunkTile;
                return concat<T>(lds_read_tr<T>(lds_ptr(toff[dblk & 3][0]) + cimm), lds_read_tr<T>(lds_ptr(toff[dblk & 3][1]) + cimm));
              },
              [&](int i, vec8<T> a) { dq[i % kNB] = mfma(a, dsb[i / kNB], dq[i % kNB]); });
        }
      }
    }
    wait_all_vmem();
    __syncthreads();
  }

  if (qrow >= lq) return;
  const int64_t orow = qs.row0 + qrow;
  if (p.dq_acc == nullptr) {
    T* ob = (T*)p.dq + qbatch * p.dq_st.batch + orow * p.dq_st.row + (int64_t)h * p.dq_st.head;
    store_rows16<T, false, kNB>(ob, dq, p.scale, g, p.D, true);
  } else {
    float* ab = p.dq_acc + qbatch * p.dq_acc_st.batch + orow * p.dq_acc_st.row + (int64_t)h * p.dq_acc_st.head;
#pragma unroll
    for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int d0 = 32 * dblk + 8 * jj + 4 * g;
        if (d0 < p.D) {
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
// dK / dV
// =====================================================================================
constexpr int kBgQ = 32;                             // query rows per Q/dO tile
constexpr int kBgQChunk = kBgQ * 256;                // [32][128] chunk tile: 8 KiB
constexpr int kBgQTile = 2 * kBgQChunk;              // [32][256]: 16 KiB
constexpr int kBgQStages = 4;                        // LDS ring: three tiles in flight behind the one being computed — one
                                                     // wave per SIMD and 32 - 48 MFMAs per tile do not cover a DMA round trip
constexpr int kBgOffDo = kBgQStages * kBgQTile;      // dO stages behind the Q stages (64 KiB)
constexpr int kBgOffStat = 2 * kBgOffDo;             // 128 KiB: per stage lse[64 slots, 32 used], delta[64 slots]
constexpr int kBgStatBytes = 2 * 256;
constexpr int kBgKvSmem = kBgOffStat + kBgQStages * kBgStatBytes;

// 4-byte LDS-DMA (buffer_load_dword ... lds): lane L's dword lands at lds_wave_base + 4 L, out-of-range lanes write zeros
__device__ __forceinline__ void dma_load32(dma_rsrc_t r, int lds_wave_base, int voffset) {
  asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dword %0, %1, 0 offen lds"
               :
               : "v"(voffset), "s"(r.w), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base))
               : "m0");
}
// s_waitcnt vmcnt(n), n < 64 (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14])
template <int n>
__device__ __forceinline__ void wait_vmem64() {
  __builtin_amdgcn_s_waitcnt(0x0F70 | (n & 15) | ((n >> 4) << 14));
}

// kWhich: 0 = dV, 1 = dK, 2 = both.  Head dims > 192 take one launch per tensor: with both accumulator sets (2 x 128
// registers) next to the K_w / V_w operands (2 x 64) the arch-VGPR half of the register file overflows (hipcc: 290
// accumulator-register moves and 80 scratch reloads per tile); a launch that owns one tensor fits, at the price of
// computing S twice (80 instead of 64 MFMAs per tile and wave).  kWhich = 2 at kQ = 3 (head dims <= 192: 2 x 96 + 2 x 48
// registers, 48 MFMAs per tile and wave instead of 60) is written but NOT launched (-DRFA_BG_FUSED_KV=1): hipcc still
// allocates it with 1099 accumulator-register moves and 212 bytes of scratch reloaded inside the tile loop (round 4).
template <typename T, int kWhich, int kQ>
__global__ __launch_bounds__(kBgThreads) void dkdv_big_kernel(const BwdParams p) {
  constexpr int kNK = 4 * kQ, kNB = 2 * kQ;
  constexpr bool kDoK = kWhich != 0, kDoV = kWhich != 1;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  if (lds_addr(smem) & 0xffff) __builtin_trap();     // the stage / fragment XORs below need a 64 KiB-aligned block
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5;
  const int l31 = lane & 31;

  int idx = blockIdx.x;
  const int G = p.H / p.Hk;
  const int hk = idx % p.Hk;
  idx /= p.Hk;
  const int nsplit = p.nsplit;
  const int qsplit = idx % nsplit;                    // which share of the key block's tiles (rfa_bwd.hip: kWide)
  idx /= nsplit;
  int kblk, b;
  split_block_batch<RFA_BATCH_FAST_KV>(idx, p.nkblk, p.B, kblk, b);
  const int h0 = hk * G;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int kwg0 = kblk * kBgRows;
  if (kwg0 >= lk) return;
  const int off = lk - lq;
  const int kw0 = kwg0 + wave * 32;
  const int krow = kw0 + l31;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;

  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + ks.row0 * p.v_st.row + (int64_t)hk * p.v_st.head;
  const T* qbase0 = (const T*)p.q + qbatch * p.q_st.batch + qs.row0 * p.q_st.row + (int64_t)h0 * p.q_st.head;
  const T* dobase0 = (const T*)p.dout + qbatch * p.dout_st.batch + qs.row0 * p.dout_st.row + (int64_t)h0 * p.dout_st.head;
  const float* lsebase0 = p.lse + qbatch * p.lse_batch + (int64_t)h0 * p.lse_head + qs.row0;
  const float* dltbase0 = p.delta + qbatch * p.delta_batch + (int64_t)h0 * p.delta_head + qs.row0;

  const bool win = p.wl >= 0 || (p.wr >= 0 && !p.causal);       // (rfa_kernels.hpp: windowed)
  const bool hi = win ? p.wr >= 0 : p.causal != 0;
  const bool lo = win && p.wl >= 0;
  const int wr = win ? p.wr : 0, wl = win ? p.wl : 0;
  int qfirst = 0;
  if (hi) {
    qfirst = kwg0 - off - wr;
    if (qfirst < 0) qfirst = 0;
  }
  int qlast = lq;
  if (lo && kwg0 + kBgRows - off + wl < qlast) qlast = kwg0 + kBgRows - off + wl;
  const int jt0 = qfirst / kBgQ;
  int jt1 = (qlast + kBgQ - 1) / kBgQ;
  if (jt1 <= jt0) jt1 = jt0;
  // this workgroup's tiles: every nsplit-th one counted from the top (interleaved, so that all workgroups walk down
  // the same tiles at about the same time); the shares' fp32 partials are summed by reduce_kernel (rfa_api.cpp)
  const int jtop = jt1 - 1 - qsplit;
  const int ntile_q = jtop >= jt0 ? (jtop - jt0) / nsplit + 1 : 0;

  // this wave's K and V rows: register B operands of S = Q K_w^T and dP = dO V_w^T
  vec8<T> kwr[kNK], vwr[kDoK ? kNK : 1];
  {
    const int kr = krow < lk ? krow : lk - 1;
    const T* kp = kbase + (int64_t)kr * p.k_st.row;
    const T* vp = vbase + (int64_t)kr * p.v_st.row;
#pragma unroll
    for (int kk = 0; kk < kNK; ++kk) {
      const int d0 = 16 * kk + 8 * g;
      kwr[kk] = d0 < p.D ? *(const vec8<T>*)(kp + d0) : zero8<T>();
      if (kDoK) vwr[kk] = d0 < p.D ? *(const vec8<T>*)(vp + d0) : zero8<T>();
    }
  }

  int voff_q[kBgQ / 8], voff_do[kBgQ / 8];
  big_dma_offsets<kBgQ>(wave, lane, (int)p.q_st.row, p.D, voff_q);
  big_dma_offsets<kBgQ>(wave, lane, (int)p.dout_st.row, p.D, voff_do);
  const int ntile = ntile_q * G;
  // Tiles are walked from the top one down, and for every tile the G heads of the group (the walk order of rfa_bwd.hip:
  // all workgroups of a launch read the same (tile, head) at about the same time).  Every wave issues the same number
  // of DMA instructions per tile — 8 tile pieces, plus one statistics row for waves 0 (lse) and 1 (delta) — so that
  // the counted waits below are exact; loads past the last tile are issued with an empty range (they write zeros).
  int ld_n = 0, ld_g = 0, ld_j = jtop > 0 ? jtop : 0;
  struct TileLoad {
    dma_rsrc_t rq, rdo, rs;
    int base, sbase;
  };
  auto prep_tile = [&]() {                               // descriptors and LDS targets of the next tile (scalar work)
    const bool valid = ld_n < ntile;
    const int j = valid ? ld_j : 0;
    const int stage = ld_n & (kBgQStages - 1);
    const T* qbase = qbase0 + (int64_t)ld_g * p.q_st.head;
    const T* dobase = dobase0 + (int64_t)ld_g * p.dout_st.head;
    const float* statbase = (wave ? dltbase0 + (int64_t)ld_g * p.delta_head : lsebase0 + (int64_t)ld_g * p.lse_head) + j * kBgQ;
    ++ld_n;
    if (++ld_g >= G) {
      ld_g = 0;
      ld_j -= nsplit;
    }
    int rows = lq - j * kBgQ;
    rows = rows < kBgQ ? rows : kBgQ;
    rows = valid && rows > 0 ? rows : 0;
    const int nq = rows > 0 ? ((rows - 1) * (int)p.q_st.row + p.D) * 2 : 0;
    const int ndo = rows > 0 ? ((rows - 1) * (int)p.dout_st.row + p.D) * 2 : 0;
    TileLoad t;
    t.rq = make_dma_rsrc(qbase + (int64_t)j * kBgQ * p.q_st.row, nq);
    t.rdo = make_dma_rsrc(dobase + (int64_t)j * kBgQ * p.dout_st.row, ndo);
    t.rs = make_dma_rsrc(statbase, rows * 4);
    t.base = lds_addr(smem) + stage * kBgQTile;
    t.sbase = lds_addr(smem) + kBgOffStat + stage * kBgStatBytes + wave * 256;
    return t;
  };
  constexpr int kPieces = 2 * (kBgQ / 8) + 1;            // per wave and tile: Q pieces, dO pieces, one statistics row
  auto issue_piece = [&](const TileLoad& t, int i) {     // i = 0 .. kPieces-1 (compile-time at every call site)
    if (i < kBgQ / 8) big_dma_piece<kBgQ>(t.rq, t.base, wave, voff_q, i);
    else if (i < 2 * (kBgQ / 8)) big_dma_piece<kBgQ>(t.rdo, t.base + kBgOffDo, wave, voff_do, i - kBgQ / 8);
    else if (wave < 2) dma_load32(t.rs, t.sbase, lane * 4);
  };
  auto load_tile = [&]() {
    const TileLoad t = prep_tile();
#pragma unroll
    for (int i = 0; i < kPieces; ++i) issue_piece(t, i);
  };
  // tile f + 1 has landed when at most the two youngest tiles' instructions are outstanding
  auto wait_next_tile = [&](bool stored) {             // stored: this tile's two dS spill stores are the youngest operations
    if (stored) {                                      // (vmcnt counts stores too and retires in issue order: rfa_bwd.hip)
      if (wave < 2) wait_vmem64<2 * (2 * (kBgQ / 8) + 1) + 2>();
      else wait_vmem64<2 * 2 * (kBgQ / 8) + 2>();
    } else {
      if (wave < 2) wait_vmem64<2 * (2 * (kBgQ / 8) + 1)>();
      else wait_vmem64<2 * 2 * (kBgQ / 8)>();
    }
  };

  const int aq0 = lds_addr(smem) + tile_off(l31, g);                 // Q / dO rows (A operands of S, dP), stage 0
  int tq0[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) tq0[hh] = lds_addr(smem) + tr_off_d<128>(lane, 0, 8 * hh + 4 * g);
  const int sa0 = lds_addr(smem) + kBgOffStat + 4 * g * 4;           // row statistics, stage 0

  const bool drop = p.drop_keep < 256;
  const uint32_t drop_j = drop ? p.k_pos0 + (uint32_t)(p.cu_k ? ks.row0 : 0) + (uint32_t)krow : 0u;
  const uint32_t drop_i0 = drop ? p.q_pos0 + (uint32_t)(p.cu_q ? qs.row0 : 0) : 0u;
  const float c = p.scale * kLog2e;
  f32x16 acck[kDoK ? kNB : 1], accv[kDoV ? kNB : 1];   // dK^T / dV^T of this wave's 32 keys
#pragma unroll
  for (int i = 0; i < kNB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (kDoK) acck[i][r] = 0.f;
      if (kDoV) accv[i][r] = 0.f;
    }

  // dS spill (D == 256, the dK launch; rfa_dqs.hip reads the blocks back — two launches, one per 128-column chunk of
  // K / dQ — instead of dq_big_kernel recomputing S and dP): block (b, h, qt = tile, kb = key / 32) of the scratch,
  // slot order and row layout of rfa_bwd.hip's kSpill instances
  const bool spill = kDoK && p.ds != nullptr;
  const int ds_lane = 16 * (16 * (l31 >> 2) + 4 * g + (l31 & 3));
  const int ds_nkb = ds_blocks(p.Sk, p.k_half);
  const int64_t ds_head_bytes = spill ? p.ds_head_blocks * kDsBlockBytes : 0;
  const char* ds_b = spill ? (const char*)p.ds + ds_base_blocks(p, b) * kDsBlockBytes : nullptr;
  const int ds_kb = __builtin_amdgcn_readfirstlane(kblk * (kBgRows / 32) + wave);

  wait_all_vmem();                                     // K_w / V_w: nothing the compiler tracks stays pending into the loop
  load_tile();
  load_tile();
  load_tile();
  wait_next_tile(false);                               // (tile 0)
  __syncthreads();

  int j = jtop, cg = 0;
  for (int f = 0; f < ntile; ++f) {
    const TileLoad nxt = prep_tile();                  // tile f + 3 goes into the stage of tile f - 1
    const int so = (f & (kBgQStages - 1)) * kBgQTile;
    const int aq = aq0 + so, tq[2] = {tq0[0] + so, tq0[1] + so};
    const int sa = sa0 + (f & (kBgQStages - 1)) * kBgStatBytes;
    const int qs0 = j * kBgQ;
    const bool active = (kw0 < lk) && (qs0 < lq) && !(hi && qs0 + 31 + off + wr < kw0) && !(lo && qs0 + off - wl > kw0 + 31);
    if (active) {
      f32x16 s, dp;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {               // dp starts at -delta[q]
        const f32x4 dl = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + 256 + 8 * jj * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { dp[4 * jj + e] = kDoK ? -dl[e] : 0.f; s[4 * jj + e] = 0.f; }
      }
      big_gemm<T, (kDoK ? 2 : 1) * kNK, 1>(
          [&](int i) {                                  // i = (0: Q tile for S, 1: dO tile for dP) * kNK + kk
            const int kk = i % kNK;
            if (!RFA_BG_X_LDS) return kwr[kk];
            return lds_read128<T>(lds_ptr(aq ^ ((kk & 7) << 5)) + (kk >> 3) * kBgQChunk + ((i / kNK) ? kBgOffDo : 0));
          },
          [&](int i, vec8<T> a) {
            if (i < kNK) s = mfma(a, kwr[i % kNK], s);
            else dp = mfma(a, vwr[kDoK ? (i % kNK) : 0], dp);
          },
          [&](int i) {                                  // the next tile's DMA pieces in the shadows of the first MFMAs
            if (RFA_BG_X_DMA && i < kPieces) issue_piece(nxt, i);
          });
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x4 ls = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + 8 * jj * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (RFA_BG_X_VALU) s[4 * jj + e] = fast_exp2(__builtin_fmaf(s[4 * jj + e], c, -kLog2e * ls[e]));
      }
      const bool need_mask = (qs0 + 32 > lq) || (hi && qs0 + off + wr < kw0 + 31) || (lo && qs0 + 31 + off - wl > kw0);
      if (RFA_BG_X_VALU && need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = qs0 + crow(r, g);
          const bool ok = (q < lq) && (!hi || krow <= q + off + wr) && (!lo || krow >= q + off - wl);
          s[r] = ok ? s[r] : 0.f;
        }
      }
      // from here on: s = dS = P (dP - delta) for the dK GEMM, pv = the (dropped, rescaled) P for the dV GEMM
      f32x16 pv;
      if (drop) {
        // the dK/dV kernel's dropout (rfa_bwd.hip): dP = keep ? dO V^T / (1 - p) : 0, dS = P (dP - delta), dV takes keep ? P / (1 - p) : 0
        const uint32_t hkey = drop_head_key(p.drop_seed, p.cu_q ? 0u : (uint32_t)b, p.head0 + (uint32_t)(h0 + cg));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x4 dl = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + 256 + 8 * jj * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * jj + e;
            const uint32_t w = drop_word(hkey, drop_i0 + (uint32_t)(qs0 + crow(r, g)), drop_j >> 2);
            const bool keep = drop_keep(w, (int)(drop_j & 3u), p.drop_keep);
            if (kDoV) pv[r] = keep ? s[r] * p.drop_scale : 0.f;   // dropped, rescaled P
            if (kDoK) {
              const float dpd = keep ? (dp[r] + dl[e]) * p.drop_scale - dl[e] : -dl[e];
              s[r] = dpd * s[r];                       // dS
            }
          }
        }
      } else if (kDoV) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pv[r] = s[r];
      }
      // dV^T += dO^T P   /   dK^T += Q^T dS: A operands by transpose reads of the dO / Q tile, B = the packed registers
      if (kDoV) {
        const vec8<T> pb[2] = {pack8<T>(pv, 0), pack8<T>(pv, 8)};
        big_gemm<T, 2 * kNB, 2>(
            [&](int i) {                                // i = ks2 * kNB + dblk
              const int dblk = i % kNB, imm = 16 * (i / kNB) * 256 + (dblk >> 2) * kBgQChunk + kBgOffDo;
              if (!RFA_BG_X_LDS) return kwr[i % kNK];
              return concat<T>(lds_read_tr<T>(lds_ptr(tq[0] ^ ((dblk & 3) << 6)) + imm), lds_read_tr<T>(lds_ptr(tq[1] ^ ((dblk & 3) << 6)) + imm));
            },
            [&](int i, vec8<T> a) { accv[i % kNB] = mfma(a, pb[i / kNB], accv[i % kNB]); });
      }
      if (kDoK) {
        if (!drop) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] *= dp[r];  // dS = P (dP - delta)
        }
        const vec8<T> pb[2] = {pack8<T>(s, 0), pack8<T>(s, 8)};
        if (spill) {
          const char* blk = ds_b + (int64_t)(h0 + cg) * ds_head_bytes + (ds_rowpart(p, j, (qs.row0 >> 5) + b, ds_nkb) + ds_kb) * kDsBlockBytes;
          const buf_rsrc_t rb = make_rsrc(blk, kDsBlockBytes);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pb[0]), rb, ds_lane, 0, 2);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pb[1]), rb, ds_lane + 128, 0, 2);
        }
        big_gemm<T, 2 * kNB, 2>(
            [&](int i) {                                // i = ks2 * kNB + dblk
              const int dblk = i % kNB, imm = 16 * (i / kNB) * 256 + (dblk >> 2) * kBgQChunk;
              if (!RFA_BG_X_LDS) return kwr[i % kNK];
              return concat<T>(lds_read_tr<T>(lds_ptr(tq[0] ^ ((dblk & 3) << 6)) + imm), lds_read_tr<T>(lds_ptr(tq[1] ^ ((dblk & 3) << 6)) + imm));
            },
            [&](int i, vec8<T> a) { acck[i % kNB] = mfma(a, pb[i / kNB], acck[i % kNB]); });
      }
    }
    if (RFA_BG_X_DMA && !active) {
#pragma unroll
      for (int i = 0; i < kPieces; ++i) issue_piece(nxt, i);
    }
    if (bool(RFA_BG_X_SYNC) & bool(RFA_BG_X_DMA)) wait_next_tile(spill && active);
    if (++cg >= G) {
      cg = 0;
      j -= nsplit;
    }
    if (RFA_BG_X_SYNC) __syncthreads();
  }
  wait_all_vmem();                                     // (the trailing empty-range loads)

  if (krow >= lk) return;
  const int64_t orow = ks.row0 + krow;
  auto store = [&](auto whichc, const f32x16 (&acc)[kNB]) {
    constexpr int which = decltype(whichc)::value;     // (the 128-wide kernel's naming: 0 = dK, 1 = dV)
    const float sc_ = which ? 1.f : p.scale;
    const Strides st = which ? p.dv_st : p.dk_st;
    const int64_t eoff = kbatch * st.batch + orow * st.row + (int64_t)hk * st.head + (int64_t)qsplit * p.kv_split_stride;
    if (p.kv_f32) {
      float* ob = (float*)(which ? p.dv : p.dk) + eoff;
#pragma unroll
      for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int d0 = 32 * dblk + 8 * jj + 4 * g;
          if (d0 < p.D) {
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = acc[dblk][4 * jj + e] * sc_;
            *(f32x4*)(ob + d0) = x;
          }
        }
    } else {
      store_rows16<T, false, kNB>((T*)(which ? p.dv : p.dk) + eoff, acc, sc_, g, p.D, true);
    }
  };
  if constexpr (kDoK) store(std::integral_constant<int, 0>{}, acck);
  if constexpr (kDoV) store(std::integral_constant<int, 1>{}, accv);
}

// =====================================================================================
// dK + dV in ONE launch (round 6)
// =====================================================================================
// The two launches above load every Q/dO tile twice and compute S twice (80 MFMAs per tile and wave for 64 of useful
// work).  Holding both accumulator sets AND both register B operands (K_w, V_w) overflows the arch-VGPR half of the
// register file (the kWhich = 2 instance above: 212 bytes of scratch in the tile loop).  This form keeps what must be in
// registers — dK^T / dV^T (2 x 128 accumulator registers: hipcc places them in the AGPR half) and K_w (64) — and puts the
// workgroup's 128 V rows into LDS ONCE (64 KiB, the arrangement of rfa_bwd.hip's 128-wide kernel): the dP GEMM reads both of
// its operands from LDS (16 more ds_read_b128 per tile and wave).  To make room the Q/dO ring shrinks from 4 stages to 2:
// with 64 MFMAs per tile (>= 2048 cycles) one tile in flight covers a DMA round trip where 32 - 48 did not.  Per tile and
// wave: 64 MFMAs, 112 LDS reads, 9 DMA pieces, one barrier — against 80 / 112 / 18 / two.
#ifndef RFA_FU_PIN_SCORES
#define RFA_FU_PIN_SCORES 0  // 1: K_w named as arch VGPRs at the top of every tile (no scratch, but MORE accumulator-register moves:
#endif                       // 486 - 494 against 503 - 506 TFLOP/s at D = 192, round 6)
#ifndef RFA_FU_SCORES_SERIAL
#define RFA_FU_SCORES_SERIAL 0  // 1: exp phase between the S and the dP chain (one score chain in accumulator registers at a time): same
#endif                          // accumulator-register traffic in hipcc's allocation, not kept
#ifndef RFA_FU_AHEAD
#define RFA_FU_AHEAD 4       // fragments read ahead of their MFMA in the fused kernel's GEMMs
#endif
constexpr int kFuStages = 2;
constexpr int kFuOffDo = kFuStages * kBgQTile;       // dO stages behind the Q stages (32 KiB)
constexpr int kFuOffV = 2 * kFuOffDo;                // 64 KiB: V rows [128][256] as two [128][128] chunk tiles
constexpr int kFuVChunk = kBgRows * 256;             // 32 KiB
constexpr int kFuOffStat = kFuOffV + 2 * kFuVChunk;  // 128 KiB
constexpr int kFuSmem = kFuOffStat + kFuStages * kBgStatBytes;

// kDrop: dropout instances (the mask hash of 16 scores per lane and tile lives in its own instance: as a run-time branch its
// registers inflate the pressure of the tile loop of every call)
template <typename T, int kQ, bool kDrop>
__global__ __launch_bounds__(kBgThreads) void dkdv_fused_big_kernel(const BwdParams p) {
  constexpr int kNK = 4 * kQ, kNB = 2 * kQ;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  if (lds_addr(smem) & 0xffff) __builtin_trap();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5;
  const int l31 = lane & 31;

  int idx = blockIdx.x;
  const int G = p.H / p.Hk;
  const int hk = idx % p.Hk;
  idx /= p.Hk;
  const int nsplit = p.nsplit;
  const int qsplit = idx % nsplit;
  idx /= nsplit;
  int kblk, b;
  split_block_batch<RFA_BATCH_FAST_KV>(idx, p.nkblk, p.B, kblk, b);
  const int h0 = hk * G;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int kwg0 = kblk * kBgRows;
  if (kwg0 >= lk) return;
  const int off = lk - lq;
  const int kw0 = kwg0 + wave * 32;
  const int krow = kw0 + l31;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;

  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + ks.row0 * p.v_st.row + (int64_t)hk * p.v_st.head;
  const T* qbase0 = (const T*)p.q + qbatch * p.q_st.batch + qs.row0 * p.q_st.row + (int64_t)h0 * p.q_st.head;
  const T* dobase0 = (const T*)p.dout + qbatch * p.dout_st.batch + qs.row0 * p.dout_st.row + (int64_t)h0 * p.dout_st.head;
  const float* lsebase0 = p.lse + qbatch * p.lse_batch + (int64_t)h0 * p.lse_head + qs.row0;
  const float* dltbase0 = p.delta + qbatch * p.delta_batch + (int64_t)h0 * p.delta_head + qs.row0;

  const bool win = p.wl >= 0 || (p.wr >= 0 && !p.causal);
  const bool hi = win ? p.wr >= 0 : p.causal != 0;
  const bool lo = win && p.wl >= 0;
  const int wr = win ? p.wr : 0, wl = win ? p.wl : 0;
  int qfirst = 0;
  if (hi) {
    qfirst = kwg0 - off - wr;
    if (qfirst < 0) qfirst = 0;
  }
  int qlast = lq;
  if (lo && kwg0 + kBgRows - off + wl < qlast) qlast = kwg0 + kBgRows - off + wl;
  const int jt0 = qfirst / kBgQ;
  int jt1 = (qlast + kBgQ - 1) / kBgQ;
  if (jt1 <= jt0) jt1 = jt0;
  const int jtop = jt1 - 1 - qsplit;
  const int ntile_q = jtop >= jt0 ? (jtop - jt0) / nsplit + 1 : 0;

  // the workgroup's 128 V rows -> LDS (one DMA burst; rows past the end of the sequence read as zeros), this wave's K rows
  // -> registers (B operand of S = Q K_w^T)
  {
    int voff_v[kBgRows / 8];
    big_dma_offsets<kBgRows>(wave, lane, (int)p.v_st.row, p.D, voff_v);
    int rows = lk - kwg0;
    rows = rows < kBgRows ? rows : kBgRows;
    const dma_rsrc_t rv = make_dma_rsrc(vbase + (int64_t)kwg0 * p.v_st.row, ((rows - 1) * (int)p.v_st.row + p.D) * 2);
    big_dma_tile<kBgRows>(rv, lds_addr(smem) + kFuOffV, wave, voff_v);
  }
  vec8<T> kwr[kNK];
  {
    const int kr = krow < lk ? krow : lk - 1;
    const T* kp = kbase + (int64_t)kr * p.k_st.row;
#pragma unroll
    for (int kk = 0; kk < kNK; ++kk) {
      const int d0 = 16 * kk + 8 * g;
      kwr[kk] = d0 < p.D ? *(const vec8<T>*)(kp + d0) : zero8<T>();
    }
  }

  int voff_q[kBgQ / 8], voff_do[kBgQ / 8];
  big_dma_offsets<kBgQ>(wave, lane, (int)p.q_st.row, p.D, voff_q);
  big_dma_offsets<kBgQ>(wave, lane, (int)p.dout_st.row, p.D, voff_do);
  const int ntile = ntile_q * G;
  int ld_n = 0, ld_g = 0, ld_j = jtop > 0 ? jtop : 0;
  struct TileLoad {
    dma_rsrc_t rq, rdo, rs;
    int base, sbase;
  };
  auto prep_tile = [&]() {
    const bool valid = ld_n < ntile;
    const int j = valid ? ld_j : 0;
    const int stage = ld_n & (kFuStages - 1);
    const T* qbase = qbase0 + (int64_t)ld_g * p.q_st.head;
    const T* dobase = dobase0 + (int64_t)ld_g * p.dout_st.head;
    const float* statbase = (wave ? dltbase0 + (int64_t)ld_g * p.delta_head : lsebase0 + (int64_t)ld_g * p.lse_head) + j * kBgQ;
    ++ld_n;
    if (++ld_g >= G) {
      ld_g = 0;
      ld_j -= nsplit;
    }
    int rows = lq - j * kBgQ;
    rows = rows < kBgQ ? rows : kBgQ;
    rows = valid && rows > 0 ? rows : 0;
    const int nq = rows > 0 ? ((rows - 1) * (int)p.q_st.row + p.D) * 2 : 0;
    const int ndo = rows > 0 ? ((rows - 1) * (int)p.dout_st.row + p.D) * 2 : 0;
    TileLoad t;
    t.rq = make_dma_rsrc(qbase + (int64_t)j * kBgQ * p.q_st.row, nq);
    t.rdo = make_dma_rsrc(dobase + (int64_t)j * kBgQ * p.dout_st.row, ndo);
    t.rs = make_dma_rsrc(statbase, rows * 4);
    t.base = lds_addr(smem) + stage * kBgQTile;
    t.sbase = lds_addr(smem) + kFuOffStat + stage * kBgStatBytes + wave * 256;
    return t;
  };
  constexpr int kPieces = 2 * (kBgQ / 8) + 1;
  auto issue_piece = [&](const TileLoad& t, int i) {
    if (i < kBgQ / 8) big_dma_piece<kBgQ>(t.rq, t.base, wave, voff_q, i);
    else if (i < 2 * (kBgQ / 8)) big_dma_piece<kBgQ>(t.rdo, t.base + kFuOffDo, wave, voff_do, i - kBgQ / 8);
    else if (wave < 2) dma_load32(t.rs, t.sbase, lane * 4);
  };

  const int aq0 = lds_addr(smem) + tile_off(l31, g);
  const int av = lds_addr(smem) + kFuOffV + 32 * wave * 256 + tile_off(l31, g);      // this wave's V rows (B operand of dP)
  int tq0[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) tq0[hh] = lds_addr(smem) + tr_off_d<128>(lane, 0, 8 * hh + 4 * g);
  const int sa0 = lds_addr(smem) + kFuOffStat + 4 * g * 4;

  constexpr bool drop = kDrop;
  const uint32_t drop_j = drop ? p.k_pos0 + (uint32_t)(p.cu_k ? ks.row0 : 0) + (uint32_t)krow : 0u;
  const uint32_t drop_i0 = drop ? p.q_pos0 + (uint32_t)(p.cu_q ? qs.row0 : 0) : 0u;
  const float c = p.scale * kLog2e;
  f32x16 acck[kNB], accv[kNB];
#pragma unroll
  for (int i = 0; i < kNB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acck[i][r] = 0.f;
      accv[i][r] = 0.f;
    }

  const bool spill = p.ds != nullptr;
  const int ds_lane = 16 * (16 * (l31 >> 2) + 4 * g + (l31 & 3));
  const int ds_nkb = ds_blocks(p.Sk, p.k_half);
  const int64_t ds_head_bytes = spill ? p.ds_head_blocks * kDsBlockBytes : 0;
  const char* ds_b = spill ? (const char*)p.ds + ds_base_blocks(p, b) * kDsBlockBytes : nullptr;
  const int ds_kb = __builtin_amdgcn_readfirstlane(kblk * (kBgRows / 32) + wave);

  {
    const TileLoad t0 = prep_tile();
#pragma unroll
    for (int i = 0; i < kPieces; ++i) issue_piece(t0, i);
  }
  wait_all_vmem();                                     // V rows, K_w, tile 0
  __syncthreads();

  int j = jtop, cg = 0;
  for (int f = 0; f < ntile; ++f) {
    const TileLoad nxt = prep_tile();                  // tile f + 1 goes into the other stage
    const int so = (f & (kFuStages - 1)) * kBgQTile;
    const int aq = aq0 + so, tq[2] = {tq0[0] + so, tq0[1] + so};
    const int sa = sa0 + (f & (kFuStages - 1)) * kBgStatBytes;
    const int qs0 = j * kBgQ;
    const bool active = (kw0 < lk) && (qs0 < lq) && !(hi && qs0 + 31 + off + wr < kw0) && !(lo && qs0 + off - wl > kw0 + 31);
    if (active) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#if !RFA_FU_SCORES_SERIAL
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x4 dl = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + 256 + 8 * jj * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[4 * jj + e] = -dl[e];
      }
#endif
#if RFA_FU_PIN_SCORES
      // In a kernel that uses AGPRs every MFMA accumulates in AGPRs: dK^T, dV^T (192 at kQ = 3) and the two score chains (32).
      // Left alone, hipcc ALSO parks K_w there (48: an MFMA-only operand), over-subscribes the 256 AGPRs and pays for it with
      // 209 v_accvgpr_write + 50 v_accvgpr_read per tile.  An empty asm that names K_w as arch VGPRs keeps it where it is.
#pragma unroll
      for (int kk = 0; kk < kNK; ++kk) asm volatile("" : "+v"(kwr[kk]));
#endif
      // S = Q K_w^T: the next tile's DMA pieces in the shadows of its first MFMAs
      big_gemm<T, kNK, 1>(
          [&](int kk) { return lds_read128<T>(lds_ptr(aq ^ ((kk & 7) << 5)) + (kk >> 3) * kBgQChunk); },
          [&](int kk, vec8<T> a) { s = mfma(a, kwr[kk], s); },
          [&](int i) {
            if (i < kPieces) issue_piece(nxt, i);
          });
#if RFA_FU_SCORES_SERIAL
      // P first (the S chain's accumulator registers are dead once P sits in arch VGPRs), THEN the dP chain: one score chain
      // in accumulator registers at a time
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x4 ls = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + 8 * jj * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[4 * jj + e] = fast_exp2(__builtin_fmaf(s[4 * jj + e], c, -kLog2e * ls[e]));
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x4 dl = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + 256 + 8 * jj * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[4 * jj + e] = -dl[e];
      }
#endif
      // dP - delta = dO V_w^T: both operands from LDS, read kAhead MFMAs ahead
      {
        constexpr int kAhead = RFA_FU_AHEAD < kNK ? RFA_FU_AHEAD : kNK;
        vec8<T> a[kNK], w[kNK];
        auto fa = [&](int kk) { return lds_read128<T>(lds_ptr(aq ^ ((kk & 7) << 5)) + (kk >> 3) * kBgQChunk + kFuOffDo); };
        auto fw = [&](int kk) { return lds_read128<T>(lds_ptr(av ^ ((kk & 7) << 5)) + (kk >> 3) * kFuVChunk); };
#pragma unroll
        for (int i = 0; i < kAhead; ++i) { a[i] = fa(i); w[i] = fw(i); }
#pragma unroll
        for (int i = 0; i < kNK; ++i) {
          if (i + kAhead < kNK) { a[i + kAhead] = fa(i + kAhead); w[i + kAhead] = fw(i + kAhead); }
          dp = mfma(a[i], w[i], dp);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * kAhead, 0);
#pragma unroll
        for (int i = 0; i < kNK - kAhead; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
      }
#if !RFA_FU_SCORES_SERIAL
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x4 ls = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + 8 * jj * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[4 * jj + e] = fast_exp2(__builtin_fmaf(s[4 * jj + e], c, -kLog2e * ls[e]));
      }
#endif
      const bool need_mask = (qs0 + 32 > lq) || (hi && qs0 + off + wr < kw0 + 31) || (lo && qs0 + 31 + off - wl > kw0);
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // (bitwise, not short-circuit: with && / || hipcc builds a branch per element — 34 branches per masked tile)
          const int q = qs0 + crow(r, g);
          const bool ok = (q < lq) & (!hi | (krow <= q + off + wr)) & (!lo | (krow >= q + off - wl));
          s[r] = ok ? s[r] : 0.f;
        }
      }
      // pbv = the (dropped, rescaled) P for the dV GEMM; s becomes dS = P (dP - delta) for the dK GEMM
      vec8<T> pbv[2];
      if (drop) {
        const uint32_t hkey = drop_head_key(p.drop_seed, p.cu_q ? 0u : (uint32_t)b, p.head0 + (uint32_t)(h0 + cg));
        f32x16 pv;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x4 dl = *(__attribute__((address_space(3))) f32x4*)(lds_ptr(sa) + 256 + 8 * jj * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * jj + e;
            const uint32_t w = drop_word(hkey, drop_i0 + (uint32_t)(qs0 + crow(r, g)), drop_j >> 2);
            const bool keep = drop_keep(w, (int)(drop_j & 3u), p.drop_keep);
            pv[r] = keep ? s[r] * p.drop_scale : 0.f;
            const float dpd = keep ? (dp[r] + dl[e]) * p.drop_scale - dl[e] : -dl[e];
            s[r] = dpd * s[r];
          }
        }
        pbv[0] = pack8<T>(pv, 0);
        pbv[1] = pack8<T>(pv, 8);
      } else {
        pbv[0] = pack8<T>(s, 0);
        pbv[1] = pack8<T>(s, 8);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= dp[r];
      }
      const vec8<T> pbk[2] = {pack8<T>(s, 0), pack8<T>(s, 8)};
      if (spill) {
        const char* blk = ds_b + (int64_t)(h0 + cg) * ds_head_bytes + (ds_rowpart(p, j, (qs.row0 >> 5) + b, ds_nkb) + ds_kb) * kDsBlockBytes;
        const buf_rsrc_t rb = make_rsrc(blk, kDsBlockBytes);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pbk[0]), rb, ds_lane, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pbk[1]), rb, ds_lane + 128, 0, 2);
      }
      // dV^T += dO^T P, dK^T += Q^T dS: A operands by transpose reads of the dO / Q tile
      big_gemm<T, 2 * kNB, 2>(
          [&](int i) {
            const int dblk = i % kNB, imm = 16 * (i / kNB) * 256 + (dblk >> 2) * kBgQChunk + kFuOffDo;
            return concat<T>(lds_read_tr<T>(lds_ptr(tq[0] ^ ((dblk & 3) << 6)) + imm), lds_read_tr<T>(lds_ptr(tq[1] ^ ((dblk & 3) << 6)) + imm));
          },
          [&](int i, vec8<T> a) { accv[i % kNB] = mfma(a, pbv[i / kNB], accv[i % kNB]); });
      big_gemm<T, 2 * kNB, 2>(
          [&](int i) {
            const int dblk = i % kNB, imm = 16 * (i / kNB) * 256 + (dblk >> 2) * kBgQChunk;
            return concat<T>(lds_read_tr<T>(lds_ptr(tq[0] ^ ((dblk & 3) << 6)) + imm), lds_read_tr<T>(lds_ptr(tq[1] ^ ((dblk & 3) << 6)) + imm));
          },
          [&](int i, vec8<T> a) { acck[i % kNB] = mfma(a, pbk[i / kNB], acck[i % kNB]); });
    } else {
#pragma unroll
      for (int i = 0; i < kPieces; ++i) issue_piece(nxt, i);
    }
    // tile f + 1 must have landed; this tile's two dS spill stores (the youngest operations) may stay in flight
    if (spill && active) wait_vmem64<2>();
    else wait_all_vmem();
    if (++cg >= G) {
      cg = 0;
      j -= nsplit;
    }
    __syncthreads();
  }
  wait_all_vmem();

  if (krow >= lk) return;
  const int64_t orow = ks.row0 + krow;
  auto store = [&](auto whichc, const f32x16 (&acc)[kNB]) {
    constexpr int which = decltype(whichc)::value;     // 0 = dK, 1 = dV
    const float sc_ = which ? 1.f : p.scale;
    const Strides st = which ? p.dv_st : p.dk_st;
    const int64_t eoff = kbatch * st.batch + orow * st.row + (int64_t)hk * st.head + (int64_t)qsplit * p.kv_split_stride;
    if (p.kv_f32) {
      float* ob = (float*)(which ? p.dv : p.dk) + eoff;
#pragma unroll
      for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int d0 = 32 * dblk + 8 * jj + 4 * g;
          if (d0 < p.D) {
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = acc[dblk][4 * jj + e] * sc_;
            *(f32x4*)(ob + d0) = x;
          }
        }
    } else {
      store_rows16<T, false, kNB>((T*)(which ? p.dv : p.dk) + eoff, acc, sc_, g, p.D, true);
    }
  };
  store(std::integral_constant<int, 0>{}, acck);
  store(std::integral_constant<int, 1>{}, accv);
}

// ------------------------------------------------------------------------------------
static inline int big_len(int S, int half) { return half ? (S + 1) / 2 : S; }

template <typename K, typename P>
static int launch_big(K kernel, const P& p, int64_t nblocks, int smem, std::atomic<unsigned long long>& done, hipStream_t stream) {
  if (int rc = opt_in_dynamic_lds((const void*)kernel, smem, done)) return rc;
  if (nblocks <= 0) return kLaunchOk;
  hipLaunchKernelGGL(kernel, dim3((unsigned)nblocks), dim3(kBgThreads), smem, stream, p);
  return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchFailed;
}

// kQ = 3: head dims <= 192 run without the k-steps / column blocks of the all-padding last quarter (25 % of the MFMAs of
// a 256-wide row); the LDS tile stays [rows][256] (its missing chunks are zero-filled by the DMA's range check)
template <typename T, int kQ>
static int launch_fwd_big_t(const FwdParams& p, int64_t n, hipStream_t stream) {
  static std::atomic<unsigned long long> done{0};
  return launch_big(fwd_big_kernel<T, kQ>, p, n, kBgFwdSmem, done, stream);
}
int launch_fwd_big(const FwdParams& p0, int dtype, hipStream_t stream) {
  FwdParams p = p0;
  p.nqblk = (big_len(p.Sq, p.q_half) + kBgRows - 1) / kBgRows;
  const int64_t n = (int64_t)p.nqblk * p.H * p.B;
  if (p.D <= 192) return dtype == 0 ? launch_fwd_big_t<bf16_t, 3>(p, n, stream) : launch_fwd_big_t<f16_t, 3>(p, n, stream);
  return dtype == 0 ? launch_fwd_big_t<bf16_t, 4>(p, n, stream) : launch_fwd_big_t<f16_t, 4>(p, n, stream);
}

template <typename T, int kQ>
static int launch_dq_big_t(const BwdParams& p, int64_t n, hipStream_t stream) {
  static std::atomic<unsigned long long> done{0};
  return launch_big(dq_big_kernel<T, kQ>, p, n, kBgFwdSmem, done, stream);
}
int launch_bwd_dq_big(const BwdParams& p0, int dtype, hipStream_t stream) {
  BwdParams p = p0;
  p.nqblk = (big_len(p.Sq, p.q_half) + kBgRows - 1) / kBgRows;
  const int64_t n = (int64_t)p.nqblk * p.H * p.B;
  if (p.D <= 192) return dtype == 0 ? launch_dq_big_t<bf16_t, 3>(p, n, stream) : launch_dq_big_t<f16_t, 3>(p, n, stream);
  return dtype == 0 ? launch_dq_big_t<bf16_t, 4>(p, n, stream) : launch_dq_big_t<f16_t, 4>(p, n, stream);
}

#ifndef RFA_BG_FUSED2
#define RFA_BG_FUSED2 1      // round 6: dK and dV in one launch with the V rows in LDS (dkdv_fused_big_kernel); 0: one launch per tensor
#endif
template <typename T, int kQ>
static int launch_dkdv_big_t(const BwdParams& p, int64_t n, hipStream_t stream) {
  static std::atomic<unsigned long long> done[3];
#if RFA_BG_FUSED2
  // head dims <= 192 (kQ = 3: 2 x 96 accumulator registers + 48 of K_w): one launch.  At kQ = 4 the two accumulator sets fill
  // the AGPR half exactly, hipcc has no register left to park arch values in and the tile loop goes to scratch (55 scratch
  // operations per tile: measured 293 against 450 TFLOP/s) — D > 192 keeps one launch per tensor.
  if constexpr (kQ == 3) {
    if (p.drop_keep < 256) return launch_big(dkdv_fused_big_kernel<T, kQ, true>, p, n, kFuSmem, done[2], stream);
    static std::atomic<unsigned long long> done_plain{0};
    return launch_big(dkdv_fused_big_kernel<T, kQ, false>, p, n, kFuSmem, done_plain, stream);
  }
#endif
#if RFA_BG_FUSED_KV
  if constexpr (kQ == 3) return launch_big(dkdv_big_kernel<T, 2, kQ>, p, n, kBgKvSmem, done[0], stream);
#endif
  if (int rc = launch_big(dkdv_big_kernel<T, 0, kQ>, p, n, kBgKvSmem, done[0], stream)) return rc;
  return launch_big(dkdv_big_kernel<T, 1, kQ>, p, n, kBgKvSmem, done[1], stream);
}
int launch_bwd_dkdv_big(const BwdParams& p0, int dtype, hipStream_t stream) {
  BwdParams p = p0;
  p.nkblk = (big_len(p.Sk, p.k_half) + kBgRows - 1) / kBgRows;
  if (p.nsplit < 1) p.nsplit = 1;
  const int64_t n = (int64_t)p.nkblk * p.Hk * p.B * p.nsplit;
  if (p.D <= 192) return dtype == 0 ? launch_dkdv_big_t<bf16_t, 3>(p, n, stream) : launch_dkdv_big_t<f16_t, 3>(p, n, stream);
  return dtype == 0 ? launch_dkdv_big_t<bf16_t, 4>(p, n, stream) : launch_dkdv_big_t<f16_t, 4>(p, n, stream);
}

}  // namespace rfa
