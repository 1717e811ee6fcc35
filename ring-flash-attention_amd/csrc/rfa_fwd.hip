// rfa_fwd.hip — flash-attention forward for gfx950 (MI355X), dense + varlen, GQA, causal
// (bottom-right aligned), optional fused online merge into fp32 (out_acc, lse_acc).
//
// Replaces flash_attn._flash_attn_forward / _flash_attn_varlen_forward as called from
// /root/reference/ring_flash_attn/zigzag_ring_flash_attn.py:52 (and siblings), and — in
// accumulate mode — ring_flash_attn/utils.py:32-73 (update_out_and_lse).
//
// Structure (one workgroup = 8 waves = 256 query rows of one head; KV tile = 64 keys):
//   * each wave owns 32 query rows; its Q fragment lives in registers for the whole kernel
//   * K/V tiles go global -> LDS by DMA (buffer_load ... lds; D < 128: through registers), double
//     buffered, one barrier per tile; LDS images are XOR-swizzled (rfa_common.hpp)
//   * S^T = K·Q^T   (A = K rows from LDS via ds_read_b128, B = Q registers): a lane owns ONE
//     query row (column of S^T) -> row max / row sum are in-lane, one cross-half exchange
//   * O^T += V^T·P^T (A = V^T via ds_read_b64_tr_b16, B = P straight from the S^T registers)
//   * epilogue: normalise, write out/lse, or merge into (out_acc, lse_acc) in fp32.
#include <type_traits>

#include "rfa_common.hpp"
#include "rfa_kernels.hpp"

// ---- tuning knobs (A/B'd on hardware with tools/ab_variants.py; defaults = best measured) ----
#ifndef RFA_FWD_DEFER
#define RFA_FWD_DEFER 8      // >0: skip the O/l rescale while the row max grew by <= this many log2 units
                             // (A/B: 0 -> 0.676 ms, 4 -> 0.648 ms, 8 -> 0.635 ms; accuracy unchanged, see the
                             //  'spike keys' self-test cases that force the rescale branch mid-loop)
#endif
#ifndef RFA_FWD_AHEAD
#define RFA_FWD_AHEAD 6      // K fragments read ahead of their MFMA in the S GEMM (round 4 A/B, one box, two passes each:
                             // 2 -> 0.5015 ms, 4 -> 0.4857, 6 -> 0.4835, 8 -> 0.4855)
#endif
#ifndef RFA_FWD_AHEAD_V
#define RFA_FWD_AHEAD_V 3    // persistent forward: V^T fragment pairs read ahead of their MFMA in the P·V GEMM (pinned)
#endif
#ifndef RFA_FWD_LEAN
#define RFA_FWD_LEAN 1       // 1: the deferred-rescale test is ONE compare of the tile's row max against a threshold kept per
                             // row (running max + DEFER / c) and the scaled running max (-m c, the addend of the exponent FMA) is
                             // kept in a register, both updated only where the rescale branch runs: 2 VALU instructions per tile
                             // between the half-wave exchange and the exponentials instead of 7 + a wait state
#endif
#ifndef RFA_FWD_MAXFORM
#define RFA_FWD_MAXFORM 1    // the half-wave exchange of the row max / row sum: 0 ds_bpermute (__shfl_xor: an LDS round trip plus
                             // six address instructions per tile), 1 v_permlane32_swap (rfa_common.hpp: max_xor32).  Same bits.
                             // Round 4 A/B (profiles/history/r04_fwd_variants.txt): 0 -> 0.4880 ms, 1 -> 0.4795.  Also measured there and
                             // NOT kept, all bit-identical: the 31-deep max chain as four independent chains (0.4831: six more
                             // canonicalising v_max), the first sub-tile's chain in the MFMA shadows of the second sub-tile's
                             // S GEMM (0.4846), one barrier per TWO tiles on a 4-stage ring (0.4889), s_setprio 1 around the S
                             // GEMM / the P·V GEMM / both (0.4842 / 0.4849 / 0.4872 against 0.4846), and the row sums as a fifth
                             // MFMA per k-step with an all-ones A operand instead of 32 v_add per tile (36 MFMAs and 113 VALU
                             // per tile instead of 32 and 145: 0.4959 against 0.4857 — the matrix pipe's time is not free); with a
                             // 3-stage ring, the tile's barrier between the S GEMM and the softmax instead of behind P·V, so that
                             // a wave runs from P·V straight into the next S GEMM (0.5328 against 0.5123: the common restart
                             // behind the barrier is worth more than the seam it removes).
#endif
// (Round 4: a software-pipelined tile loop — S of tile j+1 issued inside the softmax of tile j in every wave, two S
//  register sets, the guide's "att[2]" technique — was built here and removed again: it needs 32 more registers than
//  the 256 a wave of this 2-waves-per-SIMD kernel has; hipcc then keeps the Q fragments in scratch and reloads them
//  for every tile: numerically identical, 0.54 -> 1.50 ms.  DESIGN.md section 7.)
#ifndef RFA_FWD_PACKED_VALU
#define RFA_FWD_PACKED_VALU 0   // 1: scale / subtract / row sum as v_pk_fma_f32 / v_pk_add_f32 (22 instructions fewer per
                                // tile pair, but 17 v_mov per tile to pair registers and 235 instead of 218 VGPRs:
                                // A/B on one box 0.607 - 0.615 vs 0.598 - 0.602 ms, i.e. 1.5 % slower)
#endif

namespace rfa {

#ifndef RFA_FWD_X_LOAD
#define RFA_FWD_X_LOAD 1
#endif
// waves per workgroup (32 q rows each) are a template parameter kW of the kernel: 8 (256 query rows, one workgroup per
// CU) is the tuned form; 4 (128 rows, two independent workgroups per CU) is launched when the 8-wave grid would leave
// the chip under-filled — twice the workgroups, half the K/V tile reuse (measured: (B 1, S 2048, 16 heads) 0.050 ->
// 0.040 ms, (B 8, S 1024, 32 heads) 0.117 -> 0.111 ms, but S >= 4096 at 32 heads 3 - 6 % slower; rfa_api.cpp picks)
constexpr int kFwdWavesMax = 8;
#ifndef RFA_FWD_KV
#define RFA_FWD_KV 64        // keys per tile (64 | 128): 128 halves the barriers / DMA waits per MFMA at twice the LDS and 32 more registers
#endif
#ifndef RFA_FWD_YOUNG_PRIO
#define RFA_FWD_YOUNG_PRIO 0 // 1: waves 4-7 (the half dispatched second, the loser of every VALU arbitration) run at s_setprio 1
#endif
constexpr int kFwdKV = RFA_FWD_KV;          // keys per tile
constexpr int kFwdSub = kFwdKV / 32;        // 32-key sub-tiles per tile
#ifndef RFA_FWD_STAGES
#define RFA_FWD_STAGES 2     // LDS ring depth: tile j+STAGES-1 is in flight (DMA) while tile j is computed
#endif
constexpr int kFwdStages = RFA_FWD_STAGES;
template <int kD> constexpr int fwd_smem() { return 2 * kFwdStages * kFwdKV * HeadGeo<kD>::kRowBytes; }   // K[stages] + V[stages]

// kD: compiled head dim (128 or 64: half the MFMAs, half the LDS bytes per tile); kFullD: D == kD (LDS-DMA
// staging, no conditional loads), otherwise D < kD is zero padded through the register staging path
// kDrop: dropout on the probabilities that enter P·V (instances without a window only)
template <typename T, int kD, bool kFullD, bool kWin, bool kDrop = false, int kW = kFwdWavesMax>
__global__ __launch_bounds__(kW * 64, 2) void fwd_kernel(const FwdParams p) {
  constexpr int kFwdWaves = kW, kFwdThreads = kW * 64, kFwdQRows = kW * 32;   // (query rows per workgroup)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  // LDS map: K stages at [0, stages*tile), V stages behind them
  typedef HeadGeo<kD> Geo;
  constexpr int kRowBytes = Geo::kRowBytes;                  // (shadows the 128-wide namespace constant)
  constexpr int kNK = Geo::kKSteps, kNB = Geo::kDBlocks;
  constexpr int kFwdTileBytes = kFwdKV * kRowBytes;          // 16 KiB (8 KiB at kD = 64)
  constexpr int kFwdShare = kFwdTileBytes / 1024 / kFwdWaves;   // 1 KiB DMA pieces (4 physical rows) per wave
  constexpr int kChunks = Geo::kLay / 8;                     // 16-byte chunks per logical (LDS) row

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5;
  const int l31 = lane & 31;

  // ---- work decode: kv-head fastest so that workgroups sharing K/V sit on one XCD ----
  int idx = blockIdx.x;
  const int G = p.H / p.Hk;
  const int hk = idx % p.Hk;
  idx /= p.Hk;
  const int gq = idx % G;
  idx /= G;
  const int nsplit = p.kv_nsplit > 1 ? p.kv_nsplit : 1;      // split-KV launch: which share of the key tiles
  const int split = idx % nsplit;
  idx /= nsplit;
  int qblk_i, b;
  split_block_batch<RFA_BATCH_FAST_Q>(idx, p.nqblk, p.B, qblk_i, b);
  const int qblk = p.nqblk - 1 - qblk_i;            // heavy (late) causal blocks first
  const int h = hk * G + gq;

  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int qwg0 = qblk * kFwdQRows;
  if (qwg0 >= lq) return;
  const int off = lk - lq;                 // bottom-right causal alignment
  const int qw0 = qwg0 + wave * 32;
  const int qrow = qw0 + l31;
  const int qrow_c = qrow < lq ? qrow : lq - 1;

  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;

  const T* qbase = (const T*)p.q + qbatch * p.q_st.batch + (qs.row0 + qrow_c) * p.q_st.row +
                   (int64_t)h * p.q_st.head;
  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row +
                   (int64_t)hk * p.k_st.head;
  const T* vbase = (const T*)p.v + kbatch * p.v_st.batch + ks.row0 * p.v_st.row +
                   (int64_t)hk * p.v_st.head;

  // ---- Q fragment (B operand of S^T = K Q^T): lane (q = l31, g) holds d = 16kk + 8g .. +7
  vec8<T> qf[kNK];
#pragma unroll
  for (int kk = 0; kk < kNK; ++kk) {
    const int d0 = 16 * kk + 8 * g;
    qf[kk] = (kFullD || d0 < p.D) ? *(const vec8<T>*)(qbase + d0) : zero8<T>();
  }

  // ---- KV range of this workgroup
  const int qend = (qwg0 + kFwdQRows < lq) ? qwg0 + kFwdQRows : lq;
  // Attention band: query row i sees keys [i + off - wl, i + off + wr], each bound only if set.  Windowed calls
  // (a left bound, or a right bound without `causal`) run the kWin instances; the default instances keep the plain
  // causal logic (hi = causal, wr = 0, no left bound) so that their hot loops stay free of the extra predicates.
  const bool hi = kWin ? p.wr >= 0 : p.causal != 0;
  const bool lo = kWin && p.wl >= 0;
  const int wr = kWin ? p.wr : 0, wl = kWin ? p.wl : 0;
  int kmax = lk;
  if (hi && qend + off + wr < kmax) kmax = qend + off + wr;
  int ntiles = kmax > 0 ? (kmax + kFwdKV - 1) / kFwdKV : 0;
  int kmin = lo ? qwg0 + off - wl : 0;
  kmin = kmin > 0 ? kmin : 0;
  int jt0 = (kmin / kFwdKV) / kFwdStages * kFwdStages;   // first tile (aligned to the LDS ring: stage = j % stages)
  if (nsplit > 1) {
    // this workgroup's contiguous share of the tiles [jt0, ntiles), a multiple of the ring depth long (stage = j % stages)
    int chunk = (ntiles - jt0 + nsplit - 1) / nsplit;
    chunk = (chunk + kFwdStages - 1) / kFwdStages * kFwdStages;
    jt0 += split * chunk;
    ntiles = jt0 + chunk < ntiles ? jt0 + chunk : ntiles;          // (an empty share: jt0 >= ntiles -> no tile, l = 0)
  }

  // ---- K/V tile staging.  Raw buffer loads: the per-thread byte offsets are fixed for the whole kernel,
  // the tile advance lives in the scalar descriptor, rows past the end of the sequence read as zero.
  // With D == 128 the tile goes global -> LDS directly (buffer_load ... lds, no staging registers and no
  // LDS-write instructions): a wave-instruction fills 64 consecutive 16-byte slots = 4 tile rows, so lane
  // L of the DMA for row group c = wave + kFwdWaves i lands in row 4c + L/16, physical chunk L%16 and must FETCH
  // the logical chunk the swizzle puts there.  D < 128 needs the chunks beyond D zeroed and goes
  // global -> registers -> LDS (issue early / write late).
  constexpr bool kDma = kFullD;
  constexpr int kRowsPerPass = kFwdThreads / kChunks;      // register path: one 16-byte chunk per thread and pass
  const int sc = tid % kChunks;
  const int sr = tid / kChunks;
  const bool sd_ok = (kFullD && kD == Geo::kLay) || sc * 8 < p.D;   // (kD = 96: the layout's chunks 12..15 are not part of the head)
  vec8<T> kreg[kFwdShare], vreg[kFwdShare];
  int voff_k[kFwdShare], voff_v[kFwdShare];
#pragma unroll
  for (int i = 0; i < kFwdShare; ++i) {
    int row = sr + kRowsPerPass * i, chunk = sc;
    if (kDma) dma_lane_src<kD>(wave + kFwdWaves * i, lane, row, chunk);
    voff_k[i] = (row * (int)p.k_st.row + chunk * 8) * 2;
    voff_v[i] = (row * (int)p.v_st.row + chunk * 8) * 2;
  }
  // (round 6: the tile addresses are running 64-bit scalars — load_tile() is called for consecutive tiles jt0, jt0 + 1, ... —
  //  instead of two base + j * 64 * stride products per tile in front of the tile's first fragment read)
  const int64_t k_tile_e = (int64_t)kFwdKV * p.k_st.row, v_tile_e = (int64_t)kFwdKV * p.v_st.row;
  const T* ld_k = kbase + jt0 * k_tile_e;
  const T* ld_v = vbase + jt0 * v_tile_e;
  auto load_tile = [&](int j, auto stage) {
    constexpr int kStage = decltype(stage)::value;
    int rows = lk - j * kFwdKV;
    rows = rows < kFwdKV ? rows : kFwdKV;
    const int nk = rows > 0 ? ((rows - 1) * (int)p.k_st.row + p.D) * 2 : 0;
    const int nv = rows > 0 ? ((rows - 1) * (int)p.v_st.row + p.D) * 2 : 0;
    const T* kt = ld_k;
    const T* vt = ld_v;
    ld_k += k_tile_e;
    ld_v += v_tile_e;
    if (kDma) {
      const dma_rsrc_t rk = make_dma_rsrc(kt, nk), rv = make_dma_rsrc(vt, nv);
#pragma unroll
      for (int i = 0; i < kFwdShare; ++i) {
        const int dst = lds_addr(smem) + kStage * kFwdTileBytes + (wave + kFwdWaves * i) * 1024;
        if (RFA_FWD_X_LOAD) dma_load128(rk, dst, voff_k[i]);                      // (RFA_FWD_X_LOAD: measurement only,
        if (RFA_FWD_X_LOAD == 1) dma_load128(rv, dst + kFwdStages * kFwdTileBytes, voff_v[i]);   //  0 no tile loads, 2 K only)
      }
    } else {
      const buf_rsrc_t rk = make_rsrc(kt, nk), rv = make_rsrc(vt, nv);
#pragma unroll
      for (int i = 0; i < kFwdShare; ++i) {
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
      for (int i = 0; i < kFwdShare; ++i) {
        const int o = tile_off_d<kD>(sr + kRowsPerPass * i, sc);
        lds_write128<T>(smem + kStage * kFwdTileBytes + o, kreg[i]);
        lds_write128<T>(smem + (kFwdStages + kStage) * kFwdTileBytes + o, vreg[i]);
      }
    }
  };

  // ---- per-lane LDS addresses (absolute, finished once; the loop only adds instruction immediates)
  // K fragment (A operand): row = 32t + l31, chunk = 2kk + g
  int koff[kNK];
#pragma unroll
  for (int kk = 0; kk < kNK; ++kk) {
    koff[kk] = lds_addr(smem) + tile_off_d<kD>(l31, 2 * kk + g);   // + t * 32 rows
    pin_vgpr(koff[kk]);
  }
  // V^T fragment via transpose read: rows rb = 32t + 16ks + 8hh + 4g (+ i>>2).  The row bits below the swizzle
  // period (16 rows at kD = 128: hh, g; 32 rows at kD = 64: ks, hh, g) are part of the per-lane address, the
  // rest is an instruction immediate.
  constexpr int kVK = Geo::kSwzRows / 16;                  // k-steps that need their own address
  int voff[kNB][kVK][2];
#pragma unroll
  for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
    for (int kv = 0; kv < kVK; ++kv)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        voff[dblk][kv][hh] = lds_addr(smem) + kFwdStages * kFwdTileBytes + tr_off_d<kD>(lane, dblk, 16 * kv + 8 * hh + 4 * g);   // (incl. the V region's base: the immediates stay below 64 KiB with 3 stages too)
        pin_vgpr(voff[dblk][kv][hh]);
      }

  // dropout: global positions of this lane's query row and of key 0 of the sequence (packed input: absolute rows)
  const uint32_t drop_key = kDrop ? drop_head_key(p.drop_seed, p.cu_q ? 0u : (uint32_t)b, p.head0 + (uint32_t)h) : 0u;
  const uint32_t drop_i = kDrop ? p.q_pos0 + (uint32_t)(p.cu_q ? qs.row0 : 0) + (uint32_t)qrow : 0u;
  const uint32_t drop_j0 = kDrop ? p.k_pos0 + (uint32_t)(p.cu_k ? ks.row0 : 0) : 0u;
  const float c = p.scale * kLog2e;
  float m = -INFINITY;
  float mthr = -INFINITY, mc_run = 0.f;      // RFA_FWD_LEAN: rescale threshold m + DEFER / c, and m c (0 while m is -inf)
  float lsum = 0.f;
  f32x16 o[kNB];
#pragma unroll
  for (int i = 0; i < kNB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

  // Always stage tile 0 (rows past the end read as zero, so this is safe even when ntiles == 0): keeping
  // the prologue unconditional leaves ONE path into the loop, on which every earlier global load
  // (Q fragment included) has provably landed — otherwise hipcc's waitcnt pass merges the
  // "nothing waited yet" path into the loop header and drains the tile prefetch (vmcnt(0))
  // in front of the first MFMA of every tile.
  // The DMA path keeps kFwdStages - 1 tiles in flight; the register path (D < 128) only one.
  constexpr int kDist = kDma ? kFwdStages - 1 : 1;
  load_tile(jt0, std::integral_constant<int, 0>{});
  write_tile(std::integral_constant<int, 0>{});
  wait_all_vmem();          // Q fragment loads too: nothing the compiler tracks may stay pending into the loop
  __syncthreads();
  if (kDist == 2) load_tile(jt0 + 1, std::integral_constant<int, 1>{});   // (rows past the end: descriptor range 0)
#if RFA_FWD_YOUNG_PRIO
  if (wave >= kFwdWaves / 2) __builtin_amdgcn_s_setprio(1);
#endif

  // One KV tile.  The LDS stage is a compile-time constant (the tile loop is unrolled by the ring depth),
  // so every LDS address in here is a per-lane table entry plus an instruction immediate.
  auto tile_step = [&](int j, auto stage) {
    constexpr int kStage = decltype(stage)::value;
    constexpr int kbo = kStage * kFwdTileBytes;                   // K stage
    constexpr int vbo = kStage * kFwdTileBytes;                   // V stage, relative to the V region (whose base is part of voff)
    typedef std::integral_constant<int, (kStage + kDist) % kFwdStages> fill_t;   // stage refilled now
    typedef std::integral_constant<int, (kStage + 1) % kFwdStages> next_t;
    const bool more = j + kDist < ntiles;
    if (more) load_tile(j + kDist, fill_t{});

    const int kt0 = j * kFwdKV;
    const bool active = (qw0 < lq) && !(hi && kt0 > qw0 + 31 + off + wr) &&
                        !(lo && kt0 + kFwdKV - 1 < qw0 + off - wl);
    if (active) {
      // ---------------- S^T = K Q^T ----------------
      f32x16 s[kFwdSub];
#pragma unroll
      for (int t = 0; t < kFwdSub; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
      {
        // kFwdSub kNK K fragments (sub-tiles x kNK k-steps), read kAhead ahead of their MFMA
        constexpr int kAhead = RFA_FWD_AHEAD;
        constexpr int kN = kFwdSub * kNK;
        vec8<T> a[kN];
        auto fa = [&](int i) { return lds_read128<T>(lds_ptr(koff[i % kNK]) + kbo + (i / kNK) * 32 * kRowBytes); };
#pragma unroll
        for (int i = 0; i < kAhead; ++i) a[i] = fa(i);
#pragma unroll
        for (int i = 0; i < kN; ++i) {
          if (i + kAhead < kN) a[i + kAhead] = fa(i + kAhead);
          s[i / kNK] = mfma(a[i], qf[i % kNK], s[i / kNK]);
        }
        // pin the issue order: kAhead reads, then read/MFMA pairs, then the MFMA tail
        __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
        for (int i = 0; i < kN - kAhead; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
      }
      // ---------------- mask ----------------
      const bool need_mask = (kt0 + kFwdKV > lk) || (hi && kt0 + kFwdKV - 1 > qw0 + off + wr) ||
                             (lo && kt0 < qw0 + 31 + off - wl);
      if (need_mask) {
        const int lim = hi ? ((qrow + off + wr < lk - 1) ? qrow + off + wr : lk - 1) : lk - 1;
        const int lim_lo = lo ? qrow + off - wl : -0x40000000;
#pragma unroll
        for (int t = 0; t < kFwdSub; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt0 + 32 * t + crow(r, g);
            if (key > lim || (kWin && key < lim_lo)) s[t][r] = -INFINITY;
          }
      }
      // ---------------- online softmax ----------------
      float mloc = s[0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[0][r]);
#pragma unroll
      for (int t = 1; t < kFwdSub; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[t][r]);
      mloc = RFA_FWD_MAXFORM ? max_xor32(mloc) : fmaxf(mloc, shfl_xor32(mloc));
      // deferred rescale (RFA_FWD_DEFER > 0): while no row of the wave grew its max by more than
      // DEFER log2 units, keep the stale max — P is then bounded by 2^DEFER instead of 1, still
      // exact in fp32 / same relative precision in bf16 — and skip the 64-register O rescale.
      // RFA_FWD_LEAN: "grew by more than DEFER" is mloc > mthr with mthr = m + DEFER / c (-inf while m is -inf: a row
      // without a visible key so far has nothing to rescale), and mc = m c is carried instead of recomputed per tile.
      bool rescale = true;
      if (RFA_FWD_LEAN) {
        rescale = !__all(mloc <= mthr);
      } else {
        const float mnew0 = fmaxf(m, mloc);
        if (RFA_FWD_DEFER > 0) rescale = !__all((mnew0 - m) * c <= (float)RFA_FWD_DEFER);
      }
      if (rescale) {
        const float mnew = fmaxf(m, mloc);
        const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = fast_exp2(m * c - msafe * c);
        m = mnew;
        mc_run = msafe * c;
        mthr = mnew + (RFA_FWD_DEFER > 0 ? (float)RFA_FWD_DEFER / c : 0.f);      // (-inf stays -inf)
        lsum *= alpha;
#pragma unroll
        for (int i = 0; i < kNB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      }
      const float mc = RFA_FWD_LEAN ? mc_run : ((m == -INFINITY) ? 0.f : m) * c;
#if RFA_FWD_PACKED_VALU
      // scale / subtract and the row sum as whole-vector expressions: hipcc turns them into v_pk_fma_f32 / v_pk_add_f32
      // (two fp32 per lane and instruction) — 32 VALU instructions fewer per tile than the element-wise form
      f32x16 psum16;
#pragma unroll
      for (int t = 0; t < kFwdSub; ++t) {
        s[t] = s[t] * c - mc;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = fast_exp2(s[t][r]);
        psum16 = t ? psum16 + s[t] : s[t];
      }
      f32x4 ps4;
#pragma unroll
      for (int e = 0; e < 4; ++e) ps4[e] = (psum16[e] + psum16[4 + e]) + (psum16[8 + e] + psum16[12 + e]);
      lsum += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
#else
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < kFwdSub; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(__builtin_fmaf(s[t][r], c, -mc));
          s[t][r] = pv;
          psum += pv;
        }
      lsum += psum;
#endif
      if (kDrop) {
        // the row sum (and lse) are those of the undropped softmax; only what enters P·V is masked.  A lane holds,
        // per (t, mm), 4 consecutive keys of its row: one mask word, or the bytes of two when the sequence's
        // first key position is not a multiple of 4 (wave-uniform)
        const int mis = __builtin_amdgcn_readfirstlane((int)(drop_j0 & 3u));
#pragma unroll
        for (int t = 0; t < kFwdSub; ++t)
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

      // ---------------- O^T += V^T P^T ----------------
#pragma unroll
      for (int t = 0; t < kFwdSub; ++t)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          const vec8<T> pb = pack8<T>(s[t], 8 * ks2);
          constexpr int kPer = Geo::kSwzRows;                       // rows covered by the address table
          const int rows = 32 * t + 16 * ks2;
          const int imm = vbo + (rows / kPer) * kPer * kRowBytes;
          const int kv = (rows % kPer) / 16;
#pragma unroll
          for (int dblk = 0; dblk < kNB; ++dblk) {
            vec4<T> lo = lds_read_tr<T>(lds_ptr(voff[dblk][kv][0]) + imm);
            vec4<T> hi = lds_read_tr<T>(lds_ptr(voff[dblk][kv][1]) + imm);
            o[dblk] = mfma(concat<T>(lo, hi), pb, o[dblk]);
          }
        }
    }
    if (!kDma && more) write_tile(next_t{});
    if (kDma) {
      // tile j+1 must have landed before the barrier; the 4 DMA instructions of tile j+2 (if issued) may fly on
      if (kDist == 2 && more) wait_vmem<2 * kFwdShare>();
      else wait_all_vmem();
    }
    __syncthreads();
  };
  if (kFwdStages == 3) {
    for (int j = jt0; j < ntiles; j += 3) {
      tile_step(j, std::integral_constant<int, 0>{});
      if (j + 1 < ntiles) tile_step(j + 1, std::integral_constant<int, 1>{});
      if (j + 2 < ntiles) tile_step(j + 2, std::integral_constant<int, 2 % kFwdStages>{});
    }
  } else {
    for (int j = jt0; j < ntiles; j += 2) {
      tile_step(j, std::integral_constant<int, 0>{});
      if (j + 1 < ntiles) tile_step(j + 1, std::integral_constant<int, 1>{});
    }
  }

  // ---------------- epilogue ----------------
  if (qrow >= lq) return;
  const float l = RFA_FWD_MAXFORM ? sum_xor32(lsum) : lsum + shfl_xor32(lsum);
  const bool has = l > 0.f;
  const float inv = has ? (kDrop ? p.drop_scale : 1.f) / l : 0.f;      // kept probabilities are scaled by 1 / (1 - p)
  const float blse = has ? m * p.scale + __logf(l) : INFINITY;   // natural log
  const int64_t orow = qs.row0 + qrow;

  if (p.out_acc == nullptr && nsplit == 1) {
    T* ob = (T*)p.out + qbatch * p.out_st.batch + orow * p.out_st.row + (int64_t)h * p.out_st.head;
    store_rows16<T, kFullD, kNB>(ob, o, inv, g, p.D, true);
    if (g == 0) p.lse[qbatch * p.lse_batch + (int64_t)h * p.lse_head + orow] = blse;
  } else {
    // split-KV launch: the normalised partial of this split goes to its slot of the workspace (laid out like
    // out_acc / lse_acc, rfa_api.cpp), overwriting; combine_kernel merges the slots into the call's outputs
    float* ab = (nsplit > 1 ? p.part_out + (int64_t)split * p.part_out_split : p.out_acc) +
                qbatch * p.out_acc_st.batch + orow * p.out_acc_st.row + (int64_t)h * p.out_acc_st.head;
    float* lp = (nsplit > 1 ? p.part_lse + (int64_t)split * p.part_lse_split : p.lse_acc) +
                qbatch * p.lse_acc_batch + (int64_t)h * p.lse_acc_head + orow;
    if (p.acc_init || nsplit > 1) {
#pragma unroll
      for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int d0 = 32 * dblk + 8 * jj + 4 * g;
          if (kFullD || d0 < p.D) {
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = o[dblk][4 * jj + e] * inv;
            *(f32x4*)(ab + d0) = x;
          }
        }
      if (g == 0) *lp = has ? blse : -INFINITY;
    } else if (has) {
      // merge (out_old, lse_old) with (O/l, blse):  lse' = logaddexp, weights exp(x - lse')
      const float lold = *lp;
      const float mx = fmaxf(lold, blse);
      const float eo = __expf(lold - mx);      // lold = -inf -> 0
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
          if (kFullD || d0 < p.D) {
            f32x4 x = *(f32x4*)(ab + d0);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = x[e] * wo + o[dblk][4 * jj + e] * wb;
            *(f32x4*)(ab + d0) = x;
          }
        }
      if (g == 0) *lp = lnew;   // both half-lanes read *lp earlier in program order
    }
  }
}

// =====================================================================================
// Persistent forward (round 6, second session): head dim 128, dense input, plain outputs
// =====================================================================================
// A 256-row workgroup fills its CU (8 waves, 64 KiB of LDS, two waves per SIMD), so nothing overlaps the ~10 us every
// workgroup spends outside its tile loop: dispatch, the Q fragment loads and the first K/V tile (an HBM burst of ~100 KB per
// CU), the epilogue's stores.  The headline launch has 1024 workgroups = FOUR such rounds per CU (the launch-plan estimate
// prices a round at 8 tile-times; S = 32768 amortises it over 4 x the tiles and runs 6 % faster per FLOP).  Here the grid is
// one workgroup per CU and each walks its share of the (batch, head, query block) items; the NEXT item's first K/V tile is
// fetched by the LDS-DMA of the current item's last tile step, its Q fragments are loaded in front of the epilogue (into the
// registers the last S GEMM has read), and the epilogue's stores stay in flight across the seam (a counted vmcnt: loads
// retire in order, the 9 stores are the youngest operations).  Items are dealt in passes over the launch's
// heaviest-first order, every other pass reversed, so that each workgroup's items add up to the same number of tiles (a causal
// launch: exactly, when the passes are even).  Same arithmetic in the same order per query row as fwd_kernel: bit-identical.
template <typename T>
__global__ __launch_bounds__(kFwdWavesMax * 64, 2) void fwd_persist_kernel(const FwdParams p) {
  constexpr int kD = 128, kW = kFwdWavesMax, kRows = kW * 32;
  typedef HeadGeo<kD> Geo;
  constexpr int kRowBytes = Geo::kRowBytes;
  constexpr int kNK = Geo::kKSteps, kNB = Geo::kDBlocks;
  constexpr int kTileBytes = kFwdKV * kRowBytes;            // 16 KiB
  constexpr int kShare = kTileBytes / 1024 / kW;            // 1 KiB DMA pieces per wave and tensor
  static_assert(kFwdStages == 2 && kFwdKV == 64, "the persistent forward: two LDS stages of 64 keys");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5;
  const int l31 = lane & 31;
  const int G = p.H / p.Hk;

  // dense input: every (batch, head) has the same spans
  const SeqSpan qs = resolve_span(nullptr, 0, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(nullptr, 0, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int off = lk - lq;                                   // >= 0 (rfa_api.cpp): every query block sees at least one tile
  const bool hi = p.causal != 0;

  // ---- per-lane constants
  int voff_k[kShare], voff_v[kShare];
#pragma unroll
  for (int i = 0; i < kShare; ++i) {
    int row, chunk;
    dma_lane_src<kD>(wave + kW * i, lane, row, chunk);
    voff_k[i] = (row * (int)p.k_st.row + chunk * 8) * 2;
    voff_v[i] = (row * (int)p.v_st.row + chunk * 8) * 2;
  }
  int koff[kNK];
#pragma unroll
  for (int kk = 0; kk < kNK; ++kk) {
    koff[kk] = lds_addr(smem) + tile_off_d<kD>(l31, 2 * kk + g);
    pin_vgpr(koff[kk]);
  }
  int voff[kNB][2];
#pragma unroll
  for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      voff[dblk][hh] = lds_addr(smem) + kFwdStages * kTileBytes + tr_off_d<kD>(lane, dblk, 8 * hh + 4 * g);
      pin_vgpr(voff[dblk][hh]);
    }
  // (per-item lane values — the Q load offset, the store offsets of the epilogue — are formed from an opaque copy of the lane id
  //  where they are used: hoisted out of the item loop they would be live across the tile loop, whose 218 registers leave
  //  hipcc just enough room to read the V fragments ahead of their MFMAs)
  auto opaque_lane = [&]() {
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    return l;
  };
  const float c = p.scale * kLog2e;

  // ---- items: (kv head fastest, query head in group, query block heaviest first, batch) as fwd_kernel numbers its workgroups
  const int nitems = p.nqblk * p.H * p.B;
  const int grid = gridDim.x, bx = blockIdx.x;
  struct Item {
    int b, h, hk, qwg0, ntiles;
  };
  auto item_at = [&](int pass, Item& it) -> bool {
    int idx = pass * grid + ((pass & 1) ? grid - 1 - bx : bx);
    if (idx >= nitems) return false;
    it.hk = idx % p.Hk;
    idx /= p.Hk;
    const int gq = idx % G;
    idx /= G;
    int qblk_i;
    split_block_batch<RFA_BATCH_FAST_Q>(idx, p.nqblk, p.B, qblk_i, it.b);
    it.h = it.hk * G + gq;
    it.qwg0 = (p.nqblk - 1 - qblk_i) * kRows;
    const int qend = (it.qwg0 + kRows < lq) ? it.qwg0 + kRows : lq;
    int kmax = lk;
    if (hi && qend + off < kmax) kmax = qend + off;
    it.ntiles = (kmax + kFwdKV - 1) / kFwdKV;
    return true;
  };
  auto kv_base = [&](const Item& it, const T*& kb, const T*& vb) {
    kb = (const T*)p.k + (int64_t)it.b * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)it.hk * p.k_st.head;
    vb = (const T*)p.v + (int64_t)it.b * p.v_st.batch + ks.row0 * p.v_st.row + (int64_t)it.hk * p.v_st.head;
  };
  // a K/V tile (`rows` valid rows at kt / vt; rows past the end of the sequence read as zero) -> LDS stage `stage`
  const int64_t k_tile_e = (int64_t)kFwdKV * p.k_st.row, v_tile_e = (int64_t)kFwdKV * p.v_st.row;
  auto dma_rows = [&](const T* kt, const T* vt, int rows, int stage) {
    const int nk = rows > 0 ? ((rows - 1) * (int)p.k_st.row + kD) * 2 : 0;
    const int nv = rows > 0 ? ((rows - 1) * (int)p.v_st.row + kD) * 2 : 0;
    const dma_rsrc_t rk = make_dma_rsrc(kt, nk);
    const dma_rsrc_t rv = make_dma_rsrc(vt, nv);
#pragma unroll
    for (int i = 0; i < kShare; ++i) {
      const int dst = lds_addr(smem) + stage * kTileBytes + (wave + kW * i) * 1024;
      dma_load128(rk, dst, voff_k[i]);
      dma_load128(rv, dst + kFwdStages * kTileBytes, voff_v[i]);
    }
  };
  // this wave's 32 Q rows of an item as B-operand fragments: lane (q = l31, g) holds d = 16 kk + 8 g .. +7; rows past the
  // end of the sequence are outside the descriptor and read as zero (their results are never stored)
  vec8<T> qf[kNK];
  auto load_q = [&](const Item& it) {
    const int qw0 = it.qwg0 + wave * 32;
    int rows = lq - qw0;
    rows = rows < 32 ? rows : 32;
    const T* qb = (const T*)p.q + (int64_t)it.b * p.q_st.batch + (qs.row0 + qw0) * p.q_st.row + (int64_t)it.h * p.q_st.head;
    const buf_rsrc_t rq = make_rsrc(rows > 0 ? qb : (const T*)p.q, rows > 0 ? ((rows - 1) * (int)p.q_st.row + kD) * 2 : 0);
    const int ll = opaque_lane();
    const int voff_q = ((ll & 31) * (int)p.q_st.row + 8 * (ll >> 5)) * 2;    // this lane's first 16 bytes inside its wave's 32 Q rows
#pragma unroll
    for (int kk = 0; kk < kNK; ++kk) qf[kk] = buffer_load128<T>(rq, voff_q + 32 * kk);
  };

  Item cur;
  if (!item_at(0, cur)) return;
  const T *kb, *vb;
  kv_base(cur, kb, vb);
  load_q(cur);
  dma_rows(kb, vb, lk < kFwdKV ? lk : kFwdKV, 0);
  wait_all_vmem();
  __syncthreads();
  int s0 = 0;                                                // LDS stage that holds the current item's tile 0

  for (int pass = 0;; ++pass) {
    Item nxt;
    const bool has_next = item_at(pass + 1, nxt);
    const T *nkb = kb, *nvb = vb;
    if (has_next) kv_base(nxt, nkb, nvb);
    const int ntiles = cur.ntiles;
    const T *run_k = kb + k_tile_e, *run_v = vb + v_tile_e;   // the tile the next step fetches (tile 1 of the current item)
    const int qw0 = cur.qwg0 + wave * 32;
    const int qrow = qw0 + l31;

    float m = -INFINITY, mthr = -INFINITY, mc_run = 0.f, lsum = 0.f;
    f32x16 o[kNB];
#pragma unroll
    for (int i = 0; i < kNB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

    auto tile_step = [&](int j, auto stage) {
      constexpr int kStage = decltype(stage)::value;
      constexpr int kbo = kStage * kTileBytes, vbo = kStage * kTileBytes;
      const bool last = j + 1 >= ntiles;
      if (!last || has_next) {                                       // tile j + 1, or the NEXT item's first tile
        const int jn = last ? 0 : j + 1;
        int rows = lk - jn * kFwdKV;
        rows = rows < kFwdKV ? rows : kFwdKV;
        dma_rows(last ? nkb : run_k, last ? nvb : run_v, rows, kStage ^ 1);
        run_k += k_tile_e;
        run_v += v_tile_e;
      }
      const int kt0 = j * kFwdKV;
      const bool active = (qw0 < lq) && !(hi && kt0 > qw0 + 31 + off);
      if (active) {
        f32x16 s[kFwdSub];
#pragma unroll
        for (int t = 0; t < kFwdSub; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
        {
        constexpr int kAhead = RFA_FWD_AHEAD;
        constexpr int kN = kFwdSub * kNK;
        vec8<T> a[kN];
        auto fa = [&](int i) { return lds_read128<T>(lds_ptr(koff[i % kNK]) + kbo + (i / kNK) * 32 * kRowBytes); };
#pragma unroll
        for (int i = 0; i < kAhead; ++i) a[i] = fa(i);
#pragma unroll
        for (int i = 0; i < kN; ++i) {
          if (i + kAhead < kN) a[i + kAhead] = fa(i + kAhead);
          s[i / kNK] = mfma(a[i], qf[i % kNK], s[i / kNK]);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
        for (int i = 0; i < kN - kAhead; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
        }
        const bool need_mask = (kt0 + kFwdKV > lk) || (hi && kt0 + kFwdKV - 1 > qw0 + off);
        if (need_mask) {
          const int lim = hi ? ((qrow + off < lk - 1) ? qrow + off : lk - 1) : lk - 1;
#pragma unroll
          for (int t = 0; t < kFwdSub; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kt0 + 32 * t + crow(r, g);
              if (key > lim) s[t][r] = -INFINITY;
            }
        }
        float mloc = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[0][r]);
#pragma unroll
        for (int t = 1; t < kFwdSub; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[t][r]);
        mloc = max_xor32(mloc);
        if (!__all(mloc <= mthr)) {                          // deferred rescale, as fwd_kernel (RFA_FWD_LEAN)
          const float mnew = fmaxf(m, mloc);
          const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
          const float alpha = fast_exp2(m * c - msafe * c);
          m = mnew;
          mc_run = msafe * c;
          mthr = mnew + (RFA_FWD_DEFER > 0 ? (float)RFA_FWD_DEFER / c : 0.f);
          lsum *= alpha;
#pragma unroll
          for (int i = 0; i < kNB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        const float mc = mc_run;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < kFwdSub; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pv = fast_exp2(__builtin_fmaf(s[t][r], c, -mc));
            s[t][r] = pv;
            psum += pv;
          }
        lsum += psum;
        // O^T += V^T P^T: the V^T fragments read kAheadV MFMAs ahead, the interleave pinned (left to itself hipcc issues each
        // pair of transpose reads right in front of its MFMA in this kernel — an lgkmcnt(0) per MFMA)
        {
          vec8<T> pb[2 * kFwdSub];
#pragma unroll
          for (int x = 0; x < 2 * kFwdSub; ++x) pb[x] = pack8<T>(s[x >> 1], 8 * (x & 1));
          constexpr int kNV = 2 * kFwdSub * kNB;               // [t][ks2][dblk]
          constexpr int kAheadV = RFA_FWD_AHEAD_V;
          vec8<T> va[kNV];
          auto fv = [&](int i) {
            const int dblk = i % kNB, x = i / kNB;
            const int imm = vbo + 16 * x * kRowBytes;          // rows 32 t + 16 ks2 = 16 x
            return concat<T>(lds_read_tr<T>(lds_ptr(voff[dblk][0]) + imm), lds_read_tr<T>(lds_ptr(voff[dblk][1]) + imm));
          };
#pragma unroll
          for (int i = 0; i < kAheadV; ++i) va[i] = fv(i);
#pragma unroll
          for (int i = 0; i < kNV; ++i) {
            if (i + kAheadV < kNV) va[i + kAheadV] = fv(i + kAheadV);
            o[i % kNB] = mfma(va[i], pb[i / kNB], o[i % kNB]);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * kAheadV, 1);
#pragma unroll
          for (int i = 0; i < kNV - kAheadV; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, kAheadV, 1);
        }
      }
      if (!last) {
        wait_all_vmem();                                     // tile j + 1 has landed before the barrier publishes it
        __syncthreads();
      }
    };
    int j = 0;
    if (s0) {
      tile_step(0, std::integral_constant<int, 1>{});
      j = 1;
    }
    for (; j < ntiles; j += 2) {
      tile_step(j, std::integral_constant<int, 0>{});
      if (j + 1 < ntiles) tile_step(j + 1, std::integral_constant<int, 1>{});
    }

    // the next item's Q fragments, into the registers the last S GEMM has read for the last time: in flight under the
    // epilogue.  (NOT inside the last tile step: a load the compiler tracks, issued inside the tile loop, is 'pending' on the
    // loop's back edge as far as hipcc's waitcnt pass can tell, and it drains the DMA prefetch with a vmcnt(0) in front of
    // the S GEMM of every tile.)
    if (has_next) load_q(nxt);
    // ---- epilogue: normalise, out and lse through range-checked buffer stores (rows past the end of the sequence fall
    // outside the descriptors): exactly 9 store instructions per wave whatever the rows' validity — the seam's wait counts them
    {
      const float l = sum_xor32(lsum);
      const bool has = l > 0.f;
      const float inv = has ? 1.f / l : 0.f;
      const float blse = has ? m * p.scale + __logf(l) : INFINITY;
      int rows = lq - qw0;
      rows = rows < 32 ? (rows > 0 ? rows : 0) : 32;
      T* ob = (T*)p.out + (int64_t)cur.b * p.out_st.batch + (qs.row0 + qw0) * p.out_st.row + (int64_t)cur.h * p.out_st.head;
      const buf_rsrc_t ro = make_rsrc(rows > 0 ? ob : (T*)p.out, rows > 0 ? ((rows - 1) * (int)p.out_st.row + kD) * 2 : 0);
      const int ll = opaque_lane();
      const int vo = ((ll & 31) * (int)p.out_st.row + 8 * (ll >> 5)) * 2;
#pragma unroll
      for (int dblk = 0; dblk < kNB; ++dblk)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          f32x4 x0, x1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x0[e] = o[dblk][8 * mm + e] * inv;
            x1[e] = o[dblk][8 * mm + 4 + e] * inv;
          }
          const vec4<T> h0 = __builtin_convertvector(x0, vec4<T>);
          const vec4<T> h1 = __builtin_convertvector(x1, vec4<T>);
          i32x2 a_ = __builtin_bit_cast(i32x2, h0), b_ = __builtin_bit_cast(i32x2, h1);
          auto r0 = __builtin_amdgcn_permlane32_swap(a_[0], b_[0], false, false);
          auto r1 = __builtin_amdgcn_permlane32_swap(a_[1], b_[1], false, false);
          u32x4 w;
          w[0] = r0[0]; w[1] = r1[0]; w[2] = r0[1]; w[3] = r1[1];
          __builtin_amdgcn_raw_buffer_store_b128(w, ro, vo + (32 * dblk + 16 * mm) * 2, 0, 0);
        }
      float* lb = p.lse + (int64_t)cur.b * p.lse_batch + (int64_t)cur.h * p.lse_head + qs.row0 + qw0;
      const buf_rsrc_t rl = make_rsrc(rows > 0 ? lb : p.lse, rows * 4);
      // (lanes of the upper half-wave hold the same value: they store it too, to the same address — no exec mask, one instruction)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, blse), rl, (ll & 31) * 4, 0, 0);
    }
    if (!has_next) break;
    // the seam: the next item's tile 0 (DMA) and Q fragments are older than the 9 stores above
    wait_vmem<9>();
    // (the fragments re-defined by an empty asm: whatever wait hipcc itself wants for them it places HERE, once per item —
    //  without it the S GEMM of every tile carries a vmcnt in front of each MFMA)
#pragma unroll
    for (int kk = 0; kk < kNK; ++kk) asm volatile("" : "+v"(qf[kk]));
    __syncthreads();
    s0 = (s0 + ntiles) & 1;
    cur = nxt;
    kb = nkb;
    vb = nvb;
  }
}

template <typename T>
static int launch_fwd_persist(const FwdParams& p, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  if (int rc = opt_in_dynamic_lds((const void*)fwd_persist_kernel<T>, fwd_smem<128>(), attr_done)) return rc;
  const int64_t nitems = (int64_t)p.nqblk * p.H * p.B;
  if (nitems <= 0) return 0;
#ifdef RFA_PERSIST_ALL                                          // (measurement: one item per workgroup — the tile loop without seams)
  const int64_t grid = nitems;
#else
  const int64_t grid = nitems < p.persist_grid ? nitems : p.persist_grid;
#endif
  hipLaunchKernelGGL((fwd_persist_kernel<T>), dim3((unsigned)grid), dim3(kFwdWavesMax * 64), fwd_smem<128>(), stream, p);
  return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchFailed;
}

template <typename T, int kD, bool kFullD, bool kWin, bool kDrop = false, int kW = kFwdWavesMax>
static int launch_fwd_w(const FwdParams& p, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  if (int rc = opt_in_dynamic_lds((const void*)fwd_kernel<T, kD, kFullD, kWin, kDrop, kW>, fwd_smem<kD>(), attr_done)) return rc;
  const int64_t nblocks = (int64_t)p.nqblk * p.H * p.B * (p.kv_nsplit > 1 ? p.kv_nsplit : 1);
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL((fwd_kernel<T, kD, kFullD, kWin, kDrop, kW>), dim3((unsigned)nblocks), dim3(kW * 64), fwd_smem<kD>(), stream, p);
  return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchFailed;
}
template <typename T, int kD, bool kFullD, bool kWin, bool kDrop = false>
static int launch_fwd_t(const FwdParams& p, hipStream_t stream) {
  // the 128-row form exists for the LDS-DMA instances without window / dropout (rfa_api.cpp: fwd_rows_for)
  if constexpr (kFullD && !kWin && !kDrop) {
    if (p.qrows == 128) return launch_fwd_w<T, kD, kFullD, kWin, kDrop, 4>(p, stream);
  }
  return launch_fwd_w<T, kD, kFullD, kWin, kDrop, kFwdWavesMax>(p, stream);
}

template <typename T, bool kWin, bool kDrop>
static int launch_fwd_d(const FwdParams& p, hipStream_t stream) {
  if (p.D == 128) return launch_fwd_t<T, 128, true, kWin, kDrop>(p, stream);
  if constexpr (!kWin && !kDrop) {          // 64 < D <= 96: the 128-wide layout with three quarters of the MFMA work
    if (p.D == 96) return launch_fwd_t<T, 96, true, kWin, kDrop>(p, stream);
    if (p.D > 64 && p.D < 96) return launch_fwd_t<T, 96, false, kWin, kDrop>(p, stream);
  }
  if (p.D > 64) return launch_fwd_t<T, 128, false, kWin, kDrop>(p, stream);
  if (p.D == 64) return launch_fwd_t<T, 64, true, kWin, kDrop>(p, stream);
  return launch_fwd_t<T, 64, false, kWin, kDrop>(p, stream);
}

int launch_fwd(const FwdParams& p, int dtype, hipStream_t stream) {
  if (p.persist_grid > 0)                                      // rfa_api.cpp: head dim 128, dense, plain outputs, 256-row form, no shares
    return dtype == 0 ? launch_fwd_persist<bf16_t>(p, stream) : launch_fwd_persist<f16_t>(p, stream);
  if (p.drop_keep < 256)                                       // (rfa_api.cpp rejects dropout together with a window)
    return dtype == 0 ? launch_fwd_d<bf16_t, false, true>(p, stream) : launch_fwd_d<f16_t, false, true>(p, stream);
  if (windowed(p.causal, p.wl, p.wr))
    return dtype == 0 ? launch_fwd_d<bf16_t, true, false>(p, stream) : launch_fwd_d<f16_t, true, false>(p, stream);
  return dtype == 0 ? launch_fwd_d<bf16_t, false, false>(p, stream) : launch_fwd_d<f16_t, false, false>(p, stream);
}

int fwd_qrows_per_block() { return kFwdWavesMax * 32; }

}  // namespace rfa
