// rfa_dqs.hip — dQ from spilled dS (gfx950): the second half of the 5-GEMM backward.
//
// The split-by-output backward (rfa_bwd.hip) pays 7 GEMM-units for 5 because dq_kernel recomputes S and
// dP.  Atomics cannot fix that on this chip: the dQ partials of a 128-key workgroup are 4.3 GB of fp32
// atomic adds per headline backward and gfx950 retires 1.26 TB/s of them (tools/atomic_probe.hip,
// profiles/history/r02_atomic_probe.txt) — 3.4 ms against a 1.6 ms backward.  What the chip does have is HBM:
// dkdv_kernel already holds dS = P∘(dP−Δ) of every (32 query x 32 key) block as two packed 16-byte MFMA
// operands per lane, so it stores them (2 KB per block, 16-byte stores, 2.1 GB bf16 for the headline
// causal S = 8192) and this kernel streams them back exactly once:
//        dQ[q, :] = scale · Σ_keys dS[q, key] · K[key, :]          (one GEMM, HBM-bound)
// instead of re-deriving them with two more GEMMs + exp.  Deterministic (no atomics), same rounding of dS
// to the io dtype that the dK GEMM uses.
//
// Scratch layout (written by dkdv_kernel<kSpill>, read here): blocks of 2048 bytes indexed
//        [batch or packed sequence][q head][qt = query row / 32][kb = key / 32]
// (row / key counted inside the sequence — or inside its selected half; the qt / kb extents are those of the
//  longest (half) sequence of the call, ds_blocks() in rfa_kernels.hpp, so packed input needs no offset table;
//  dense causal calls pack the rows triangularly — row qt keeps only the key blocks a causal query block can
//  see, ds_row_off() / ds_row_len() in rfa_kernels.hpp — which halves the scratch of a square launch)
// holding dS[32 q][32 keys] as 128 slots of 16 bytes; slot p = 16·(key>>2) + 8·i + 4·g + (key&3) contains
// (for key = 0..31 of the block, i = 0/1, g = 0/1) the 8 values q = 16i + 4g + {0..3}, 16i + 8 + 4g + {0..3}
// of that key.  This is the image that makes the transposed LDS read below conflict-free after a
// lane-linear (1 KB contiguous per instruction) DMA.
//
// Structure: one workgroup = 8 waves = 256 query rows of one head; a wave owns 32 rows (its dQ^T
// accumulators: lane = query row).  Per 64-key tile: the K tile (16 KB, shared, XOR-swizzled like every
// tile in this library) and each wave's own 4 KB of dS go global -> LDS by DMA through a 3-stage ring
// (two tiles in flight: the kernel is bandwidth-bound, 48 KB per tile and workgroup);
//        dQ^T[d, q] += K^T[d, keys] · dS^T[keys, q]
// with both operands fetched by ds_read_b64_tr_b16 (k index = natural key order).
#include <type_traits>

#include "rfa_common.hpp"
#include "rfa_kernels.hpp"

#ifndef RFA_DQS_NT
#define RFA_DQS_NT 1         // 1: the dS stream is fetched with the non-temporal policy (0.41 -> 0.35 ms)
#endif
#ifndef RFA_DQS_STAGES4
#define RFA_DQS_STAGES4 4    // ring depth of the 4-wave (128-row) form: 4 = three tiles in flight (128 KiB), 5 = four (160 KiB)
#endif

namespace rfa {

constexpr int kDsKV = 64;                              // keys per tile = 2 dS blocks per wave
constexpr int kDsKBytes = kDsKV * kRowBytes;           // 16 KiB K tile
constexpr int kDsWaveBytes = 2 * kDsBlockBytes;        // 4 KiB of dS per wave and tile
// Two forms: 8 waves = 256 query rows per workgroup through a 3-stage ring (144 KiB: the headline's grid of 1024
// workgroups), and 4 waves = 128 rows through a 4-stage ring (128 KiB, three tiles in flight) for grids that would leave
// half of the CUs without a workgroup — the stream is HBM-bound, what counts is the bytes in flight over the whole chip
// (a llama3 head group at 2048 tokens per rank: 128 workgroups of 256 rows -> 256 of 128 rows)
template <int kW> struct DsGeo {
  static constexpr int kThreads = kW * 64;
  static constexpr int kRows = kW * 32;                // query rows per workgroup
  static constexpr int kStages = kW == 8 ? 3 : RFA_DQS_STAGES4;
  static constexpr int kSBytes = kW * kDsWaveBytes;    // 32 / 16 KiB of dS per tile
  static constexpr int kSmem = kStages * (kDsKBytes + kSBytes);
  static constexpr int kKPieces = 16 / kW;             // 1 KiB K pieces per wave and tile
  static constexpr int kDmaPerTile = kKPieces + 4;     // DMA instructions per wave and tile
};

__device__ __forceinline__ void dma_load128_stream(dma_rsrc_t r, int lds_wave_base, int voffset) {
#if RFA_DQS_NT
  asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen nt lds"
               :
               : "v"(voffset), "s"(r.w), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base))
               : "m0");
#else
  dma_load128(r, lds_wave_base, voffset);
#endif
}

template <typename T, int kW>
__global__ __launch_bounds__(DsGeo<kW>::kThreads, 2) void dq_ds_kernel(const BwdParams p) {
  typedef DsGeo<kW> Geo;
  constexpr int kDsRows = Geo::kRows, kDsStages = Geo::kStages, kDsSBytes = Geo::kSBytes;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  // LDS map: K stages at [0, 3 x 16K), dS stages behind them ([stage][wave][4K])
  constexpr int kOffS = kDsStages * kDsKBytes;

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

  // dense, or packed sequences (cu_seqlens), whole or the front / back half of every sequence
  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int lq = qs.len, lk = ks.len;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;
  const int qwg0 = qblk * kDsRows;
  if (qwg0 >= lq) return;
  const int off = lk - lq;
  const int qw0 = qwg0 + wave * 32;
  const int qrow = qw0 + l31;

  const T* kbase = (const T*)p.k + kbatch * p.k_st.batch + ks.row0 * p.k_st.row + (int64_t)hk * p.k_st.head;
  // scratch layout: (b, h, qt, kb) with the extents of the LONGEST sequence (p.Sq, p.Sk = max_seqlen for packed
  // input; = the sequence length when dense), of which this sequence uses the first ceil(lk / 32) key blocks of
  // its first ceil(lq / 32) rows of blocks
  const int nKb = ds_blocks(p.Sk, p.k_half);
  // this wave's run of dS blocks: (b, h, qt = qw0 / 32, kb = 0 .. ceil(lk/32)-1), contiguous
  // (rows are rectangular, p.ds_c >= nKb, or packed triangular for dense causal calls: rfa_kernels.hpp)
  const int qt = qw0 >> 5;
  const char* srun = (const char*)p.ds + (ds_base_blocks(p, b) + (int64_t)h * p.ds_head_blocks +
                                          ds_rowpart(p, qt, (qs.row0 >> 5) + b, nKb)) * (int64_t)kDsBlockBytes;
  int run = (lk + 31) >> 5;                              // blocks of this row that can hold data
  const int rlen = ds_rowlen(p, qt, nKb);
  run = run < rlen ? run : rlen;
  const dma_rsrc_t rs = make_dma_rsrc(srun, qw0 < lq ? run * kDsBlockBytes : 0);

  const int qend = (qwg0 + kDsRows < lq) ? qwg0 + kDsRows : lq;
  int kmax = lk;                                        // (windowed calls are not eligible for the spill path)
  if (p.causal && qend + off < kmax) kmax = qend + off;
  const int ntiles = kmax > 0 ? (kmax + kDsKV - 1) / kDsKV : 0;

  // K tile DMA: as in the forward kernel (lane L of the piece for row group c = wave + 8 i lands in row
  // 4c + L/16, physical chunk L%16 and fetches the logical chunk the swizzle puts there)
  int voff_k[Geo::kKPieces];
#pragma unroll
  for (int i = 0; i < Geo::kKPieces; ++i) {
    const int row = 4 * (wave + kW * i) + (lane >> 4);
    const int chunk = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (wave & 3));
    // (p.D < 128: the second 128-column chunk of a head dim 136 .. 248 — the lanes of its missing 16-byte chunks point past
    //  every descriptor range and write zeros, as in rfa_bigd.hip)
    voff_k[i] = chunk * 8 < p.D ? (row * (int)p.k_st.row + chunk * 8) * 2 : 0x7ffffff0;
  }
  const int voff_s = lane * 16;
  auto load_tile = [&](int j, auto stage) {
    constexpr int kStage = decltype(stage)::value;
    int rows = lk - j * kDsKV;
    rows = rows < kDsKV ? rows : kDsKV;
    const int nk = rows > 0 ? ((rows - 1) * (int)p.k_st.row + p.D) * 2 : 0;
    const dma_rsrc_t rk = make_dma_rsrc(kbase + (int64_t)j * kDsKV * p.k_st.row, nk);
#pragma unroll
    for (int i = 0; i < Geo::kKPieces; ++i)
      dma_load128(rk, lds_addr(smem) + kStage * kDsKBytes + (wave + kW * i) * 1024, voff_k[i]);
    const int sdst = lds_addr(smem) + kOffS + kStage * kDsSBytes + wave * kDsWaveBytes;
#pragma unroll
    for (int d = 0; d < 4; ++d)          // blocks kb = 2j, 2j+1 (past the run: descriptor range -> zeros)
      dma_load128_stream(rs, sdst + d * 1024, voff_s + j * kDsWaveBytes + d * 1024);
  };

  // ---- per-lane LDS addresses
  // K^T fragment (A operand: m = d = 32 dblk + l31, k = key 16 ks + 8 g + e): transpose reads at rows
  // rb = 16 ks + 8 g + 4 hh
  int aoff[4][2];
#pragma unroll
  for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      aoff[dblk][hh] = lds_addr(smem) + (8 * g + 4 * hh + ((lane & 15) >> 2)) * kRowBytes +
                       tr_lane_off(lane, dblk, (2 * g + hh) & 3);
      pin_vgpr(aoff[dblk][hh]);
    }
  // dS^T fragment (B operand: n = q = l31, k = key 16 ks' + 8 g + e inside a block): the 4 x 16 block
  // (keys k0 .. k0+3) x (16 queries of half i) is gathered from slots 16 (k0/4) + 8 i + 4 (c&1) + r, 8-byte
  // chunk (c >> 1), with r = row supplied by this lane, c = its chunk of 4 queries
  int boff;
  {
    const int i = (lane >> 4) & 1, c = lane & 3, r = (lane >> 2) & 3;
    boff = lds_addr(smem) + kOffS + wave * kDsWaveBytes + 512 * g + 16 * (8 * i + 4 * (c & 1) + r) + 8 * (c >> 1);
    pin_vgpr(boff);
  }

  f32x16 dq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

  typedef std::integral_constant<int, 0> st0;
  typedef std::integral_constant<int, 1> st1;
  typedef std::integral_constant<int, 2> st2;
  typedef std::integral_constant<int, 3> st3;
  typedef std::integral_constant<int, 4> st4;
  constexpr int kFly = (kDsStages - 2) * Geo::kDmaPerTile;   // DMA instructions that may stay in flight behind the tile awaited
  load_tile(0, st0{});
  if (ntiles >= kDsStages - 1) {
    load_tile(1, st1{});
    if (kDsStages > 3) load_tile(2, st2{});
    if (kDsStages > 4) load_tile(3, st3{});
    wait_vmem<kFly>();
  } else {
    if (ntiles > 1) load_tile(1, st1{});
    if (kDsStages > 4 && ntiles > 2) load_tile(2, st2{});
    wait_all_vmem();
  }
  __syncthreads();

  auto tile_step = [&](int j, auto stage) {
    constexpr int kStage = decltype(stage)::value;
    constexpr int kbo = kStage * kDsKBytes;
    constexpr int sbo = kStage * kDsSBytes;
    typedef std::integral_constant<int, (kStage + kDsStages - 1) % kDsStages> fill_t;
    const bool more = j + kDsStages - 1 < ntiles;
    if (more) load_tile(j + kDsStages - 1, fill_t{});
    if (qw0 < lq) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int kw0 = j * kDsKV + 32 * blk;
        // exactly the predicate under which dkdv_kernel wrote this block
        const bool active = (kw0 < lk) && !(p.causal && qw0 + 31 + off < kw0);
        if (active) {
#pragma unroll
          for (int ks2 = 0; ks2 < 2; ++ks2) {
            const int ks = 2 * blk + ks2;
            const int simm = sbo + blk * kDsBlockBytes + 1024 * ks2;
            const vec4<T> b0 = lds_read_tr<T>(lds_ptr(boff) + simm);
            const vec4<T> b1 = lds_read_tr<T>(lds_ptr(boff) + simm + 256);
            const vec8<T> bf = concat<T>(b0, b1);
#pragma unroll
            for (int dblk = 0; dblk < 4; ++dblk) {
              const int imm = kbo + 16 * ks * kRowBytes;
              const vec4<T> lo = lds_read_tr<T>(lds_ptr(aoff[dblk][0]) + imm);
              const vec4<T> hi = lds_read_tr<T>(lds_ptr(aoff[dblk][1]) + imm);
              dq[dblk] = mfma(concat<T>(lo, hi), bf, dq[dblk]);
            }
          }
        }
      }
    }
    if (more) wait_vmem<kFly>();     // tile j+1 has landed; the DMA instructions of the tiles behind it may fly on
    else wait_all_vmem();
    __syncthreads();
  };
  for (int j = 0; j < ntiles; j += kDsStages) {
    tile_step(j, st0{});
    if (j + 1 < ntiles) tile_step(j + 1, st1{});
    if (j + 2 < ntiles) tile_step(j + 2, st2{});
    if (kDsStages > 3 && j + 3 < ntiles) tile_step(j + 3, st3{});
    if (kDsStages > 4 && j + 4 < ntiles) tile_step(j + 4, st4{});
  }

  if (qrow >= lq) return;
  if (p.dq_acc == nullptr) {
    T* ob = (T*)p.dq + qbatch * p.dq_st.batch + (qs.row0 + qrow) * p.dq_st.row + (int64_t)h * p.dq_st.head;
    store_rows16<T, false>(ob, dq, p.scale, g, p.D, true);
  } else {
    float* ab = p.dq_acc + qbatch * p.dq_acc_st.batch + (qs.row0 + qrow) * p.dq_acc_st.row +
                (int64_t)h * p.dq_acc_st.head;
#pragma unroll
    for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int d0 = 32 * dblk + 8 * jj + 4 * g;
        if (d0 >= p.D) continue;
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

template <typename T, int kW>
static int launch_dq_ds_w(BwdParams p, hipStream_t stream) {
  typedef DsGeo<kW> Geo;
  static std::atomic<unsigned long long> attr_done{0};
  if (int rc = opt_in_dynamic_lds((const void*)dq_ds_kernel<T, kW>, Geo::kSmem, attr_done)) return rc;
  const int lq = p.q_half ? (p.Sq + 1) / 2 : p.Sq;       // (the longest (half) sequence: rfa_api.cpp eff_len)
  p.nqblk = (lq + Geo::kRows - 1) / Geo::kRows;
  const int64_t nblocks = (int64_t)p.nqblk * p.H * p.B;
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL((dq_ds_kernel<T, kW>), dim3((unsigned)nblocks), dim3(Geo::kThreads), Geo::kSmem, stream, p);
  return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchFailed;
}

template <typename T>
static int launch_dq_ds_t(const BwdParams& p, hipStream_t stream) {
  // 128-row workgroups when the 256-row grid would leave CUs without one (one workgroup per CU in both forms)
  const int lq = p.q_half ? (p.Sq + 1) / 2 : p.Sq;
  const int64_t wgs8 = (int64_t)((lq + 255) / 256) * p.H * p.B;
#ifndef RFA_DQS_FORM
#define RFA_DQS_FORM 0       // tuning: 8 / 4 = always that form
#endif
  const bool small = RFA_DQS_FORM == 4 || (RFA_DQS_FORM == 0 && wgs8 < 256);
  return small ? launch_dq_ds_w<T, 4>(p, stream) : launch_dq_ds_w<T, 8>(p, stream);
}

int launch_bwd_dq_from_ds(const BwdParams& p, int dtype, hipStream_t stream) {
  if (p.D > kHeadDim) {
    // head dims 136 .. 256 (dS written by rfa_bigd.hip's dK launch): dQ[:, c] = scale dS K[:, c] per 128-column chunk c —
    // the same kernel twice, on column-offset views of K and dQ (the second chunk has D - 128 columns)
    for (int c = 0; c < 2; ++c) {
      BwdParams pc = p;
      pc.D = c == 0 ? kHeadDim : p.D - kHeadDim;
      pc.k = (const char*)p.k + (size_t)c * kHeadDim * 2;
      if (p.dq_acc) pc.dq_acc = p.dq_acc + c * kHeadDim;
      else pc.dq = (char*)p.dq + (size_t)c * kHeadDim * 2;
      if (int rc = dtype == 0 ? launch_dq_ds_t<bf16_t>(pc, stream) : launch_dq_ds_t<f16_t>(pc, stream)) return rc;
    }
    return kLaunchOk;
  }
  return dtype == 0 ? launch_dq_ds_t<bf16_t>(p, stream) : launch_dq_ds_t<f16_t>(p, stream);
}

}  // namespace rfa
