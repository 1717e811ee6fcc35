// rfa_kernels.hpp — host-visible parameter blocks and launchers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rfa_common.hpp"

#include <atomic>

namespace rfa {

// windowed instances are needed for a left bound, or for a right bound that is not the plain causal one
inline bool windowed(int causal, int wl, int wr) { return wl >= 0 || (wr >= 0 && !causal); }

// launcher status codes (rfa_api.cpp maps them to rfa_status)
enum { kLaunchOk = 0, kLaunchFailed = -1, kLaunchAttrFailed = -2 };

// Dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) is a PER-DEVICE attribute of a kernel:
// `done` is a bit mask indexed by device ordinal, so a process driving several GPUs opts every device in,
// the call is made once per (kernel, device), concurrent first launches are benign (the attribute is
// idempotent), and a failure is reported instead of surfacing later as a generic launch error.
inline int opt_in_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return kLaunchAttrFailed;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return kLaunchOk;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return kLaunchAttrFailed;
  }
  done.fetch_or(bit, std::memory_order_release);
  return kLaunchOk;
}

struct FwdParams {
  const void *q, *k, *v;
  void* out;
  float* lse;
  float* out_acc;
  float* lse_acc;
  const int32_t *cu_q, *cu_k;
  Strides q_st, k_st, v_st, out_st, out_acc_st;
  int64_t lse_batch, lse_head, lse_acc_batch, lse_acc_head;
  int B, H, Hk, D, Sq, Sk;
  int q_half, k_half;
  int causal, acc_init;
  int wl, wr;          // visible keys of query i: i + (Sk - Sq) - wl <= j <= i + (Sk - Sq) + wr; -1 = unbounded
                       // (causal is folded in by the API layer: wr = 0)
  int nqblk;
  int qrows;           // query rows per workgroup the launch was sized for: 256 (8 waves) or 128 (4 waves, small grids)
  int persist_grid;    // > 0: the persistent 256-row form (fwd_persist_kernel) with this many workgroups
  // split-KV launches (short query ranges against long key ranges on under-filled grids): the key tiles of a workgroup's
  // range are divided between kv_nsplit workgroups; split s writes a NORMALISED partial (out fp32, lse; -inf = no key)
  // to part_out + s * part_out_split / part_lse + s * part_lse_split (laid out like out_acc / lse_acc), and
  // combine_kernel (rfa_aux.hip) merges the splits into the call's real outputs
  int kv_nsplit;
  float* part_out;
  float* part_lse;
  int64_t part_out_split, part_lse_split;
  float scale;
  // dropout (rfa_common.hpp: drop_word): keep threshold 0..256 (256 = off), scale of kept probabilities, seed and
  // the offsets that turn local (head, query position, key position) into global ones
  unsigned drop_keep;
  float drop_scale;
  unsigned long long drop_seed;
  unsigned q_pos0, k_pos0, head0;
};

struct PreParams {
  const void *dout, *out;
  float* delta;
  const int32_t* cu_q;
  Strides dout_st, out_st;
  int64_t delta_batch, delta_head;
  int B, H, D, Sq, q_half;
};

struct BwdParams {
  const void *dout, *q, *k, *v;
  const float *lse, *delta;
  void *dq;            // io dtype (or nullptr when dq_acc)
  float* dq_acc;
  void *dk, *dv;       // per-q-head outputs, io dtype: head index = q head
  const int32_t *cu_q, *cu_k;
  Strides dout_st, q_st, k_st, v_st, dq_st, dq_acc_st, dk_st, dv_st;
  int64_t lse_batch, lse_head, delta_batch, delta_head;
  int B, H, Hk, D, Sq, Sk;
  int q_half, k_half;
  int causal, acc_init;
  int wl, wr;          // attention window as in FwdParams (causal: wr = 0)
  int kv_f32;          // dk / dv point to fp32 buffers (overwritten), strides in fp32 elements
  void* ds;            // dS spill scratch (rfa_dqs.hip) or nullptr: dkdv_kernel stores its packed dS blocks there
  int ds_tri, ds_c;    // scratch rows are triangular (dense causal calls): row qt holds key blocks 0 .. min(nKb, qt + ds_c) - 1
  int ds_packed;       // packed (cu_seqlens) layout of the scratch: [head][global query-block row][key block], see ds_rowpart()
  int64_t ds_head_blocks;   // blocks per head (dense: per (batch, head))
  int nqblk, nkblk;
  float scale;
  // 256-key dK/dV form (dkdv_kernel kWide): the tile range of a key block is shared by nsplit workgroups; split s
  // stores its partial kv_split_stride elements behind split 0's (rfa_api.cpp lays them out for reduce_kernel)
  int wide, nsplit;
  int64_t kv_split_stride;
  int kv_part_f32;     // dk / dv point to fp32 PARTIALS in the workspace (split launches): fp32 stores, strides in fp32 elements
  int kv_accum;        // fp32 stores ADD to what the partial holds (later query-head fractions of a chunked dS hand-off)
  // balanced causal schedule of the 256-key form (dkdv_kernel kBal; rfa_api.cpp: bwd_dkdv_plan): B * Hk * nkblk / 2 pair
  // slots of 8 waves x 2 tensors x D / 32 x 4 KiB (fp32 accumulators in lane order) and one flag word per pair
  // (0 = nobody yet, 1 = the first arriver is publishing, 2 = published), zeroed by a kernel of the launcher's before every launch
  int bal;
  void* pair_ws;
  unsigned* pair_flags;
  // dropout (rfa_common.hpp: drop_word): keep threshold 0..256 (256 = off), scale of kept probabilities, seed and
  // the offsets that turn local (head, query position, key position) into global ones
  unsigned drop_keep;
  float drop_scale;
  unsigned long long drop_seed;
  unsigned q_pos0, k_pos0, head0;
};

// dst[b, row, hk, :] (=|+=) sum_g src[b, row, hk*G+g, :]
struct ReduceParams {
  const void* src;     // partials (io dtype, or fp32 when src_f32), head index = q head
  void* dst;           // io dtype (or nullptr)
  float* dst_acc;      // fp32 accumulate target (or nullptr)
  // optional second tensor reduced by the same launch (blockIdx.z = 1): dV next to dK
  const void* src2;
  void* dst2;
  float* dst_acc2;
  const int32_t* cu_k;
  Strides src_st, dst_st, dst_acc_st, dst2_st, dst_acc2_st;
  int B, Hk, G, D, Sk, k_half, acc_init;
  int src_f32;
  int64_t g_stride;    // 0: member g of head hk is source head hk*G+g; else: head hk, g_stride elements further per g
};

// combine the kv_nsplit partial (out, lse) pairs of a split-KV forward launch into the call's outputs
struct CombineParams {
  const float* part_out;    // (nsplit, ...) laid out like out_acc: strides part_st, split stride part_out_split
  const float* part_lse;
  Strides part_st;
  int64_t part_lse_batch, part_lse_head, part_out_split, part_lse_split;
  int nsplit;
  void* out;                // io dtype (plain mode) or nullptr
  float* lse;
  float* out_acc;           // fp32 accumulate mode or nullptr
  float* lse_acc;
  Strides out_st, out_acc_st;
  int64_t lse_batch, lse_head, lse_acc_batch, lse_acc_head;
  const int32_t* cu_q;
  int B, H, D, Sq, q_half, acc_init;
};

struct MergeParams {
  float* out_acc;
  float* lse_acc;
  const void* block_out;
  const float* block_lse;
  Strides out_acc_st, block_out_st;
  int64_t lse_acc_batch, lse_acc_head, block_lse_batch, block_lse_head;
  int64_t lse_acc_row, block_lse_row;
  int B, H, D, S, acc_init;
};

int launch_fwd(const FwdParams& p, int dtype, hipStream_t stream);
int fwd_qrows_per_block();
// head dims 129 .. 256 (rfa_bigd.hip): same parameter blocks; the launchers set their own nqblk / nkblk
constexpr int kMaxHeadDim = 256;
int launch_fwd_big(const FwdParams& p, int dtype, hipStream_t stream);
int launch_bwd_dq_big(const BwdParams& p, int dtype, hipStream_t stream);
int launch_bwd_dkdv_big(const BwdParams& p, int dtype, hipStream_t stream);

int launch_preprocess(const PreParams& p, int dtype, hipStream_t stream);
int launch_bwd_dq(const BwdParams& p, int dtype, hipStream_t stream);
int launch_bwd_dkdv(const BwdParams& p, int dtype, hipStream_t stream);
// dQ = scale * dS K from the dS blocks a preceding launch_bwd_dkdv (with p.ds set) stored; dense, D == 128
int launch_bwd_dq_from_ds(const BwdParams& p, int dtype, hipStream_t stream);
constexpr int kDsBlockBytes = 2048;
// rows / columns of 32 x 32 dS blocks the spill scratch reserves per (sequence, head): the longest (half) sequence
__host__ __device__ inline int ds_blocks(int S, int half) { return ((half ? (S + 1) / 2 : S) + 31) >> 5; }   // one (32 query x 32 key) block of dS in the io dtype
// Layout of the dS scratch of one (sequence, head): rows qt = query row / 32 of 2 KiB blocks kb = key / 32.
//   rectangular (packed input, non-causal): every row holds nKb blocks;
//   triangular (dense causal: block (qt, kb) is visited iff kb < qt + c, c = ((31 + lk - lq) >> 5) + 1): row qt holds
//   its first ds_row_len(qt) = clamp(qt + c, 0, nKb) blocks, rows packed back to back — half the scratch of a
//   square causal launch.  ds_row_off(nQt, ...) is the number of blocks per (sequence, head).
__host__ __device__ inline int ds_row_len(int qt, int nKb, int c, int tri) {
  if (!tri) return nKb;
  const int n = qt + c;
  return n < 0 ? 0 : (n > nKb ? nKb : n);
}
__host__ __device__ inline int64_t ds_row_off(int qt, int nKb, int c, int tri) {
  if (!tri) return (int64_t)qt * nKb;
  const int iA = c < 0 ? -c : 0;                        // first row with a non-negative length
  int iB = nKb - c;                                     // first row at full length
  iB = iB < iA ? iA : iB;
  const int e1 = qt < iB ? qt : iB;
  const int64_t n1 = e1 > iA ? e1 - iA : 0;
  const int64_t n2 = qt > iB ? qt - iB : 0;
  return n1 * (iA + c) + n1 * (n1 - 1) / 2 + n2 * nKb;
}
// Where the dS blocks of query-block row qt (counted inside its (half) sequence) live, in blocks from the scratch base:
//     ds_base_blocks(p, b) + hc * p.ds_head_blocks + ds_rowpart(p, qt, g0, nKb)          (hc: head index inside the launch)
//   dense:  [batch][head][rows], rows rectangular or packed triangular (ds_row_off above);
//   packed: [head][g][nKb] with the GLOBAL row index g = g0 + qt, g0 = (first packed row of the (half) sequence >> 5)
//           + sequence index — unique per (sequence, row) because ceil(len / 32) <= floor(end / 32) - floor(start / 32) + 1.
//           The scratch then holds total_q / 32 + B rows per head instead of B x max_seqlen / 32: bounded by the PACKED
//           row count whatever the mix of sequence lengths (one long and two short sequences used to cost three
//           times the longest: 10.5 GB instead of 3.9 GB for the varlen benchmark's second pattern).
__host__ __device__ inline int64_t ds_base_blocks(const BwdParams& p, int b) {
  return p.ds_packed ? 0 : (int64_t)b * p.H * p.ds_head_blocks;
}
__host__ __device__ inline int64_t ds_rowpart(const BwdParams& p, int qt, int64_t g0, int nKb) {
  return p.ds_packed ? (g0 + qt) * (int64_t)nKb : ds_row_off(qt, nKb, p.ds_c, 1);
}
__host__ __device__ inline int ds_rowlen(const BwdParams& p, int qt, int nKb) {
  return p.ds_packed ? nKb : ds_row_len(qt, nKb, p.ds_c, 1);
}
int bwd_dq_rows_per_block();
int bwd_dkdv_keys_per_block(bool wide);
int launch_reduce(const ReduceParams& p, int dtype, hipStream_t stream);
int launch_merge(const MergeParams& p, int dtype, hipStream_t stream);
int launch_combine(const CombineParams& p, int dtype, hipStream_t stream);
int launch_cast(void* dst, const float* src, int64_t n, int dtype, hipStream_t stream);
int launch_lse_relayout(float* dst, const float* src, const int32_t* cu, int B, int H,
                        int max_seqlen, int64_t packed_head_stride, int64_t packed_row_stride,
                        bool flatten, hipStream_t stream);

}  // namespace rfa
