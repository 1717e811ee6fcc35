/*
 * rfa.h — C ABI of librfa_hip.so, the MI355X (gfx950) attention operator library.
 *
 * This is the drop-in boundary of the hot path.  Each entry point replaces one private
 * function of the (un-vendored, CUDA-only) `flash_attn.flash_attn_interface` module that the
 * reference calls, or one eager/Triton helper of the reference itself:
 *
 *   rfa_fwd            <- flash_attn._flash_attn_forward          (call sites
 *                         /root/reference/ring_flash_attn/ring_flash_attn.py:53,
 *                         zigzag_ring_flash_attn.py:52) and
 *                         flash_attn._flash_attn_varlen_forward   (ring_flash_attn_varlen.py:77,
 *                         zigzag_ring_flash_attn_varlen.py:137, llama3_flash_attn_varlen.py:147)
 *                         plus, when `out_acc/lse_acc` are given, the fused form of
 *                         update_out_and_lse (ring_flash_attn/utils.py:32-73).
 *   rfa_bwd_preprocess <- the rowsum(dO*O) prologue inside flash_attn._flash_attn_backward.
 *   rfa_bwd            <- flash_attn._flash_attn_backward          (ring_flash_attn.py:131,
 *                         zigzag_ring_flash_attn.py:156) and _flash_attn_varlen_backward
 *                         (ring_flash_attn_varlen.py:169, zigzag_ring_flash_attn_varlen.py:275,
 *                         llama3_flash_attn_varlen.py:282), plus, when `*_acc` are given, the
 *                         fp32 `dq += ...`, `dk += ...` accumulation of
 *                         zigzag_ring_flash_attn.py:164-187 / ring_flash_attn.py:134-141.
 *   rfa_merge          <- _update_out_and_lse (ring_flash_attn/utils.py:32-50) as a stand-alone
 *                         kernel (also covers the slice_ variant, utils.py:65-70).
 *   rfa_lse_flatten / rfa_lse_unflatten
 *                      <- triton_utils.flatten_varlen_lse / unflatten_varlen_lse
 *                         (ring_flash_attn/triton_utils.py:39-67,103-137).
 *   rfa_cast           <- `out.to(q.dtype)` / `dq.to(q.dtype)` tails (zigzag_ring_flash_attn.py:86,199).
 *
 * Conventions
 *   - Plain C, no torch types.  All pointers are DEVICE pointers owned by the caller.  The
 *     library never allocates, frees or synchronises; every launch goes to the `stream`
 *     argument (a hipStream_t passed as void*).
 *   - All strides are in ELEMENTS of the tensor's own dtype.  The innermost (head_dim) stride
 *     must be 1 and every row start must be 16-byte aligned.
 *   - Dense mode (cu_seqlens_* == NULL):  q is (B, Sq, H, D), k/v are (B, Sk, Hk, D), lse is
 *     (B, H, Sq).  Varlen mode: q is (Tq, H, D), k/v (Tk, Hk, D), lse is (H, Tq); `B` is the
 *     number of packed sequences, `Sq`/`Sk` the max sequence lengths, *_batch strides unused.
 *   - Causal masks are bottom-right aligned: key j is visible to query i iff
 *     j <= i + (len_k - len_q)  (flash_attn >= 2.1 semantics).
 *   - Rows with no visible key produce out = 0 and lse = +inf (flash_attn semantics); in
 *     accumulate mode such rows leave the accumulators untouched.
 *   - `q_half` / `k_half` select, per packed (or dense) sequence, the whole sequence (0), its
 *     front half (1) or its back half (2) WITHOUT gathering — the offset arithmetic that
 *     replaces get_half_index/get_half_lse (zigzag_ring_flash_attn_varlen.py:24-71).  Outputs
 *     (out, lse, dq, delta) are addressed like q; dk/dv like k.
 *   - Return value: 0 on success, negative rfa_status otherwise; rfa_strerror() explains.
 */
#ifndef RFA_H_
#define RFA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFA_ABI_VERSION 6

typedef enum {
  RFA_OK = 0,
  RFA_ERR_NULL = -1,        /* required pointer is NULL                       */
  RFA_ERR_DTYPE = -2,       /* dtype not RFA_BF16 / RFA_F16                    */
  RFA_ERR_HEAD_DIM = -3,    /* head_dim not a multiple of 8 or > 256           */
  RFA_ERR_HEADS = -4,       /* H not a multiple of Hk                          */
  RFA_ERR_SHAPE = -5,       /* negative / zero extents                         */
  RFA_ERR_ALIGN = -6,       /* pointer or stride breaks the 16-byte contract   */
  RFA_ERR_LAUNCH = -7,      /* hipLaunchKernel reported an error               */
  RFA_ERR_ARGS = -8,        /* inconsistent flag / pointer combination         */
  RFA_ERR_ATTR = -9         /* per-device dynamic-LDS opt-in of a kernel failed */
} rfa_status;

typedef enum { RFA_BF16 = 0, RFA_F16 = 1 } rfa_dtype;

enum { RFA_HALF_FULL = 0, RFA_HALF_FRONT = 1, RFA_HALF_BACK = 2 };

/* (batch, row, head) element strides of a (B, S, H, D) / (T, H, D) tensor */
typedef struct {
  int64_t batch;
  int64_t row;
  int64_t head;
} rfa_strides;

typedef struct {
  /* inputs */
  const void *q, *k, *v;
  rfa_strides q_st, k_st, v_st;
  /* plain outputs (io dtype out, fp32 lse).  Used when out_acc == NULL. */
  void *out;
  rfa_strides out_st;
  float *lse;              /* (B,H,Sq) or (H,Tq) */
  int64_t lse_batch, lse_head; /* element strides; row stride is 1 */
  /* fused online-merge accumulators (fp32).  When non-NULL the kernel merges its block
   * result into them instead of writing out/lse:  out_acc has out's logical shape,
   * lse_acc has lse's.  acc_init != 0: first block of a ring — overwrite, do not read. */
  float *out_acc;
  rfa_strides out_acc_st;
  float *lse_acc;
  int64_t lse_acc_batch, lse_acc_head;
  int32_t acc_init;
  /* varlen */
  const int32_t *cu_seqlens_q, *cu_seqlens_k; /* (B+1,) int32 device, or NULL */
  int32_t q_half, k_half;
  /* shape */
  int32_t B, H, Hk, D, Sq, Sk;
  float softmax_scale;
  int32_t causal;
  int32_t dtype; /* rfa_dtype */
  /* Local (sliding window) attention, flash_attn semantics (window_size_left / window_size_right of
   * _flash_attn_forward, forwarded by /root/reference/ring_flash_attn/llama3_flash_attn_varlen.py:147 and
   * adapters/hf_adapter.py:121-128): query i sees keys j with
   *     i + (len_k - len_q) - window_left <= j <= i + (len_k - len_q) + window_right ;
   * a negative value leaves that side unbounded; `causal` forces window_right = 0.  Only honoured when
   * `window` is non-zero, so that a zero-initialised struct means "no window". */
  int32_t window, window_left, window_right;
  /* Dropout (flash_attn's dropout_p; the reference forwards it on the llama3 path,
   * /root/reference/ring_flash_attn/llama3_flash_attn_varlen.py:131,135,266).  dropout_p = 0: off.  An attention
   * probability is kept with probability keep/256, keep = round((1 - dropout_p) * 256), and kept ones are scaled by
   * 256 / keep, keep = round((1 - dropout_p) * 256) — the reciprocal of the probability the mask really keeps with, so that
   * E[dropout(P)] = P for every p —; lse is that of the undropped softmax (flash_attn semantics).  DEVIATION from
   * flash_attn, deliberate: flash_attn also quantises its threshold to 8 bits but keeps scaling by 1 / (1 - dropout_p),
   * which is biased by up to 0.4 % (p = 0.17: 256 / 212 = 1.2075 here, 1 / 0.83 = 1.2048 there); its Philox mask cannot
   * be reproduced without the package anyway, so no bit-level parity is lost (tests/test_oracle.py pins the factor
   * and the unbiasedness against values written out by hand, not against the oracle).  The keep mask is a pure
   * function of (dropout_seed, batch, head_offset + head, q_pos_offset + query position, k_pos_offset + key
   * position) — csrc/rfa_common.hpp: drop_word — where a position is the row inside the dense sequence, or the
   * absolute row of the packed tensor for cu_seqlens input; a rank that holds rows [a, b) of a longer stream passes
   * a / the first gathered key row as offsets and gets the same bits as an unsharded call.  The backward must be
   * given the forward's values.  Not available together with a window (RFA_ERR_ARGS). */
  float dropout_p;
  uint64_t dropout_seed;
  int64_t q_pos_offset, k_pos_offset;
  int32_t head_offset;
  /* Kernel form (tuning / tests; 0 = chosen by the library): RFA_FWD_8x32 = 8 waves x 32 query rows, two waves per SIMD
   * (csrc/rfa_fwd.hip: every head dim, windows, dropout; 256 query rows per workgroup — RFA_FWD_AUTO also launches the
   * same kernel with 4 waves / 128 rows on grids that would under-fill the chip, RFA_FWD_4x32 asks for that form wherever
   * it exists: head dim 128 / 64, no window, no dropout).  Value 2 named a 4-wave x 64-row, one-wave-per-SIMD experiment
   * that lost to the 8 x 32 form by 7 - 13 % in two rounds and was removed in round 6: RFA_ERR_ARGS.
   * RFA_FWD_P8x32 (ABI 6): the PERSISTENT 256-row form — one workgroup per CU walks its share of the (batch, head, query
   * block) items, the next item's first K/V tile and Q fragments fetched under the current item's last tile and epilogue.
   * Head dim 128, dense input (no cu_seqlens) with Sk >= Sq, plain outputs (no out_acc), no window, no dropout, no split-KV
   * shares; where a call is not eligible the field reads as RFA_FWD_AUTO.  RFA_FWD_AUTO takes it for launches of at least
   * two items per CU.  Bit-identical to the 8 x 32 form. */
  int32_t fwd_form;
  /* ABI 5 — split-KV launches.  A call with few query rows and many keys (a llama3 head group at 2048 tokens per rank
   * against the gathered keys of 8 ranks: 256 key tiles per workgroup, half the CUs without one) is launched with the
   * key tiles of every workgroup divided between kv_nsplit workgroups; each writes a normalised partial (out, lse) to
   * `workspace` and a second, streaming kernel combines the partials into the call's outputs (plain or accumulate mode
   * alike).  workspace: rfa_fwd_workspace_bytes() bytes, or NULL (= never split).  kv_nsplit: 0 = chosen from the
   * shapes, 1 = off, 2..8 forced (tests).  total_q: varlen only, number of packed q rows addressed (sizes the
   * workspace).  Eligible: head dim 128 or 64 exactly, no window, no dropout. */
  void *workspace;
  int32_t kv_nsplit;
  int64_t total_q;
} rfa_fwd_args;

enum { RFA_FWD_AUTO = 0, RFA_FWD_8x32 = 1, RFA_FWD_RETIRED_2 = 2, RFA_FWD_4x32 = 3, RFA_FWD_P8x32 = 4 };

typedef struct {
  const void *dout, *out; /* (B,Sq,H,D) io dtype */
  rfa_strides dout_st, out_st;
  float *delta;           /* same layout contract as lse */
  int64_t delta_batch, delta_head;
  const int32_t *cu_seqlens_q;
  int32_t q_half;
  int32_t B, H, D, Sq;
  int32_t dtype;
} rfa_bwd_preprocess_args;

typedef struct {
  const void *dout, *q, *k, *v;
  rfa_strides dout_st, q_st, k_st, v_st;
  const float *lse;   /* GLOBAL log-sum-exp of the rows (natural log)            */
  int64_t lse_batch, lse_head;
  const float *delta; /* rowsum(dout*out), from rfa_bwd_preprocess              */
  int64_t delta_batch, delta_head;
  /* plain outputs in io dtype (used when the matching *_acc pointer is NULL) */
  void *dq, *dk, *dv;
  rfa_strides dq_st, dk_st, dv_st;
  /* fp32 accumulators: dq_acc += dq ; dk_acc += dk ; dv_acc += dv  (acc_init: overwrite) */
  float *dq_acc, *dk_acc, *dv_acc;
  rfa_strides dq_acc_st, dk_acc_st, dv_acc_st;
  int32_t acc_init;
  /* workspace for dK/dV partials (already summed over the query heads of a K/V group),
   * rfa_bwd_workspace_bytes() bytes.  Unsplit launches: 2 * total_k*Hk*D elements of the io dtype (one
   * rounding per block, the rounding point of flash_attn's block gradients).  Launches whose key blocks
   * share their query range between `nsplit` workgroups (256-key kernel form, see dkdv_form): nsplit fp32
   * partials per element, summed and rounded ONCE by the reduction pass.
   * ALWAYS size it with rfa_bwd_workspace_bytes(); may be NULL whenever that returns 0 (single-phase
   * calls that write dk/dv or overwrite dk_acc/dv_acc on launches that need no such split). */
  void *workspace;
  const int32_t *cu_seqlens_q, *cu_seqlens_k;
  int32_t q_half, k_half;
  int32_t B, H, Hk, D, Sq, Sk;
  int64_t total_k;    /* varlen: number of packed k rows addressed (Tk); dense: B*Sk */
  float softmax_scale;
  int32_t causal;
  int32_t deterministic; /* accepted; this implementation is always deterministic */
  int32_t dtype;
  /* 0 = everything.  RFA_BWD_COMPUTE: dQ + dK/dV partials into `workspace` only;
   * RFA_BWD_REDUCE: add / copy the partials of a previous COMPUTE call into dk_acc/dv_acc or
   * dk/dv.  Splitting lets a ring step overlap the compute with the arrival of the
   * dk/dv accumulators it will add into (zigzag_ring_flash_attn.py:172-187).  With phases != 0
   * the workspace is always required. */
  int32_t phases;
  /* Optional dS spill scratch (io dtype, rfa_bwd_ds_scratch_bytes() bytes; NULL = not used).  When given and
   * rfa_bwd_ds_scratch_bytes() is non-zero for the call, the backward runs 5 GEMMs instead of 7: the dK/dV
   * kernel stores dS = P*(dP - delta) of every (32 query x 32 key) block it visits and dQ is computed by a
   * streaming GEMM over those blocks instead of recomputing S and dP (csrc/rfa_dqs.hip).  The contents are
   * only meaningful between the two kernels of one call (or between a RFA_BWD_SKIP_DQ call and the matching
   * RFA_BWD_SKIP_DKDV call).  Eligible: D == 128 (or 256), no bounded left window — dense or packed
   * (cu_seqlens: the scratch is then laid out with the extents of the longest sequence, Sq / Sk = max_seqlen). */
  void *ds_scratch;
  int32_t window, window_left, window_right; /* as in rfa_fwd_args (a bounded window_left is not eligible for ds_scratch) */
  /* dK/dV launch plan (ABI 4: part of the call instead of process environment, so that the COMPUTE and the
   * REDUCE phase of one backward — and rfa_bwd_workspace_bytes() — can never disagree).
   *   dkdv_form   RFA_DKDV_AUTO: chosen from the shapes; RFA_DKDV_128: a workgroup owns 128 keys;
   *               RFA_DKDV_256: 256 keys (head dim 128 without a window only, otherwise ignored)
   *               RFA_DKDV_BAL (ABI 6): 256 keys in the balanced causal schedule — every workgroup of a dense causal
   *               self-attention block (Sq == Sk, a multiple of 512 rows, head dim 128 or 64, single-phase call that writes
   *               dk / dv or overwrites dk_acc / dv_acc) does the same amount of work, key blocks of the lower half are
   *               shared by two workgroups that add their partials between themselves: no reduction pass.  Where the
   *               call is not eligible the field is read as RFA_DKDV_AUTO.  RFA_DKDV_256 with dkdv_nsplit > 0 names the
   *               shared-range plan instead.
   *   dkdv_nsplit 0: chosen from the shapes; 1..8: workgroups sharing the query range of a key block
   *               (256-key form only)
   * A zero-initialised struct means "auto".  rfa_bwd_plan() reports what a call will run. */
  int32_t dkdv_form, dkdv_nsplit;
  /* Measurement aid (NULL = off): 4 caller-owned hipEvent_t handles, recorded on `stream` before the first
   * launch of the call and after its first kernel, its second kernel and the reduction pass (launches a call
   * does not make record their event right behind the previous one).  Kernel order: dK/dV then dQ when the call
   * runs the dS-spill form (rfa_bwd_plan: five_gemm), dQ then dK/dV otherwise.  This is how bench.py times the
   * kernels INSIDE the real step instead of in isolation. */
  void **prof_events;
  /* dropout: as in rfa_fwd_args, with the forward's values (a call with dropout runs the 7-GEMM form) */
  float dropout_p;
  uint64_t dropout_seed;
  int64_t q_pos_offset, k_pos_offset;
  int32_t head_offset;
  /* ABI 5: size of the buffer `ds_scratch` points to.  0 = at least rfa_bwd_ds_scratch_bytes().  A SMALLER buffer does
   * not switch the 5-GEMM form off: the call then runs it in HEAD-GROUP CHUNKS that reuse the one buffer — chunks of
   * whole K/V heads (with their query heads) or, where even one K/V head's query heads do not fit, fractions of ONE
   * K/V head's query heads, whose dK/dV shares are accumulated in the fp32 partials — each chunk one dK/dV launch and
   * one dQ launch, in stream order.  Peak scratch is then independent of the head count and the long-context cases
   * (S = 32768, 32 heads: 34 GB of dS) keep the 5-GEMM form (rfa_bwd_ds_chunks() reports the chunking; head dim 128
   * only; a buffer below one query head's share, rfa_bwd_ds_scratch_min_bytes(), falls back to the 7-GEMM form). */
  int64_t ds_scratch_bytes;
  /* ABI 5, varlen: number of packed q rows addressed (Tq); 0 = total_k.  Sizes the packed layout of ds_scratch:
   * H * (total_q / 32 + B) * ceil(max_seqlen_k / 32) blocks of 2 KiB — bounded by the packed row count, not by
   * B x the longest sequence. */
  int64_t total_q;
} rfa_bwd_args;

enum { RFA_DKDV_AUTO = 0, RFA_DKDV_128 = 1, RFA_DKDV_256 = 2, RFA_DKDV_BAL = 3 };

enum {
  RFA_BWD_ALL = 0,
  RFA_BWD_COMPUTE = 1,
  RFA_BWD_REDUCE = 2,
  /* measurement aids, valid together with RFA_BWD_COMPUTE: launch only one of the two kernels */
  RFA_BWD_SKIP_DKDV = 4,
  RFA_BWD_SKIP_DQ = 8,
  /* dk_acc / dv_acc are OVERWRITTEN with this block's dK/dV (fp32) while dq_acc still follows
   * acc_init — for schedules in which every dK/dV accumulator slot receives exactly one block
   * (all-gather / reduce-scatter exchange).  In a single-phase call the dK/dV kernel then stores
   * fp32 straight into dk_acc / dv_acc: no workspace, no reduction pass (unless the launch is split,
   * see `workspace`). */
  RFA_BWD_KV_OVERWRITE = 16
};

typedef struct {
  float *out_acc;          /* (B,S,H,D) fp32, updated in place */
  rfa_strides out_acc_st;
  float *lse_acc;          /* (B,H,S) fp32, updated in place   */
  int64_t lse_acc_batch, lse_acc_head;
  const void *block_out;   /* io dtype */
  rfa_strides block_out_st;
  const float *block_lse;  /* (B,H,S) */
  int64_t block_lse_batch, block_lse_head;
  int32_t B, H, D, S;      /* S rows are merged (pass slice pointers for a row range) */
  int32_t acc_init;
  int32_t dtype;
  /* element stride between consecutive rows of lse_acc / block_lse; 0 means 1.  The
   * reference keeps its running lse as (B,S,H,1) (utils.py:41,64): batch=S*H, head=1, row=H. */
  int64_t lse_acc_row, block_lse_row;
} rfa_merge_args;

int rfa_abi_version(void);
/* identity of this build: the first 16 hex digits of the sha256 over the library's sources, compiled into the binary by
 * the build recipe (ring-flash-attention_amd/build.py).  Measurement files (profiles/r*_traffic.json) record the id of
 * the library they were collected on; bench.py only quotes them for a library with the same id. */
const char *rfa_build_id(void);
const char *rfa_strerror(int status);

int rfa_fwd(const rfa_fwd_args *args, void *stream);
/* bytes of rfa_fwd_args.workspace the call can use (0: the call is never split); *nsplit (may be NULL) = the number of
 * key-range shares the call runs with when given that workspace.  Pure function of the arguments. */
int64_t rfa_fwd_workspace_bytes(const rfa_fwd_args *args, int32_t *nsplit);
int rfa_bwd_preprocess(const rfa_bwd_preprocess_args *args, void *stream);
int64_t rfa_bwd_workspace_bytes(const rfa_bwd_args *args);
/* the launch plan of a call: *form = RFA_DKDV_128 / RFA_DKDV_256 / RFA_DKDV_BAL, *nsplit >= 1, *five_gemm = 1 when the call
 * (with its ds_scratch) runs the dS-spill form.  Pure function of the arguments. */
int rfa_bwd_plan(const rfa_bwd_args *args, int32_t *form, int32_t *nsplit, int32_t *five_gemm);
/* bytes of ds_scratch the whole hand-off of the call uses — dense: B*H*ceil(Sq/32)*ceil(Sk/32)*2048 (dense causal: the
 * visited triangle only); packed input: H*(total_q/32 + B)*ceil(max_seqlen_k/32)*2048; with q_half / k_half: ceil(S/2)
 * instead of S — or 0 if it is not eligible */
int64_t rfa_bwd_ds_scratch_bytes(const rfa_bwd_args *args);
/* the smallest ds_scratch with which the call still runs the 5-GEMM form (one query head's dS; 0: not eligible) */
int64_t rfa_bwd_ds_scratch_min_bytes(const rfa_bwd_args *args);
/* the chunking of the call's dS hand-off for its (ds_scratch, ds_scratch_bytes): *nchunks = launches pairs (0: the
 * call runs the 7-GEMM form, 1: one pair over all heads), *kv_heads / *q_heads = K/V heads and query heads PER K/V HEAD
 * of one chunk, *chunk_bytes = scratch bytes one chunk uses.  Pure function of the arguments. */
int rfa_bwd_ds_chunks(const rfa_bwd_args *args, int32_t *nchunks, int32_t *kv_heads, int32_t *q_heads, int64_t *chunk_bytes);
int rfa_bwd(const rfa_bwd_args *args, void *stream);
int rfa_merge(const rfa_merge_args *args, void *stream);

/* dst(io dtype) = (io)src(fp32), n elements, both contiguous */
int rfa_cast(void *dst, const float *src, int64_t n, int32_t dtype, void *stream);

/* dst[b, row, h, :] = sum over s < nslots of src[s][b, row, h, :]: io dtype in, summed in fp32, io dtype out.
 * The owner-side sum of the per-rank dK/dV contributions of the all-to-all exchange form — the reference's
 * `dk = dk_comm_buffer + block_dk` accumulation (zigzag_ring_flash_attn.py:164-187) after the blocks have
 * travelled in the io dtype instead of fp32 accumulators — written straight into the (possibly strided:
 * a slice of a packed kv gradient) destination, instead of a sum + cast + copy triple. */
typedef struct {
  const void *src;         /* slot 0; slot s starts slot_stride elements further */
  int64_t slot_stride;
  int32_t nslots;
  void *dst;
  rfa_strides src_st, dst_st;   /* element strides (batch, row, head) inside one slot / of dst; D contiguous */
  int32_t B, S, H, D;
  int32_t dtype;
} rfa_sum_slots_args;
int rfa_sum_slots(const rfa_sum_slots_args *args, void *stream);

/* LSE re-layout, fp32 (triton_utils.py:39-67,103-137).  The padded tensor is a contiguous
 * (B, H, max_seqlen); the packed tensor has element (h, t) at h*head_stride + t*row_stride, so
 * it covers both the (H, T) result of flatten and the (T, H, 1) input of unflatten.
 * Padding positions of the unflatten destination are left untouched (as the reference does). */
int rfa_lse_flatten(float *dst_packed, const float *src_padded, const int32_t *cu_seqlens,
                    int32_t B, int32_t H, int32_t max_seqlen, int64_t dst_head_stride,
                    int64_t dst_row_stride, void *stream);
int rfa_lse_unflatten(float *dst_padded, const float *src_packed, const int32_t *cu_seqlens,
                      int32_t B, int32_t H, int32_t max_seqlen, int64_t src_head_stride,
                      int64_t src_row_stride, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RFA_H_ */
