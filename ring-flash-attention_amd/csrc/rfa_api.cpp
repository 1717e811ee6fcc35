// rfa_api.cpp — the C ABI of librfa_hip.so (see include/rfa.h for the contract and for the
// reference call sites each entry point replaces).  Pure argument checking + parameter
// translation; no allocation, no synchronisation, launches only on the caller's stream.
#include "../../include/rfa.h"
#include "rfa_kernels.hpp"

#include <algorithm>
#include <atomic>
#include <vector>

using namespace rfa;

namespace {

inline Strides cv(const rfa_strides& s) { return Strides{s.batch, s.row, s.head}; }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// element strides must keep every row start 16-byte aligned
inline bool stride_ok(const rfa_strides& s, int elt_bytes) {
  const int64_t q = 16 / elt_bytes;
  return (s.row % q) == 0 && (s.head % q) == 0 && (s.batch % q) == 0;
}

inline int check_common(int dtype, int H, int Hk, int D, int B) {
  if (dtype != RFA_BF16 && dtype != RFA_F16) return RFA_ERR_DTYPE;
  if (D <= 0 || D > kMaxHeadDim || (D % 8) != 0) return RFA_ERR_HEAD_DIM;
  if (H <= 0 || Hk <= 0 || (H % Hk) != 0) return RFA_ERR_HEADS;
  if (B < 0) return RFA_ERR_SHAPE;
  return RFA_OK;
}

inline int launch_status(int rc) {
  return rc == kLaunchOk ? RFA_OK : (rc == kLaunchAttrFailed ? RFA_ERR_ATTR : RFA_ERR_LAUNCH);
}

inline int eff_len(int S, int half) { return half == RFA_HALF_FULL ? S : (S + 1) / 2; }

// dropout_p -> keep threshold in 1/256 (256 = nothing dropped) — csrc/rfa_common.hpp: drop_keep
inline unsigned drop_threshold(float p) {
  if (!(p > 0.f)) return 256u;
  const int k = (int)((1.f - p) * 256.f + 0.5f);
  return (unsigned)(k < 0 ? 0 : (k > 255 ? 255 : k));      // p > 0 always drops something
}
// kept probabilities are rescaled by the reciprocal of the probability the mask REALLY keeps with (keep / 256, the
// quantised threshold above), not by 1 / (1 - p): E[dropout(P)] = P for every p (oracle/flash_attn_ref.py: drop_rescale)
inline float drop_rescale(float p) {
  const unsigned k = drop_threshold(p);
  return k >= 256u ? 1.f : (k == 0u ? 0.f : 256.f / (float)k);
}
inline bool drop_args_ok(float p, int window, int wl, int wr, int causal) {
  if (!(p >= 0.f) || p >= 1.f) return false;
  const bool win = window && (wl >= 0 || (wr >= 0 && !causal));
  return !(p > 0.f && win);
}

}  // namespace

// the library is built with -fvisibility=hidden: the C ABI below (include/rfa.h) is everything a loader can bind —
// no rfa::launch_* internals, no kernel stubs (tests/test_abi.py::test_library_exports_only_the_c_abi)
#pragma GCC visibility push(default)
extern "C" {

int rfa_abi_version(void) { return RFA_ABI_VERSION; }

#ifndef RFA_BUILD_ID
#define RFA_BUILD_ID "unstamped"
#endif
const char* rfa_build_id(void) { return RFA_BUILD_ID; }

const char* rfa_strerror(int status) {
  switch (status) {
    case RFA_OK: return "ok";
    case RFA_ERR_NULL: return "a required pointer is NULL";
    case RFA_ERR_DTYPE: return "unsupported dtype (only bf16 / fp16)";
    case RFA_ERR_HEAD_DIM: return "head_dim must be a multiple of 8 and <= 256";
    case RFA_ERR_HEADS: return "nheads must be a positive multiple of nheads_k";
    case RFA_ERR_SHAPE: return "negative or inconsistent extent";
    case RFA_ERR_ALIGN: return "pointer/stride violates the 16-byte alignment contract";
    case RFA_ERR_LAUNCH: return "HIP kernel launch failed";
    case RFA_ERR_ARGS: return "inconsistent flag / pointer combination (or dropout together with a window, dropout_p outside [0, 1))";
    case RFA_ERR_ATTR: return "could not opt the kernel into its dynamic LDS size on this device";
    default: return "unknown rfa status";
  }
}

// ---- launch plans (round 6: a cost estimate instead of shape thresholds) ------------------------------------------------
// Rounds 2-5 chose the forward form and the dK/dV plan with constants found on a handful of shapes (`wgs8 >= 112 &&
// wgs8 <= 192 && tiles >= 128`, `wgs128 * (ns + 1) <= 640`, `sq / (ns + 1) >= 2048`, ...).  tools/plan_sweep.py, which times
// the chosen form against every form that can be forced on BOTH sides of those boundaries, found the cliffs the round-5
// review predicted: forward launches of 96 / 160 / 192 / 224 workgroups against long key chains 8 - 26 % behind the best
// form, and backward launches with 1 - 4 K/V heads (a llama3 head group, any GQA model with few K/V heads) up to 1.9 x
// behind (profiles/r06_plan_sweep_before.md).  Both plans now come from ONE estimate of a launch's makespan:
//     workgroups are jobs of `tiles + kPlanWgOverhead` tile-times (prologue, epilogue, fill / drain), dealt heaviest
//     first (the kernels' blockIdx order) to the chip's 256 workgroup slots (one 8-wave workgroup per CU);
//     makespan x (1 + 0.4 busy^2): the tile time grows with the fraction of the chip that is busy (the board clocks to
//     its power cap: 1.33 us per forward tile with 96 CUs busy, 1.85 with 256 — profiles/r06_power_limiters.md);
//     + kPlanSecondPass tile-times when the shares' partials need a second kernel (combine / reduce).
// The plan is the candidate with the smallest estimate (ties: fewer shares).  A pure function of the call's shapes; the
// result is memoised per shape (a 256-entry table of atomics: no lock, no allocation).
// Fitted on / checked against tools/plan_sweep.py --dump (63 shapes x every forced form): worst chosen / best 1.06 where
// the old rules reached 1.26 (forward) and 1.90 (backward); tests/test_gpu_plan_rules.py keeps it that way.
namespace {

constexpr int kPlanSlots = 256;            // CUs: one 8-wave workgroup each (LDS and registers allow no second one)
constexpr double kPlanBusyGain = 0.4;

// makespan of `n` jobs (sizes[i] tile-times each + ovh), heaviest first, on kPlanSlots equal machines; *work = sum of all
double plan_makespan(std::vector<int>& sizes, double ovh, double* work) {
  std::sort(sizes.begin(), sizes.end(), [](int x, int y) { return x > y; });
  double total = 0;
  for (int v : sizes) total += v + ovh;
  *work = total;
  if (sizes.empty()) return 0;
  if ((int)sizes.size() <= kPlanSlots) return sizes[0] + ovh;
  // min-heap of machine loads
  std::vector<double> load(kPlanSlots, 0.0);
  auto cmp = [](double x, double y) { return x > y; };
  for (int v : sizes) {
    std::pop_heap(load.begin(), load.end(), cmp);
    load.back() += v + ovh;
    std::push_heap(load.begin(), load.end(), cmp);
  }
  double mk = 0;
  for (double l : load) mk = l > mk ? l : mk;
  return mk;
}
double plan_cost(std::vector<int>& sizes, double ovh, double second_pass) {
  double work = 0;
  const double mk = plan_makespan(sizes, ovh, &work);
  if (mk <= 0) return 0;
  double busy = work / (mk * kPlanSlots);
  busy = busy > 1 ? 1 : busy;
  return mk * (1 + kPlanBusyGain * busy * busy) + second_pass;
}

// memo: (56-bit hash of the shape key) << 8 | plan code; 0 = empty
std::atomic<uint64_t> g_plan_memo[256];
inline uint64_t plan_hash(const int64_t* k, int n) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < n; ++i) {
    h ^= (uint64_t)k[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 33;
  }
  h &= 0x00FFFFFFFFFFFFFFull;
  return h ? h : 1;
}
inline bool plan_lookup(uint64_t h, int* code) {
  const uint64_t w = g_plan_memo[h & 255].load(std::memory_order_relaxed);
  if ((w >> 8) != h) return false;
  *code = (int)(w & 255);
  return true;
}
inline void plan_store(uint64_t h, int code) { g_plan_memo[h & 255].store((h << 8) | (uint64_t)(code & 255), std::memory_order_relaxed); }

}  // namespace

// Forward plan: query rows per workgroup (256 = 8 waves, 128 = 4 waves, two workgroups per CU) and split-KV shares.
//   * many workgroups (>= 384 of 256 rows: 1.5 rounds of the chip and more): no shares; 128-row workgroups for sequences
//     <= 1024 (+6 % at 1024, +13 % at 512, -2 % from 2048 on: round 4)
//   * sequences <= 1024 otherwise: the 128-row form, shares by the round-4 rule (fill 512 slots + 25 %, >= 32 tiles each)
//   * very few workgroups (<= 48 of 256 rows: one or two llama3 head groups) against >= 64 key tiles: the 128-row form,
//     shares until 256 workgroups exist (>= 16 tiles each) — a lone 4-wave workgroup walks its tiles 1.4 x faster than an
//     8-wave one does twice the rows (0.96 vs 1.33 us per tile), and there are CUs to spare
//   * fewer than 64 key tiles, or 257 .. 383 workgroups of 256 rows: the 128-row form, no shares (round 4's rules, confirmed
//     by the round-6 sweep)
//   * everything between: the 256-row form with the share count of the smallest estimated makespan
struct FwdPlan { int rows, ns, persist; };
// compute units of the current device (the persistent forward's grid; 256 when no device can be asked: plan queries on a host)
static int device_cus() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return kPlanSlots;
  }
  int v = cached[dev & 63].load(std::memory_order_relaxed);
  if (v > 0) return v;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    n = kPlanSlots;
  }
  cached[dev & 63].store(n, std::memory_order_relaxed);
  return n;
}
// the persistent 256-row forward (rfa_fwd.hip: fwd_persist_kernel): what a call must look like
static bool fwd_persist_eligible(const rfa_fwd_args* a) {
  if (a->D != kHeadDim || a->cu_seqlens_q != nullptr || a->out_acc != nullptr || a->dropout_p > 0.f) return false;
  if (a->window && (a->window_left >= 0 || (a->window_right >= 0 && !a->causal))) return false;
  if (a->B <= 0 || a->Sq <= 0 || a->Sk <= 0) return false;
  return eff_len(a->Sk, a->k_half) >= eff_len(a->Sq, a->q_half);
}
static FwdPlan fwd_plan_base(const rfa_fwd_args* a);
// Measured (profiles/r06_persistent_forward.md; bit-identical to the 8 x 32 form everywhere): at the headline the seam
// overlap buys nothing (0.4926 against 0.4905 ms: the dispatcher's own overlap of a finishing and a starting workgroup is as
// good), from 16384 rows on the static deal loses 1 %; sequences of 1024 .. 2048 rows in launches of >= 2 items per CU gain
// 1 - 4 % over the best other form (B 8 x S 1024: 0.0861 against 0.0898 for the 128-row form and 0.0941 for the 256-row one).
static FwdPlan fwd_plan(const rfa_fwd_args* a) {
  FwdPlan pl = fwd_plan_base(a);
  pl.persist = 0;
  if (!fwd_persist_eligible(a) || a->kv_nsplit > 1) return pl;
  const int sq = eff_len(a->Sq, a->q_half);
  const int64_t items = (int64_t)a->B * a->H * ((sq + 255) / 256);
  const bool by_name = a->fwd_form == RFA_FWD_P8x32;
  bool by_rule = a->fwd_form == RFA_FWD_AUTO && pl.ns == 1 && items >= 2 * (int64_t)kPlanSlots && sq >= 1024 && sq <= 2048;
  if (by_rule) {
    // the static deal must balance: the tiles every workgroup gets from its passes (every other pass reversed, as
    // fwd_persist_kernel deals them) within 4 % of the mean — whole pairs of passes over a causal launch do exactly; a
    // partial last pass does not (S 1280 x B 6: 3.75 passes, measured 27 % behind the 256-row form)
    const int sk = eff_len(a->Sk, a->k_half), off = sk - sq, nq = (sq + 255) / 256, grid = device_cus();
    std::vector<int64_t> load((size_t)grid, 0);
    int64_t total = 0;
    for (int64_t idx = 0; idx < items; ++idx) {
      const int64_t pass = idx / grid, r = idx % grid;
      const int64_t w = (pass & 1) ? grid - 1 - r : r;
      const int64_t blk = idx / ((int64_t)a->H);               // (kv head, query head in group) are the fastest digits
      const int qblk = nq - 1 - (int)(blk % nq);                // (batch slowest: RFA_BATCH_FAST_Q = 0)
      const int qend = (qblk + 1) * 256 < sq ? (qblk + 1) * 256 : sq;
      int kmax = sk;
      if (a->causal && qend + off < kmax) kmax = qend + off;
      const int nt = (kmax + 63) / 64 + 8;                     // + the estimate's per-item overhead
      load[(size_t)w] += nt;
      total += nt;
    }
    int64_t mx = 0;
    for (int64_t l : load) mx = l > mx ? l : mx;
    by_rule = mx * grid <= total + total / 25;
  }
  if (by_name || by_rule) {
    pl.rows = 256;
    pl.ns = 1;
    pl.persist = 1;
  }
  return pl;
}
static FwdPlan fwd_plan_base(const rfa_fwd_args* a) {
  FwdPlan pl{fwd_qrows_per_block(), 1, 0};
  const bool win = a->window && (a->window_left >= 0 || (a->window_right >= 0 && !a->causal));
  if ((a->D != kHeadDim && a->D != kHeadDim / 2) || win || a->dropout_p > 0.f) return pl;   // one form, no shares
  if (a->B <= 0 || a->Sq <= 0 || a->Sk <= 0) return pl;
  const int sq = eff_len(a->Sq, a->q_half), sk = eff_len(a->Sk, a->k_half);
  const int64_t w8 = (int64_t)a->B * a->H * ((sq + 255) / 256), w128 = (int64_t)a->B * a->H * ((sq + 127) / 128);
  const int tiles = (sk + 63) / 64;
  const bool can_split = a->kv_nsplit != 1;
  // a form / share count asked for by name (tuning, tests) is taken as given
  if (a->fwd_form == RFA_FWD_8x32 || a->fwd_form == RFA_FWD_4x32 || a->kv_nsplit > 1) {
    pl.rows = a->fwd_form == RFA_FWD_4x32 ? 128 : 256;
    pl.ns = a->kv_nsplit > 1 ? (a->kv_nsplit > 8 ? 8 : a->kv_nsplit) : 1;
    if (a->fwd_form == RFA_FWD_AUTO) {                      // shares forced, form free: the form the rules below would take
      rfa_fwd_args b = *a;
      b.kv_nsplit = 0;
      pl.rows = fwd_plan_base(&b).rows;
    }
    return pl;
  }
  if (w8 >= 384) {
    pl.rows = sq <= 1024 ? 128 : 256;
    return pl;
  }
  if (sq <= 1024) {
    pl.rows = 128;
    if (can_split && w128 < 384 && tiles >= 64) {
      int ns = 1;
      while (ns < 8 && w128 * (ns + 1) <= 640 && tiles / (ns + 1) >= 32) ++ns;
      pl.ns = ns;
    }
    return pl;
  }
  if (w8 <= 48 && tiles >= 64) {
    pl.rows = 128;
    if (can_split) {
      int ns = (int)(256 / w128);
      ns = ns < 1 ? 1 : (ns > 8 ? 8 : ns);
      while (ns > 1 && tiles / ns < 16) --ns;
      pl.ns = ns;
    }
    return pl;
  }
  if (tiles < 64) {
    // short key chains on an under-filled chip (S 2048, 16 heads: 128 workgroups of <= 32 tiles): nothing to share — shares of
    // < 32 tiles do not pay for the combine pass — and the 128-row form's two workgroups per CU balance the causal triangle
    // (round 4's rule; round 6 re-measured: 0.0385 ms against 0.0455 with two shares of 256-row workgroups, 0.0493 with one)
    pl.rows = 128;
    return pl;
  }
  if (w8 > 256) {
    // 257 .. 383 workgroups of 256 rows: one round of the chip plus a short second one.  The 128-row form (two workgroups
    // per CU, a light one's partner speeds up when it leaves) balances that better than any share count of the 256-row
    // form: S 6144, 12 heads, causal 0.130 vs 0.144 ms (round 4's rule for this range, kept: profiles/r06_plan_sweep_after.md)
    pl.rows = 128;
    return pl;
  }
  pl.rows = 256;
  if (!can_split) return pl;
  const int64_t key[8] = {1, a->B, a->H, sq, sk, a->causal ? 1 : 0, 0, 0};
  const uint64_t h = plan_hash(key, 8);
  int code;
  if (plan_lookup(h, &code)) {
    pl.ns = code;
    return pl;
  }
  // workgroups of the 256-row form: block i of every (batch, head) sees the key tiles below its causal edge
  const int off = sk - sq, nq = (sq + 255) / 256;
  const int64_t mult = (int64_t)a->B * a->H;
  double best = -1;
  static const int cand[] = {1, 2, 3, 4, 6, 8};
  for (int ns : cand) {
    if (ns > 1 && tiles / ns < 16) continue;
    std::vector<int> sizes;
    sizes.reserve((size_t)(nq * ns * mult));
    for (int i = 0; i < nq; ++i) {
      const int qend = (i + 1) * 256 < sq ? (i + 1) * 256 : sq;
      int kmax = sk;
      if (a->causal && qend + off < kmax) kmax = qend + off;
      const int nt = kmax > 0 ? (kmax + 63) / 64 : 0;
      int chunk = (nt + ns - 1) / ns;
      chunk = (chunk + 1) / 2 * 2;                           // (shares are a multiple of the LDS ring depth long: fwd_kernel)
      for (int sidx = 0; sidx < ns; ++sidx) {
        const int lo = sidx * chunk, hi = lo + chunk < nt ? lo + chunk : nt;
        const int n = hi > lo ? hi - lo : 0;
        for (int64_t m = 0; m < mult; ++m) sizes.push_back(n);
      }
    }
    const double c = plan_cost(sizes, 8.0, ns > 1 ? 8.0 : 0.0);
    if (best < 0 || c < 0.98 * best) {
      best = c;
      pl.ns = ns;
    }
  }
  plan_store(h, pl.ns);
  return pl;
}
static int fwd_kv_nsplit(const rfa_fwd_args* a) { return fwd_plan(a).ns; }
static int64_t fwd_rows_total(const rfa_fwd_args* a) {
  if (a->cu_seqlens_q != nullptr) return a->total_q > 0 ? a->total_q : 0;
  return (int64_t)a->B * a->Sq;
}

int64_t rfa_fwd_workspace_bytes(const rfa_fwd_args* a, int32_t* nsplit) {
  if (nsplit) *nsplit = 1;
  if (!a || check_common(a->dtype, a->H, a->Hk, a->D, a->B)) return 0;
  const int ns = fwd_kv_nsplit(a);
  const int64_t rows = fwd_rows_total(a);
  if (ns <= 1 || rows <= 0) return 0;
  if (nsplit) *nsplit = ns;
  return (int64_t)ns * rows * a->H * ((int64_t)a->D + 1) * 4;      // fp32 partial outs + partial lse
}

int rfa_fwd(const rfa_fwd_args* a, void* stream) {
  if (!a) return RFA_ERR_NULL;
  int rc = check_common(a->dtype, a->H, a->Hk, a->D, a->B);
  if (rc) return rc;
  if (a->Sq < 0 || a->Sk < 0) return RFA_ERR_SHAPE;
  if (a->B == 0 || a->Sq == 0) return RFA_OK;
  if (!a->q || !a->k || !a->v) return RFA_ERR_NULL;
  if (a->out_acc) {
    if (!a->lse_acc) return RFA_ERR_NULL;
  } else if (!a->out || !a->lse) {
    return RFA_ERR_NULL;
  }
  if ((a->cu_seqlens_q == nullptr) != (a->cu_seqlens_k == nullptr)) return RFA_ERR_ARGS;
  if (!drop_args_ok(a->dropout_p, a->window, a->window_left, a->window_right, a->causal)) return RFA_ERR_ARGS;
  if (!aligned16(a->q) || !aligned16(a->k) || !aligned16(a->v)) return RFA_ERR_ALIGN;
  if (!stride_ok(a->q_st, 2) || !stride_ok(a->k_st, 2) || !stride_ok(a->v_st, 2)) return RFA_ERR_ALIGN;
  if (a->out_acc) {
    if (!aligned16(a->out_acc) || !stride_ok(a->out_acc_st, 4)) return RFA_ERR_ALIGN;
  } else {
    if (!aligned16(a->out) || !stride_ok(a->out_st, 2)) return RFA_ERR_ALIGN;
  }
  FwdParams p{};
  p.q = a->q; p.k = a->k; p.v = a->v;
  p.out = a->out; p.lse = a->lse;
  p.out_acc = a->out_acc; p.lse_acc = a->lse_acc;
  p.cu_q = a->cu_seqlens_q; p.cu_k = a->cu_seqlens_k;
  p.q_st = cv(a->q_st); p.k_st = cv(a->k_st); p.v_st = cv(a->v_st);
  p.out_st = cv(a->out_st); p.out_acc_st = cv(a->out_acc_st);
  p.lse_batch = a->lse_batch; p.lse_head = a->lse_head;
  p.lse_acc_batch = a->lse_acc_batch; p.lse_acc_head = a->lse_acc_head;
  p.B = a->B; p.H = a->H; p.Hk = a->Hk; p.D = a->D; p.Sq = a->Sq; p.Sk = a->Sk;
  p.q_half = a->q_half; p.k_half = a->k_half;
  p.causal = a->causal ? 1 : 0; p.acc_init = a->acc_init ? 1 : 0;
  p.wl = (a->window && a->window_left >= 0) ? a->window_left : -1;
  p.wr = a->causal ? 0 : ((a->window && a->window_right >= 0) ? a->window_right : -1);
  p.scale = a->softmax_scale;
  p.drop_keep = drop_threshold(a->dropout_p);
  p.drop_scale = drop_rescale(a->dropout_p);
  p.drop_seed = a->dropout_seed;
  p.q_pos0 = (unsigned)a->q_pos_offset; p.k_pos0 = (unsigned)a->k_pos_offset; p.head0 = (unsigned)a->head_offset;
  // 256 query rows per workgroup (8 waves) or 128 (4 waves, two workgroups per CU), and the split-KV shares: fwd_plan()
  const FwdPlan plan = fwd_plan(a);
  int rows = plan.rows;
  // split-KV (needs the caller's workspace): the 128-row form with the key tiles of a workgroup shared by kv_nsplit
  const int ns = (a->workspace != nullptr && fwd_rows_total(a) > 0) ? plan.ns : 1;
  if (a->kv_nsplit < 0) return RFA_ERR_ARGS;
  if (ns <= 1 && plan.ns > 1) {                              // planned with shares but called without a workspace: the unsplit plan
    rfa_fwd_args b = *a;
    b.kv_nsplit = 1;
    rows = fwd_plan(&b).rows;
  }
  CombineParams cb{};
  if (ns > 1) {
    if (!aligned16(a->workspace)) return RFA_ERR_ALIGN;
    const int64_t rt = fwd_rows_total(a);
    // partial layout: out (ns, rows_total, H, D) fp32, lse (ns, [B,] H, rows) fp32 behind it — addressed by the kernel
    // through the accumulate-mode fields (the call's own accumulators, if any, are the combine kernel's business)
    p.kv_nsplit = ns;
    p.part_out = (float*)a->workspace;
    p.part_lse = p.part_out + (int64_t)ns * rt * a->H * a->D;
    p.part_out_split = rt * a->H * a->D;
    p.part_lse_split = rt * a->H;
    p.out_acc_st = Strides{a->cu_seqlens_q ? 0 : (int64_t)a->Sq * a->H * a->D, (int64_t)a->H * a->D, (int64_t)a->D};
    p.lse_acc_batch = a->cu_seqlens_q ? 0 : (int64_t)a->H * a->Sq;
    p.lse_acc_head = a->cu_seqlens_q ? rt : (int64_t)a->Sq;
    cb.part_out = p.part_out; cb.part_lse = p.part_lse; cb.part_st = p.out_acc_st;
    cb.part_lse_batch = p.lse_acc_batch; cb.part_lse_head = p.lse_acc_head;
    cb.part_out_split = p.part_out_split; cb.part_lse_split = p.part_lse_split;
    cb.nsplit = ns;
    cb.out = a->out; cb.lse = a->lse; cb.out_acc = a->out_acc; cb.lse_acc = a->lse_acc;
    cb.out_st = cv(a->out_st); cb.out_acc_st = cv(a->out_acc_st);
    cb.lse_batch = a->lse_batch; cb.lse_head = a->lse_head;
    cb.lse_acc_batch = a->lse_acc_batch; cb.lse_acc_head = a->lse_acc_head;
    cb.cu_q = a->cu_seqlens_q;
    cb.B = a->B; cb.H = a->H; cb.D = a->D; cb.Sq = a->Sq; cb.q_half = a->q_half; cb.acc_init = a->acc_init ? 1 : 0;
  }
  p.qrows = rows;
  p.nqblk = (eff_len(a->Sq, a->q_half) + rows - 1) / rows;
  p.persist_grid = (plan.persist && rows == 256 && ns <= 1 && plan.ns <= 1) ? device_cus() : 0;
  if (a->fwd_form < RFA_FWD_AUTO || a->fwd_form > RFA_FWD_P8x32 || a->fwd_form == RFA_FWD_RETIRED_2) return RFA_ERR_ARGS;
  if (a->D > kHeadDim) return launch_status(launch_fwd_big(p, a->dtype, (hipStream_t)stream));
  if (int rc2 = launch_fwd(p, a->dtype, (hipStream_t)stream)) return launch_status(rc2);
  if (ns > 1 && launch_combine(cb, a->dtype, (hipStream_t)stream)) return RFA_ERR_LAUNCH;
  return RFA_OK;
}

int rfa_bwd_preprocess(const rfa_bwd_preprocess_args* a, void* stream) {
  if (!a) return RFA_ERR_NULL;
  int rc = check_common(a->dtype, a->H, a->H, a->D, a->B);
  if (rc) return rc;
  if (a->Sq < 0) return RFA_ERR_SHAPE;
  if (a->B == 0 || a->Sq == 0) return RFA_OK;
  if (!a->dout || !a->out || !a->delta) return RFA_ERR_NULL;
  if (!aligned16(a->dout) || !aligned16(a->out) || !stride_ok(a->dout_st, 2) || !stride_ok(a->out_st, 2))
    return RFA_ERR_ALIGN;
  PreParams p{};
  p.dout = a->dout; p.out = a->out; p.delta = a->delta; p.cu_q = a->cu_seqlens_q;
  p.dout_st = cv(a->dout_st); p.out_st = cv(a->out_st);
  p.delta_batch = a->delta_batch; p.delta_head = a->delta_head;
  p.B = a->B; p.H = a->H; p.D = a->D; p.Sq = a->Sq; p.q_half = a->q_half;
  return launch_preprocess(p, a->dtype, (hipStream_t)stream) ? RFA_ERR_LAUNCH : RFA_OK;
}

// The dK/dV kernel sums the query heads of a K/V group itself, so GQA needs no scratch.  A workspace
// (bf16/fp16 partials, (rows, Hk, D) x 2) is only used when the result is ADDED to fp32 accumulators or
// when compute and reduction are issued as two phases (ring steps: compute overlaps the arrival of the
// accumulators).
static bool bwd_single_phase(const rfa_bwd_args* a) {
  return (a->phases & (RFA_BWD_COMPUTE | RFA_BWD_REDUCE)) == 0;
}
// single-phase call whose dK/dV accumulators are overwritten: the kernel stores fp32 directly
static bool bwd_kv_direct(const rfa_bwd_args* a) {
  return a->dk_acc != nullptr && bwd_single_phase(a) && (a->acc_init || (a->phases & RFA_BWD_KV_OVERWRITE));
}

// Which dK/dV kernel form a call runs — a pure function of the call's arguments (shapes + the dkdv_form /
// dkdv_nsplit fields), so that rfa_bwd_workspace_bytes, a BWD_COMPUTE and its BWD_REDUCE call agree: the 256-key
// workgroup form needs head dim 128 or 64 (exactly) and no window; its workgroups are B * Hk * ceil(Sk / 256), so the Q/dO tile
// range of a key block is shared by up to 4 workgroups until the launch has about two workgroups per CU (a causal
// launch needs that many for its heavy-first order to balance), each with at least 8 tiles.  Launches that stay
// small even so run the 128-key form (twice the workgroups).
struct DkdvPlan { int wide, nsplit, bal; };     // bal: the balanced causal schedule of the 256-key form (dkdv_kernel kBal)
struct DsChunks { int nchunks, hc, gc; int64_t chunk_bytes; };     // hc K/V heads x gc query heads per K/V head per chunk
static DsChunks bwd_ds_chunking(const rfa_bwd_args* a);
// The dK/dV plan by estimated makespan (see "launch plans" above).  Jobs: one workgroup per (batch, K/V head, key block
// [, share of its query tiles]); a key block that starts at key k0 walks the 64-row Q/dO tiles from its causal start
// max(0, k0 - off) to the end of the sequence for each of the G query heads of its K/V head; `ns` workgroups take every
// ns-th of those tiles.  The 128-key form has twice the workgroups, each tile-step with half the MFMAs but the same
// LDS-DMA staging and barrier: 0.7 of the 256-key form's tile time (fitted).  Shares need the fp32 partials + reduce_kernel
// pass: 12 tile-times.  Returns wide (0 / 1) and the share count; only_wide: the best share count of the 256-key form.
// Packed (cu_seqlens) input: the sequences' lengths live on the device — the estimate takes B sequences of the mean
// length total / B (bounded by max_seqlen), which is what the old rule's "about total_k / 256 key blocks" did.
static int bwd_dkdv_cost_plan(const rfa_bwd_args* a, int hk_launch, bool only_wide, int* ns_out, int* bal_out = nullptr) {
  int sk = eff_len(a->Sk, a->k_half), sq = eff_len(a->Sq, a->q_half);
  if (a->cu_seqlens_k != nullptr && a->total_k > 0 && a->B > 0) {
    const int64_t mean_k = (a->total_k + a->B - 1) / a->B;
    const int64_t mean_q = a->total_q > 0 ? (a->total_q + a->B - 1) / a->B : mean_k;
    sk = (int)(mean_k < sk ? mean_k : sk);
    sq = (int)(mean_q < sq ? mean_q : sq);
    if (a->k_half != RFA_HALF_FULL) sk = (sk + 1) / 2;
    if (a->q_half != RFA_HALF_FULL) sq = (sq + 1) / 2;
  }
  const bool big = a->D > kHeadDim;          // rfa_bigd.hip: 128-key workgroups (one wave per SIMD), 32-row Q/dO tiles; ONE form
  const int G = a->H / a->Hk;
  const int64_t key[8] = {2, a->B, hk_launch, G, sq, sk, a->causal ? 1 : 0, (big ? 4 : 0) | (a->D == 64 ? 2 : 0) | (only_wide ? 1 : 0)};
  const uint64_t h = plan_hash(key, 8);
  int code;
  if (plan_lookup(h, &code)) {
    *ns_out = code & 15;
    if (bal_out) *bal_out = (code >> 5) & 1;
    return (code >> 4) & 1;
  }
  if (bal_out) *bal_out = 0;
  const int off = sk - sq;
  const int64_t mult = (int64_t)a->B * hk_launch;
  const int trows = big ? 32 : 64;
  const int ntq = (sq + trows - 1) / trows;
  auto estimate = [&](int keys, int ns) {
    const int nkb = (sk + keys - 1) / keys;
    std::vector<int> sizes;
    sizes.reserve((size_t)(nkb * ns * mult));
    for (int kb = 0; kb < nkb; ++kb) {
      int qfirst = a->causal ? kb * keys - off : 0;
      qfirst = qfirst < 0 ? 0 : qfirst;
      const int nt = ntq - qfirst / trows > 0 ? ntq - qfirst / trows : 0;
      for (int sidx = 0; sidx < ns; ++sidx) {
        const int n = nt > sidx ? (nt - 1 - sidx) / ns + 1 : 0;        // tiles nt-1-sidx, nt-1-sidx-ns, ... (dkdv_kernel)
        for (int64_t m = 0; m < mult; ++m) sizes.push_back(n * G);
      }
    }
    // constants: the wide head dims keep round 6's first fit; the 128-wide kernels were re-fitted on the 80 shapes of
    // profiles/r06_plan_sweep_final.md after the batch index became the fastest digit of the dK/dV work decode (which made
    // the heaviest-first assumption of this estimate TRUE for multi-batch launches and the 128-key form — twice the
    // workgroups — the best plan of several of them): worst chosen / best 1.03, mean 1.001
    const double ovh = big ? 12.0 : (keys == 128 ? 6.0 : 8.0);
    double work = 0;
    const double mk = plan_makespan(sizes, ovh, &work);
    if (mk <= 0) return 0.0;
    double busy = work / (mk * kPlanSlots);
    busy = busy > 1 ? 1 : busy;
    return mk * (keys == 128 && !big ? 0.65 : 1.0) * (1 + (big ? kPlanBusyGain : 0.3) * busy * busy) + (ns > 1 ? (big ? 12.0 : 8.0) : 0.0);
  };
  double best = -1;
  int wide = 0, best_ns = 1;
  if (big) {
    for (int ns = 1; ns <= 8; ++ns) {
      if (ns == 5 || ns == 7 || (ns > 1 && ntq / ns < 8)) continue;     // (shares of at least 8 tiles of 32 rows, as rounds 3-5)
      const double c = estimate(128, ns);
      if (best < 0 || c < 0.98 * best) {
        best = c;
        best_ns = ns;
      }
    }
    plan_store(h, best_ns);
    *ns_out = best_ns;
    return 0;
  }
  if (!only_wide) best = estimate(128, 1);
  for (int ns = 1; ns <= 8; ++ns) {
    if (ns == 5 || ns == 7) continue;
    if (ns > 1 && (sq < 1024 ? ns > 2 : ntq / ns < 2)) continue;       // (shares of at least 2 tiles; short sequences: two at most)
    const double c = estimate(256, ns);
    if (best < 0 || c < 0.98 * best) {
      best = c;
      wide = 1;
      best_ns = ns;
    }
  }
  // the balanced causal schedule, where the SHAPE allows it (the caller adds the call-level conditions: bwd_bal_eligible):
  // B * Hk * nkb equal workgroups of (T/2 + 2) G tile-times; half of them cross one key-block seam (a second prologue /
  // epilogue) and every pair exchanges one partial: 14 tile-times of overhead on average instead of 8, no second pass
  int bal = 0;
  if (!only_wide && !big && a->causal && sq == sk && sk >= 512 && sk % 512 == 0) {
    const int nkb = sk / 256;
    std::vector<int> sizes((size_t)(mult * nkb), (2 * nkb + 2) * G);
    double work = 0;
    const double mk = plan_makespan(sizes, 14.0, &work);
    double busy = work / (mk * kPlanSlots);
    busy = busy > 1 ? 1 : busy;
    bal = mk * (1 + 0.3 * busy * busy) < 0.98 * best ? 1 : 0;
  }
  plan_store(h, (bal << 5) | (wide << 4) | best_ns);
  *ns_out = best_ns;
  if (bal_out) *bal_out = bal;
  return wide;
}

// The balanced causal schedule (rfa_bwd.hip: dkdv_kernel kBal): every workgroup of a dense causal self-attention block
// does the same T/2 + 2 tiles' worth of work, only the lower half of the key blocks is shared (by exactly two workgroups
// that add their partials between themselves): no nsplit fp32 partials per key, no reduce_kernel pass.  Needs the whole
// call in one launch that writes the final dK/dV (single phase, plain outputs or overwritten accumulators, all heads
// at once) and a sequence of whole PAIRS of 256-key blocks.
static bool bwd_bal_eligible(const rfa_bwd_args* a, bool whole_call) {
  if ((a->D != kHeadDim && a->D != 64) || !a->causal || a->cu_seqlens_q != nullptr || a->dropout_p > 0.f) return false;
  if (a->window && (a->window_left >= 0 || a->window_right >= 0)) return false;
  const int lq = eff_len(a->Sq, a->q_half), lk = eff_len(a->Sk, a->k_half);
  if (lq != lk || lk < 512 || (lk % 512) != 0) return false;
  if (!bwd_single_phase(a) || (a->phases & (RFA_BWD_SKIP_DKDV | RFA_BWD_SKIP_DQ))) return false;
  if (a->dk_acc != nullptr && !bwd_kv_direct(a)) return false;
  return whole_call;                           // (not a head-group chunk of a chunked dS hand-off)
}
// hk_launch: K/V heads of ONE dK/dV launch (= Hk, or the K/V heads of a chunk of a chunked dS hand-off)
static DkdvPlan bwd_dkdv_plan_for(const rfa_bwd_args* a, int hk_launch, bool whole_call) {
  DkdvPlan pl{0, 1, 0};
  if (a->D > kHeadDim) {
    // rfa_bigd.hip: 128-key workgroups that occupy a CU each (one wave per SIMD); the tile range of a key block is shared by
    // the number of workgroups with the smallest estimated makespan (fp32 partials, summed by reduce_kernel; round 6: the
    // same estimate as the 128-wide kernels' plan instead of "until about two workgroups per CU")
    int ns = 1;
    (void)bwd_dkdv_cost_plan(a, hk_launch, false, &ns);
    if (a->dkdv_nsplit > 0) ns = a->dkdv_nsplit > 8 ? 8 : a->dkdv_nsplit;
    pl.nsplit = ns;
    return pl;
  }
  const bool win = a->window && (a->window_left >= 0 || (a->window_right >= 0 && !a->causal));
  if (a->dkdv_form == RFA_DKDV_128 || (a->D != kHeadDim && a->D != 64) || win || a->dropout_p > 0.f) return pl;
  // a plan asked for by name (tuning, tests; the REDUCE half of a two-phase call repeats its COMPUTE half's plan)
  if (a->dkdv_form == RFA_DKDV_BAL && bwd_bal_eligible(a, whole_call)) {
    pl.wide = pl.bal = 1;
    return pl;
  }
  if (a->dkdv_form == RFA_DKDV_256 || a->dkdv_nsplit > 0) {
    pl.wide = 1;
    if (a->dkdv_nsplit > 0) {
      pl.nsplit = a->dkdv_nsplit > 8 ? 8 : a->dkdv_nsplit;
    } else {
      rfa_bwd_args b = *a;                                   // form forced, shares free: the count the estimate gives the 256-key form
      b.dkdv_form = RFA_DKDV_AUTO;
      int best_ns = 1;
      (void)bwd_dkdv_cost_plan(&b, hk_launch, true, &best_ns);
      pl.nsplit = best_ns;
    }
    return pl;
  }
  int ns = 1, bal = 0;
  pl.wide = bwd_dkdv_cost_plan(a, hk_launch, false, &ns, &bal);
  pl.nsplit = pl.wide ? ns : 1;
  if (bal && bwd_bal_eligible(a, whole_call)) {
    pl.wide = pl.bal = 1;
    pl.nsplit = 1;
  }
  return pl;
}
static DkdvPlan bwd_dkdv_plan(const rfa_bwd_args* a) {
  const DsChunks ch = bwd_ds_chunking(a);
  return bwd_dkdv_plan_for(a, ch.nchunks > 1 ? ch.hc : a->Hk, ch.nchunks <= 1);
}
static bool bwd_needs_ws(const rfa_bwd_args* a) {
  if (!bwd_single_phase(a)) return true;
  if (bwd_dkdv_plan(a).bal) return true;
  if (bwd_dkdv_plan(a).nsplit > 1) return true;
  if (bwd_ds_chunking(a).gc < a->H / a->Hk) return true;      // query-head fractions of a K/V head accumulate in fp32 partials
  return a->dk_acc != nullptr && !bwd_kv_direct(a);
}

static int64_t bwd_ds_head_blocks(const rfa_bwd_args* a);
static bool bwd_spill_eligible(const rfa_bwd_args* a) {
  if (!((a->cu_seqlens_q == nullptr) == (a->cu_seqlens_k == nullptr) && a->D >= kHeadDim && a->D <= 2 * kHeadDim &&
        a->B > 0 && a->Sq > 0 && a->Sk > 0 && !(a->dropout_p > 0.f) &&
        !(a->window && (a->window_left >= 0 || (a->window_right >= 0 && !a->causal)))))
    return false;
  // one head's share of the scratch stays below 4 GiB: the dK/dV kernel addresses a dS block by a 32-bit scalar offset from
  // its head's base (a single dense causal sequence of 65536 rows — 4.3 GB per head — runs the 7-GEMM form)
  return bwd_ds_head_blocks(a) < ((int64_t)1 << 21);
}

// dS scratch rows: packed triangular for dense causal calls (block (qt, kb) is visited iff kb < qt + c), else
// rectangular, expressed as c >= nKb (rfa_kernels.hpp: ds_row_off)
static int bwd_ds_c(const rfa_bwd_args* a) {
  const int nkb = ds_blocks(a->Sk, a->k_half);
  if (!a->causal || a->cu_seqlens_q != nullptr) return nkb;
  const int off = eff_len(a->Sk, a->k_half) - eff_len(a->Sq, a->q_half);
  const int c = ((31 + off) >> 5) + 1;                   // (arithmetic shift: off may be negative)
  return c > nkb ? nkb : c;
}

// blocks of the dS scratch per head: dense = one (batch, head) — rows rectangular or packed triangular —, packed
// (cu_seqlens) = ALL sequences of a head: total_q / 32 + B rows of ceil(max_seqlen_k / 32) blocks (rfa_kernels.hpp:
// ds_rowpart — bounded by the packed row count instead of B x the longest sequence)
static int64_t bwd_ds_head_blocks(const rfa_bwd_args* a) {
  const int nkb = ds_blocks(a->Sk, a->k_half);
  if (a->cu_seqlens_q != nullptr) {
    const int64_t tq = a->total_q > 0 ? a->total_q : a->total_k;
    return ((tq >> 5) + a->B) * (int64_t)nkb;
  }
  return ds_row_off(ds_blocks(a->Sq, a->q_half), nkb, bwd_ds_c(a), 1);
}
// dS bytes of ONE query head (all batches / packed sequences)
static int64_t bwd_ds_head_bytes(const rfa_bwd_args* a) {
  return (a->cu_seqlens_q != nullptr ? 1 : (int64_t)a->B) * bwd_ds_head_blocks(a) * kDsBlockBytes;
}

int64_t rfa_bwd_ds_scratch_bytes(const rfa_bwd_args* a) {
  if (!a || !bwd_spill_eligible(a)) return 0;
  return (int64_t)a->H * bwd_ds_head_bytes(a);
}

int64_t rfa_bwd_ds_scratch_min_bytes(const rfa_bwd_args* a) {
  if (!a || !bwd_spill_eligible(a)) return 0;
  // head dim 256 (rfa_bigd.hip) has no chunked form; two-phase calls (COMPUTE / REDUCE) chunk by whole K/V heads only
  if (a->D != kHeadDim) return rfa_bwd_ds_scratch_bytes(a);
  const int G = a->H / a->Hk;
  return (bwd_single_phase(a) ? 1 : G) * bwd_ds_head_bytes(a);
}

// How the dS hand-off of a call is cut to fit the scratch it was given: all heads at once when they fit; else chunks
// of hc whole K/V heads (hc the largest divisor of Hk whose query heads fit); else — not even one K/V head's G query
// heads fit — chunks of gc query heads of ONE K/V head (gc the largest divisor of G that fits), whose dK/dV shares are
// accumulated in the fp32 partials of the workspace (kv_accum).  nchunks = 0: the 7-GEMM form.
static DsChunks bwd_ds_chunking(const rfa_bwd_args* a) {
  DsChunks none{0, a->Hk, a->H / a->Hk, 0};
  if (a->ds_scratch == nullptr || !bwd_spill_eligible(a)) return none;
  const int G = a->H / a->Hk;
  const int64_t per = bwd_ds_head_bytes(a);
  const int64_t full = per * a->H;
  const int64_t avail = a->ds_scratch_bytes > 0 ? a->ds_scratch_bytes : full;
  if (per <= 0) return none;
  if (avail >= full) return DsChunks{1, a->Hk, G, full};
  if (a->D != kHeadDim) return none;
  const int64_t fit = avail / per;                       // query heads whose dS fit
  if (fit >= G) {
    int hc = 1;
    for (int d = 1; d <= a->Hk; ++d)
      if (a->Hk % d == 0 && (int64_t)d * G <= fit) hc = d;
    return DsChunks{a->Hk / hc, hc, G, per * hc * G};
  }
  if (fit < 1 || !bwd_single_phase(a)) return none;
  int gc = 1;
  for (int d = 1; d <= G; ++d)
    if (G % d == 0 && d <= fit) gc = d;
  return DsChunks{a->Hk * (G / gc), 1, gc, per * gc};
}

int rfa_bwd_ds_chunks(const rfa_bwd_args* a, int32_t* nchunks, int32_t* kv_heads, int32_t* q_heads, int64_t* chunk_bytes) {
  if (!a) return RFA_ERR_NULL;
  if (int rc = check_common(a->dtype, a->H, a->Hk, a->D, a->B)) return rc;
  const DsChunks ch = bwd_ds_chunking(a);
  if (nchunks) *nchunks = ch.nchunks;
  if (kv_heads) *kv_heads = ch.hc;
  if (q_heads) *q_heads = ch.gc;
  if (chunk_bytes) *chunk_bytes = ch.chunk_bytes;
  return RFA_OK;
}

int rfa_bwd_plan(const rfa_bwd_args* a, int32_t* form, int32_t* nsplit, int32_t* five_gemm) {
  if (!a) return RFA_ERR_NULL;
  const DkdvPlan pl = bwd_dkdv_plan(a);
  if (form) *form = pl.bal ? RFA_DKDV_BAL : pl.wide ? RFA_DKDV_256 : RFA_DKDV_128;
  if (nsplit) *nsplit = pl.nsplit;
  if (five_gemm) *five_gemm = bwd_ds_chunking(a).nchunks > 0 ? 1 : 0;
  return RFA_OK;
}

static int64_t bal_flag_bytes(int64_t pairs) { return (pairs * 4 + 255) / 256 * 256; }

int64_t rfa_bwd_workspace_bytes(const rfa_bwd_args* a) {
  if (!a || !bwd_needs_ws(a)) return 0;
  // balanced schedule: one fp32 pair slot (dK + dV accumulators of 256 keys) per key block of the lower half of every
  // (batch, K/V head) + one flag word per pair (rounded up to 256 bytes, in front of the slots)
  if (bwd_dkdv_plan(a).bal) {
    const int64_t pairs = (int64_t)a->B * a->Hk * (eff_len(a->Sk, a->k_half) / 512);
    return bal_flag_bytes(pairs) + pairs * 2 * 256 * (int64_t)a->D * 4;
  }
  // unsplit: one io-dtype partial per element; split launches: nsplit fp32 partials
  const int ns = bwd_dkdv_plan(a).nsplit;
  const bool f32 = ns > 1 || bwd_ds_chunking(a).gc < a->H / a->Hk;
  return 2 * a->total_k * (int64_t)a->Hk * a->D * (f32 ? 4 * ns : 2);
}

int rfa_bwd(const rfa_bwd_args* a, void* stream) {
  if (!a) return RFA_ERR_NULL;
  int rc = check_common(a->dtype, a->H, a->Hk, a->D, a->B);
  if (rc) return rc;
  if (a->Sq < 0 || a->Sk < 0) return RFA_ERR_SHAPE;
  if (a->B == 0 || a->Sq == 0 || a->Sk == 0) return RFA_OK;
  if (!a->dout || !a->q || !a->k || !a->v || !a->lse || !a->delta) return RFA_ERR_NULL;
  if (!a->dq && !a->dq_acc) return RFA_ERR_NULL;
  if ((a->dk_acc == nullptr) != (a->dv_acc == nullptr)) return RFA_ERR_ARGS;
  if (!a->dk_acc && (!a->dk || !a->dv)) return RFA_ERR_NULL;
  if ((a->cu_seqlens_q == nullptr) != (a->cu_seqlens_k == nullptr)) return RFA_ERR_ARGS;
  if (a->dkdv_form < RFA_DKDV_AUTO || a->dkdv_form > RFA_DKDV_BAL || a->dkdv_nsplit < 0) return RFA_ERR_ARGS;
  if (!drop_args_ok(a->dropout_p, a->window, a->window_left, a->window_right, a->causal)) return RFA_ERR_ARGS;
  const bool ws = bwd_needs_ws(a);
  if (ws && !a->workspace) return RFA_ERR_NULL;
  if (!aligned16(a->dout) || !aligned16(a->q) || !aligned16(a->k) || !aligned16(a->v))
    return RFA_ERR_ALIGN;
  if (!stride_ok(a->dout_st, 2) || !stride_ok(a->q_st, 2) || !stride_ok(a->k_st, 2) || !stride_ok(a->v_st, 2))
    return RFA_ERR_ALIGN;
  if (a->dq_acc && (!aligned16(a->dq_acc) || !stride_ok(a->dq_acc_st, 4))) return RFA_ERR_ALIGN;
  if (!a->dq_acc && (!aligned16(a->dq) || !stride_ok(a->dq_st, 2))) return RFA_ERR_ALIGN;
  if (a->dk_acc && (!aligned16(a->dk_acc) || !aligned16(a->dv_acc) || !stride_ok(a->dk_acc_st, 4) ||
                    !stride_ok(a->dv_acc_st, 4)))
    return RFA_ERR_ALIGN;
  if (!a->dk_acc && (!aligned16(a->dk) || !aligned16(a->dv) || !stride_ok(a->dk_st, 2) || !stride_ok(a->dv_st, 2)))
    return RFA_ERR_ALIGN;

  hipStream_t st = (hipStream_t)stream;
  BwdParams p{};
  p.dout = a->dout; p.q = a->q; p.k = a->k; p.v = a->v; p.lse = a->lse; p.delta = a->delta;
  p.dq = a->dq; p.dq_acc = a->dq_acc;
  p.cu_q = a->cu_seqlens_q; p.cu_k = a->cu_seqlens_k;
  p.dout_st = cv(a->dout_st); p.q_st = cv(a->q_st); p.k_st = cv(a->k_st); p.v_st = cv(a->v_st);
  p.dq_st = cv(a->dq_st); p.dq_acc_st = cv(a->dq_acc_st);
  p.lse_batch = a->lse_batch; p.lse_head = a->lse_head;
  p.delta_batch = a->delta_batch; p.delta_head = a->delta_head;
  p.B = a->B; p.H = a->H; p.Hk = a->Hk; p.D = a->D; p.Sq = a->Sq; p.Sk = a->Sk;
  p.q_half = a->q_half; p.k_half = a->k_half;
  p.causal = a->causal ? 1 : 0; p.acc_init = a->acc_init ? 1 : 0;
  p.wl = (a->window && a->window_left >= 0) ? a->window_left : -1;
  p.wr = a->causal ? 0 : ((a->window && a->window_right >= 0) ? a->window_right : -1);
  p.scale = a->softmax_scale;
  p.drop_keep = drop_threshold(a->dropout_p);
  p.drop_scale = drop_rescale(a->dropout_p);
  p.drop_seed = a->dropout_seed;
  p.q_pos0 = (unsigned)a->q_pos_offset; p.k_pos0 = (unsigned)a->k_pos_offset; p.head0 = (unsigned)a->head_offset;
  p.nqblk = (eff_len(a->Sq, a->q_half) + bwd_dq_rows_per_block() - 1) / bwd_dq_rows_per_block();
  const DsChunks chunks = bwd_ds_chunking(a);
  const int Gfull = a->H / a->Hk;
  const bool frac = chunks.nchunks > 0 && chunks.gc < Gfull;       // query-head fractions of a K/V head per launch
  const DkdvPlan plan = bwd_dkdv_plan(a);
  p.wide = plan.wide; p.nsplit = plan.nsplit;
  p.nkblk = (eff_len(a->Sk, a->k_half) + bwd_dkdv_keys_per_block(plan.wide) - 1) / bwd_dkdv_keys_per_block(plan.wide);

  Strides ws_st{};
  const bool part_f32 = plan.nsplit > 1 || frac;
  if (plan.bal) {
    // balanced schedule: the kernel writes the FINAL dK / dV itself (io dtype, or the overwritten fp32 accumulators);
    // the workspace holds the pair flags and the pair slots
    const int64_t pairs = (int64_t)a->B * a->Hk * (p.nkblk / 2);
    p.bal = 1;
    p.pair_flags = (unsigned*)a->workspace;
    p.pair_ws = (char*)a->workspace + bal_flag_bytes(pairs);
    if (a->dk_acc) {
      p.dk = a->dk_acc; p.dv = a->dv_acc;
      p.dk_st = cv(a->dk_acc_st); p.dv_st = cv(a->dv_acc_st);
      p.kv_f32 = 1;
    } else {
      p.dk = a->dk; p.dv = a->dv;
      p.dk_st = cv(a->dk_st); p.dv_st = cv(a->dv_st);
    }
  } else if (bwd_kv_direct(a) && !part_f32) {
    p.dk = a->dk_acc; p.dv = a->dv_acc;
    p.dk_st = cv(a->dk_acc_st); p.dv_st = cv(a->dv_acc_st);
    p.kv_f32 = 1;
  } else if (ws) {
    // partials: (rows, Hk, nsplit, D) contiguous; dense rows = b*Sk + row (own batch stride).  The kernel addresses
    // K/V head hk of split s at element (hk * nsplit + s) * D of a row, reduce_kernel reads the nsplit entries of
    // a K/V head as its "group" (G = nsplit, head stride D).  Split launches — and launches that cover a fraction of
    // a K/V head's query heads (chunked dS hand-off: the later fractions ADD to the partial) — keep fp32 partials
    // (summed, then rounded once: the rounding an unsplit launch does); the others the io dtype.
    const int64_t ns = plan.nsplit;
    const int64_t esz = part_f32 ? 4 : 2;
    ws_st.head = a->D;
    ws_st.row = (int64_t)a->Hk * ns * a->D;
    ws_st.batch = a->cu_seqlens_k ? 0 : (int64_t)a->Sk * a->Hk * ns * a->D;
    p.dk = a->workspace;
    p.dv = (char*)a->workspace + a->total_k * (int64_t)a->Hk * ns * a->D * esz;
    p.dk_st = ws_st; p.dv_st = ws_st;
    p.dk_st.head = p.dv_st.head = ns * a->D;
    p.kv_split_stride = a->D;
    p.kv_f32 = p.kv_part_f32 = part_f32 ? 1 : 0;
  } else {
    p.dk = a->dk; p.dv = a->dv;
    p.dk_st = cv(a->dk_st); p.dv_st = cv(a->dv_st);
  }
  auto mark = [&](int i) {
    if (a->prof_events && a->prof_events[i]) (void)hipEventRecord((hipEvent_t)a->prof_events[i], st);
  };
  mark(0);
  const bool do_compute = bwd_single_phase(a) || (a->phases & RFA_BWD_COMPUTE);
  const bool do_reduce = bwd_single_phase(a) || (a->phases & RFA_BWD_REDUCE);
  const int kv_init = (a->acc_init || (a->phases & RFA_BWD_KV_OVERWRITE)) ? 1 : 0;
  if (do_compute) {
    if (chunks.nchunks > 0) {
      // 5-GEMM form: dK/dV kernel first (it stores dS), then dQ streams dS back — once over all heads, or (a scratch
      // smaller than the whole hand-off) chunk by chunk over head groups that reuse the one scratch in stream order
      if (!aligned16(a->ds_scratch)) return RFA_ERR_ALIGN;
      if (chunks.nchunks > 1 && (a->phases & (RFA_BWD_SKIP_DKDV | RFA_BWD_SKIP_DQ))) return RFA_ERR_ARGS;
      p.ds = a->ds_scratch;
      p.ds_c = bwd_ds_c(a);
      p.ds_tri = p.ds_c < ds_blocks(a->Sk, a->k_half) ? 1 : 0;
      p.ds_packed = a->cu_seqlens_q != nullptr ? 1 : 0;
      p.ds_head_blocks = bwd_ds_head_blocks(a);
      const int nfrac = Gfull / chunks.gc;
      const int64_t kv_esz = p.kv_f32 ? 4 : 2;
      for (int c = 0; c < chunks.nchunks; ++c) {
        BwdParams pc = p;
        if (chunks.nchunks > 1) {
          const int hk0 = frac ? c / nfrac : c * chunks.hc;      // first K/V head of the chunk
          const int f = frac ? c % nfrac : 0;                    // which fraction of that K/V head's query heads
          const int64_t qh0 = (int64_t)hk0 * Gfull + (int64_t)f * chunks.gc;   // first query head
          pc.Hk = chunks.hc;
          pc.H = chunks.hc * chunks.gc;
          pc.q = (const char*)p.q + qh0 * p.q_st.head * 2;
          pc.dout = (const char*)p.dout + qh0 * p.dout_st.head * 2;
          pc.lse = p.lse + qh0 * p.lse_head;
          pc.delta = p.delta + qh0 * p.delta_head;
          if (p.dq_acc) pc.dq_acc = p.dq_acc + qh0 * p.dq_acc_st.head;
          else pc.dq = (char*)p.dq + qh0 * p.dq_st.head * 2;
          pc.k = (const char*)p.k + (int64_t)hk0 * p.k_st.head * 2;
          pc.v = (const char*)p.v + (int64_t)hk0 * p.v_st.head * 2;
          pc.dk = (char*)p.dk + (int64_t)hk0 * p.dk_st.head * kv_esz;
          pc.dv = (char*)p.dv + (int64_t)hk0 * p.dv_st.head * kv_esz;
          pc.kv_accum = f > 0 ? 1 : 0;
        }
        if (!(a->phases & RFA_BWD_SKIP_DKDV))
          if (int rc2 = launch_bwd_dkdv(pc, a->dtype, st)) return launch_status(rc2);
        if (c == 0) mark(1);
        if (!(a->phases & RFA_BWD_SKIP_DQ))
          if (int rc2 = launch_bwd_dq_from_ds(pc, a->dtype, st)) return launch_status(rc2);
      }
      mark(2);
    } else {
      if (!(a->phases & RFA_BWD_SKIP_DQ))
        if (int rc2 = launch_bwd_dq(p, a->dtype, st)) return launch_status(rc2);
      mark(1);
      if (!(a->phases & RFA_BWD_SKIP_DKDV))
        if (int rc2 = launch_bwd_dkdv(p, a->dtype, st)) return launch_status(rc2);
      mark(2);
    }
  } else {
    mark(1);
    mark(2);
  }
  if (ws && do_reduce && !plan.bal) {
    // dK and dV in one launch (grid z)
    ReduceParams r{};
    r.src = p.dk; r.src2 = p.dv;
    r.src_st = ws_st;
    r.src_f32 = part_f32 ? 1 : 0;
    r.cu_k = a->cu_seqlens_k;
    r.B = a->B; r.Hk = a->Hk; r.G = plan.nsplit; r.D = a->D; r.Sk = a->Sk;   // query-head groups are already summed
    r.k_half = a->k_half; r.acc_init = kv_init;
    if (a->dk_acc) {
      r.dst_acc = a->dk_acc; r.dst_acc2 = a->dv_acc;
      r.dst_acc_st = cv(a->dk_acc_st); r.dst_acc2_st = cv(a->dv_acc_st);
    } else {
      r.dst = a->dk; r.dst2 = a->dv;
      r.dst_st = cv(a->dk_st); r.dst2_st = cv(a->dv_st);
    }
    if (launch_reduce(r, a->dtype, st)) return RFA_ERR_LAUNCH;
  }
  mark(3);
  return RFA_OK;
}

int rfa_merge(const rfa_merge_args* a, void* stream) {
  if (!a) return RFA_ERR_NULL;
  int rc = check_common(a->dtype, a->H, a->H, a->D, a->B);
  if (rc) return rc;
  if (a->S < 0) return RFA_ERR_SHAPE;
  if (a->B == 0 || a->S == 0) return RFA_OK;
  if (!a->out_acc || !a->lse_acc || !a->block_out || !a->block_lse) return RFA_ERR_NULL;
  if (!aligned16(a->out_acc) || !aligned16(a->block_out) || !stride_ok(a->out_acc_st, 4) ||
      !stride_ok(a->block_out_st, 2))
    return RFA_ERR_ALIGN;
  MergeParams p{};
  p.out_acc = a->out_acc; p.lse_acc = a->lse_acc; p.block_out = a->block_out; p.block_lse = a->block_lse;
  p.out_acc_st = cv(a->out_acc_st); p.block_out_st = cv(a->block_out_st);
  p.lse_acc_batch = a->lse_acc_batch; p.lse_acc_head = a->lse_acc_head;
  p.block_lse_batch = a->block_lse_batch; p.block_lse_head = a->block_lse_head;
  p.lse_acc_row = a->lse_acc_row ? a->lse_acc_row : 1;
  p.block_lse_row = a->block_lse_row ? a->block_lse_row : 1;
  p.B = a->B; p.H = a->H; p.D = a->D; p.S = a->S; p.acc_init = a->acc_init ? 1 : 0;
  return launch_merge(p, a->dtype, (hipStream_t)stream) ? RFA_ERR_LAUNCH : RFA_OK;
}

int rfa_sum_slots(const rfa_sum_slots_args* a, void* stream) {
  if (!a) return RFA_ERR_NULL;
  int rc = check_common(a->dtype, a->H, a->H, a->D, a->B);
  if (rc) return rc;
  if (a->S < 0 || a->nslots < 1) return RFA_ERR_SHAPE;
  if (a->B == 0 || a->S == 0) return RFA_OK;
  if (!a->src || !a->dst) return RFA_ERR_NULL;
  if (!aligned16(a->src) || !aligned16(a->dst) || !stride_ok(a->src_st, 2) || !stride_ok(a->dst_st, 2) ||
      (a->slot_stride % 8) != 0)
    return RFA_ERR_ALIGN;
  ReduceParams r{};
  r.src = a->src; r.src_st = cv(a->src_st);
  r.dst = a->dst; r.dst_st = cv(a->dst_st);
  r.B = a->B; r.Hk = a->H; r.G = a->nslots; r.D = a->D; r.Sk = a->S;
  r.k_half = RFA_HALF_FULL; r.acc_init = 1;
  r.g_stride = a->slot_stride;
  if (launch_reduce(r, a->dtype, (hipStream_t)stream)) return RFA_ERR_LAUNCH;
  return RFA_OK;
}

int rfa_cast(void* dst, const float* src, int64_t n, int32_t dtype, void* stream) {
  if (dtype != RFA_BF16 && dtype != RFA_F16) return RFA_ERR_DTYPE;
  if (n < 0) return RFA_ERR_SHAPE;
  if (n == 0) return RFA_OK;
  if (!dst || !src) return RFA_ERR_NULL;
  if (!aligned16(dst) || !aligned16(src)) return RFA_ERR_ALIGN;
  return launch_cast(dst, src, n, dtype, (hipStream_t)stream) ? RFA_ERR_LAUNCH : RFA_OK;
}

int rfa_lse_flatten(float* dst, const float* src, const int32_t* cu_seqlens, int32_t B, int32_t H,
                    int32_t max_seqlen, int64_t dst_head_stride, int64_t dst_row_stride, void* stream) {
  if (B < 0 || H < 0 || max_seqlen < 0) return RFA_ERR_SHAPE;
  if (B == 0 || H == 0 || max_seqlen == 0) return RFA_OK;
  if (!dst || !src || !cu_seqlens) return RFA_ERR_NULL;
  return launch_lse_relayout(dst, src, cu_seqlens, B, H, max_seqlen, dst_head_stride, dst_row_stride,
                             true, (hipStream_t)stream) ? RFA_ERR_LAUNCH : RFA_OK;
}

int rfa_lse_unflatten(float* dst, const float* src, const int32_t* cu_seqlens, int32_t B, int32_t H,
                      int32_t max_seqlen, int64_t src_head_stride, int64_t src_row_stride, void* stream) {
  if (B < 0 || H < 0 || max_seqlen < 0) return RFA_ERR_SHAPE;
  if (B == 0 || H == 0 || max_seqlen == 0) return RFA_OK;
  if (!dst || !src || !cu_seqlens) return RFA_ERR_NULL;
  return launch_lse_relayout(dst, src, cu_seqlens, B, H, max_seqlen, src_head_stride, src_row_stride,
                             false, (hipStream_t)stream) ? RFA_ERR_LAUNCH : RFA_OK;
}

}  // extern "C"
#pragma GCC visibility pop
