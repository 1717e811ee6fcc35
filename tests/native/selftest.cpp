// tests/native/selftest.cpp — torch-free GPU self test of librfa_hip.so through its C ABI.
//
//   1. hardware-layout probes (MFMA 32x32x16 operand/accumulator maps, ds_read_b64_tr_b16 with
//      the library's swizzle) — the assumptions documented in csrc/rfa_common.hpp
//   2. parity of rfa_fwd / rfa_bwd / rfa_merge against the C oracle (oracle/attn_ref.c) over
//      dense/varlen, causal/non-causal, GQA, ragged tails, half-selection, accumulate mode
//   3. (--perf) timing of the headline shape + sampled-row spot checks at full size
//
// Build: see ring-flash-attention_amd/build.py (target "selftest").  Exit code 0 = all passed.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rfa.h"
#include "../../ring-flash-attention_amd/csrc/rfa_common.hpp"

extern "C" {
int rfa_ref_fwd(const float*, const float*, const float*, float*, float*, int, int, int, int, int, int,
                const int32_t*, const int32_t*, int64_t, float, int);
int rfa_ref_bwd(const float*, const float*, const float*, const float*, const float*, const float*, float*,
                float*, float*, int, int, int, int, int, int, const int32_t*, const int32_t*, int64_t,
                int64_t, float, int);
int rfa_ref_merge(float*, float*, const float*, const float*, int, int, int, int);
}

#define HIPCHECK(x)                                                                        \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)

static int g_fail = 0;

// ---------------------------------------------------------------- bf16 helpers (host)
static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffff) > 0x7f800000) return 0x7fc0;
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 2654435761ull + 88172645463325252ull) {}
  uint32_t next() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (uint32_t)(s >> 32);
  }
  float uni() { return (next() >> 8) * (1.0f / 16777216.0f); }
  float normal() {
    float u1 = uni() + 1e-7f, u2 = uni();
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
  }
};

// fills bf16 device-format vector and its float image
static void fill_normal(std::vector<uint16_t>& h, std::vector<float>& f, size_t n, Rng& r, float sc = 1.f) {
  h.resize(n); f.resize(n);
  for (size_t i = 0; i < n; ++i) { h[i] = f2bf(r.normal() * sc); f[i] = bf2f(h[i]); }
}

template <typename T>
static T* dupload(const std::vector<T>& v) {
  T* d = nullptr;
  HIPCHECK(hipMalloc(&d, v.size() * sizeof(T) + 256));
  HIPCHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
template <typename T>
static T* dalloc(size_t n, int fill = 0xff) {
  T* d = nullptr;
  HIPCHECK(hipMalloc(&d, n * sizeof(T) + 256));
  HIPCHECK(hipMemset(d, fill, n * sizeof(T)));   // poison (0xff.. = NaN for fp32 / bf16)
  return d;
}
template <typename T>
static std::vector<T> ddownload(const T* d, size_t n) {
  std::vector<T> v(n);
  HIPCHECK(hipMemcpy(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
  return v;
}

// ================================================================= probes
using namespace rfa;

__global__ void mfma_probe(const bf16_t* A /*32x16*/, const bf16_t* B /*16x32*/, float* C /*32x32*/) {
  const int l = threadIdx.x, g = l >> 5, l31 = l & 31;
  vec8<bf16_t> a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = A[l31 * 16 + 8 * g + e];        // A[m = l31][k = 8g+e]
    b[e] = B[(8 * g + e) * 32 + l31];      // B[k = 8g+e][n = l31]
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = mfma(a, b, c);
  for (int r = 0; r < 16; ++r) C[crow(r, g) * 32 + l31] = c[r];   // row m = crow(r,g), col n = l31
}

// LDS tile [64][128] u16 with value = row*128 + col, swizzled with tile_off.
__global__ void tr_probe(uint16_t* out /* [16 rb][4 dblk][64 lanes][4] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_t* smem = (lds_t*)smem_raw;
  const int lane = threadIdx.x;
  for (int idx = lane; idx < 64 * 16; idx += 64) {
    const int row = idx >> 4, chunk = idx & 15;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v;
    for (int e = 0; e < 8; ++e) v[e] = (short)(row * 128 + chunk * 8 + e);
    *(__attribute__((address_space(3))) s16x8*)(smem + tile_off(row, chunk)) = v;
  }
  __syncthreads();
  for (int rbi = 0; rbi < 16; ++rbi) {
    const int rb = 4 * rbi;
    for (int dblk = 0; dblk < 4; ++dblk) {
      const int off = (rb + ((lane & 15) >> 2)) * kRowBytes + tr_lane_off(lane, dblk, (rb >> 2) & 3);
      s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off));
      uint16_t* o = out + ((rbi * 4 + dblk) * 64 + lane) * 4;
      o[0] = (uint16_t)v[0]; o[1] = (uint16_t)v[1]; o[2] = (uint16_t)v[2]; o[3] = (uint16_t)v[3];
    }
  }
}

static void run_probes() {
  // ---- MFMA
  {
    Rng r(1);
    std::vector<uint16_t> A(32 * 16), B(16 * 32);
    std::vector<float> Af(32 * 16), Bf(16 * 32);
    for (size_t i = 0; i < A.size(); ++i) { Af[i] = (float)((int)(r.next() % 9) - 4); A[i] = f2bf(Af[i]); }
    for (size_t i = 0; i < B.size(); ++i) { Bf[i] = (float)((int)(r.next() % 7) - 3); B[i] = f2bf(Bf[i]); }
    uint16_t* dA = dupload(A); uint16_t* dB = dupload(B);
    float* dC = dalloc<float>(32 * 32);
    hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, (const bf16_t*)dA, (const bf16_t*)dB, dC);
    HIPCHECK(hipDeviceSynchronize());
    auto C = ddownload(dC, 32 * 32);
    int bad = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        float ref = 0;
        for (int k = 0; k < 16; ++k) ref += Af[m * 16 + k] * Bf[k * 32 + n];
        if (C[m * 32 + n] != ref) ++bad;
      }
    printf("[probe] mfma_f32_32x32x16_bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    if (bad) ++g_fail;
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  // ---- transpose read
  {
    const size_t n = 16 * 4 * 64 * 4;
    uint16_t* dout = dalloc<uint16_t>(n);
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 64 * 256, 0, dout);
    HIPCHECK(hipDeviceSynchronize());
    auto o = ddownload(dout, n);
    int bad = 0;
    for (int rbi = 0; rbi < 16; ++rbi)
      for (int dblk = 0; dblk < 4; ++dblk)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const int expect = (4 * rbi + j) * 128 + 32 * dblk + (lane & 31);
            const int got = o[((rbi * 4 + dblk) * 64 + lane) * 4 + j];
            if (got != expect) {
              if (bad < 12)
                printf("   tr mismatch rb=%d dblk=%d lane=%d j=%d: got (row %d col %d) want (row %d col %d)\n",
                       4 * rbi, dblk, lane, j, got / 128, got % 128, expect / 128, expect % 128);
              ++bad;
            }
          }
    printf("[probe] ds_read_b64_tr_b16 + swizzle: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    if (bad) ++g_fail;
    hipFree(dout);
  }
}

// ================================================================= attention cases
struct Case {
  const char* name;
  int B, H, Hk, D, Sq, Sk, causal;
  std::vector<int> cu;     // non-empty => varlen with cu_q == cu_k == cu (B = cu.size()-1)
  std::vector<int> cuk;    // optional distinct cu_k
  int spike = 0;           // >0: plant keys k[j] = 3*q[i] so single rows' maxima jump by >> 8 log2
                           // units at chosen tiles (forces the rare deferred-rescale branch of the
                           // online softmax in the middle of the K loop; dense, H == Hk only)
};

struct ErrStat { double maxerr = 0, maxref = 0; size_t nan = 0; };
static ErrStat cmp(const float* got, const float* ref, size_t n) {
  ErrStat e;
  for (size_t i = 0; i < n; ++i) {
    if (!(got[i] == got[i])) { if (ref[i] == ref[i]) ++e.nan; continue; }
    if (isinf(ref[i])) { if (got[i] != ref[i]) ++e.nan; continue; }
    e.maxerr = fmax(e.maxerr, fabs((double)got[i] - ref[i]));
    e.maxref = fmax(e.maxref, fabs((double)ref[i]));
  }
  return e;
}
static bool report(const char* what, const ErrStat& e, double atol, double rtol) {
  const bool ok = e.nan == 0 && e.maxerr <= atol + rtol * e.maxref;
  printf("    %-5s max|err| %.3e (max|ref| %.3e, bad/nan %zu)  %s\n", what, e.maxerr, e.maxref, e.nan,
         ok ? "ok" : "FAIL");
  return ok;
}
static std::vector<float> bfvec(const std::vector<uint16_t>& h) {
  std::vector<float> f(h.size());
  for (size_t i = 0; i < h.size(); ++i) f[i] = bf2f(h[i]);
  return f;
}

static rfa_strides st_dense(int S, int H, int D) { return rfa_strides{(int64_t)S * H * D, (int64_t)H * D, D}; }

static void run_case(const Case& c, uint64_t seed) {
  const bool varlen = !c.cu.empty();
  const int B = varlen ? (int)c.cu.size() - 1 : c.B;
  const std::vector<int>& cuq = c.cu;
  const std::vector<int>& cuk = c.cuk.empty() ? c.cu : c.cuk;
  const int64_t Tq = varlen ? cuq.back() : (int64_t)B * c.Sq;
  const int64_t Tk = varlen ? cuk.back() : (int64_t)B * c.Sk;
  int Sq = c.Sq, Sk = c.Sk;
  if (varlen) {
    Sq = Sk = 0;
    for (int b = 0; b < B; ++b) { Sq = std::max(Sq, cuq[b + 1] - cuq[b]); Sk = std::max(Sk, cuk[b + 1] - cuk[b]); }
  }
  const int H = c.H, Hk = c.Hk, D = c.D;
  const float scale = 1.0f / sqrtf((float)D);
  printf("[case] %s: B=%d H=%d Hk=%d D=%d Sq=%d Sk=%d causal=%d %s\n", c.name, B, H, Hk, D, Sq, Sk, c.causal,
         varlen ? "varlen" : "dense");
  Rng r(seed);
  std::vector<uint16_t> q, k, v, dout;
  std::vector<float> qf, kf, vf, dof;
  fill_normal(q, qf, (size_t)Tq * H * D, r);
  fill_normal(k, kf, (size_t)Tk * Hk * D, r);
  fill_normal(v, vf, (size_t)Tk * Hk * D, r);
  fill_normal(dout, dof, (size_t)Tq * H * D, r);
  if (c.spike && !varlen && H == Hk) {
    // (query row, key row) pairs spread over first / middle / last KV tiles and both sub-tiles
    const int pairs[][2] = {{5, 70}, {40, 3}, {100, Sk - 1}, {Sq - 1, Sk / 2 + 1}, {Sq / 2, 130}, {17, 200 % Sk}};
    for (auto& pr : pairs) {
      const int i = pr[0] % Sq, j = pr[1] % Sk;
      for (int h = 0; h < H; ++h)
        for (int d = 0; d < D; ++d) {
          const size_t qi = ((size_t)i * H + h) * D + d, kj = ((size_t)j * Hk + h) * D + d;
          k[kj] = f2bf(3.0f * qf[qi]);
          kf[kj] = bf2f(k[kj]);
        }
    }
  }

  // ---- oracle
  std::vector<float> ro((size_t)Tq * H * D), rl((size_t)Tq * H), rdq((size_t)Tq * H * D),
      rdk((size_t)Tk * Hk * D), rdv((size_t)Tk * Hk * D);
  rfa_ref_fwd(qf.data(), kf.data(), vf.data(), ro.data(), rl.data(), B, H, Hk, D, Sq, Sk,
              varlen ? cuq.data() : nullptr, varlen ? cuk.data() : nullptr, Tq, scale, c.causal);
  rfa_ref_bwd(dof.data(), qf.data(), kf.data(), vf.data(), ro.data(), rl.data(), rdq.data(), rdk.data(),
              rdv.data(), B, H, Hk, D, Sq, Sk, varlen ? cuq.data() : nullptr, varlen ? cuk.data() : nullptr,
              Tq, Tk, scale, c.causal);

  // ---- device
  uint16_t *dq_ = dupload(q), *dk_ = dupload(k), *dv_ = dupload(v), *ddo = dupload(dout);
  uint16_t* dout_o = dalloc<uint16_t>((size_t)Tq * H * D);
  float* dlse = dalloc<float>((size_t)Tq * H);
  float* ddelta = dalloc<float>((size_t)Tq * H);
  uint16_t* gdq = dalloc<uint16_t>((size_t)Tq * H * D);
  uint16_t* gdk = dalloc<uint16_t>((size_t)Tk * Hk * D);
  uint16_t* gdv = dalloc<uint16_t>((size_t)Tk * Hk * D);
  int32_t *dcuq = nullptr, *dcuk = nullptr;
  if (varlen) { dcuq = dupload(cuq); dcuk = dupload(cuk); }

  rfa_fwd_args fa;
  memset(&fa, 0, sizeof(fa));
  fa.q = dq_; fa.k = dk_; fa.v = dv_; fa.out = dout_o; fa.lse = dlse;
  fa.q_st = st_dense(Sq, H, D); fa.k_st = st_dense(Sk, Hk, D); fa.v_st = fa.k_st; fa.out_st = fa.q_st;
  fa.lse_batch = varlen ? 0 : (int64_t)H * Sq; fa.lse_head = varlen ? Tq : Sq;
  fa.cu_seqlens_q = dcuq; fa.cu_seqlens_k = dcuk;
  fa.B = B; fa.H = H; fa.Hk = Hk; fa.D = D; fa.Sq = Sq; fa.Sk = Sk;
  fa.softmax_scale = scale; fa.causal = c.causal; fa.dtype = RFA_BF16;
  int rc = rfa_fwd(&fa, nullptr);
  if (rc) { printf("    rfa_fwd failed: %s\n", rfa_strerror(rc)); ++g_fail; return; }
  HIPCHECK(hipDeviceSynchronize());
  auto go = bfvec(ddownload(dout_o, (size_t)Tq * H * D));
  auto gl = ddownload(dlse, (size_t)Tq * H);
  // lse layouts: dense (B,H,Sq) / varlen (H,Tq) — identical to the oracle's
  bool ok = true;
  ok &= report("out", cmp(go.data(), ro.data(), go.size()), 4e-3, 1.2e-2);
  ok &= report("lse", cmp(gl.data(), rl.data(), gl.size()), 2e-3, 2e-4);

  rfa_bwd_preprocess_args pa;
  memset(&pa, 0, sizeof(pa));
  pa.dout = ddo; pa.out = dout_o; pa.dout_st = fa.q_st; pa.out_st = fa.q_st; pa.delta = ddelta;
  pa.delta_batch = fa.lse_batch; pa.delta_head = fa.lse_head; pa.cu_seqlens_q = dcuq;
  pa.B = B; pa.H = H; pa.D = D; pa.Sq = Sq; pa.dtype = RFA_BF16;
  rc = rfa_bwd_preprocess(&pa, nullptr);
  if (rc) { printf("    rfa_bwd_preprocess failed: %s\n", rfa_strerror(rc)); ++g_fail; return; }

  rfa_bwd_args ba;
  memset(&ba, 0, sizeof(ba));
  ba.dout = ddo; ba.q = dq_; ba.k = dk_; ba.v = dv_;
  ba.dout_st = fa.q_st; ba.q_st = fa.q_st; ba.k_st = fa.k_st; ba.v_st = fa.k_st;
  ba.lse = dlse; ba.lse_batch = fa.lse_batch; ba.lse_head = fa.lse_head;
  ba.delta = ddelta; ba.delta_batch = fa.lse_batch; ba.delta_head = fa.lse_head;
  ba.dq = gdq; ba.dk = gdk; ba.dv = gdv; ba.dq_st = fa.q_st; ba.dk_st = fa.k_st; ba.dv_st = fa.k_st;
  ba.cu_seqlens_q = dcuq; ba.cu_seqlens_k = dcuk;
  ba.B = B; ba.H = H; ba.Hk = Hk; ba.D = D; ba.Sq = Sq; ba.Sk = Sk; ba.total_k = Tk; ba.total_q = Tq;
  ba.softmax_scale = scale; ba.causal = c.causal; ba.dtype = RFA_BF16;
  void* ws = nullptr;
  const int64_t wsb = rfa_bwd_workspace_bytes(&ba);
  if (wsb) { HIPCHECK(hipMalloc(&ws, wsb)); HIPCHECK(hipMemset(ws, 0xff, wsb)); }
  ba.workspace = ws;
  rc = rfa_bwd(&ba, nullptr);
  if (rc) { printf("    rfa_bwd failed: %s\n", rfa_strerror(rc)); ++g_fail; return; }
  HIPCHECK(hipDeviceSynchronize());
  auto gq = bfvec(ddownload(gdq, (size_t)Tq * H * D));
  auto gk = bfvec(ddownload(gdk, (size_t)Tk * Hk * D));
  auto gv = bfvec(ddownload(gdv, (size_t)Tk * Hk * D));
  ok &= report("dq", cmp(gq.data(), rdq.data(), gq.size()), 5e-3, 2e-2);
  ok &= report("dk", cmp(gk.data(), rdk.data(), gk.size()), 5e-3, 2e-2);
  ok &= report("dv", cmp(gv.data(), rdv.data(), gv.size()), 5e-3, 2e-2);
  if (!ok) ++g_fail;

  hipFree(dq_); hipFree(dk_); hipFree(dv_); hipFree(ddo); hipFree(dout_o); hipFree(dlse); hipFree(ddelta);
  hipFree(gdq); hipFree(gdk); hipFree(gdv);
  if (ws) hipFree(ws);
  if (dcuq) { hipFree(dcuq); hipFree(dcuk); }
}

// ---- accumulate-mode + half-selection test: emulate one zigzag rank's three step kinds on one
// GPU.  q (S rows), kv blocks A,B,C (S rows each); out must equal attention of q over the
// concatenation [A (full, causal) | B front half (all q) | C (only back-half q rows)], which the
// oracle computes per row-set with plain dense calls + rfa_ref_merge.
static void run_acc_case(int S, int H, int Hk, int D, uint64_t seed, bool varlen) {
  printf("[case] fused-merge / half-select (%s): S=%d H=%d Hk=%d D=%d\n", varlen ? "varlen 2 seqs" : "dense",
         S, H, Hk, D);
  const float scale = 1.0f / sqrtf((float)D);
  const int B = 1;
  // varlen: two sequences [0,S1) [S1,S)  (both even)
  const int S1 = varlen ? (S / 4) * 2 : S;
  std::vector<int> cu = varlen ? std::vector<int>{0, S1, S} : std::vector<int>{};
  const int nseq = varlen ? 2 : 1;
  Rng r(seed);
  std::vector<uint16_t> q, kA, vA, kB, vB, kC, vC;
  std::vector<float> qf, kAf, vAf, kBf, vBf, kCf, vCf;
  fill_normal(q, qf, (size_t)S * H * D, r);
  fill_normal(kA, kAf, (size_t)S * Hk * D, r); fill_normal(vA, vAf, (size_t)S * Hk * D, r);
  fill_normal(kB, kBf, (size_t)S * Hk * D, r); fill_normal(vB, vBf, (size_t)S * Hk * D, r);
  fill_normal(kC, kCf, (size_t)S * Hk * D, r); fill_normal(vC, vCf, (size_t)S * Hk * D, r);

  // ---------- oracle: per sequence, build gathered tensors and merge
  std::vector<float> ref_out((size_t)S * H * D, 0.f), ref_lse((size_t)H * S, 0.f);   // lse (H,S)
  for (int sq = 0; sq < nseq; ++sq) {
    const int s0 = varlen ? cu[sq] : 0, s1 = varlen ? cu[sq + 1] : S, L = s1 - s0, Lh = L / 2;
    auto slice = [&](const std::vector<float>& x, int heads, int a, int bnd) {
      return std::vector<float>(x.begin() + (size_t)a * heads * D, x.begin() + (size_t)bnd * heads * D);
    };
    std::vector<float> ql = slice(qf, H, s0, s1);
    // step 0: causal over A
    std::vector<float> o0((size_t)L * H * D), l0((size_t)H * L);
    auto ka = slice(kAf, Hk, s0, s1), va = slice(vAf, Hk, s0, s1);
    rfa_ref_fwd(ql.data(), ka.data(), va.data(), o0.data(), l0.data(), 1, H, Hk, D, L, L, 0, 0, 0, scale, 1);
    // step "<= rank": all q over front half of B
    std::vector<float> o1((size_t)L * H * D), l1((size_t)H * L);
    auto kb = slice(kBf, Hk, s0, s0 + Lh), vb = slice(vBf, Hk, s0, s0 + Lh);
    rfa_ref_fwd(ql.data(), kb.data(), vb.data(), o1.data(), l1.data(), 1, H, Hk, D, L, Lh, 0, 0, 0, scale, 0);
    rfa_ref_merge(o0.data(), l0.data(), o1.data(), l1.data(), 1, L, H, D);
    // step "> rank": back half q over all of C
    std::vector<float> qh = slice(qf, H, s0 + Lh, s1);
    std::vector<float> o2((size_t)(L - Lh) * H * D), l2((size_t)H * (L - Lh));
    auto kc = slice(kCf, Hk, s0, s1), vc = slice(vCf, Hk, s0, s1);
    rfa_ref_fwd(qh.data(), kc.data(), vc.data(), o2.data(), l2.data(), 1, H, Hk, D, L - Lh, L, 0, 0, 0, scale, 0);
    // merge into rows [Lh, L)
    std::vector<float> ob(o0.begin() + (size_t)Lh * H * D, o0.end()), lb((size_t)H * (L - Lh));
    for (int h = 0; h < H; ++h) for (int i = 0; i < L - Lh; ++i) lb[(size_t)h * (L - Lh) + i] = l0[(size_t)h * L + Lh + i];
    rfa_ref_merge(ob.data(), lb.data(), o2.data(), l2.data(), 1, L - Lh, H, D);
    std::copy(ob.begin(), ob.end(), o0.begin() + (size_t)Lh * H * D);
    for (int h = 0; h < H; ++h) for (int i = 0; i < L - Lh; ++i) l0[(size_t)h * L + Lh + i] = lb[(size_t)h * (L - Lh) + i];
    std::copy(o0.begin(), o0.end(), ref_out.begin() + (size_t)s0 * H * D);
    for (int h = 0; h < H; ++h) for (int i = 0; i < L; ++i) ref_lse[(size_t)h * S + s0 + i] = l0[(size_t)h * L + i];
  }

  // ---------- device: three rfa_fwd calls in accumulate mode, no gathers
  uint16_t *dq_ = dupload(q), *dkA = dupload(kA), *dvA = dupload(vA), *dkB = dupload(kB), *dvB = dupload(vB),
           *dkC = dupload(kC), *dvC = dupload(vC);
  float* oacc = dalloc<float>((size_t)S * H * D);
  float* lacc = dalloc<float>((size_t)S * H);
  int32_t* dcu = varlen ? dupload(cu) : nullptr;
  int Smax = S;
  if (varlen) Smax = std::max(S1, S - S1);
  rfa_fwd_args fa;
  memset(&fa, 0, sizeof(fa));
  fa.q = dq_; fa.q_st = st_dense(S, H, D); fa.k_st = st_dense(S, Hk, D); fa.v_st = fa.k_st;
  fa.out_acc = oacc; fa.out_acc_st = fa.q_st; fa.lse_acc = lacc; fa.lse_acc_batch = (int64_t)H * S; fa.lse_acc_head = S;
  fa.cu_seqlens_q = dcu; fa.cu_seqlens_k = dcu;
  fa.B = varlen ? nseq : B; fa.H = H; fa.Hk = Hk; fa.D = D; fa.Sq = Smax; fa.Sk = Smax;
  fa.softmax_scale = scale; fa.dtype = RFA_BF16;
  int rc;
  fa.k = dkA; fa.v = dvA; fa.causal = 1; fa.acc_init = 1; fa.q_half = 0; fa.k_half = 0;
  rc = rfa_fwd(&fa, nullptr);
  fa.k = dkB; fa.v = dvB; fa.causal = 0; fa.acc_init = 0; fa.q_half = 0; fa.k_half = RFA_HALF_FRONT;
  rc |= rfa_fwd(&fa, nullptr);
  fa.k = dkC; fa.v = dvC; fa.causal = 0; fa.acc_init = 0; fa.q_half = RFA_HALF_BACK; fa.k_half = 0;
  rc |= rfa_fwd(&fa, nullptr);
  if (rc) { printf("    rfa_fwd(acc) failed: %s\n", rfa_strerror(rc)); ++g_fail; return; }
  HIPCHECK(hipDeviceSynchronize());
  auto go = ddownload(oacc, (size_t)S * H * D);
  auto gl = ddownload(lacc, (size_t)S * H);
  bool ok = true;
  ok &= report("out", cmp(go.data(), ref_out.data(), go.size()), 4e-3, 1.2e-2);
  ok &= report("lse", cmp(gl.data(), ref_lse.data(), gl.size()), 2e-3, 2e-4);

  // ---------- stand-alone merge kernel on the same data (dense only): merge o1 block into o0
  if (!varlen) {
    uint16_t* bo = dalloc<uint16_t>((size_t)S * H * D);
    float* bl = dalloc<float>((size_t)S * H);
    float* oacc2 = dalloc<float>((size_t)S * H * D);
    float* lacc2 = dalloc<float>((size_t)S * H);
    rfa_fwd_args f2 = fa;
    f2.out_acc = nullptr; f2.lse_acc = nullptr; f2.out = bo; f2.out_st = fa.q_st; f2.lse = bl;
    f2.lse_batch = (int64_t)H * S; f2.lse_head = S; f2.q_half = 0; f2.k_half = 0;
    rfa_merge_args ma;
    memset(&ma, 0, sizeof(ma));
    ma.out_acc = oacc2; ma.out_acc_st = fa.q_st; ma.lse_acc = lacc2; ma.lse_acc_batch = (int64_t)H * S; ma.lse_acc_head = S;
    ma.block_out = bo; ma.block_out_st = fa.q_st; ma.block_lse = bl; ma.block_lse_batch = (int64_t)H * S; ma.block_lse_head = S;
    ma.B = 1; ma.H = H; ma.D = D; ma.S = S; ma.dtype = RFA_BF16;
    f2.k = dkA; f2.v = dvA; f2.causal = 1; rc = rfa_fwd(&f2, nullptr);
    ma.acc_init = 1; rc |= rfa_merge(&ma, nullptr);
    f2.k = dkB; f2.v = dvB; f2.causal = 0; f2.k_half = RFA_HALF_FRONT; rc |= rfa_fwd(&f2, nullptr);
    ma.acc_init = 0; rc |= rfa_merge(&ma, nullptr);
    // slice variant: rows [S/2, S) merged with block over C
    f2.k = dkC; f2.v = dvC; f2.k_half = 0; f2.q_half = RFA_HALF_BACK; rc |= rfa_fwd(&f2, nullptr);
    rfa_merge_args mb = ma;
    const int Lh = S / 2;
    mb.out_acc = oacc2 + (size_t)Lh * H * D; mb.lse_acc = lacc2 + Lh;
    mb.block_out = bo + (size_t)Lh * H * D; mb.block_lse = bl + Lh; mb.S = S - Lh;
    rc |= rfa_merge(&mb, nullptr);
    if (rc) { printf("    merge path failed: %s\n", rfa_strerror(rc)); ++g_fail; }
    HIPCHECK(hipDeviceSynchronize());
    auto go2 = ddownload(oacc2, (size_t)S * H * D);
    auto gl2 = ddownload(lacc2, (size_t)S * H);
    ok &= report("m.out", cmp(go2.data(), ref_out.data(), go2.size()), 6e-3, 1.5e-2);
    ok &= report("m.lse", cmp(gl2.data(), ref_lse.data(), gl2.size()), 2e-3, 2e-4);
    hipFree(bo); hipFree(bl); hipFree(oacc2); hipFree(lacc2);
  }
  if (!ok) ++g_fail;
  hipFree(dq_); hipFree(dkA); hipFree(dvA); hipFree(dkB); hipFree(dvB); hipFree(dkC); hipFree(dvC);
  hipFree(oacc); hipFree(lacc);
  if (dcu) hipFree(dcu);
}

// backward accumulate mode: dq_acc/dk_acc/dv_acc over two calls == sum of two plain calls
static void run_bwd_acc_case(int S, int H, int Hk, int D, uint64_t seed) {
  printf("[case] backward accumulate mode: S=%d H=%d Hk=%d D=%d\n", S, H, Hk, D);
  const float scale = 1.0f / sqrtf((float)D);
  Rng r(seed);
  std::vector<uint16_t> q, k, v, dout, out;
  std::vector<float> qf, kf, vf, dof, of_;
  fill_normal(q, qf, (size_t)S * H * D, r);
  fill_normal(k, kf, (size_t)S * Hk * D, r);
  fill_normal(v, vf, (size_t)S * Hk * D, r);
  fill_normal(dout, dof, (size_t)S * H * D, r);
  // a consistent (out, lse): attention of q over k,v (non causal)
  std::vector<float> ro((size_t)S * H * D), rl((size_t)S * H), rdq((size_t)S * H * D), rdk((size_t)S * Hk * D),
      rdv((size_t)S * Hk * D);
  rfa_ref_fwd(qf.data(), kf.data(), vf.data(), ro.data(), rl.data(), 1, H, Hk, D, S, S, 0, 0, 0, scale, 0);
  out.resize(ro.size()); of_.resize(ro.size());
  for (size_t i = 0; i < ro.size(); ++i) { out[i] = f2bf(ro[i]); of_[i] = bf2f(out[i]); }
  rfa_ref_bwd(dof.data(), qf.data(), kf.data(), vf.data(), of_.data(), rl.data(), rdq.data(), rdk.data(),
              rdv.data(), 1, H, Hk, D, S, S, 0, 0, 0, 0, scale, 0);
  uint16_t *dq_ = dupload(q), *dk_ = dupload(k), *dv_ = dupload(v), *ddo = dupload(dout), *dout_o = dupload(out);
  float* dlse = dupload(rl);
  float* ddelta = dalloc<float>((size_t)S * H);
  float* aq = dalloc<float>((size_t)S * H * D);
  float* ak = dalloc<float>((size_t)S * Hk * D);
  float* av = dalloc<float>((size_t)S * Hk * D);
  rfa_bwd_preprocess_args pa;
  memset(&pa, 0, sizeof(pa));
  pa.dout = ddo; pa.out = dout_o; pa.dout_st = st_dense(S, H, D); pa.out_st = pa.dout_st; pa.delta = ddelta;
  pa.delta_batch = (int64_t)H * S; pa.delta_head = S; pa.B = 1; pa.H = H; pa.D = D; pa.Sq = S; pa.dtype = RFA_BF16;
  int rc = rfa_bwd_preprocess(&pa, nullptr);
  rfa_bwd_args ba;
  memset(&ba, 0, sizeof(ba));
  ba.dout = ddo; ba.q = dq_; ba.k = dk_; ba.v = dv_;
  ba.dout_st = pa.dout_st; ba.q_st = pa.dout_st; ba.k_st = st_dense(S, Hk, D); ba.v_st = ba.k_st;
  ba.lse = dlse; ba.lse_batch = (int64_t)H * S; ba.lse_head = S;
  ba.delta = ddelta; ba.delta_batch = (int64_t)H * S; ba.delta_head = S;
  ba.dq_acc = aq; ba.dk_acc = ak; ba.dv_acc = av; ba.dq_acc_st = pa.dout_st; ba.dk_acc_st = ba.k_st; ba.dv_acc_st = ba.k_st;
  ba.B = 1; ba.H = H; ba.Hk = Hk; ba.D = D; ba.Sq = S; ba.Sk = S; ba.total_k = S;
  ba.softmax_scale = scale; ba.causal = 0; ba.dtype = RFA_BF16;
  void* ws = nullptr;
  HIPCHECK(hipMalloc(&ws, rfa_bwd_workspace_bytes(&ba)));
  ba.workspace = ws;
  ba.acc_init = 1; rc |= rfa_bwd(&ba, nullptr);
  ba.acc_init = 0; rc |= rfa_bwd(&ba, nullptr);      // twice => 2x
  if (rc) { printf("    bwd acc failed: %s\n", rfa_strerror(rc)); ++g_fail; return; }
  HIPCHECK(hipDeviceSynchronize());
  auto gq = ddownload(aq, (size_t)S * H * D);
  auto gk = ddownload(ak, (size_t)S * Hk * D);
  auto gv = ddownload(av, (size_t)S * Hk * D);
  for (auto& x : rdq) x *= 2.f;
  for (auto& x : rdk) x *= 2.f;
  for (auto& x : rdv) x *= 2.f;
  bool ok = true;
  ok &= report("dq", cmp(gq.data(), rdq.data(), gq.size()), 1e-2, 2e-2);
  ok &= report("dk", cmp(gk.data(), rdk.data(), gk.size()), 1e-2, 2e-2);
  ok &= report("dv", cmp(gv.data(), rdv.data(), gv.size()), 1e-2, 2e-2);
  if (!ok) ++g_fail;
  hipFree(dq_); hipFree(dk_); hipFree(dv_); hipFree(ddo); hipFree(dout_o); hipFree(dlse); hipFree(ddelta);
  hipFree(aq); hipFree(ak); hipFree(av); hipFree(ws);
}

// ================================================================= perf + full-size spot check
static void run_perf(int S, int H, int Hk, int D, int iters) {
  printf("[perf] B=1 S=%d H=%d Hk=%d D=%d causal bf16, %d iters\n", S, H, Hk, D, iters);
  const float scale = 1.0f / sqrtf((float)D);
  Rng r(42);
  std::vector<uint16_t> q, k, v, dout;
  std::vector<float> qf, kf, vf, dof;
  fill_normal(q, qf, (size_t)S * H * D, r);
  fill_normal(k, kf, (size_t)S * Hk * D, r);
  fill_normal(v, vf, (size_t)S * Hk * D, r);
  fill_normal(dout, dof, (size_t)S * H * D, r);
  uint16_t *dq_ = dupload(q), *dk_ = dupload(k), *dv_ = dupload(v), *ddo = dupload(dout);
  uint16_t* o = dalloc<uint16_t>((size_t)S * H * D);
  float* lse = dalloc<float>((size_t)S * H);
  float* delta = dalloc<float>((size_t)S * H);
  uint16_t* gdq = dalloc<uint16_t>((size_t)S * H * D);
  uint16_t* gdk = dalloc<uint16_t>((size_t)S * Hk * D);
  uint16_t* gdv = dalloc<uint16_t>((size_t)S * Hk * D);
  rfa_fwd_args fa;
  memset(&fa, 0, sizeof(fa));
  fa.q = dq_; fa.k = dk_; fa.v = dv_; fa.out = o; fa.lse = lse;
  fa.q_st = st_dense(S, H, D); fa.k_st = st_dense(S, Hk, D); fa.v_st = fa.k_st; fa.out_st = fa.q_st;
  fa.lse_batch = (int64_t)H * S; fa.lse_head = S;
  fa.B = 1; fa.H = H; fa.Hk = Hk; fa.D = D; fa.Sq = S; fa.Sk = S; fa.softmax_scale = scale; fa.causal = 1; fa.dtype = RFA_BF16;
  rfa_bwd_preprocess_args pa;
  memset(&pa, 0, sizeof(pa));
  pa.dout = ddo; pa.out = o; pa.dout_st = fa.q_st; pa.out_st = fa.q_st; pa.delta = delta;
  pa.delta_batch = fa.lse_batch; pa.delta_head = fa.lse_head; pa.B = 1; pa.H = H; pa.D = D; pa.Sq = S; pa.dtype = RFA_BF16;
  rfa_bwd_args ba;
  memset(&ba, 0, sizeof(ba));
  ba.dout = ddo; ba.q = dq_; ba.k = dk_; ba.v = dv_;
  ba.dout_st = fa.q_st; ba.q_st = fa.q_st; ba.k_st = fa.k_st; ba.v_st = fa.k_st;
  ba.lse = lse; ba.lse_batch = fa.lse_batch; ba.lse_head = fa.lse_head;
  ba.delta = delta; ba.delta_batch = fa.lse_batch; ba.delta_head = fa.lse_head;
  ba.dq = gdq; ba.dk = gdk; ba.dv = gdv; ba.dq_st = fa.q_st; ba.dk_st = fa.k_st; ba.dv_st = fa.k_st;
  ba.B = 1; ba.H = H; ba.Hk = Hk; ba.D = D; ba.Sq = S; ba.Sk = S; ba.total_k = S; ba.softmax_scale = scale; ba.causal = 1;
  ba.dtype = RFA_BF16;
  void* ws = nullptr;
  {
    rfa_bwd_args b3 = ba; b3.phases = RFA_BWD_COMPUTE;   // phased calls always need the workspace
    const int64_t wsb = rfa_bwd_workspace_bytes(&b3);
    if (wsb) HIPCHECK(hipMalloc(&ws, wsb));
  }
  ba.workspace = ws;

  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
  const double fwd_flop = 4.0 * S * (double)S * D * H / 2.0;
  auto timeit = [&](const char* name, double flop, auto&& fn) {
    fn();
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fn();
    HIPCHECK(hipEventRecord(e1, 0));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("    %-12s %8.3f ms   %8.1f TFLOP/s (algorithmic)   %.1f%% of 2.5 PF\n", name, ms, flop / ms * 1e-9,
           flop / ms * 1e-9 / 2500.0 * 100.0);
    return ms;
  };
  float tf = timeit("fwd", fwd_flop, [&] { if (rfa_fwd(&fa, nullptr)) { printf("fwd err\n"); exit(3); } });
  float tp = timeit("bwd-pre", 0.0, [&] { rfa_bwd_preprocess(&pa, nullptr); });
  {
    rfa_bwd_args b2 = ba;
    b2.phases = RFA_BWD_COMPUTE | RFA_BWD_SKIP_DKDV;
    timeit("bwd:dq", 0.5 * fwd_flop, [&] { rfa_bwd(&b2, nullptr); });
    b2.phases = RFA_BWD_COMPUTE | RFA_BWD_SKIP_DQ;
    timeit("bwd:dkdv", 2.0 * fwd_flop, [&] { rfa_bwd(&b2, nullptr); });
  }
  float tb = timeit("bwd", 2.5 * fwd_flop, [&] { if (rfa_bwd(&ba, nullptr)) { printf("bwd err\n"); exit(3); } });
  const float tot = tf + tp + tb;
  printf("    fwd+bwd      %8.3f ms   %8.1f it/s   %8.1f TFLOP/s   %.1f%% of 2.5 PF\n", tot, 1000.f / tot,
         3.5 * fwd_flop / tot * 1e-9, 3.5 * fwd_flop / tot * 1e-9 / 2500.0 * 100.0);

  // ---- spot checks at full size (sampled rows, fp64 on host)
  HIPCHECK(hipDeviceSynchronize());
  auto go = bfvec(ddownload(o, (size_t)S * H * D));
  auto gl = ddownload(lse, (size_t)S * H);
  auto gq = bfvec(ddownload(gdq, (size_t)S * H * D));
  auto gk = bfvec(ddownload(gdk, (size_t)S * Hk * D));
  auto gv = bfvec(ddownload(gdv, (size_t)S * Hk * D));
  const int G = H / Hk;
  double eo = 0, el = 0, eq = 0, ek = 0, ev = 0, mq = 0, mk = 0, mv = 0;
  Rng rs(7);
  std::vector<double> sc(S), acc(D);
  for (int smp = 0; smp < 24; ++smp) {
    const int h = rs.next() % H, i = (smp < 4) ? (smp * (S - 1) / 3) : rs.next() % S, hk = h / G;
    double m = -1e300;
    for (int j = 0; j <= i; ++j) {
      double dot = 0;
      for (int d = 0; d < D; ++d) dot += (double)qf[((size_t)i * H + h) * D + d] * kf[((size_t)j * Hk + hk) * D + d];
      sc[j] = dot * scale; m = fmax(m, sc[j]);
    }
    double l = 0; for (int d = 0; d < D; ++d) acc[d] = 0;
    for (int j = 0; j <= i; ++j) { double p = exp(sc[j] - m); l += p; for (int d = 0; d < D; ++d) acc[d] += p * vf[((size_t)j * Hk + hk) * D + d]; }
    const double L = m + log(l);
    el = fmax(el, fabs(L - gl[(size_t)h * S + i]));
    double delta_i = 0;
    for (int d = 0; d < D; ++d) {
      eo = fmax(eo, fabs(acc[d] / l - go[((size_t)i * H + h) * D + d]));
      delta_i += (double)dof[((size_t)i * H + h) * D + d] * go[((size_t)i * H + h) * D + d];
    }
    // dq row
    std::vector<double> dqr(D, 0.0);
    for (int j = 0; j <= i; ++j) {
      double dp = 0;
      for (int d = 0; d < D; ++d) dp += (double)dof[((size_t)i * H + h) * D + d] * vf[((size_t)j * Hk + hk) * D + d];
      const double ds = exp(sc[j] - L) * (dp - delta_i) * scale;
      for (int d = 0; d < D; ++d) dqr[d] += ds * kf[((size_t)j * Hk + hk) * D + d];
    }
    for (int d = 0; d < D; ++d) { eq = fmax(eq, fabs(dqr[d] - gq[((size_t)i * H + h) * D + d])); mq = fmax(mq, fabs(dqr[d])); }
  }
  // dk/dv rows: need lse/delta of every q row of the group -> use the GPU's lse/out (validated above)
  for (int smp = 0; smp < 6; ++smp) {
    const int hk = rs.next() % Hk, j = (smp < 2) ? smp * (S - 1) : rs.next() % S;
    std::vector<double> dkr(D, 0.0), dvr(D, 0.0);
    for (int gq_ = 0; gq_ < G; ++gq_) {
      const int h = hk * G + gq_;
      for (int i = j; i < S; ++i) {
        double dot = 0, dp = 0, dl = 0;
        for (int d = 0; d < D; ++d) {
          const double qv = qf[((size_t)i * H + h) * D + d], dov = dof[((size_t)i * H + h) * D + d];
          dot += qv * kf[((size_t)j * Hk + hk) * D + d];
          dp += dov * vf[((size_t)j * Hk + hk) * D + d];
          dl += dov * go[((size_t)i * H + h) * D + d];
        }
        const double p = exp(dot * scale - gl[(size_t)h * S + i]);
        const double ds = p * (dp - dl) * scale;
        for (int d = 0; d < D; ++d) {
          dkr[d] += ds * qf[((size_t)i * H + h) * D + d];
          dvr[d] += p * dof[((size_t)i * H + h) * D + d];
        }
      }
    }
    for (int d = 0; d < D; ++d) {
      ek = fmax(ek, fabs(dkr[d] - gk[((size_t)j * Hk + hk) * D + d])); mk = fmax(mk, fabs(dkr[d]));
      ev = fmax(ev, fabs(dvr[d] - gv[((size_t)j * Hk + hk) * D + d])); mv = fmax(mv, fabs(dvr[d]));
    }
  }
  printf("    spot-check  out %.2e  lse %.2e  dq %.2e (max %.2e)  dk %.2e (max %.2e)  dv %.2e (max %.2e)\n", eo, el,
         eq, mq, ek, mk, ev, mv);
  if (eo > 1e-2 || el > 2e-3 || eq > 5e-3 + 2e-2 * mq || ek > 5e-3 + 2.5e-2 * mk || ev > 5e-3 + 2.5e-2 * mv) {
    printf("    spot-check FAIL\n");
    ++g_fail;
  }
  hipFree(dq_); hipFree(dk_); hipFree(dv_); hipFree(ddo); hipFree(o); hipFree(lse); hipFree(delta);
  hipFree(gdq); hipFree(gdk); hipFree(gdv);
  if (ws) hipFree(ws);
}

int main(int argc, char** argv) {
  bool perf = false, quick = false, perf_only = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--perf")) perf = true;
    if (!strcmp(argv[i], "--quick")) quick = true;
    if (!strcmp(argv[i], "--perf-only")) perf = perf_only = true;
  }
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s (%s), %d CUs, abi %d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, rfa_abi_version());
  if (perf_only) {
    run_perf(8192, 32, 8, 128, 20);
    printf("%s (%d failing groups)\n", g_fail ? "SELFTEST FAILED" : "SELFTEST PASSED", g_fail);
    return g_fail ? 1 : 0;
  }
  run_probes();

  std::vector<Case> cases = {
      {"tiny-noncausal", 1, 1, 1, 128, 64, 64, 0, {}, {}},
      {"tiny-causal", 1, 1, 1, 128, 64, 64, 1, {}, {}},
      {"one-block", 1, 2, 2, 128, 256, 256, 1, {}, {}},
      {"multi-block-gqa", 2, 4, 2, 128, 512, 512, 1, {}, {}},
      {"ragged-ref-fixture(239x2)", 1, 5, 5, 128, 478, 478, 1, {}, {}},      // 3824/8 rows per rank
      {"sq<sk causal (bottom-right)", 1, 3, 1, 128, 200, 456, 1, {}, {}},
      {"sq>sk causal (empty rows)", 1, 2, 2, 128, 300, 100, 1, {}, {}},
      {"noncausal rect", 1, 4, 4, 128, 333, 590, 0, {}, {}},
      {"d64", 1, 2, 1, 64, 257, 257, 1, {}, {}},
      {"d8 (llama3 test dim)", 1, 5, 5, 8, 529, 529, 1, {}, {}},
      {"d96 gqa causal (three-block instances, LDS-DMA)", 2, 4, 2, 96, 515, 515, 1, {}, {}},
      {"d80 rect noncausal (three-block instances, register staging)", 1, 3, 1, 80, 200, 456, 0, {}, {}},
      {"varlen ref fixture/8", 0, 5, 5, 128, 0, 0, 1, {0, 16, 156, 530}, {}},
      {"varlen noncausal", 0, 4, 2, 128, 0, 0, 0, {0, 120, 1248, 1500}, {}},
      {"varlen q!=k (llama3 style)", 0, 4, 2, 128, 0, 0, 1, {0, 100, 356}, {0, 300, 812}},
      {"spike keys, non-causal (forces mid-loop rescale)", 1, 2, 2, 128, 300, 520, 0, {}, {}, 1},
      {"spike keys, causal", 1, 3, 3, 128, 640, 640, 1, {}, {}, 1},
      {"d256 gqa causal (rfa_bigd.hip)", 2, 4, 2, 256, 515, 515, 1, {}, {}},
      {"d192 rect noncausal (zero-padded second chunk)", 1, 2, 1, 192, 200, 456, 0, {}, {}},
      {"d256 varlen", 0, 4, 2, 256, 0, 0, 1, {0, 120, 1248, 1500}, {}},
  };
  if (quick) cases.resize(4);
  uint64_t seed = 100;
  for (auto& c : cases) run_case(c, seed++);
  run_acc_case(512, 4, 2, 128, 900, false);
  run_acc_case(478, 5, 5, 128, 901, false);
  run_acc_case(640, 4, 4, 128, 902, true);
  run_bwd_acc_case(384, 4, 2, 128, 903);
  run_bwd_acc_case(300, 3, 3, 128, 904);
  run_acc_case(512, 4, 2, 256, 905, false);
  run_bwd_acc_case(384, 4, 2, 256, 906);

  if (perf) {
    run_perf(8192, 32, 8, 128, 20);
    run_perf(8192, 32, 32, 128, 20);
  }
  printf("%s (%d failing groups)\n", g_fail ? "SELFTEST FAILED" : "SELFTEST PASSED", g_fail);
  return g_fail ? 1 : 0;
}
