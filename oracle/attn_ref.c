/*
 * oracle/attn_ref.c — TEST INFRASTRUCTURE ONLY (the parity oracle, never the product path).
 *
 * Plain-C restatement of the four private entry points of the third-party package the
 * reference delegates all arithmetic to:
 *     flash_attn.flash_attn_interface._flash_attn_forward / _flash_attn_backward /
 *     _flash_attn_varlen_forward / _flash_attn_varlen_backward
 * (PyPI `flash-attn`, Dao-AILab/flash-attention; NOT vendored under /root/reference and NOT
 * version-pinned there — pyproject.toml:1-22 lists no dependencies.  The reference's call
 * sites need the >= 2.7 behaviour: 4-tuple return, (nheads, total) varlen LSE, bottom-right
 * aligned causal mask — zigzag_ring_flash_attn.py:30-57, ring_flash_attn_varlen.py:83-88.)
 *
 * Restated algorithm (FlashAttention-2, Dao 2023, Alg. 1 & 2 — here without tiling, since
 * tiling does not change the mathematical result):
 *   forward :  S = scale * Q K^T  (+ causal mask: key j visible to query i iff
 *              j <= i + (len_k - len_q));  lse_i = log sum_j exp(S_ij);  O = softmax(S) V;
 *              GQA: q head h reads kv head h / (H/Hk);  rows with no visible key:
 *              O = 0, lse = +inf.
 *   backward:  P = exp(S - lse);  D_i = sum_d dO_id O_id;  dP = dO V^T;  dS = P o (dP - D);
 *              dQ = scale dS K;  dK = scale dS^T Q;  dV = P^T dO  (dK, dV summed over the
 *              q heads of a GQA group).
 * All arithmetic in double, inputs/outputs float.  Tensors are contiguous:
 *   dense : q (B,Sq,H,D)  k,v (B,Sk,Hk,D)  out (B,Sq,H,D)  lse (B,H,Sq)
 *   varlen: q (Tq,H,D)    k,v (Tk,Hk,D)    out (Tq,H,D)    lse (H,Tq), cu_seqlens (B+1)
 * "parity unpinned" note: the reference's own tests hold no numeric golden vectors for this
 * boundary (test/utils.py:15-38 only prints diffs); this file is pinned instead against an
 * independent fp64 softmax-attention written with torch autograd (tests/test_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int64_t q0, lq, k0, lk; /* first row and length of the q / k span of one sequence */
} span_t;

static span_t get_span(const int32_t *cu_q, const int32_t *cu_k, int b, int Sq, int Sk) {
  span_t s;
  if (cu_q) {
    s.q0 = cu_q[b];
    s.lq = cu_q[b + 1] - cu_q[b];
    s.k0 = cu_k[b];
    s.lk = cu_k[b + 1] - cu_k[b];
  } else {
    s.q0 = (int64_t)b * Sq;
    s.lq = Sq;
    s.k0 = (int64_t)b * Sk;
    s.lk = Sk;
  }
  return s;
}

/* lse index of (b, h, local row i) */
static int64_t lse_index(int varlen, int b, int h, int64_t i, int H, int Sq, int64_t Tq, int64_t q0) {
  return varlen ? (int64_t)h * Tq + q0 + i : ((int64_t)b * H + h) * Sq + i;
}

int rfa_ref_fwd(const float *q, const float *k, const float *v, float *out, float *lse, int B, int H,
                int Hk, int D, int Sq, int Sk, const int32_t *cu_q, const int32_t *cu_k, int64_t Tq,
                float scale, int causal) {
  const int G = H / Hk;
  const int varlen = cu_q != NULL;
  if (H % Hk) return -1;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    for (int h = 0; h < H; ++h) {
      const span_t s = get_span(cu_q, cu_k, b, Sq, Sk);
      const int hk = h / G;
      const int64_t off = s.lk - s.lq;
      double *sc = (double *)malloc(sizeof(double) * (size_t)(s.lk > 0 ? s.lk : 1));
      double *acc = (double *)malloc(sizeof(double) * (size_t)D);
      for (int64_t i = 0; i < s.lq; ++i) {
        const float *qi = q + ((s.q0 + i) * H + h) * D;
        int64_t jmax = s.lk; /* exclusive */
        if (causal && i + off + 1 < jmax) jmax = i + off + 1;
        float *oi = out + ((s.q0 + i) * H + h) * D;
        const int64_t li = lse_index(varlen, b, h, i, H, Sq, Tq, s.q0);
        if (jmax <= 0) {
          for (int d = 0; d < D; ++d) oi[d] = 0.f;
          lse[li] = INFINITY;
          continue;
        }
        double m = -INFINITY;
        for (int64_t j = 0; j < jmax; ++j) {
          const float *kj = k + ((s.k0 + j) * Hk + hk) * D;
          double dot = 0.0;
          for (int d = 0; d < D; ++d) dot += (double)qi[d] * (double)kj[d];
          sc[j] = dot * (double)scale;
          if (sc[j] > m) m = sc[j];
        }
        double l = 0.0;
        for (int d = 0; d < D; ++d) acc[d] = 0.0;
        for (int64_t j = 0; j < jmax; ++j) {
          const double p = exp(sc[j] - m);
          l += p;
          const float *vj = v + ((s.k0 + j) * Hk + hk) * D;
          for (int d = 0; d < D; ++d) acc[d] += p * (double)vj[d];
        }
        for (int d = 0; d < D; ++d) oi[d] = (float)(acc[d] / l);
        lse[li] = (float)(m + log(l));
      }
      free(sc);
      free(acc);
    }
  }
  return 0;
}

/* dq (like q), dk, dv (like k) are fully overwritten. */
int rfa_ref_bwd(const float *dout, const float *q, const float *k, const float *v, const float *out,
                const float *lse, float *dq, float *dk, float *dv, int B, int H, int Hk, int D,
                int Sq, int Sk, const int32_t *cu_q, const int32_t *cu_k, int64_t Tq, int64_t Tk,
                float scale, int causal) {
  const int G = H / Hk;
  const int varlen = cu_q != NULL;
  if (H % Hk) return -1;
  const int64_t nq = varlen ? Tq : (int64_t)B * Sq;
  const int64_t nk = varlen ? Tk : (int64_t)B * Sk;
  memset(dq, 0, sizeof(float) * (size_t)(nq * H * D));
  memset(dk, 0, sizeof(float) * (size_t)(nk * Hk * D));
  memset(dv, 0, sizeof(float) * (size_t)(nk * Hk * D));
  /* one thread owns one (b, kv head): no write races on dk/dv */
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    for (int hk = 0; hk < Hk; ++hk) {
      const span_t s = get_span(cu_q, cu_k, b, Sq, Sk);
      const int64_t off = s.lk - s.lq;
      double *dkacc = (double *)calloc((size_t)((s.lk > 0 ? s.lk : 1) * D), sizeof(double));
      double *dvacc = (double *)calloc((size_t)((s.lk > 0 ? s.lk : 1) * D), sizeof(double));
      double *dqacc = (double *)malloc(sizeof(double) * (size_t)D);
      for (int gq = 0; gq < G; ++gq) {
        const int h = hk * G + gq;
        for (int64_t i = 0; i < s.lq; ++i) {
          const float *qi = q + ((s.q0 + i) * H + h) * D;
          const float *doi = dout + ((s.q0 + i) * H + h) * D;
          const float *oi = out + ((s.q0 + i) * H + h) * D;
          int64_t jmax = s.lk;
          if (causal && i + off + 1 < jmax) jmax = i + off + 1;
          if (jmax <= 0) continue;
          const double L = (double)lse[lse_index(varlen, b, h, i, H, Sq, Tq, s.q0)];
          double delta = 0.0;
          for (int d = 0; d < D; ++d) delta += (double)doi[d] * (double)oi[d];
          for (int d = 0; d < D; ++d) dqacc[d] = 0.0;
          for (int64_t j = 0; j < jmax; ++j) {
            const float *kj = k + ((s.k0 + j) * Hk + hk) * D;
            const float *vj = v + ((s.k0 + j) * Hk + hk) * D;
            double dot = 0.0, dp = 0.0;
            for (int d = 0; d < D; ++d) {
              dot += (double)qi[d] * (double)kj[d];
              dp += (double)doi[d] * (double)vj[d];
            }
            const double p = exp(dot * (double)scale - L);
            const double ds = p * (dp - delta) * (double)scale;
            double *dkj = dkacc + j * D, *dvj = dvacc + j * D;
            for (int d = 0; d < D; ++d) {
              dqacc[d] += ds * (double)kj[d];
              dkj[d] += ds * (double)qi[d];
              dvj[d] += p * (double)doi[d];
            }
          }
          float *dqi = dq + ((s.q0 + i) * H + h) * D;
          for (int d = 0; d < D; ++d) dqi[d] = (float)dqacc[d];
        }
      }
      for (int64_t j = 0; j < s.lk; ++j) {
        float *dkj = dk + ((s.k0 + j) * Hk + hk) * D;
        float *dvj = dv + ((s.k0 + j) * Hk + hk) * D;
        for (int d = 0; d < D; ++d) {
          dkj[d] = (float)dkacc[j * D + d];
          dvj[d] = (float)dvacc[j * D + d];
        }
      }
      free(dkacc);
      free(dvacc);
      free(dqacc);
    }
  }
  return 0;
}

/* online merge of two partial results — restates _update_out_and_lse
 * (/root/reference/ring_flash_attn/utils.py:40-48):
 *   out <- out - sigmoid(block_lse - lse) * (out - block_out)
 *   lse <- lse - logsigmoid(lse - block_lse)
 * out (B,S,H,D), lse (B,H,S), block_out (B,S,H,D), block_lse (B,H,S); all contiguous. */
int rfa_ref_merge(float *out, float *lse, const float *block_out, const float *block_lse, int B, int S,
                  int H, int D) {
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h)
      for (int i = 0; i < S; ++i) {
        const int64_t li = ((int64_t)b * H + h) * S + i;
        const double a = lse[li], bl = block_lse[li];
        const double sig = 1.0 / (1.0 + exp(-(bl - a)));
        /* logsigmoid(x) = -log1p(exp(-x)) */
        const double x = a - bl;
        const double logsig = x >= 0 ? -log1p(exp(-x)) : x - log1p(exp(x));
        float *o = out + (((int64_t)b * S + i) * H + h) * D;
        const float *bo = block_out + (((int64_t)b * S + i) * H + h) * D;
        for (int d = 0; d < D; ++d) o[d] = (float)((double)o[d] - sig * ((double)o[d] - (double)bo[d]));
        lse[li] = (float)(a - logsig);
      }
  return 0;
}

int rfa_ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
