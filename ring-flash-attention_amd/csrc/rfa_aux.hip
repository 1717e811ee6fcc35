// rfa_aux.hip — the HBM-bound side kernels of the ring attention path (gfx950).
//
//   preprocess_kernel : Δ = rowsum(dO ∘ O)            (prologue of flash_attn's backward)
//   reduce_kernel     : dK/dV partial sums (io dtype or fp32 partials) + fp32 accumulate / cast
//                       (flash_attn's dk_expanded.sum + the reference's `dk += dk_buffer`,
//                       /root/reference/ring_flash_attn/zigzag_ring_flash_attn.py:182-187)
//   merge_kernel      : stand-alone online merge of (out, lse) pairs
//                       (/root/reference/ring_flash_attn/utils.py:32-73)
//   cast_kernel       : fp32 -> io dtype              (zigzag_ring_flash_attn.py:86,199)
//   lse_relayout      : (B,H,max_seqlen) <-> packed   (ring_flash_attn/triton_utils.py)
// All are pure streaming kernels: 16-byte vector accesses, one pass, no LDS.
#include "rfa_common.hpp"
#include "rfa_kernels.hpp"

namespace rfa {

// ------------------------------------------------------------------------------------
// Δ[b,h,row] = Σ_d dO·O.   16 lanes per (row, head): 16 x 8 = 128 elements.
// grid: x = ceil(rows*H / 16) blocks of 256 threads (16 row-heads per block), y = B
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const PreParams p) {
  const int b = blockIdx.y;
  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  const int row = (int)(item / p.H);
  const int h = (int)(item % p.H);
  if (row >= qs.len) return;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t arow = qs.row0 + row;
  float acc = 0.f;
  for (int d = sub * 8; d < p.D; d += 128) {                 // (one pass for head dims <= 128)
    const vec8<T> a = *(const vec8<T>*)((const T*)p.dout + qbatch * p.dout_st.batch +
                                        arow * p.dout_st.row + (int64_t)h * p.dout_st.head + d);
    const vec8<T> o = *(const vec8<T>*)((const T*)p.out + qbatch * p.out_st.batch +
                                        arow * p.out_st.row + (int64_t)h * p.out_st.head + d);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)o[e];
  }
#pragma unroll
  for (int s = 8; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
  if (sub == 0) p.delta[qbatch * p.delta_batch + (int64_t)h * p.delta_head + arow] = acc;
}

// ------------------------------------------------------------------------------------
// dst[b,row,hk,:] (=|+=) Σ_g src[b,row,hk*G+g,:]   (TS = source type: io dtype, or fp32 partials of a split launch)
// one thread per 8-element chunk; grid x = ceil(rows*Hk*16/256), y = B, z = tensor (0: src/dst, 1: src2/dst2 —
// dK and dV of one backward are reduced by ONE launch)
// ------------------------------------------------------------------------------------
template <typename T, typename TS>
__global__ __launch_bounds__(256) void reduce_kernel(const ReduceParams p) {
  const int b = blockIdx.y;
  const bool second = blockIdx.z != 0;
  const SeqSpan ks = resolve_span(p.cu_k, b, p.Sk, p.k_half);
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int sub = (int)(item & 15);
  const int64_t rh = item >> 4;
  const int row = (int)(rh / p.Hk);
  const int hk = (int)(rh % p.Hk);
  if (row >= ks.len) return;
  const int64_t kbatch = p.cu_k ? 0 : (int64_t)b;
  const int64_t arow = ks.row0 + row;
  for (int d = sub * 8; d < p.D; d += 128) {                 // (one pass for head dims <= 128)
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const TS* sp = (const TS*)(second ? p.src2 : p.src) + kbatch * p.src_st.batch + arow * p.src_st.row +
                 (int64_t)(p.g_stride ? hk : hk * p.G) * p.src_st.head + d;
  const int64_t gstep = p.g_stride ? p.g_stride : p.src_st.head;
  for (int gq = 0; gq < p.G; ++gq) {
    // io-dtype partials: flash_attn also rounds each block's dK/dV to the io dtype before they are added up in
    // fp32 (the reference's `dk += block_dk`); fp32 partials (the workgroups sharing one key block of a split
    // launch) are rounded once, after the sum
    const vec8<TS> v = *(const vec8<TS>*)(sp + (int64_t)gq * gstep);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
  }
  float* dacc = second ? p.dst_acc2 : p.dst_acc;
  if (dacc) {
    const Strides st = second ? p.dst_acc2_st : p.dst_acc_st;
    float* dp = dacc + kbatch * st.batch + arow * st.row + (int64_t)hk * st.head + d;
    f32x4 x0, x1;
    if (p.acc_init) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { x0[e] = acc[e]; x1[e] = acc[4 + e]; }
    } else {
      x0 = *(f32x4*)dp;
      x1 = *(f32x4*)(dp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x0[e] += acc[e]; x1[e] += acc[4 + e]; }
    }
    *(f32x4*)dp = x0;
    *(f32x4*)(dp + 4) = x1;
  } else {
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    f32x8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = acc[e];
    const Strides st = second ? p.dst2_st : p.dst_st;
    T* dp = (T*)(second ? p.dst2 : p.dst) + kbatch * st.batch + arow * st.row + (int64_t)hk * st.head + d;
    *(vec8<T>*)dp = __builtin_convertvector(x, vec8<T>);
  }
  }
}

// ------------------------------------------------------------------------------------
// merge: out' = out − σ(blse − lse)(out − bout),  lse' = lse − logσ(lse − blse)
// (== weights exp(x − logaddexp)).  16 lanes per (row, head).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void merge_kernel(const MergeParams p) {
  const int b = blockIdx.y;
  const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  const int row = (int)(item / p.H);
  const int h = (int)(item % p.H);
  if (row >= p.S) return;
  float* lp = p.lse_acc + (int64_t)b * p.lse_acc_batch + (int64_t)h * p.lse_acc_head + row * p.lse_acc_row;
  const float blse = p.block_lse[(int64_t)b * p.block_lse_batch + (int64_t)h * p.block_lse_head +
                                 row * p.block_lse_row];
  float wo, wb, lnew;
  if (p.acc_init) {
    wo = 0.f; wb = 1.f; lnew = blse;
  } else {
    const float lold = *lp;
    const float mx = fmaxf(lold, blse);
    if (mx == -INFINITY) {
      wo = 0.f; wb = 0.f; lnew = -INFINITY;
    } else if (mx == INFINITY) {
      // flash_attn marks empty rows with +inf; the reference formula then yields
      // out = block_out (blse=+inf) or out (lold=+inf), lse = +inf.
      wo = (lold == INFINITY) ? 1.f : 0.f; wb = 1.f - wo; lnew = INFINITY;
    } else {
      const float eo = __expf(lold - mx), eb = __expf(blse - mx);
      const float den = eo + eb;
      wo = eo / den; wb = eb / den; lnew = mx + __logf(den);
    }
  }
  for (int d = sub * 8; d < p.D; d += 128) {                 // (one pass for head dims <= 128)
    float* op = p.out_acc + (int64_t)b * p.out_acc_st.batch + (int64_t)row * p.out_acc_st.row +
                (int64_t)h * p.out_acc_st.head + d;
    const vec8<T> bv = *(const vec8<T>*)((const T*)p.block_out + (int64_t)b * p.block_out_st.batch +
                                         (int64_t)row * p.block_out_st.row +
                                         (int64_t)h * p.block_out_st.head + d);
    f32x4 x0, x1;
    if (p.acc_init) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { x0[e] = (float)bv[e]; x1[e] = (float)bv[4 + e]; }
    } else {
      x0 = *(f32x4*)op;
      x1 = *(f32x4*)(op + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x0[e] = x0[e] * wo + (float)bv[e] * wb;
        x1[e] = x1[e] * wo + (float)bv[4 + e] * wb;
      }
    }
    *(f32x4*)op = x0;
    *(f32x4*)(op + 4) = x1;
  }
  // every lane of the 16-lane group has already read *lp (same wave, program order)
  if (sub == 0) *lp = lnew;
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(T* dst, const float* src, int64_t n) {
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      f32x8 x;
      const f32x4 a = *(const f32x4*)(src + i);
      const f32x4 c = *(const f32x4*)(src + i + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[e] = a[e]; x[4 + e] = c[e]; }
      *(vec8<T>*)(dst + i) = __builtin_convertvector(x, vec8<T>);
    } else {
      for (int64_t k = i; k < n; ++k) dst[k] = (T)src[k];
    }
  }
}

// flatten: packed[h, cu[b] + i] = padded[b, h, i];  unflatten: the inverse (pad is untouched)
__global__ __launch_bounds__(256) void lse_relayout_kernel(float* dst, const float* src,
                                                           const int32_t* cu, int H, int max_seqlen,
                                                           int64_t phs, int64_t prs, int flatten) {
  const int b = blockIdx.z, h = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int start = cu[b];
  const int len = cu[b + 1] - start;
  if (i >= len || i >= max_seqlen) return;
  const int64_t pidx = (int64_t)h * phs + (int64_t)(start + i) * prs;
  const int64_t didx = ((int64_t)b * H + h) * max_seqlen + i;
  if (flatten) dst[pidx] = src[didx];
  else dst[didx] = src[pidx];
}

// ------------------------------------------------------------------------------------
static inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -1; }

int launch_preprocess(const PreParams& p, int dtype, hipStream_t stream) {
  const int64_t items = (int64_t)p.Sq * p.H;
  if (items <= 0 || p.B <= 0) return 0;
  dim3 grid((unsigned)((items + 15) / 16), (unsigned)p.B);
  if (dtype == 0) hipLaunchKernelGGL(preprocess_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(preprocess_kernel<f16_t>, grid, dim3(256), 0, stream, p);
  return ok();
}

int launch_reduce(const ReduceParams& p, int dtype, hipStream_t stream) {
  const int64_t items = (int64_t)p.Sk * p.Hk * 16;
  if (items <= 0 || p.B <= 0) return 0;
  dim3 grid((unsigned)((items + 255) / 256), (unsigned)p.B, p.src2 ? 2u : 1u);
  if (p.src_f32) {
    if (dtype == 0) hipLaunchKernelGGL((reduce_kernel<bf16_t, float>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((reduce_kernel<f16_t, float>), grid, dim3(256), 0, stream, p);
  } else {
    if (dtype == 0) hipLaunchKernelGGL((reduce_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((reduce_kernel<f16_t, f16_t>), grid, dim3(256), 0, stream, p);
  }
  return ok();
}

// ------------------------------------------------------------------------------------
// combine the partial (out, lse) pairs of a split-KV forward launch (rfa_fwd.hip: kv_nsplit):
//     lse = logsumexp_s lse_s,   out = sum_s exp(lse_s - lse) out_s           (out_s normalised, lse_s = -inf: no key)
// and deliver the result like the forward's own epilogue would have: plain (out io dtype, lse; +inf for rows without a
// key) or merged into the fp32 (out_acc, lse_acc) pair with the same formula as the fused epilogue (acc_init: overwrite).
// 16 lanes per (row, head), 8 columns per lane and pass.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void combine_kernel(const CombineParams p) {
  const int b = blockIdx.y;
  const SeqSpan qs = resolve_span(p.cu_q, b, p.Sq, p.q_half);
  const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  const int row = (int)(item / p.H);
  const int h = (int)(item % p.H);
  if (row >= qs.len) return;
  const int64_t qbatch = p.cu_q ? 0 : (int64_t)b;
  const int64_t arow = qs.row0 + row;
  const float* lp0 = p.part_lse + qbatch * p.part_lse_batch + (int64_t)h * p.part_lse_head + arow;
  float mx = -INFINITY;
  for (int s = 0; s < p.nsplit; ++s) mx = fmaxf(mx, lp0[(int64_t)s * p.part_lse_split]);
  const bool has = mx > -INFINITY;
  float den = 0.f;
  for (int s = 0; s < p.nsplit; ++s) den += has ? __expf(lp0[(int64_t)s * p.part_lse_split] - mx) : 0.f;
  const float blse = has ? mx + __logf(den) : -INFINITY;
  // weights of the combined block inside the caller's accumulators (plain mode: the block IS the result)
  float wo = 0.f, wb = 1.f, lnew = blse;
  float* la = nullptr;
  if (p.out_acc != nullptr) {
    la = p.lse_acc + qbatch * p.lse_acc_batch + (int64_t)h * p.lse_acc_head + arow;
    if (!p.acc_init) {
      const float lold = *la;
      const float m2 = fmaxf(lold, blse);
      if (m2 > -INFINITY) {
        const float eo = __expf(lold - m2), eb = __expf(blse - m2);
        wo = eo / (eo + eb);
        wb = eb / (eo + eb);
        lnew = m2 + __logf(eo + eb);
      } else {
        wo = 1.f; wb = 0.f; lnew = lold;
      }
    }
  }
  const float* pp = p.part_out + qbatch * p.part_st.batch + arow * p.part_st.row + (int64_t)h * p.part_st.head;
  for (int d = sub * 8; d < p.D; d += 128) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int s = 0; s < p.nsplit; ++s) {
      const float ls = lp0[(int64_t)s * p.part_lse_split];
      const float w = has && ls > -INFINITY ? __expf(ls - mx) / den : 0.f;
      const f32x4 x0 = *(const f32x4*)(pp + (int64_t)s * p.part_out_split + d);
      const f32x4 x1 = *(const f32x4*)(pp + (int64_t)s * p.part_out_split + d + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[e] += w * x0[e]; acc[4 + e] += w * x1[e]; }
    }
    if (p.out_acc != nullptr) {
      float* ap = p.out_acc + qbatch * p.out_acc_st.batch + arow * p.out_acc_st.row + (int64_t)h * p.out_acc_st.head + d;
      f32x4 y0, y1;
      if (p.acc_init) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { y0[e] = acc[e]; y1[e] = acc[4 + e]; }
      } else {
        y0 = *(f32x4*)ap;
        y1 = *(f32x4*)(ap + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { y0[e] = y0[e] * wo + acc[e] * wb; y1[e] = y1[e] * wo + acc[4 + e] * wb; }
      }
      *(f32x4*)ap = y0;
      *(f32x4*)(ap + 4) = y1;
    } else {
      typedef float f32x8 __attribute__((ext_vector_type(8)));
      f32x8 x;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = acc[e];
      T* op = (T*)p.out + qbatch * p.out_st.batch + arow * p.out_st.row + (int64_t)h * p.out_st.head + d;
      *(vec8<T>*)op = __builtin_convertvector(x, vec8<T>);
    }
  }
  if (sub == 0) {
    if (p.out_acc != nullptr) *la = lnew;
    else p.lse[qbatch * p.lse_batch + (int64_t)h * p.lse_head + arow] = has ? blse : INFINITY;
  }
}

int launch_combine(const CombineParams& p, int dtype, hipStream_t stream) {
  const int rows = p.q_half ? (p.Sq + 1) / 2 : p.Sq;
  const int64_t items = (int64_t)rows * p.H;
  if (items <= 0 || p.B <= 0) return 0;
  dim3 grid((unsigned)((items + 15) / 16), (unsigned)p.B);
  if (dtype == 0) hipLaunchKernelGGL(combine_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(combine_kernel<f16_t>, grid, dim3(256), 0, stream, p);
  return ok();
}

int launch_merge(const MergeParams& p, int dtype, hipStream_t stream) {
  const int64_t items = (int64_t)p.S * p.H;
  if (items <= 0 || p.B <= 0) return 0;
  dim3 grid((unsigned)((items + 15) / 16), (unsigned)p.B);
  if (dtype == 0) hipLaunchKernelGGL(merge_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(merge_kernel<f16_t>, grid, dim3(256), 0, stream, p);
  return ok();
}

int launch_cast(void* dst, const float* src, int64_t n, int dtype, hipStream_t stream) {
  if (n <= 0) return 0;
  int64_t blocks = (n + 2047) / 2048;
  if (blocks > 4096) blocks = 4096;
  if (dtype == 0) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, (bf16_t*)dst, src, n);
  else hipLaunchKernelGGL(cast_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, (f16_t*)dst, src, n);
  return ok();
}

int launch_lse_relayout(float* dst, const float* src, const int32_t* cu, int B, int H,
                        int max_seqlen, int64_t phs, int64_t prs, bool flatten, hipStream_t stream) {
  if (B <= 0 || H <= 0 || max_seqlen <= 0) return 0;
  dim3 grid((unsigned)((max_seqlen + 255) / 256), (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(lse_relayout_kernel, grid, dim3(256), 0, stream, dst, src, cu, H, max_seqlen,
                     phs, prs, flatten ? 1 : 0);
  return ok();
}

}  // namespace rfa
