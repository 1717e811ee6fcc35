// atomic_probe.hip — how fast can gfx950 absorb the fp32 atomic-add stream a fused (5-GEMM) attention
// backward would emit for dQ?  Stand-alone (no torch): hipcc --offload-arch=gfx950 -O3 tools/atomic_probe.hip
//
// Traffic model = the dK/dV kernel of csrc/rfa_bwd.hip at the headline shape (S = 8192, H = 32, Hk = 8,
// D = 128, causal): 512 workgroups (64 key blocks x 8 kv heads) x 8 waves; a workgroup walks the 64-row
// query tiles from the last one down to its causal start, and for every tile the G = 4 heads of its kv
// group; per (tile, head) it adds a 64 x 128 fp32 partial dQ into dq_acc[S][H][D] — wave (par, dblk) owns
// rows 32 par .. +31, columns 32 dblk .. +31 in MFMA C layout: 16 wave-instructions, each touching two
// 128-byte row segments (lanes 0-31 / 32-63).  4.3 GB of atomic traffic per backward.
//
// Modes:  0 atomics agent scope (sc1)      1 atomics, no scope bits (executed in the issuing XCD's L2 —
//         only coherent if a head's workgroups share an XCD; speed reference)      2 plain stores
//         3 no memory traffic (MFMA filler only)
// Filler: N dependent-chain MFMAs per wave per tile (0 = pure traffic; 40 = the fused kernel's count).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int S = 8192, H = 32, HK = 8, D = 128, G = H / HK;

template <int kMode>
__device__ __forceinline__ void emit(float* p, float v) {
  if (kMode == 0) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (kMode == 1) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (kMode == 2) *(volatile float*)p = v;
}

template <int kMode, int kFill>
__global__ __launch_bounds__(512, 2) void probe(float* dq, float* sink, int ordered) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int par = wave >> 2, dblk = wave & 3, g = lane >> 5, l31 = lane & 31;
  int idx = blockIdx.x;
  const int hk = idx % HK;
  const int kblk = idx / HK;
  const int jt0 = kblk * 2, jt1 = S / 64;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
  float* base = dq + (size_t)(32 * par + 4 * g) * H * D + 32 * dblk + l31;
  for (int jj = jt1 - 1; jj >= jt0; --jj) {
    // ordered: every workgroup is at the same tile at the same time (the kernel's walk);
    // unordered: start offsets staggered by key block
    const int j = ordered ? jj : jt0 + (jj - jt0 + kblk * 7) % (jt1 - jt0);
    for (int cg = 0; cg < G; ++cg) {
      const int h = hk * G + cg;
      if (kFill > 0) {
#pragma unroll
        for (int i = 0; i < kFill; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
      }
      if (kMode != 3) {
        float* p = base + (size_t)j * 64 * H * D + h * D;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2);
          emit<kMode>(p + (size_t)row * H * D, 1.0f);
        }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[tid] = s;
}

template <int kMode, int kFill>
static float run(float* dq, float* sink, int ordered, int reps, bool verify) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const size_t n = (size_t)S * H * D;
  CHECK(hipMemset(dq, 0, n * 4));
  hipLaunchKernelGGL((probe<kMode, kFill>), dim3(512), dim3(512), 0, 0, dq, sink, ordered);   // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(dq, 0, n * 4));
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<kMode, kFill>), dim3(512), dim3(512), 0, 0, dq, sink, ordered);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  if (verify && (kMode == 0 || kMode == 1)) {
    // element (row, h, d) receives `reps` adds from every key block whose causal range contains the row's tile
    std::vector<float> host(n);
    CHECK(hipMemcpy(host.data(), dq, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int row = 0; row < S; row += 37)
      for (int h = 0; h < H; ++h)
        for (int d = 0; d < D; d += 5) {
          const int tile = row / 64;
          const int nblk = tile / 2 + 1;   // key blocks with jt0 = 2 kblk <= tile
          const float want = (float)reps * nblk;
          if (host[((size_t)row * H + h) * D + d] != want) ++bad;
        }
    printf("   verify mode %d: %zu mismatches\n", kMode, bad);
  }
  return ms;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  float *dq, *sink;
  const size_t n = (size_t)S * H * D;
  CHECK(hipMalloc(&dq, n * 4));
  CHECK(hipMalloc(&sink, 4096));
  const double gb = 0.0;
  (void)gb;
  // bytes of atomic traffic per launch
  double tiles = 0;
  for (int kblk = 0; kblk < 64; ++kblk) tiles += (S / 64 - 2 * kblk);
  const double bytes = tiles * HK * G * 64.0 * 128 * 4;
  printf("atomic traffic per launch: %.2f GB (%.0f wave-instructions)\n", bytes / 1e9, bytes / 256);
#define RUN(mode, fill, ord, ver)                                                                  \
  {                                                                                                \
    float ms = run<mode, fill>(dq, sink, ord, reps, ver);                                          \
    printf("mode %d fill %2d %s: %.3f ms  -> %.2f TB/s of adds\n", mode, fill, ord ? "ordered  " : "unordered", ms, \
           mode == 3 ? 0.0 : bytes / ms / 1e9);                                                    \
  }
  RUN(0, 0, 1, true);
  RUN(0, 0, 0, false);
  RUN(1, 0, 1, true);
  RUN(1, 0, 0, false);
  RUN(2, 0, 1, false);
  RUN(3, 40, 1, false);
  RUN(0, 40, 1, false);
  RUN(1, 40, 1, false);
  RUN(2, 40, 1, false);
  RUN(0, 40, 0, false);
  return 0;
}
