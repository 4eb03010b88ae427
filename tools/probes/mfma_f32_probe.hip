// How fast do dependent v_mfma_f32_32x32x2_f32 chains issue?  (waves per SIMD) x (independent accumulators per wave)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ void k(float* out, int iters, const float* __restrict__ src) {
  v16f c[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[threadIdx.x + i * 64]; b[i] = src[threadIdx.x + 512 + i * 64]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[(t + i) & 7], c[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
int run(int threads, int blocks_per_cu, float* out, const float* src) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int iters = 2000, blocks = 256 * blocks_per_cu;
  k<NACC><<<blocks, threads>>>(out, 10, src);
  CK(hipEventRecord(e0)); k<NACC><<<blocks, threads>>>(out, iters, src); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double mfmas_per_simd = (double)iters * 8 * NACC * (threads / 64) * blocks_per_cu / 4.0;
  double cyc = ms * 1e-3 * 2.4e9 / mfmas_per_simd;
  double tf = (double)blocks * (threads / 64) * iters * 8 * NACC * 2.0 * 32 * 32 * 2 / (ms * 1e-3) / 1e12;
  printf("waves/SIMD=%d accs/wave=%d : %.1f cycles(@2.4GHz)/MFMA/SIMD  %.1f TFLOPS\n", threads / 256 * blocks_per_cu, NACC, cyc, tf);
  return 0;
}
int main() {
  float *out, *src; CK(hipMalloc(&out, 256 * 8 * 1024 * 4)); CK(hipMalloc(&src, 4096 * 4)); CK(hipMemset(src, 0, 4096 * 4));
  run<1>(256, 1, out, src); run<2>(256, 1, out, src); run<4>(256, 1, out, src);
  run<1>(512, 1, out, src); run<2>(512, 1, out, src);
  run<1>(512, 2, out, src); run<1>(256, 4, out, src); run<2>(512, 2, out, src);
  return 0;
}
