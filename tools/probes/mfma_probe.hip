// Layout + numerics probe for the gfx950 MFMA forms the hot path uses.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_probe.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// A: [16][64] i8 row-major, B: [16][64] i8 (n-major, k contiguous), D: [16][16] i32
__global__ void k_i8_16(const int8_t* A, const int8_t* B, int* D) {
  int l = threadIdx.x;
  v4i a = *(const v4i*)(A + (l & 15) * 64 + (l >> 4) * 16);
  v4i b = *(const v4i*)(B + (l & 15) * 64 + (l >> 4) * 16);
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// A: [32][32], B: [32][32] (n-major), D: [32][32]
__global__ void k_i8_32(const int8_t* A, const int8_t* B, int* D) {
  int l = threadIdx.x;
  v4i a = *(const v4i*)(A + (l & 31) * 32 + (l >> 5) * 16);
  v4i b = *(const v4i*)(B + (l & 31) * 32 + (l >> 5) * 16);
  v16i c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// f32 32x32x2 chained over K: A [32][K], B [32][K] (n-major); D[32][32]; init C = bias[n]
__global__ void k_f32_32(const float* A, const float* B, const float* bias, float* D, int K) {
  int l = threadIdx.x;
  v16f c;
  for (int r = 0; r < 16; ++r) c[r] = bias[l & 31];
  for (int k = 0; k < K; k += 2) {
    float a = A[(l & 31) * K + k + (l >> 5)];
    float b = B[(l & 31) * K + k + (l >> 5)];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_f32_16(const float* A, const float* B, const float* bias, float* D, int K) {
  int l = threadIdx.x;
  v4f c;
  for (int r = 0; r < 4; ++r) c[r] = bias[l & 15];
  for (int k = 0; k < K; k += 4) {
    float a = A[(l & 15) * K + k + (l >> 4)];
    float b = B[(l & 15) * K + k + (l >> 4)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void k_dot4(const int* a, int* out) {
  int l = threadIdx.x;
  out[l] = __builtin_amdgcn_sdot4(a[l], 0x01010101, 5, false);
}
// correctly-rounded division + rint semantics check
__global__ void k_div(const float* x, const float* d, float* q, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) q[i] = rintf(x[i] / d[i]);
}

// --- micro benchmarks -------------------------------------------------------
__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_mfma_peak_i8(int* out, int iters) {
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
  v16i c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 0; c2[r] = 0; c3[r] = 0; }
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
  }
  int s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_mfma_peak_f32(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  v16f c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 0; c2[r] = 0; c3[r] = 0; }
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float ev_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  srand(1);
  // ---- i8 16x16x64
  {
    std::vector<int8_t> A(16 * 64), B(16 * 64); std::vector<int> D(256), R(256);
    for (auto& v : A) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : B) v = (int8_t)(rand() % 255 - 127);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int s = 0; for (int k = 0; k < 64; ++k) s += (int)A[i * 64 + k] * (int)B[j * 64 + k]; R[i * 16 + j] = s; }
    int8_t *dA, *dB; int* dD; CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    k_i8_16<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
    printf("mfma_i32_16x16x64_i8 layout: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
  }
  {
    std::vector<int8_t> A(32 * 32), B(32 * 32); std::vector<int> D(1024), R(1024);
    for (auto& v : A) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : B) v = (int8_t)(rand() % 255 - 127);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 32; ++k) s += (int)A[i * 32 + k] * (int)B[j * 32 + k]; R[i * 32 + j] = s; }
    int8_t *dA, *dB; int* dD; CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dD, 4096));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    k_i8_32<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += D[i] != R[i];
    printf("mfma_i32_32x32x32_i8 layout: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
  }
  // ---- f32 chains: bitwise equal to an fmaf chain in ascending k, C-in = bias?
  for (int which = 0; which < 2; ++which) {
    int T = which ? 16 : 32, K = 256;
    std::vector<float> A(T * K), B(T * K), bias(T), D(T * T), R(T * T);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
    for (auto& v : bias) v = (rand() / (float)RAND_MAX - 0.5f);
    for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) { float s = bias[j]; for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[j * K + k], s); R[i * T + j] = s; }
    float *dA, *dB, *db, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&db, T * 4)); CK(hipMalloc(&dD, T * T * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bias.data(), T * 4, hipMemcpyHostToDevice));
    if (which) k_f32_16<<<1, 64>>>(dA, dB, db, dD, K); else k_f32_32<<<1, 64>>>(dA, dB, db, dD, K);
    CK(hipMemcpy(D.data(), dD, T * T * 4, hipMemcpyDeviceToHost));
    int bad = 0; double maxd = 0; for (int i = 0; i < T * T; ++i) { bad += (D[i] != R[i]); maxd = fmax(maxd, fabs((double)D[i] - R[i])); }
    printf("mfma_f32_%s chain vs ascending-k fmaf chain: %s (%d/%d bit mismatches, max abs diff %.3g)\n", which ? "16x16x4" : "32x32x2", bad ? "DIFFERS" : "BIT-EXACT", bad, T * T, maxd);
  }
  // ---- sdot4
  {
    std::vector<int> a(64), o(64); for (int i = 0; i < 64; ++i) a[i] = (int)0x80FF7F01 + i;
    int *da, *dout; CK(hipMalloc(&da, 256)); CK(hipMalloc(&dout, 256)); CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice));
    k_dot4<<<1, 64>>>(da, dout); CK(hipMemcpy(o.data(), dout, 256, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 64; ++i) { int8_t* b = (int8_t*)&a[i]; int s = 5 + b[0] + b[1] + b[2] + b[3]; bad += s != o[i]; }
    printf("sdot4(x, 0x01010101): %s\n", bad ? "MISMATCH" : "OK");
  }
  // ---- division / rint
  {
    int n = 1 << 20; std::vector<float> x(n), d(n), q(n);
    for (int i = 0; i < n; ++i) { x[i] = (rand() / (float)RAND_MAX) * 40.f; d[i] = 0.001f + (rand() / (float)RAND_MAX) * 0.3f; }
    float *dx, *dd, *dq; CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dd, n * 4)); CK(hipMalloc(&dq, n * 4));
    CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, d.data(), n * 4, hipMemcpyHostToDevice));
    k_div<<<n / 256, 256>>>(dx, dd, dq, n); CK(hipMemcpy(q.data(), dq, n * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < n; ++i) bad += (q[i] != rintf(x[i] / d[i]));
    printf("rintf(x/d) vs host IEEE: %d/%d mismatches\n", bad, n);
  }
  // ---- stream copy
  {
    size_t bytes = (size_t)2 << 30; float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) k_copy<<<2048, 256>>>(a, b, bytes / 16);
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) k_copy<<<2048, 256>>>(a, b, bytes / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = ev_ms(e0, e1) / 10; printf("stream copy 2GiB->2GiB: %.3f ms  %.1f GB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e6);
    CK(hipFree(a)); CK(hipFree(b));
  }
  // ---- MFMA peaks
  {
    int* o; CK(hipMalloc(&o, 1024 * 256 * 4)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 20000; k_mfma_peak_i8<<<1024, 256>>>(o, 100);
    CK(hipEventRecord(e0)); k_mfma_peak_i8<<<1024, 256>>>(o, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    double ops = 1024.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 32; float ms = ev_ms(e0, e1);
    printf("i8 32x32x32 MFMA peak: %.1f TOPS (%.3f ms)\n", ops / ms / 1e9, ms);
    float* of = (float*)o; k_mfma_peak_f32<<<1024, 256>>>(of, 100);
    iters = 4000; CK(hipEventRecord(e0)); k_mfma_peak_f32<<<1024, 256>>>(of, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    ops = 1024.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 2; ms = ev_ms(e0, e1);
    printf("f32 32x32x2 MFMA peak: %.1f TFLOPS (%.3f ms)\n", ops / ms / 1e9, ms);
  }
  return 0;
}
