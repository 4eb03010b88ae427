// What limits a 5-accumulator i8 MFMA loop with 2 waves per SIMD?  Variants: operands fixed / rotating register sets /
// read from LDS each step / + s_barrier per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(int* out, int steps, const int8_t* __restrict__ src) {
  __shared__ __attribute__((aligned(16))) int8_t lds[5 * 26624 + 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 5 * 26624 / 4; i += 512) ((int*)lds)[i] = i * 2654435761u;
  __syncthreads();
  v16i acc[5];
  for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  v4i fa[2][5], fb[2];
  for (int ks = 0; ks < 2; ++ks) { for (int i = 0; i < 5; ++i) fa[ks][i] = v4i{lane, i, ks, 3}; fb[ks] = v4i{lane, 9, ks, 5}; }
  for (int s = 0; s < steps; ++s) {
    if (MODE >= 2) {   // 12 fragment reads from LDS per step
      const int8_t* stg = lds + (s % 5) * 26624;
      for (int ks = 0; ks < 2; ++ks) {
        for (int i = 0; i < 5; ++i) fa[ks][i] = *(const v4i*)(stg + ((i * 32 + (lane & 31)) * 64 + ((ks * 2 + (lane >> 5)) ^ ((lane >> 2) & 3)) * 16));
        fb[ks] = *(const v4i*)(stg + 10240 + ((wave * 32 + (lane & 31)) * 64) + ks * 32 + (lane >> 5) * 16);
      }
    } else if (MODE == 1) {
      for (int ks = 0; ks < 2; ++ks) { for (int i = 0; i < 5; ++i) fa[ks][i][0] += s; fb[ks][1] ^= s; }
    }
    if (MODE >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if (MODE >= 3) __builtin_amdgcn_s_barrier();
    if (MODE >= 4) {   // 4 DMA instructions per wave per step = 32 KiB per workgroup, into the stage freed one step ago
      int8_t* dstg = lds + ((s + 4) % 5) * 26624;
      const int8_t* g = src + ((size_t)(blockIdx.x * 37 + s) % 200) * 32768 + wave * 4096 + lane * 16;
      size_t jstride = 1024;
      if (MODE >= 5) {   // the conv's gather: 16 rows x 64 B per instruction; A rows 384 B apart (pixels), B rows 3456 B apart (filters)
        const int row = lane >> 2, ch = lane & 3;
        const size_t tap = (size_t)((s % 9) / 3 * 354 + (s % 9) % 3) * 384 + (s / 9 % 6) * 64;
        g = src + (size_t)(blockIdx.x % 220) * 160 * 384 + tap + (size_t)(wave * 2 * 16 + row) * 384 + ch * 16;
        jstride = 16 * 384;
      }
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + j * jstride),
                                         (__attribute__((address_space(3))) void*)(dstg + (wave * 4 + j) * 832), 16, 0, 0);
    }
    for (int ks = 0; ks < 2; ++ks)
      for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks][i], fb[ks], acc[i], 0, 0, 0);
  }
  int sum = 0; for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}
static int8_t* src;
template <int MODE> int run(int blocks, int* out, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int steps = 2000;
  k<MODE><<<blocks, 512>>>(out, 10, src);
  CK(hipEventRecord(e0)); k<MODE><<<blocks, 512>>>(out, steps, src); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double mf = (double)steps * 10 * 2;  // MFMAs per SIMD (2 waves)
  printf("%-34s blocks=%d: %.1f cycles(@2.4GHz)/MFMA/SIMD, %.2f us per step, %.0f TOPS chip-equivalent\n", what, blocks,
         ms * 1e-3 * 2.4e9 / mf, ms * 1e3 / steps, (double)blocks * 8 * steps * 10 * 65536.0 / (ms * 1e-3) / 1e12);
  return 0;
}
int main() {
  int* out; CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&src, 64 << 20)); CK(hipMemset(src, 1, 64 << 20));
  for (int blocks : {1, 220, 256}) {
    run<0>(blocks, out, "fixed operands");
    run<1>(blocks, out, "operands touched by VALU each step");
    run<2>(blocks, out, "12 ds_read_b128 per step");
    run<3>(blocks, out, "12 ds_read_b128 + s_barrier per step");
    run<4>(blocks, out, "... + 32 KiB LDS-DMA ring per step");
    run<5>(blocks, out, "... DMA gathers 64 B row segments");
  }
  return 0;
}
