// Skeleton of a 160 x 256 block tile computed by 4 waves (one per SIMD), each wave a 160 x 64 register tile (5 x 2 MFMA
// accumulators, 0.7 LDS fragment reads per MFMA), BK = 64 chunks through an S-stage LDS-DMA ring, fragments for chunk s+1
// read while chunk s is multiplied.  Reports the i8 MFMA rate this structure reaches.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int MT = 5, NT = 2, NW = 4, BM = MT * 32, BN = NW * NT * 32, BK = 64, STAGE = (BM + BN) * BK;
constexpr int LA = 3, LB = 4, LPS = LA + LB;

__device__ inline int swz(int row, int ch) { return row * 64 + ((ch ^ ((row >> 2) & 3)) << 4); }

template <int S, int MODE>
__global__ __launch_bounds__(256, 1) void k(int* out, int steps, const int8_t* __restrict__ srcp) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < S * STAGE / 4; i += 256) ((int*)lds)[i] = i * 2654435761u;
  __syncthreads();
  v16i acc[MT][NT];
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  const int8_t* src[LPS]; int dst[LPS];
  for (int j = 0; j < LPS; ++j) {
    const bool is_a = j < LA;
    const int blk = is_a ? (wave * LA + j) % 10 : wave * LB + (j - LA);
    const int p = blk * 64 + lane, row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
    src[j] = is_a ? srcp + (size_t)(blockIdx.x % 220) * 160 * 384 + (size_t)row * 384 + c * 16
                  : srcp + (40 << 20) + (size_t)row * 3456 + c * 16;
    dst[j] = is_a ? blk * 1024 : BM * BK + blk * 1024;
  }
  int offA[MT][2], offB[NT][2];
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ks * 2 + (lane >> 5);
    for (int i = 0; i < MT; ++i) offA[i][ks] = swz(i * 32 + (lane & 31), ch);
    for (int j = 0; j < NT; ++j) offB[j][ks] = BM * BK + swz((wave * NT + j) * 32 + (lane & 31), ch);
  }
  int i_step = 0;
  auto issue = [&]() {
    const int tapoff = ((i_step % 9) / 3 * 354 + (i_step % 9) % 3) * 384 + (i_step / 9 % 6) * 64;
    int8_t* stage = lds + (i_step % S) * STAGE;
    for (int j = 0; j < LPS; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (j < LA ? tapoff : (i_step % 54) * 64)),
                                       (__attribute__((address_space(3))) void*)(stage + dst[j]), 16, 0, 0);
    ++i_step;
  };
  if (MODE >= 5) {   // 5: every DMA instruction reads 1 KiB contiguous; 6: only the B half is contiguous; 7: only B is loaded at all
    for (int j = 0; j < LPS; ++j) {
      const bool is_a = j < LA;
      if (MODE == 5 || !is_a) src[j] = srcp + (is_a ? (size_t)(blockIdx.x % 220) * 61440 + (wave * LA + j) % 10 * 1024 : (40 << 20) + (size_t)(wave * LB + j - LA) * 1024) + lane * 16;
    }
  }
  auto issue_lin = [&]() {
    const int aoff = (i_step % 6) * 10240, boff = (i_step % 54) * 16384;
    int8_t* stage = lds + (i_step % S) * STAGE;
    for (int j = (MODE == 7 ? LA : 0); j < LPS; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (j < LA ? aoff : boff)),
                                       (__attribute__((address_space(3))) void*)(stage + dst[j]), 16, 0, 0);
    ++i_step;
  };
  if (MODE >= 5) {
    for (int p = 0; p < S - 1; ++p) issue_lin();
    for (int s = 0; s < steps; ++s) {
      if (MODE == 7) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 3) * LB) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 3) * LPS) : "memory");
      __builtin_amdgcn_s_barrier();
      issue_lin();
    }
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
    return;
  }
  for (int p = 0; p < S - 1; ++p) issue();
  v4i fa[2][2][MT], fb[2][2][NT];
  auto read = [&](int set, const int8_t* stg) {
    for (int ks = 0; ks < 2; ++ks) {
      for (int i = 0; i < MT; ++i) fa[set][ks][i] = *(const v4i*)(stg + offA[i][ks]);
      for (int j = 0; j < NT; ++j) fb[set][ks][j] = *(const v4i*)(stg + offB[j][ks]);
    }
  };
  auto mm = [&](int set) {
    for (int ks = 0; ks < 2; ++ks)
      for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][ks][i], fb[set][ks][j], acc[i][j], 0, 0, 0);
  };
  if (MODE == 0) {        // reads of chunk s after the barrier of step s (the current kernel's order)
    for (int s = 0; s < steps; ++s) {
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * LPS) : "memory");
      __builtin_amdgcn_s_barrier();
      issue();
      read(0, lds + (s % S) * STAGE);
      mm(0);
    }
  } else {                // chunk s+1's fragments are read during chunk s's MFMAs
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * LPS) : "memory");
    __builtin_amdgcn_s_barrier();
    read(0, lds);
    for (int s = 0; s < steps; s += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (MODE != 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 3) * LPS) : "memory");
        if (MODE != 4) __builtin_amdgcn_s_barrier();
        if (MODE != 2) issue();
        if (MODE != 3) read(h ^ 1, lds + ((s + h + 1) % S) * STAGE);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 3) mm(h);
      }
    }
  }
  int sum = 0;
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}
static int8_t* src;
template <int S, int MODE> int run(int blocks, int* out, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int steps = 2000;
  CK(hipFuncSetAttribute((const void*)k<S, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, S * STAGE));
  k<S, MODE><<<blocks, 256, S * STAGE>>>(out, 10, src);
  CK(hipGetLastError());
  CK(hipEventRecord(e0)); k<S, MODE><<<blocks, 256, S * STAGE>>>(out, steps, src); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double mf = (double)steps * 2 * MT * NT;  // MFMAs per SIMD (1 wave)
  printf("%-44s S=%d blocks=%d: %.1f cycles(@2.4GHz)/MFMA/SIMD, %.2f us per step, %.0f TOPS\n", what, S, blocks,
         ms * 1e-3 * 2.4e9 / mf, ms * 1e3 / steps, (double)blocks * 4 * mf * 65536.0 / (ms * 1e-3) / 1e12);
  return 0;
}
int main() {
  int* out; CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&src, 64 << 20)); CK(hipMemset(src, 1, 64 << 20));
  for (int blocks : {1, 220, 256}) {
    run<5, 0>(blocks, out, "reads after barrier");
    run<5, 1>(blocks, out, "reads one chunk ahead");
    run<4, 1>(blocks, out, "reads one chunk ahead");
    run<4, 2>(blocks, out, "  same without the DMA ring");
    run<4, 3>(blocks, out, "  DMA ring + barrier only");
    run<4, 4>(blocks, out, "  everything but the barrier");
    run<4, 5>(blocks, out, "  DMA only, all sources contiguous");
    run<4, 6>(blocks, out, "  DMA only, B contiguous, A gathered");
    run<4, 7>(blocks, out, "  DMA only, B only (contiguous)");
  }
  return 0;
}
