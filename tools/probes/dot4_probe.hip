// Issue rate of v_dot4_i32_i8 (sdot4) against v_fma_f32 and v_mad_i32_i24 on gfx950: one wave per SIMD slot, 8 independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dot4_probe tools/probes/dot4_probe.hip && /tmp/dot4_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void probe(int* out, int n, int seed) {
    int a[8], b = seed + threadIdx.x;
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = i + threadIdx.x; f[i] = (float)i; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_sdot4(b, a[i] | 0x01010101, a[i], false);
            if (OP == 1) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
            if (OP == 2) a[i] = __mul24(a[i], b) + i;
            if (OP == 3) a[i] = __builtin_amdgcn_udot4(b, a[i] | 0x01010101, a[i], false);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (int)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (int)(t1 - t0);
}

int main() {
    int* d; hipMalloc(&d, 1 << 20);
    const int n = 4096;
    const char* names[4] = {"v_dot4_i32_i8 (sdot4)", "v_fma_f32", "v_mul_i24 + add", "v_dot4_u32_u8 (udot4)"};
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int op = 0; op < 4; ++op) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            dim3 grid(256 * 4), block(64 * waves);              // every SIMD gets `waves` waves (roughly)
            void (*k)(int*, int, int) = op == 0 ? probe<0> : op == 1 ? probe<1> : op == 2 ? probe<2> : probe<3>;
            k<<<grid, block>>>(d, n, 3);
            hipEventRecord(e0);
            k<<<grid, block>>>(d, n, 3);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
            printf("%-24s waves/block %d: %.3f ms, wave 0: %d cycles for %d instr -> %.2f cycles/instr\n", names[op], waves, ms, cyc, n * 8, (double)cyc / (n * 8));
        }
    return 0;
}
