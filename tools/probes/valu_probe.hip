// Issue rate of the VALU instructions of the requantizing epilogue on gfx950: plain vs packed f32, cvt, rndne, med3, perm.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/probes/valu_probe.hip && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void probe(int* out, int n, float seed) {
    float f[8];
    v2f p[8];
    int a[8];
    for (int i = 0; i < 8; ++i) { f[i] = seed + i + threadIdx.x; p[i] = v2f{seed + i, seed - i}; a[i] = i + threadIdx.x; }
    const v2f m2 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) f[i] = f[i] * 1.0001f;                                   // v_mul_f32
            if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
            if (OP == 4) f[i] = __builtin_rintf(f[i]);                            // v_rndne_f32
            if (OP == 5) f[i] = (float)__float_as_int(f[i]);                      // v_cvt_f32_i32
            if (OP == 6) f[i] = __builtin_amdgcn_fmed3f(f[i], 1.0f, 255.0f);      // v_med3_f32
            if (OP == 7) a[i] = __builtin_amdgcn_perm(a[i], a[(i + 1) & 7], 0x0c0c0400u);
            if (OP == 8) a[i] = __mul24(a[i], a[(i + 1) & 7]) + a[(i + 2) & 7];   // v_mad_i32_i24
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + p[i][0] + p[i][1] + (float)a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (int)(t1 - t0);
}

int main() {
    int* d; hipMalloc(&d, 1 << 22);
    const int n = 4096;
    const char* names[9] = {"v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_rndne_f32", "v_cvt_f32_i32", "v_med3_f32", "v_perm_b32", "v_mad_i32_i24"};
    void (*ks[9])(int*, int, float) = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>};
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int op = 0; op < 9; ++op) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            dim3 grid(256 * 4), block(64 * waves);
            ks[op]<<<grid, block>>>(d, n, 3.0f);
            hipEventRecord(e0);
            ks[op]<<<grid, block>>>(d, n, 3.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
            printf("%-16s waves/block %d: %.3f ms, wave 0: %.2f cycles/instr\n", names[op], waves, ms, (double)cyc / (n * 8));
        }
    return 0;
}
