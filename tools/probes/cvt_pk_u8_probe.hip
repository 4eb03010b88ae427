// v_cvt_pk_u8_f32 on gfx950: rounding mode, saturation, NaN; and its issue rate beside v_med3_f32 + v_perm_b32 (what q_pack4 uses now).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cvt_pk_u8_probe tools/probes/cvt_pk_u8_probe.hip && /tmp/cvt_pk_u8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void conv(const float* x, unsigned* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0);
}

template <int OP>
__global__ void rate(int* out, int n, float seed) {
    float f[8]; unsigned a[8];
    for (int i = 0; i < 8; ++i) { f[i] = seed + i + threadIdx.x * 0.25f; a[i] = i + threadIdx.x; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_cvt_pk_u8_f32(f[i], 1, a[i]);
            if (OP == 1) f[i] = __builtin_amdgcn_fmed3f(f[i], 1.0f, 255.0f);
            if (OP == 2) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
            if (OP == 3) f[i] = fmaxf(f[i], 0.5f);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (int)(t1 - t0);
}

int main() {
    const float xs[] = {0.0f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 2.4999f, 2.5001f, 3.5f, 126.5f, 127.5f, 254.5f, 254.51f, 255.0f, 255.49f, 255.5f, 256.0f, 300.0f, 1e9f,
                        -0.4f, -0.5f, -0.6f, -1.0f, -3.0f, -1e9f, NAN, INFINITY, -INFINITY};
    const int n = sizeof(xs) / sizeof(float);
    float* dx; unsigned* d; hipMalloc(&dx, sizeof(xs)); hipMalloc(&d, 1 << 22);
    hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
    conv<<<1, 64>>>(dx, d, n);
    unsigned h[64]; hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int rne_ok = 1;
    for (int i = 0; i < n; ++i) {
        float r = std::isnan(xs[i]) ? 0.0f : fminf(fmaxf(rintf(xs[i]), 0.0f), 255.0f);
        printf("cvt_pk_u8_f32(%g) = %u   (clamp(rint) = %g)%s\n", xs[i], h[i] & 255, r, (h[i] & 255) == (unsigned)r ? "" : "   <-- differs");
        if (!std::isnan(xs[i]) && (h[i] & 255) != (unsigned)r) rne_ok = 0;
    }
    printf("round-to-nearest-even + saturation on these inputs: %s\n", rne_ok ? "YES" : "NO");
    const char* names[4] = {"v_cvt_pk_u8_f32", "v_med3_f32", "v_fma_f32", "v_max_f32"};
    void (*ks[4])(int*, int, float) = {rate<0>, rate<1>, rate<2>, rate<3>};
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int op = 0; op < 4; ++op) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            ks[op]<<<1024, 64 * waves>>>(reinterpret_cast<int*>(d), 4096, 3.0f);
            hipEventRecord(e0);
            ks[op]<<<1024, 64 * waves>>>(reinterpret_cast<int*>(d), 4096, 3.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
            printf("%-16s %d wave(s)/SIMD: %.3f ms, wave 0: %.2f ticks/instr\n", names[op], waves, ms, (double)cyc / (4096 * 8));
        }
    return 0;
}
