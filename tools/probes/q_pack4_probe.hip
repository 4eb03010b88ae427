// q_pack4 (csrc/common.h: the requantizer of every int8 epilogue) against clamp(rintf(y / delta) + zp, lo, 255) with the IEEE division,
// on adversarial inputs: y at (k + 0.5 + eps) * delta for eps around the rounding boundary at every scale the 1e-4 test band has to
// catch, ordinary random y, values far outside the clamp, negative and tiny ones, with and without the ReLU folded into the clamp.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I quantv2x_amd/csrc -o /tmp/p tools/probes/q_pack4_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
#include "common.h"

__global__ void run(const float* y, int n, float delta, float zp, int relu, int* out) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 >= n) return;
    const float lowc = relu ? zp + 8388608.0f : 8388608.0f;
    out[i / 4] = qv2x::q_pack4(y[i], y[i + 1], y[i + 2], y[i + 3], delta, 1.0f / delta, zp, lowc);
}

int main() {
    std::mt19937_64 g(1234);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    const int n = 1 << 22;
    float* dy; int* dout;
    hipMalloc(&dy, n * 4); hipMalloc(&dout, n);
    long long total = 0, bad = 0;
    for (int trial = 0; trial < 48; ++trial) {
        const float delta = (float)std::exp(std::log(1e-4) + u(g) * (std::log(30.0) - std::log(1e-4)));
        const float zp = (float)(int)(u(g) * 256.0) * (trial % 3 ? 1.0f : 0.0f);
        const int relu = trial & 1;
        std::vector<float> y(n);
        const double eps[] = {0, 1e-9, -1e-9, 3e-8, -3e-8, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 5e-5, -5e-5, 9.9e-5, -9.9e-5, 1.01e-4, -1.01e-4, 3e-4, -3e-4, 1e-3, -1e-3};
        for (int i = 0; i < n; ++i) {
            const int kind = i % 8;
            if (kind < 5) {                                       // at a rounding boundary of a code in or near the clamp range
                const double k = std::floor(u(g) * 300.0 - 20.0) - zp;
                float v = (float)((k + 0.5 + eps[(int)(u(g) * 21.0) % 21]) * (double)delta);
                if (kind == 4) v = std::nextafterf(v, (u(g) < 0.5 ? -1.0f : 1.0f) * INFINITY);
                y[i] = v;
            } else if (kind == 5) y[i] = (float)((u(g) * 400.0 - 70.0 - zp) * delta);
            else if (kind == 6) y[i] = (float)((u(g) - 0.5) * 1e9 * delta);
            else y[i] = (float)((u(g) - 0.5) * 1e-3 * delta);
        }
        hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
        run<<<n / 4 / 256, 256>>>(dy, n, delta, zp, relu, dout);
        std::vector<int> out(n / 4);
        hipMemcpy(out.data(), dout, n, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) {
            float c = std::nearbyintf(y[i] / delta) + zp;          // IEEE division, round to nearest even: UniformAffineQuantizer.forward
            const float lo = relu ? zp : 0.0f;
            c = c < lo ? lo : (c > 255.0f ? 255.0f : c);
            const int want = (int)c, got = ((out[i / 4] >> (8 * (i & 3))) & 255) ^ 128;
            ++total; bad += want != got;
            if (want != got && bad <= 5) printf("MISMATCH y=%.9g delta=%.9g zp=%g relu=%d: want %d got %d\n", y[i], delta, zp, relu, want, got);
        }
    }
    printf("q_pack4 vs clamp(rintf(y / delta) + zp): %lld mismatches of %lld\n", bad, total);
    // the NaN / Inf contract of common.h: NaN -> the lowest code, +-Inf saturate
    for (int relu = 0; relu < 2; ++relu) {
        const float special[8] = {NAN, INFINITY, -INFINITY, 0.0f, NAN, 1e30f, -1e30f, -0.0f};
        hipMemcpy(dy, special, sizeof(special), hipMemcpyHostToDevice);
        run<<<1, 64>>>(dy, 8, 0.05f, 7.0f, relu, dout);
        int o[2]; hipMemcpy(o, dout, 8, hipMemcpyDeviceToHost);
        printf("relu=%d zp=7: q(NaN, +Inf, -Inf, 0) = %d %d %d %d | q(NaN, 1e30, -1e30, -0) = %d %d %d %d\n", relu,
               ((o[0]) & 255) ^ 128, ((o[0] >> 8) & 255) ^ 128, ((o[0] >> 16) & 255) ^ 128, ((o[0] >> 24) & 255) ^ 128,
               ((o[1]) & 255) ^ 128, ((o[1] >> 8) & 255) ^ 128, ((o[1] >> 16) & 255) ^ 128, ((o[1] >> 24) & 255) ^ 128);
    }
    return bad != 0;
}
