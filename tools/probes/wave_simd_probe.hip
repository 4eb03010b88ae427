// Which SIMD does wave w of a 512-thread (and 256-, 128-thread) workgroup run on?  (HW_REG_HW_ID: [3:0] wave slot, [5:4] SIMD, [11:8] CU)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_simd_probe tools/probes/wave_simd_probe.hip && /tmp/wave_simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned* out) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = hw;
}

int main() {
    unsigned* d; hipMalloc(&d, 1 << 20);
    for (int threads : {512, 256, 128}) {
        const int nw = threads / 64, blocks = 512;
        probe<<<blocks, threads>>>(d);
        std::vector<unsigned> h(blocks * nw);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        int same = 0, total = 0;
        printf("%d threads: SIMD of waves 0..%d of the first 6 workgroups:", threads, nw - 1);
        for (int b = 0; b < blocks; ++b) {
            if (b < 6) { printf("  ["); for (int w = 0; w < nw; ++w) printf("%u", (h[b * nw + w] >> 4) & 3); printf("]"); }
            for (int w = 0; w + 4 < nw; ++w) { same += ((h[b * nw + w] >> 4) & 3) == ((h[b * nw + w + 4] >> 4) & 3); ++total; }
        }
        if (total) printf("   wave w and w + 4 on the same SIMD: %d of %d", same, total);
        printf("\n");
    }
    return 0;
}
