// Would a "waves over PIXELS, weights through LDS" halo-patch conv keep the int8 matrix pipes busier than the shipped form (waves over
// channels, weights L2 -> registers: 53-56 cycles per MFMA per wave)?  K loop only, synthetic operands:
//   workgroup = 8 waves (2 per SIMD, one workgroup per CU), wave = ONE 32-pixel M tile x NT 32-channel N tiles (NT = 8: 256 channels);
//   per K = 32 sub-step a wave reads 1 pixel fragment + NT weight fragments (ds_read_b128 each) and issues NT MFMAs;
//   weights arrive by LDS-DMA in steps of K = 128 (NT * 4 KB per step) into a ring of three slots, the request for step s + 2 right
//   after the barrier of step s (which every wave reaches after its first sub-step of s, i.e. done with the slot of step s - 1),
//   a 24 KB "halo tile" by LDS-DMA from a 1 GiB buffer every HEVERY steps, requested AFTER the step's weights so that the in-order
//   vmcnt wait for the weights (vmcnt(3): the wave's three tile requests may stay in flight) never lands on the tile.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/p tools/probes/wide2_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NT, bool DMA_W, bool DMA_H>
__global__ __launch_bounds__(512, 2) void probe(const int8_t* __restrict__ wts, int wbytes, const int8_t* __restrict__ act, long long abytes,
                                                int* out, long long* cyc, int nsteps, int hevery) {
    constexpr int SLOT = NT * 4096;                        // bytes of one K = 128 step of weights: [4 sub-steps][NT][lane][16 B]
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t* ring = lds;                                    // [3][SLOT]
    int8_t* halo = lds + 3 * SLOT;                         // [2][24 KB]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < (3 * SLOT + 2 * 24576) / 4; i += 512) ((int*)lds)[i] = i * 2654435761u;
    __syncthreads();
    v16i acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    const unsigned ring_l = (unsigned)(uintptr_t)((__attribute__((address_space(3))) int8_t*)ring);
    const unsigned halo_l = (unsigned)(uintptr_t)((__attribute__((address_space(3))) int8_t*)halo);
    auto dma_w = [&](int step) __attribute__((always_inline)) {        // this wave's NT / 2 KB of the step's slot
        if (!DMA_W) return;
        const unsigned dst = ring_l + (step % 3) * SLOT + wave * (SLOT / 8);
        const unsigned src = (unsigned)(((long long)step * SLOT) % (wbytes - SLOT)) + wave * (SLOT / 8) + lane * 16;
#pragma unroll
        for (int j = 0; j < SLOT / 8 / 1024; ++j)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + j * 1024), "v"(src + j * 1024), "s"(wts) : "memory", "m0");
    };
    auto dma_h = [&](int tile) __attribute__((always_inline)) {        // 24 KB from somewhere in the big buffer: 3 KB per wave
        if (!DMA_H) return;
        const unsigned dst = halo_l + (tile & 1) * 24576 + wave * 3072;
        const long long base = ((long long)(blockIdx.x * 977 + tile * 131) * 24576) % (abytes - 24576);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + j * 1024), "v"((unsigned)(wave * 3072 + j * 1024 + lane * 16)), "s"(act + base) : "memory", "m0");
    };
    dma_w(0); dma_w(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_readcyclecounter();
    v4i px[2], wf[2][NT];
    auto reads = [&](int buf, int step, int sub) __attribute__((always_inline)) {
        const int8_t* hb = halo + ((step / hevery) & 1) * 24576 + ((step * 4 + sub) % 20) * 1024 + lane * 16;
        px[buf] = *(const v4i*)hb;
        const int8_t* wb = ring + (step % 3) * SLOT + sub * (SLOT / 4) + lane * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[buf][j] = *(const v4i*)(wb + j * 1024);
    };
    auto mfmas = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[buf][j], px[buf], acc[j], 0, 0, 0);
    };
    reads(0, 0, 0);
    for (int s = 0; s < nsteps; ++s) {
        // sub-step 0, then the hand-over: step s + 1's weights (requested one step ago) must have landed, every wave is done with step s - 1
        reads(1, s, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        const bool tile_pending = DMA_H && ((s - 1) % hevery == 0) && s > 0;      // a tile was requested behind step s + 1's weights
        if (tile_pending) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        dma_w(s + 2);
        if (s % hevery == 0) dma_h(s / hevery + 1);
        reads(0, s, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        reads(1, s, 3);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        reads(0, s + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_readcyclecounter();
    int sum = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[j][r];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NT, bool DW, bool DH>
static void run(const char* name, const int8_t* w, int wbytes, const int8_t* a, long long abytes, int* out, long long* cyc, int blocks, int hevery) {
    const int nsteps = 400, lds = 3 * NT * 4096 + 2 * 24576;
    hipFuncSetAttribute((const void*)probe<NT, DW, DH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NT, DW, DH><<<blocks, 512, lds>>>(w, wbytes, a, abytes, out, cyc, nsteps, hevery);
    hipEventRecord(e0);
    probe<NT, DW, DH><<<blocks, 512, lds>>>(w, wbytes, a, abytes, out, cyc, nsteps, hevery);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(blocks);
    hipMemcpy(c.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : c) mean += (double)v; mean /= blocks;
    const double mf = (double)nsteps * 4 * NT;                     // MFMAs per wave
    const double tops = (double)blocks * 8 * mf * 32768 * 2 / (ms * 1e-3) / 1e12;
    printf("%-58s NT=%d blocks=%d: %.3f ms, %.1f cycles per MFMA per wave = %.1f per MFMA per SIMD (32 = the pipe), %.0f TOP/s\n",
           name, NT, blocks, ms, mean / mf, mean / mf / 2, tops);
}

int main() {
    const int wbytes = 885 * 1024;                                 // the 384 -> 256 layer's weights: L2 resident
    const long long abytes = 1LL << 30;
    int8_t *w, *a; int* out; long long* cyc;
    hipMalloc(&w, wbytes); hipMalloc(&a, abytes); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8);
    hipMemset(w, 1, wbytes); hipMemset(a, 2, abytes);
    for (int blocks : {1, 256}) {
        run<8, false, false>("no DMA at all (LDS reads + MFMAs + barrier per K = 128)", w, wbytes, a, abytes, out, cyc, blocks, 4);
        run<8, true, false>("weights by LDS-DMA", w, wbytes, a, abytes, out, cyc, blocks, 4);
        run<8, true, true>("weights + a 24 KB tile every 4 steps (K = 512)", w, wbytes, a, abytes, out, cyc, blocks, 4);
        run<4, true, true>("the same at 128 channels per workgroup", w, wbytes, a, abytes, out, cyc, blocks, 4);
    }
    return 0;
}
