// What does it cost the VALU to READ the result registers of i8 MFMAs on gfx950?  (profiles/r06_cand_ablations.log: in the candidate stage of the
// two-stage encode the MFMAs alone cost 88 us, the epilogue's arithmetic alone 27 us, together 640 us -- "the time goes with the number of
// MFMA-result registers the VALU reads".)  One wave per SIMD (or two), NACC accumulators of v_mfma_i32_32x32x32_i8, per iteration 8 MFMAs per
// accumulator and then an "epilogue" of one v_add per register in several forms:
//   MODE 0  MFMAs only                                   MODE 1  + read every result register once (s += acc[c][r])
//   MODE 2  + read, then re-initialise it (acc = it)     MODE 3  + the same number of reads of OTHER registers (never written by an MFMA)
//   MODE 4  no MFMAs, the reads of mode 1                 MODE 5  + every second result register read twice
//   MODE 6  + four dependent VALU per result register     MODE 7  MFMAs, then 8 x s_nop 15, then the reads of mode 1
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rr tools/probes/mfma_result_read_probe.hip && /tmp/rr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE, int NACC, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void probe(int n, long long* out, int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v16i acc[NACC];
    int other[NACC][16];
    for (int c = 0; c < NACC; ++c) for (int r = 0; r < 16; ++r) { acc[c][r] = 0; other[c][r] = lane * (c + 3) + r; }
    v4i a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, 7, 11, 13};
    int s = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        if (MODE != 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(k & 1 ? b : a, k & 1 ? a : b, acc[c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 7) {
#pragma unroll
            for (int z = 0; z < 8; ++z) asm volatile("s_nop 15");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1 || MODE == 4 || MODE == 7) {
#pragma unroll
            for (int c = 0; c < NACC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[c][r];
        }
        if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < NACC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) { s += acc[c][r]; acc[c][r] = it; }
        }
        if (MODE == 3) {
#pragma unroll
            for (int c = 0; c < NACC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) { s += other[c][r]; }
            other[it % NACC][it & 15] ^= s;
        }
        if (MODE == 5) {
#pragma unroll
            for (int c = 0; c < NACC; ++c)
#pragma unroll
                for (int r = 0; r < 16; r += 2) { s += acc[c][r]; s ^= acc[c][r]; }
        }
        if (MODE == 6) {
#pragma unroll
            for (int c = 0; c < NACC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) { int t = acc[c][r]; t = t * 3 + it; t ^= t >> 3; t += lane; s += t; }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_readcyclecounter();
    for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][9] + other[c][3];
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
    if (s == 123456789) sink[0] = s;
}

template <int MODE, int NACC, int WPE> void run(long long* d, int* sink, const char* what) {
    const int n = 400, blocks = 256 * WPE;
    probe<MODE, NACC, WPE><<<blocks, 256>>>(n, d, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<MODE, NACC, WPE><<<blocks, 256>>>(n, d, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[512 * 4]; hipMemcpy(h, d, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks * 4; ++i) m += h[i];
    m /= blocks * 4.0 * n;
    printf("mode %d  %2d accumulators  %d wave(s)/SIMD: %8.3f ms  %8.0f ticks per iteration per wave  (%d MFMAs, %d result registers)  %s\n", MODE, NACC, WPE, ms, m,
           MODE == 4 ? 0 : 8 * NACC, 16 * NACC, what);
}

int main() {
    long long* d; int* sink; hipMalloc(&d, 512 * 4 * 8); hipMalloc(&sink, 64);
    run<0, 12, 1>(d, sink, "MFMAs only");
    run<4, 12, 1>(d, sink, "reads only");
    run<1, 12, 1>(d, sink, "MFMAs + one read per result register");
    run<2, 12, 1>(d, sink, "MFMAs + read + re-initialise");
    run<3, 12, 1>(d, sink, "MFMAs + the same reads of other registers");
    run<5, 12, 1>(d, sink, "MFMAs + every second result register twice");
    run<6, 12, 1>(d, sink, "MFMAs + four dependent VALU per result register");
    run<7, 12, 1>(d, sink, "MFMAs, 8 x s_nop 15, reads");
    run<0, 4, 1>(d, sink, "MFMAs only");
    run<1, 4, 1>(d, sink, "MFMAs + one read per result register");
    run<0, 4, 2>(d, sink, "MFMAs only");
    run<4, 4, 2>(d, sink, "reads only");
    run<1, 4, 2>(d, sink, "MFMAs + one read per result register");
    run<2, 4, 2>(d, sink, "MFMAs + read + re-initialise");
    run<6, 4, 2>(d, sink, "MFMAs + four dependent VALU per result register");
    return 0;
}
