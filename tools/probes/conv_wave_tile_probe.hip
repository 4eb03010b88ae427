// K loop of an int8 3x3 convolution on gfx950 as a function of the WAVE TILE: CT 32-channel tiles x PT 32-pixel tiles of accumulators per
// wave, weights (the A operand) by 16-byte buffer loads from L2 -- one per channel tile and K = 32 sub-step -- and pixels (the B operand)
// by ds_read_b128 from a halo tile in LDS -- one per pixel tile and sub-step; CT x PT MFMAs per sub-step.  Synthetic operands, no epilogue,
// no halo refill: what the matrix pipes reach when only the K loop's own operand traffic is in the instruction stream.
//   shipped halo-patch kernel:  8 waves per CU (two per SIMD), CT = 1, PT = 5   (0.2 buffer loads + 1.0 LDS reads per MFMA)
//   candidate:                  4 waves per CU (one per SIMD, accumulators in AGPRs), CT = 2, PT = 8   (0.125 + 0.5 per MFMA)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/p tools/probes/conv_wave_tile_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NW, int CT, int PT>
__global__ __launch_bounds__(NW * 64, 1) void probe(const int8_t* __restrict__ wts, int wbytes, int* out, long long* cyc, int nsteps) {
    __shared__ __attribute__((aligned(16))) int8_t lds[96 * 1024];          // a halo tile's worth (and: one workgroup per CU)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += NW * 64) ((int*)lds)[i] = i * 2654435761u;
    __syncthreads();
    v16i acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][p][r] = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wts, 0, wbytes, 0x00020000);
    const int wbase = wave * CT * 2048;                                     // this wave's channel tiles: [step][k half][lane][16 B] per tile
    auto load_a = [&](v4i (&a)[CT], int sub) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
            a[c] = (v4i)__builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (wbase + c * 2048 + sub * (NW * CT * 2048)) % (wbytes - 4096), 0);
    };
    auto load_b = [&](v4i (&b)[PT], int sub) __attribute__((always_inline)) {
        const int8_t* hb = lds + ((sub * 7) % 40) * 1024 + lane * 16;
#pragma unroll
        for (int p = 0; p < PT; ++p) b[p] = *(const v4i*)(hb + p * 2048);
    };
    // weights: a ring of four sub-steps (requested three sub-steps ahead, as the shipped kernel's ring of three steps does); pixels: the next
    // sub-step's fragments requested before this sub-step's MFMAs
    v4i ar[4][CT], b0[PT], b1[PT];
    load_a(ar[0], 0); load_a(ar[1], 1); load_a(ar[2], 2); load_b(b0, 0);
    const long long t0 = __builtin_readcyclecounter();
    const long long r0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz, constant
    for (int s = 0; s < nsteps; s += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            load_a(ar[(u + 3) & 3], s + u + 3);
            if (u & 1) load_b(b0, s + u + 1); else load_b(b1, s + u + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[c][p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ar[u][c], (u & 1) ? b1[p] : b0[p], acc[c][p], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    int keep = 0;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) keep += acc[c][p][0] + acc[c][p][9];
    if (lane == 0) { cyc[blockIdx.x * 16 + wave] = t1 - t0; cyc[4096 + blockIdx.x * 16 + wave] = r1 - r0; }
    if (keep == 123456789) out[0] = keep;
}

template <int NW, int CT, int PT>
static void run(const int8_t* w, int wbytes, int* out, long long* cyc, const char* what) {
    const int nsteps = 72 * 40;                                            // 40 items of a 256 -> 256 layer's 72 sub-steps (4 chunks x 9 taps x 2 halves)
    probe<NW, CT, PT><<<256, NW * 64>>>(w, wbytes, out, cyc, 72);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<NW, CT, PT><<<256, NW * 64>>>(w, wbytes, out, cyc, nsteps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static long long h[2 * 256 * 16]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, rt = 0;
    for (int b = 0; b < 256; ++b) for (int wv = 0; wv < NW; ++wv) { m += h[b * 16 + wv]; rt += h[4096 + b * 16 + wv]; }
    m /= 256.0 * NW; rt /= 256.0 * NW;
    const double mfmas = (double)nsteps * CT * PT;                          // per wave
    const double ops = 256.0 * NW * mfmas * 32 * 32 * 32 * 2;
    printf("%-58s %d waves per CU, wave tile %d x %d: %7.3f ms, %5.1f cycles per MFMA per wave = %5.1f per MFMA per SIMD (32 = the pipe's rate), %6.1f TOP/s; loop %.3f ms by the 100 MHz clock -> the cycle counter ran at %.2f GHz\n",
           what, NW, CT, PT, ms, m / mfmas, m / mfmas / (NW / 4), ops / (ms * 1e-3) / 1e12, rt / 1e5, m / (rt * 10.0) );
}

int main() {
    const int wbytes = 4 << 20;
    int8_t* w; int* out; long long* cyc;
    (void)hipMalloc(&w, wbytes); (void)hipMemset(w, 1, wbytes); (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 2 * 256 * 16 * 8);
    run<8, 1, 5>(w, wbytes, out, cyc, "shipped shape (two waves per SIMD)");
    run<4, 1, 5>(w, wbytes, out, cyc, "the same tile, one wave per SIMD");
    run<4, 2, 5>(w, wbytes, out, cyc, "64 channels x 160 pixels per wave");
    run<4, 2, 8>(w, wbytes, out, cyc, "64 channels x 256 pixels per wave (16 accumulators: AGPRs)");
    run<4, 1, 8>(w, wbytes, out, cyc, "32 channels x 256 pixels per wave");
    run<8, 1, 8>(w, wbytes, out, cyc, "32 channels x 256 pixels per wave, two waves per SIMD");
    run<4, 4, 4>(w, wbytes, out, cyc, "128 channels x 128 pixels per wave");
    run<8, 2, 4>(w, wbytes, out, cyc, "64 channels x 128 pixels per wave, two waves per SIMD");
    run<8, 2, 3>(w, wbytes, out, cyc, "64 channels x 96 pixels per wave, two waves per SIMD");
    run<8, 1, 6>(w, wbytes, out, cyc, "32 channels x 192 pixels per wave, two waves per SIMD");
    run<8, 1, 4>(w, wbytes, out, cyc, "32 channels x 128 pixels per wave, two waves per SIMD");
    run<16, 1, 4>(w, wbytes, out, cyc, "32 channels x 128 pixels per wave, four waves per SIMD");
    run<16, 1, 3>(w, wbytes, out, cyc, "32 channels x 96 pixels per wave, four waves per SIMD");
    run<16, 1, 2>(w, wbytes, out, cyc, "32 channels x 64 pixels per wave, four waves per SIMD");
    run<12, 1, 3>(w, wbytes, out, cyc, "32 channels x 96 pixels per wave, three waves per SIMD");
    run<12, 1, 5>(w, wbytes, out, cyc, "32 channels x 160 pixels per wave, three waves per SIMD");
    return 0;
}
