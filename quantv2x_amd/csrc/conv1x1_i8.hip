// f3: the 1x1 convolutions of the HEAL Pyramid model under QuantModel -- conv1 / conv3 and the strided shortcut of QuantBottleneck
// and QuantBasicBlock (opencood/quant/quant_block.py:68-131 over quant_layer.py:391-410) -- on v_mfma_i32_32x32x32_i8.
//
// GEMM view: rows = output channels (the WEIGHTS are the A operand), columns = output pixels, K = Cin <= 512.  One wave = 32 pixels
// x 64 channels: a lane reads its pixel's 16 K-bytes per MFMA step straight from the padded i8 BEV (the map is read once per 64
// output channels, from L2 after the first), the weights arrive pre-packed in fragment order (16 B per lane, coalesced, L2
// resident: <= 128 KB per layer).  The lane then owns ONE pixel and 16 channels per tile in four runs of four, so the epilogue is
// per-lane: the zero-point correction uses the lane's own window sum, four results requantize at a time (q_pack4) and the end of a
// residual block -- out = quant(relu(conv3(x) + shortcut)) -- is fused: the shortcut comes in as fp32 (the strided 1x1 branch, or the
// decoded feature in front of the first block) or as the block input's codes, dequantized in the epilogue.
#include "common.h"

namespace qv2x {
namespace {

struct C1Args {
    const int8_t* in; const int8_t* w; const float* scale; const int32_t* corr; const int32_t* aw; const float* bias;
    const void* res; void* out;
    int n, h, wd, cin, cout, stride, ho, wo, M, relu, out_ctotal, out_c0, res_ax, tpw;
    float out_delta, out_zp, res_delta;
};

// KS = Cin / 32 MFMA steps (2, 4, 8 or 16): straight-line code, every pixel load of a tile in flight before the first MFMA.
// The 64 x Cin weight slice of the workgroup (blockIdx.y) is staged ONCE in LDS in fragment order (conflict-free ds_read_b128) and
// shared by the four waves and by the `tpw` 32-pixel tiles each wave walks: with one slice fetch per wave tile the layer moved
// 18 MB of weights through L2 for 2 MB of activations (10 us at any size).
template <int MODE, int KS>
__global__ __launch_bounds__(256) void conv1x1_i8_kernel(const C1Args a) {
    constexpr int SP = 64 + 16;                                      // staging pitch: 64 channel bytes + 16 (bank spread)
    __shared__ __attribute__((aligned(16))) int8_t smem[2 * KS * 1024 + 4 * 32 * SP + 1024];
    int8_t* wl = smem;
    int* cst = (int*)(smem + 2 * KS * 1024 + 4 * 32 * SP);          // [aw | corr | scale | bias][64] of this slice
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int cb = blockIdx.y;                                     // 64 output channels
    {
        const v4i* wsrc = (const v4i*)(a.w + ((size_t)(cb * 2) * KS) * 1024);
#pragma unroll
        for (int i = 0; i < (2 * KS * 64) / 256; ++i) ((v4i*)wl)[i * 256 + threadIdx.x] = wsrc[i * 256 + threadIdx.x];
        const int which = threadIdx.x >> 6, c = cb * 64 + (threadIdx.x & 63);
        cst[threadIdx.x] = which == 0 ? a.aw[c] : which == 1 ? a.corr[c] : which == 2 ? __float_as_int(a.scale[c]) : __float_as_int(a.bias[c]);
    }
    int8_t* stage = smem + 2 * KS * 1024 + wave * (32 * SP);
    const float rd = 1.0f / a.out_delta, lo = a.relu ? 0.0f : -3.0e38f;
    // pixel fragments of a tile: requested ahead -- the first tile's before the barrier that publishes the weight slice, the next
    // tile's before the current one's MFMAs
    auto tile_of = [&](int it) { return __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 4 + wave) * a.tpw + it)); };
    auto fetch = [&](int tile, v4i (&dst)[KS]) {
        const int mm = min(tile * 32 + l31, a.M - 1);
        const int img = mm / (a.ho * a.wo), rem = mm - img * (a.ho * a.wo);
        const int yo = rem / a.wo, xo = rem - yo * a.wo;
        const int8_t* src = a.in + ((size_t)(img * (a.h + 2) + yo * a.stride + 1) * (a.wd + 2) + xo * a.stride + 1) * a.cin + 16 * half;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dst[ks] = *(const v4i*)(src + 32 * ks);
    };
    v4i fb[KS], fbn[KS];
    fetch(tile_of(0), fbn);
    __syncthreads();
    for (int it = 0; it < a.tpw; ++it) {
        const int tile = tile_of(it);
        if (tile * 32 >= a.M) break;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fb[ks] = fbn[ks];
        if (it + 1 < a.tpw) fetch(tile_of(it + 1), fbn);
        const int m = tile * 32 + l31;
        const bool valid = m < a.M;
        const int mm = valid ? m : a.M - 1;
        const int img = mm / (a.ho * a.wo), rem = mm - img * (a.ho * a.wo);
        const int yo = rem / a.wo, xo = rem - yo * a.wo;
        const int opix = (img * (a.ho + 2) + yo + 1) * (a.wo + 2) + xo + 1;       // padded output pixel
        v4f rf[2][4];                                               // the shortcut, requested ahead of the MFMAs
        int rw[2][4];
        if (MODE == 2 && valid) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) rf[j][g] = *(const v4f*)((const float*)a.res + (size_t)m * a.cout + cb * 64 + j * 32 + 8 * g + 4 * half);
        }
        if (MODE == 3 && valid) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) rw[j][g] = *(const int*)((const int8_t*)a.res + (size_t)opix * a.cout + cb * 64 + j * 32 + 8 * g + 4 * half);
        }
        v16i acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0;
        int xs = 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const v4i fa0 = *(const v4i*)(wl + ks * 1024 + lane * 16), fa1 = *(const v4i*)(wl + (KS + ks) * 1024 + lane * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) xs = __builtin_amdgcn_sdot4(fb[ks][q], 0x01010101, xs, false);
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa0, fb[ks], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa1, fb[ks], acc[1], 0, 0, 0);
        }
        const int tot = xs + __shfl_xor(xs, 32);                   // sum of the pixel's (code - 128) over Cin
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = j * 32 + 8 * g + 4 * half, ch = cb * 64 + cl;
                const v4i xa = *(const v4i*)(cst + cl), xc = *(const v4i*)(cst + 64 + cl);
                const v4f xsc = *(const v4f*)(cst + 128 + cl), xb = *(const v4f*)(cst + 192 + cl);
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int T = acc[j][4 * g + e] + xa[e] * tot + xc[e];
                    y[e] = xb[e] + (float)T * xsc[e];
                }
                if (MODE == 1) {                                   // disable_act_quant: the fp32 map [M][Cout]
                    if (valid) {
                        v4f o = {y[0], y[1], y[2], y[3]};
                        *(v4f*)((float*)a.out + (size_t)m * a.cout + ch) = o;
                    }
                    continue;
                }
                if (MODE == 2 && valid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = y[e] + rf[j][g][e];
                }
                if (MODE == 3 && valid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = y[e] + (float)(((rw[j][g] << (24 - 8 * e)) >> 24) + a.res_ax) * a.res_delta;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], lo);
                *(int*)(stage + l31 * SP + j * 32 + 8 * g + 4 * half) = q_pack4(y[0], y[1], y[2], y[3], a.out_delta, rd, a.out_zp);
            }
        }
        if (MODE != 1) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            // 32 pixels x 64 bytes leave as 16-byte stores: chunk id -> pixel id >> 2, 16-byte piece id & 3
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int id = lane + 64 * t, px = id >> 2, piece = id & 3;
                const size_t op = (size_t)__shfl(opix, px);         // that pixel's padded output index (lane px holds it)
                const bool ok = __shfl((int)valid, px);
                if (ok) *(v4i*)((int8_t*)a.out + op * a.out_ctotal + a.out_c0 + cb * 64 + piece * 16) = *(const v4i*)(stage + px * SP + piece * 16);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int MODE>
static void launch_c1(const C1Args& a, dim3 grid, hipStream_t st) {
    switch (a.cin >> 5) {
        case 2: conv1x1_i8_kernel<MODE, 2><<<grid, 256, 0, st>>>(a); break;
        case 4: conv1x1_i8_kernel<MODE, 4><<<grid, 256, 0, st>>>(a); break;
        case 8: conv1x1_i8_kernel<MODE, 8><<<grid, 256, 0, st>>>(a); break;
        default: conv1x1_i8_kernel<MODE, 16><<<grid, 256, 0, st>>>(a); break;
    }
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_conv1x1_i8(const qv2x_conv1x1_desc* d, const int8_t* in, const int8_t* w_frag, const float* scale, const int32_t* corr,
                               const int32_t* aw, const float* bias, const void* res, void* out, void* stream) {
    using namespace qv2x;
    if (!d || !in || !w_frag || !scale || !corr || !aw || !bias || !out) return fail(QV2X_EINVAL, "qv2x_conv1x1_i8: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || (d->stride != 1 && d->stride != 2)) return fail(QV2X_EINVAL, "qv2x_conv1x1_i8: bad shape / stride");
    if ((d->cin != 64 && d->cin != 128 && d->cin != 256 && d->cin != 512) || d->cout % 64) return fail(QV2X_EALIGN, "qv2x_conv1x1_i8: cin 64 | 128 | 256 | 512, cout %% 64");
    if (d->mode < 0 || d->mode > 3) return fail(QV2X_EINVAL, "qv2x_conv1x1_i8: mode 0..3");
    if ((d->mode >= 2) && !res) return fail(QV2X_EINVAL, "qv2x_conv1x1_i8: mode %d needs the shortcut", d->mode);
    if (((uintptr_t)in & 15) || ((uintptr_t)w_frag & 15) || ((uintptr_t)out & 15) || ((uintptr_t)res & 15) || ((uintptr_t)scale & 15) ||
        ((uintptr_t)corr & 15) || ((uintptr_t)aw & 15) || ((uintptr_t)bias & 15)) return fail(QV2X_EALIGN, "qv2x_conv1x1_i8: 16-byte aligned pointers");
    C1Args a;
    a.in = in; a.w = w_frag; a.scale = scale; a.corr = corr; a.aw = aw; a.bias = bias; a.res = res; a.out = out;
    a.n = d->n; a.h = d->h; a.wd = d->w; a.cin = d->cin; a.cout = d->cout; a.stride = d->stride;
    a.ho = (d->h - 1) / d->stride + 1; a.wo = (d->w - 1) / d->stride + 1; a.M = d->n * a.ho * a.wo;
    a.relu = d->relu; a.out_delta = d->out_delta; a.out_zp = d->out_zp; a.res_ax = 128 - d->res_zx; a.res_delta = d->res_delta;
    a.out_ctotal = d->mode == 1 ? d->cout : d->out_ctotal; a.out_c0 = d->mode == 1 ? 0 : d->out_c0;
    if (d->mode != 1) {
        if (!(d->out_delta > 0.0f)) return fail(QV2X_EINVAL, "qv2x_conv1x1_i8: out_delta must be positive");
        if (d->out_ctotal % 16 || d->out_c0 % 16 || d->out_ctotal < d->out_c0 + d->cout) return fail(QV2X_EALIGN, "qv2x_conv1x1_i8: out channel window (%% 16)");
    }
    // 32-pixel tiles per wave: as many as still leave two workgroups per CU
    const int tiles = (a.M + 31) / 32, slices = a.cout / 64;
    a.tpw = 1;
    while (a.tpw < 8 && (long long)((tiles + 8 * a.tpw - 1) / (8 * a.tpw)) * slices >= 512) a.tpw *= 2;
    dim3 grid((tiles + 4 * a.tpw - 1) / (4 * a.tpw), slices);
    hipStream_t st = (hipStream_t)stream;
    switch (d->mode) {
        case 0: launch_c1<0>(a, grid, st); break;
        case 1: launch_c1<1>(a, grid, st); break;
        case 2: launch_c1<2>(a, grid, st); break;
        default: launch_c1<3>(a, grid, st); break;
    }
    return hip_check(hipGetLastError(), "qv2x_conv1x1_i8 launch");
}
