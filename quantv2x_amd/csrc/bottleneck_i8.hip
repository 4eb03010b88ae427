// f3: one whole QuantBottleneck (quant_block.py:100-131 over resblock.py:69-128) in ONE launch, for the stride-1 blocks without a 1x1 shortcut
// (13 of the 16 of the HEAL Pyramid model):   x --conv1 1x1--> t1 --grouped 3x3--> t2 --conv3 1x1--> (+ x) --ReLU, block quantizer--> out
// Three launches of 8-16 us each spend most of their time in launch ramp and operand round trips on maps of 4 400 .. 70 400 cells; here
// t1 and t2 never leave the CU.  A workgroup owns a 2 x 32 patch of output pixels and ALL channels:
//   A  conv1 on the 4 x 34 HALO patch (136 pixels = five 32-pixel MFMA tiles, 2.1x the patch: the recompute that buys the fusion), the
//      wave's output-channel tiles with their weights in registers, results requantized into LDS; halo pixels outside the map get the
//      code of 0.0 (the zero padding of the 3x3 in the dequantized domain)
//   B  the grouped 3x3 as block-diagonal 32-channel slabs (gconv3x3_i8.hip) straight from the LDS tile, slabs dealt over the waves
//   C  conv3 on the 2 x 32 patch from LDS, (row, channel tile) pairs dealt over the waves, + the block input's codes dequantized
//      (`out += residual`), ReLU, the block's quantizer, 16-byte stores through an LDS stage
// Integer arithmetic, constants and results are those of the three separate kernels (tests compare every block's codes with the oracle).
#include "common.h"

namespace qv2x {
namespace {

struct BnArgs {
    const int8_t* x; int8_t* out;
    const int8_t* w1; const float* sc1; const int* cr1; const int* aw1; const float* bs1;
    const int8_t* w2; const float* sc2; const int* cr2; const int* aw2; const float* bs2;
    const int8_t* w3; const float* sc3; const int* cr3; const int* aw3; const float* bs3;
    int n, h, w, cin, width, cg;
    float d1, z1, d2, z2, db, zb;          // output quantizers of conv1, conv2 and of the block
    int res_ax; float res_delta;            // the block input's quantizer (128 - zx, delta)
};

constexpr int TH = 2, TW = 32, HPH = TH + 2, HPW = TW + 2, HP = HPH * HPW;        // 136 halo pixels

// KS1 = cin / 32 (2, 4, 8), NT1 = conv1 channel tiles per wave = width / 128 (1, 2, 4)
template <int KS1, int NT1>
__global__ __launch_bounds__(256) void bottleneck_i8_kernel(const BnArgs a) {
    constexpr int WIDTH = NT1 * 128, P1 = WIDTH + 16;                             // LDS pitch of t1 / t2 rows
    constexpr int KS3 = WIDTH / 32, SP = 48;
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    int8_t* t1 = smem;                                  // [HP][P1]
    int8_t* t2 = smem + HP * P1;                        // [TH * 32][P1]
    int8_t* stage = t2 + TH * 32 * P1;                  // [4 waves][32][SP]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int tiles_x = (a.w + TW - 1) / TW, tiles_y = (a.h + TH - 1) / TH;
    const int img = blockIdx.x / (tiles_x * tiles_y), trem = blockIdx.x - img * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const int planes = a.cin;

    // ---------------- A: conv1 on the halo patch -----------------------------------------------------------------------------------------
    {
        v4i wf[NT1][KS1];                               // this wave's channel tiles: wave + 4 j
#pragma unroll
        for (int j = 0; j < NT1; ++j)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) wf[j][ks] = *(const v4i*)(a.w1 + ((size_t)((wave + 4 * j) * KS1 + ks)) * 1024 + lane * 16);
        const float rd = 1.0f / a.d1;
        const int zcode = ((int)a.z1 - 128) & 0xff;
        const int zfill = zcode | (zcode << 8) | (zcode << 16) | (zcode << 24);
        // the pixel fragments of halo tile pt + 1 are requested before the MFMAs of tile pt
        auto fetch = [&](int pt, v4i (&dst)[KS1]) {
            const int hp = pt * 32 + l31;
            const int hy = hp / HPW, hx = hp - hy * HPW;
            const int yc = min(max(y0 - 1 + hy, -1), a.h), xc = min(max(x0 - 1 + hx, -1), a.w);      // stay inside the padded tensor
            const int8_t* src = a.x + ((size_t)(img * (a.h + 2) + yc + 1) * (a.w + 2) + xc + 1) * a.cin + 16 * half;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) dst[ks] = *(const v4i*)(src + 32 * ks);
        };
        v4i fb[KS1], fbn[KS1];
        fetch(0, fbn);
#pragma unroll 1
        for (int pt = 0; pt < (HP + 31) / 32; ++pt) {
            const int hp = pt * 32 + l31;
            const int hy = hp / HPW, hx = hp - hy * HPW;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool inside = hp < HP && y >= 0 && y < a.h && x >= 0 && x < a.w;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) fb[ks] = fbn[ks];
            if (pt + 1 < (HP + 31) / 32) fetch(pt + 1, fbn);
            v16i acc[NT1];
#pragma unroll
            for (int j = 0; j < NT1; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0;
            int xs = 0;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xs = __builtin_amdgcn_sdot4(fb[ks][q], 0x01010101, xs, false);
#pragma unroll
                for (int j = 0; j < NT1; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j][ks], fb[ks], acc[j], 0, 0, 0);
            }
            const int tot = xs + __shfl_xor(xs, 32);
#pragma unroll
            for (int j = 0; j < NT1; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = (wave + 4 * j) * 32 + 8 * g + 4 * half;
                    const v4i xa = *(const v4i*)(a.aw1 + ch), xc4 = *(const v4i*)(a.cr1 + ch);
                    const v4f xsc = *(const v4f*)(a.sc1 + ch), xb = *(const v4f*)(a.bs1 + ch);
                    float yv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) yv[e] = fmaxf(xb[e] + (float)(acc[j][4 * g + e] + xa[e] * tot + xc4[e]) * xsc[e], 0.0f);
                    const int packed = q_pack4(yv[0], yv[1], yv[2], yv[3], a.d1, rd, a.z1);
                    if (hp < HP) *(int*)(t1 + hp * P1 + ch) = inside ? packed : zfill;
                }
            }
        }
    }
    __syncthreads();

    // ---------------- B: grouped 3x3 from the LDS tile, slab = wave + 4 k ----------------------------------------------------------------
    {
        v4i ones;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int wv = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) wv |= (((16 * half + 4 * q + e) / a.cg) == (l31 / a.cg) ? 1 : 0) << (8 * e);
            ones[q] = wv;
        }
        const float rd = 1.0f / a.d2;
#pragma unroll 1
        for (int k = 0; k < NT1; ++k) {                  // WIDTH / 32 slabs over four waves = NT1 each
            const int slab = wave + 4 * k, c0 = slab * 32;
            v4i wg[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wg[t] = *(const v4i*)(a.w2 + (size_t)slab * (9 * 1024) + t * 1024 + lane * 16);
#pragma unroll
            for (int ty = 0; ty < TH; ++ty) {
                v16i acc, sum;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[r] = 0; sum[r] = 0; }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const v4i fb = *(const v4i*)(t1 + ((ty + t / 3) * HPW + l31 + t % 3) * P1 + c0 + half * 16);
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wg[t], fb, acc, 0, 0, 0);
                    sum = __builtin_amdgcn_mfma_i32_32x32x32_i8(ones, fb, sum, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = c0 + 8 * g + 4 * half;
                    const v4i xa = *(const v4i*)(a.aw2 + ch), xc4 = *(const v4i*)(a.cr2 + ch);
                    const v4f xsc = *(const v4f*)(a.sc2 + ch), xb = *(const v4f*)(a.bs2 + ch);
                    float yv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) yv[e] = fmaxf(xb[e] + (float)(acc[4 * g + e] + xa[e] * sum[4 * g + e] + xc4[e]) * xsc[e], 0.0f);
                    *(int*)(t2 + (ty * 32 + l31) * P1 + ch) = q_pack4(yv[0], yv[1], yv[2], yv[3], a.d2, rd, a.z2);
                }
            }
        }
    }
    __syncthreads();

    // ---------------- C: conv3 + shortcut + ReLU + the block quantizer; pairs (row, channel tile) dealt over the waves -------------------
    {
        const int ntile3 = planes >> 5, pairs = TH * ntile3;
        const float rd = 1.0f / a.db;
        int8_t* st = stage + wave * (32 * SP);
#pragma unroll 1
        for (int pr = wave; pr < pairs; pr += 4) {
            const int ty = pr / ntile3, nt = pr - ty * ntile3;
            v16i acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0;
            int xs = 0;
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks) {
                const v4i fb = *(const v4i*)(t2 + (ty * 32 + l31) * P1 + 32 * ks + 16 * half);
                const v4i fa = *(const v4i*)(a.w3 + ((size_t)(nt * KS3 + ks)) * 1024 + lane * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) xs = __builtin_amdgcn_sdot4(fb[q], 0x01010101, xs, false);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
            }
            const int tot = xs + __shfl_xor(xs, 32);
            const int y = y0 + ty, x = x0 + l31;
            const bool valid = y < a.h && x < a.w;
            const size_t pix = (size_t)(img * (a.h + 2) + min(y, a.h - 1) + 1) * (a.w + 2) + min(x, a.w - 1) + 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = nt * 32 + 8 * g + 4 * half;
                const v4i xa = *(const v4i*)(a.aw3 + ch), xc4 = *(const v4i*)(a.cr3 + ch);
                const v4f xsc = *(const v4f*)(a.sc3 + ch), xb = *(const v4f*)(a.bs3 + ch);
                const int rw = *(const int*)(a.x + pix * planes + ch);
                float yv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float yy = xb[e] + (float)(acc[4 * g + e] + xa[e] * tot + xc4[e]) * xsc[e];
                    yv[e] = fmaxf(yy + (float)(((rw << (24 - 8 * e)) >> 24) + a.res_ax) * a.res_delta, 0.0f);
                }
                *(int*)(st + l31 * SP + 8 * g + 4 * half) = q_pack4(yv[0], yv[1], yv[2], yv[3], a.db, rd, a.zb);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            {   // the row's 32 pixels x 32 bytes leave as 16-byte stores: pixel lane >> 1, piece lane & 1
                const int xo = x0 + (lane >> 1);
                if (y < a.h && xo < a.w)
                    *(v4i*)(a.out + ((size_t)(img * (a.h + 2) + y + 1) * (a.w + 2) + xo + 1) * planes + nt * 32 + (lane & 1) * 16) =
                        *(const v4i*)(st + (lane >> 1) * SP + (lane & 1) * 16);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            (void)valid;
        }
    }
}

template <int KS1, int NT1>
int launch_bn(const BnArgs& a, hipStream_t st) {
    constexpr int WIDTH = NT1 * 128, P1 = WIDTH + 16;
    constexpr int bytes = HP * P1 + TH * 32 * P1 + 4 * 32 * 48;
    // > 64 KB of dynamic LDS needs the attribute; set on every call (a host-side driver call, no process-global "done" flag to go stale
    // on a second device; not part of a captured graph)
    if (int rc = hip_check(hipFuncSetAttribute((const void*)bottleneck_i8_kernel<KS1, NT1>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "bottleneck LDS size")) return rc;
    const int tiles = a.n * ((a.h + TH - 1) / TH) * ((a.w + TW - 1) / TW);
    bottleneck_i8_kernel<KS1, NT1><<<tiles, 256, bytes, st>>>(a);
    return hip_check(hipGetLastError(), "qv2x_bottleneck_i8 launch");
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_bottleneck_i8(const qv2x_bottleneck_desc* d, const int8_t* x, const int8_t* const* w /* conv1, conv2, conv3 */,
                                  const float* const* scale, const int32_t* const* corr, const int32_t* const* aw, const float* const* bias,
                                  int8_t* out, void* stream) {
    using namespace qv2x;
    if (!d || !x || !w || !scale || !corr || !aw || !bias || !out) return fail(QV2X_EINVAL, "qv2x_bottleneck_i8: null pointer");
    for (int i = 0; i < 3; ++i)
        if (!w[i] || !scale[i] || !corr[i] || !aw[i] || !bias[i] || ((uintptr_t)w[i] & 15) || ((uintptr_t)scale[i] & 15) || ((uintptr_t)corr[i] & 15) ||
            ((uintptr_t)aw[i] & 15) || ((uintptr_t)bias[i] & 15)) return fail(QV2X_EALIGN, "qv2x_bottleneck_i8: layer %d constants null or not 16-byte aligned", i);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0) return fail(QV2X_EINVAL, "qv2x_bottleneck_i8: bad shape");
    if (d->width != 2 * d->planes || (d->planes != 64 && d->planes != 128 && d->planes != 256) || (d->cg != 4 && d->cg != 8 && d->cg != 16))
        return fail(QV2X_EINVAL, "qv2x_bottleneck_i8: planes 64 | 128 | 256, width = 2 x planes, 4 | 8 | 16 channels per group");
    if (!(d->delta1 > 0.0f) || !(d->delta2 > 0.0f) || !(d->out_delta > 0.0f)) return fail(QV2X_EINVAL, "qv2x_bottleneck_i8: quantizer steps must be positive");
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return fail(QV2X_EALIGN, "qv2x_bottleneck_i8: 16-byte aligned maps");
    BnArgs a;
    a.x = x; a.out = out;
    a.w1 = w[0]; a.sc1 = scale[0]; a.cr1 = corr[0]; a.aw1 = aw[0]; a.bs1 = bias[0];
    a.w2 = w[1]; a.sc2 = scale[1]; a.cr2 = corr[1]; a.aw2 = aw[1]; a.bs2 = bias[1];
    a.w3 = w[2]; a.sc3 = scale[2]; a.cr3 = corr[2]; a.aw3 = aw[2]; a.bs3 = bias[2];
    a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->planes; a.width = d->width; a.cg = d->cg;
    a.d1 = d->delta1; a.z1 = d->zp1; a.d2 = d->delta2; a.z2 = d->zp2; a.db = d->out_delta; a.zb = d->out_zp;
    a.res_ax = 128 - d->in_zx; a.res_delta = d->in_delta;
    hipStream_t st = (hipStream_t)stream;
    switch (d->planes) {
        case 64: return launch_bn<2, 1>(a, st);
        case 128: return launch_bn<4, 2>(a, st);
        default: return launch_bn<8, 4>(a, st);
    }
}
