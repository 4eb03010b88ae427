// a3: a CHAIN of 64-channel 3x3 QuantModule convolutions in one launch -- backbone level 0 of the V2X-Real / OPV2V
// models (opencood/models/sub_modules/base_bev_backbone.py:96-119 under quant_block.py:243-303: ZeroPad2d + Conv3x3
// stride 2 + 3 x Conv3x3, each with folded BN, ReLU and an output quantizer).
//
// Why: one such layer is 1.3 GMAC = ~0.5 us of int8 MFMA, but a launch of it costs ~8.6 us (launch -> first fetch ->
// nine K steps -> store drain, DESIGN.md "what was measured"); four of them 34 us.  Here a workgroup owns a TH x TW
// patch of the LAST layer's output and computes the whole chain for it: layer d is evaluated on the patch grown by
// (D-1-d) pixels on every side (recomputed halo), the intermediate maps never leave LDS (int8 codes, the same
// [pixel][64 B] rows with XOR-swizzled 16-byte chunks the other conv kernels use), and positions outside the image are
// written as the code of 0.0 -- exactly the zero padding the next layer of the reference sees.
//
//   * weights live in REGISTERS: a wave owns 32 output channels, i.e. 18 B fragments of v_mfma_i32_32x32x32_i8
//     (9 taps x 2 k-halves) = 72 VGPRs, loaded once per layer with 1-KiB coalesced loads from a pre-packed tensor.
//     No weight staging, no barrier inside a layer.
//   * GEMM rows are the pixels of the layer's output region linearised with the INPUT pitch (m = r * P + c), so tap
//     (dy, dx) is the pure shift m + dy * P + dx: the 32 lanes of a half-wave read 32 consecutive LDS pixels for every
//     tap (conflict-free ds_read_b128, as in conv_i8_wide.hip); the two garbage columns per row are dropped in the epilogue.
//   * the stride-2 first layer reads its input space-to-depth: four parity planes (row parity x column parity), so
//     input (2r + dy, 2c + dx) is again a pure shift inside plane (dy & 1, dx & 1).
//   * arithmetic is the single-group case of conv_i8.hip (exact i32 sums, zero-point correction from v_dot4 window sums,
//     y = bias + float(T) * scale, ReLU, q_code): bit-identical to oracle/qv2x_oracle.c:orc_conv3x3 layer by layer.
#include "common.h"

#ifndef QV2X_CHAIN_DBG
#define QV2X_CHAIN_DBG 0      // dev builds (tools/bench_chain_abl.py): 1 no epilogue math, 2 no K loop, 3 no input DMA, 4 weights loaded once
#endif

namespace qv2x {
namespace {

constexpr int CH = 64;                       // channels of every tensor in the chain
constexpr int MAXD = 4;

struct ChainArgs {
    const int8_t* in; const int8_t* w; const float* scale; const int* corr; const int* aw; const float* bias; int8_t* out;
    int n, h, w_, hin_p, win_p, tiles_x, tiles_y, relu;
    float out_delta[MAXD], out_zp[MAXD];
};

template <int V> struct IC { static constexpr int value = V; };

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// geometry of layer d of a D-layer chain over a TH x TW patch
template <int TH, int TW, int D, int S0>
struct Geo {
    static constexpr int rows(int d) { return TH + 2 * (D - 1 - d); }
    static constexpr int cols(int d) { return TW + 2 * (D - 1 - d); }
    static constexpr int pitch(int d) { return (d == 0 && S0 == 2) ? cols(0) + 1 : cols(d) + 2; }     // input pitch of layer d
    static constexpr int M(int d) { return (rows(d) - 1) * pitch(d) + cols(d); }
    static constexpr int MT(int d) { return cdiv(M(d), 32); }
    // farthest input pixel a (possibly garbage) lane of layer d touches, relative to its plane / region base
    static constexpr int reach(int d) { return MT(d) * 32 + ((d == 0 && S0 == 2) ? pitch(0) + 1 : 2 * pitch(d) + 2); }
    // layer-0 input: stride 2 -> four parity planes, each padded to 16 pixels; stride 1 -> one plane
    static constexpr int plane_px(int p) {
        return S0 == 2 ? cdiv((rows(0) + ((p >> 1) == 0 ? 1 : 0)) * pitch(0), 16) * 16 : cdiv((rows(0) + 2) * pitch(0), 16) * 16;
    }
    static constexpr int plane_off(int p) { return p == 0 ? 0 : plane_off(p - 1) + plane_px(p - 1); }
    static constexpr int NPLANE = S0 == 2 ? 4 : 1;
    static constexpr int in_px = plane_off(NPLANE - 1) + plane_px(NPLANE - 1);
    // area 0 holds the input planes, later the outputs of the odd layers; area 1 the outputs of the even layers
    static constexpr int area_px(int a) {
        int v = a == 0 ? cmax(in_px, plane_off(NPLANE - 1) + reach(0)) : 0;
        for (int d = 0; d < D; ++d) {
            if (((d & 1) ? 0 : 1) == a) v = cmax(v, rows(d) * cols(d));          // written by layer d
            if (d >= 1 && (((d - 1) & 1) ? 0 : 1) == a) v = cmax(v, reach(d));     // read by layer d
        }
        return cdiv(v, 16) * 16;
    }
};

template <int TH, int TW, int D, int S0>
__global__ __launch_bounds__(512, 1) void conv3x3_i8_chain64_kernel(const ChainArgs a) {
    using G = Geo<TH, TW, D, S0>;
    constexpr int A0 = G::area_px(0) * CH, A1 = G::area_px(1) * CH;
    __shared__ __attribute__((aligned(1024))) int8_t lds[A0 + A1];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int nt = wave & 1, mg = wave >> 1;                           // this wave: output channels [32 nt, +32), M tiles mg, mg + 4, ...
    const int txi = blockIdx.x % a.tiles_x, tyi = (blockIdx.x / a.tiles_x) % a.tiles_y, img = blockIdx.x / (a.tiles_x * a.tiles_y);
    const int y0 = tyi * TH, x0 = txi * TW;

    // ---- layer-0 input region -> LDS (area 0) by LDS-DMA: 1 KiB = 16 pixels per instruction ------------------------------
    {
        const int Yo = y0 - (D - 1), Xo = x0 - (D - 1);                // image coordinates of output (0, 0) of layer 0's region
        constexpr int NBLK = G::in_px / 16;
#pragma unroll
        for (int j = 0; j < cdiv(NBLK, 8); ++j) {
            const int blk = wave + 8 * j;
            if (blk < NBLK && QV2X_CHAIN_DBG != 3) {
                const int q = blk * 16 + (lane >> 2), slot = lane & 3;
                int p = 0;
                if (S0 == 2) p = (q >= G::plane_off(1)) + (q >= G::plane_off(2)) + (q >= G::plane_off(3));
                const int hp = q - (p == 0 ? 0 : p == 1 ? G::plane_off(1) : p == 2 ? G::plane_off(2) : G::plane_off(3));
                const int prow = hp / G::pitch(0), pcol = hp - prow * G::pitch(0);
                int yc, xc;
                if (S0 == 2) { yc = 2 * (Yo + prow) + (p >> 1); xc = 2 * (Xo + pcol) + (p & 1); }
                else         { yc = Yo + prow;                  xc = Xo + pcol; }
                yc = min(max(yc, 0), a.hin_p - 1);                      // outside the padded input: any bytes (they only feed
                xc = min(max(xc, 0), a.win_p - 1);                      // outputs that lie outside the image)
                const int sub = slot ^ ((hp >> 2) & 3);
                const int8_t* src = a.in + ((size_t)(img * a.hin_p + yc) * a.win_p + xc) * CH + sub * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lds + blk * 1024), 16, 0, 0);
            }
        }
    }

    // The WEIGHTS are the A operand of the MFMA and the pixels the B operand (out^T = W x^T): lane l then holds ONE pixel
    // (column l & 31) and 16 channels of it, co(r) = 32 nt + 8 (r >> 2) + 4 (l >> 5) + (r & 3) -- four runs of four
    // consecutive bytes of the [pixel][channel] row.  Everything that depends on the pixel (region coordinates, inside /
    // valid tests, LDS address, the window sum) is computed once per lane and tile, not once per element, and a tile is
    // written with four ds_write_b32 per lane (the first version of this kernel, pixels as rows and one ds_write_b8 per
    // element, spent ~85 VALU instructions per output element and ran 50 us for the four layers: VALU-issue-bound).
    v4i breg[18];
    int c_aw[16], c_cr[16];
    float c_sc[16], c_bs[16];
    auto load_weights = [&](int d) __attribute__((always_inline)) {
        const int8_t* wl = a.w + ((size_t)(d * 2 + nt) * 18 * 64 + lane) * 16;
#pragma unroll
        for (int t = 0; t < 18; ++t) breg[t] = *(const v4i*)(wl + (size_t)t * 1024);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = d * CH + 32 * nt + 8 * g + 4 * half;
            const v4i xa = *(const v4i*)(a.aw + c0), xc = *(const v4i*)(a.corr + c0);
            const v4f xs_ = *(const v4f*)(a.scale + c0), xb = *(const v4f*)(a.bias + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { c_aw[4 * g + e] = xa[e]; c_cr[4 * g + e] = xc[e]; c_sc[4 * g + e] = xs_[e]; c_bs[4 * g + e] = xb[e]; }
        }
    };
    load_weights(0);

    auto run_layer = [&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        constexpr int P = G::pitch(d), M = G::M(d), MT = G::MT(d), CO = G::cols(d);
        constexpr bool PLANES = (d == 0 && S0 == 2);
        const int8_t* src = lds + ((d & 1) ? A0 : 0);                  // layer 0 reads area 0, layer 1 area 1, ...
        int8_t* dst = lds + ((d & 1) ? 0 : A0);
        const float od = a.out_delta[d], oz = a.out_zp[d], rd = 1.0f / od;
        const int padb = ((int)oz - 128) & 0xff, padw = padb * 0x01010101;
        const float lowc = a.relu ? oz + 8388608.0f : 8388608.0f;        // the ReLU lives in the clamp (q_pack4)
        const int Yo = y0 - (D - 1 - d), Xo = x0 - (D - 1 - d);
        // LDS byte offset of tap t for this lane in M tile 0; tile i adds 2048 i (the swizzle has a period of 16 pixels)
        int off[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            const int shift = PLANES ? (dy >> 1) * P + (dx >> 1) : dy * P + dx;
            const int base = PLANES ? G::plane_off((dy & 1) * 2 + (dx & 1)) * CH : 0;
            const int hp = l31 + shift;
            off[tap] = base + hp * CH + ((half ^ ((hp >> 2) & 3)) << 4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // weights + constants of this layer (and the input DMA)
        if (d == 0) __builtin_amdgcn_s_barrier();                       // every wave's part of the input is in LDS

        for (int i = mg; i < MT; i += 4) {
            const int8_t* st = src + i * 2048;
            v16i acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0;
            int xs = 0;
#pragma unroll
            for (int tap = 0; tap < (QV2X_CHAIN_DBG == 2 ? 0 : 9); ++tap) {
                v4i f0, f1;
                if (QV2X_CHAIN_DBG == 6) { f0 = breg[(tap * 2 + 3) % 18]; f1 = breg[(tap * 2 + 5) % 18]; }      // no LDS reads
                else { f0 = *(const v4i*)(st + off[tap]); f1 = *(const v4i*)(st + (off[tap] ^ 32)); }
                if (QV2X_CHAIN_DBG != 5) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) xs = __builtin_amdgcn_sdot4(f0[q], 0x01010101, xs, false);
#pragma unroll
                    for (int q = 0; q < 4; ++q) xs = __builtin_amdgcn_sdot4(f1[q], 0x01010101, xs, false);
                } else xs += f0[0] + f1[3];
                if (QV2X_CHAIN_DBG != 7) {
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(breg[tap * 2], f0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(breg[tap * 2 + 1], f1, acc, 0, 0, 0);
                } else { acc[tap] += f0[1] ^ breg[tap * 2][0]; acc[tap + 1] += f1[2] ^ breg[tap * 2 + 1][1]; }
            }
            const int tot = xs + __shfl_xor(xs, 32);                    // window sum of this lane's pixel (both k-halves)
            const int m = i * 32 + l31;
            const int ro = m / P, cc = m - ro * P;
            const bool valid = m < M && cc < CO;                        // not a garbage column / row of the linearised region
            const bool inside = (unsigned)(Yo + ro) < (unsigned)a.h && (unsigned)(Xo + cc) < (unsigned)a.w_;
            const int op = ro * CO + cc, swz = (op >> 2) & 3;
            int8_t* o = dst + op * CH + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const int T = acc[r] + __mul24(c_aw[r], tot) + c_cr[r];
                    y[e] = c_bs[r] + (float)T * c_sc[r];
                }
                int pk = QV2X_CHAIN_DBG == 1 ? acc[4 * g] + acc[4 * g + 1] + acc[4 * g + 2] + acc[4 * g + 3] + tot : q_pack4(y[0], y[1], y[2], y[3], od, rd, oz, lowc);
                pk = inside ? pk : padw;                                // outside the image: the zero padding of the next layer
                if (valid) *(int*)(o + (((2 * nt + (g >> 1)) ^ swz) << 4) + 8 * (g & 1)) = pk;
            }
        }
        if (d + 1 < D && QV2X_CHAIN_DBG != 4) load_weights(d + 1);                             // in flight across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // region d complete; its input area is free
    };
    run_layer(IC<0>{});
    if constexpr (D > 1) run_layer(IC<1>{});
    if constexpr (D > 2) run_layer(IC<2>{});
    if constexpr (D > 3) run_layer(IC<3>{});

    // ---- copy the TH x TW patch of the last layer to the padded NHWC output, 16 bytes per thread -------------------------
    {
        const int8_t* reg = lds + (((D - 1) & 1) ? 0 : A0);
        constexpr int UNITS = TH * TW * 4;
#pragma unroll
        for (int k = 0; k < cdiv(UNITS, 512); ++k) {
            const int u = tid + k * 512;
            if (u < UNITS) {
                const int op = u >> 2, slot = u & 3, sub = slot ^ ((op >> 2) & 3);
                const int ro = op / TW, cc = op - ro * TW;
                const int Y = y0 + ro, X = x0 + cc;
                if (Y < a.h && X < a.w_)
                    *(v4i*)(a.out + ((size_t)(img * (a.h + 2) + Y + 1) * (a.w_ + 2) + X + 1) * CH + sub * 16) = *(const v4i*)(reg + op * CH + slot * 16);
            }
        }
    }
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_conv3x3_i8_chain64(const qv2x_chain_desc* d, const int8_t* in, const int8_t* w_chain, const float* scale,
                                       const int32_t* corr, const int32_t* aw, const float* bias, int8_t* out, void* stream) {
    using namespace qv2x;
    if (!d || !in || !w_chain || !scale || !corr || !aw || !bias || !out) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_chain64: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || (d->stride0 != 1 && d->stride0 != 2))
        return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_chain64: bad shape n=%d h=%d w=%d stride0=%d", d->n, d->h, d->w, d->stride0);
    if (d->depth < 1 || d->depth > MAXD) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_chain64: 1..%d layers", MAXD);
    if (((uintptr_t)in & 15) || ((uintptr_t)w_chain & 15) || ((uintptr_t)out & 15)) return fail(QV2X_EALIGN, "qv2x_conv3x3_i8_chain64: in / w / out must be 16-byte aligned");
    ChainArgs a;
    a.in = in; a.w = w_chain; a.scale = scale; a.corr = corr; a.aw = aw; a.bias = bias; a.out = out;
    a.n = d->n; a.h = d->h; a.w_ = d->w; a.relu = d->relu;
    // layer 0 is a padding-1 convolution of the in_h x in_w map: (in + 2 - 3) / stride + 1 outputs per axis
    if ((d->in_h + 2 - 3) / d->stride0 + 1 != d->h || (d->in_w + 2 - 3) / d->stride0 + 1 != d->w)
        return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_chain64: a %d x %d input does not give %d x %d at stride %d", d->in_h, d->in_w, d->h, d->w, d->stride0);
    a.hin_p = d->in_h + 2; a.win_p = d->in_w + 2;                     // padded size of the input tensor
    for (int l = 0; l < MAXD; ++l) {
        a.out_delta[l] = l < d->depth ? d->out_delta[l] : 1.0f;
        a.out_zp[l] = l < d->depth ? d->out_zp[l] : 0.0f;
        if (l < d->depth && !(d->out_delta[l] > 0.0f)) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_chain64: out_delta must be positive");
    }
    constexpr int TH = 5, TW = 32;
    a.tiles_x = (a.w_ + TW - 1) / TW; a.tiles_y = (a.h + TH - 1) / TH;
    const dim3 grid(a.n * a.tiles_x * a.tiles_y);
    hipStream_t st = (hipStream_t)stream;
#define QV2X_CHAIN(D, S) conv3x3_i8_chain64_kernel<TH, TW, D, S><<<grid, 512, 0, st>>>(a)
    const int key = d->depth * 10 + d->stride0;
    switch (key) {
        case 11: QV2X_CHAIN(1, 1); break;
        case 12: QV2X_CHAIN(1, 2); break;
        case 21: QV2X_CHAIN(2, 1); break;
        case 22: QV2X_CHAIN(2, 2); break;
        case 31: QV2X_CHAIN(3, 1); break;
        case 32: QV2X_CHAIN(3, 2); break;
        case 41: QV2X_CHAIN(4, 1); break;
        default: QV2X_CHAIN(4, 2); break;
    }
#undef QV2X_CHAIN
    return hip_check(hipGetLastError(), "qv2x_conv3x3_i8_chain64 launch");
}
