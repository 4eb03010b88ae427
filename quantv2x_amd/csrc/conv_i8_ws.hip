// Weights-stationary 3x3 QuantModule convolution for layers with 64 INPUT channels (K = 576): the 64 -> 64 layers of backbone level 0
// (opencood/models/sub_modules/base_bev_backbone.py:60-75 under quant/quant_block.py:243-303).  Same arithmetic as conv_i8.hip /
// conv_i8_wide.hip (exact i32 sums on v_mfma_i32_32x32x32_i8, then the fp32 requantizer of quant_layer.py:132-133), bit-identical results.
//
// Why another form (round 4).  On the halo-patch kernel of conv_i8_wide.hip a 64 -> 64 item spends 4.8k cycles in its K loop and 7.9k in
// the requantizing epilogue (10 VALU instructions per output against 576 MACs), and the two never overlap: both waves of a SIMD multiply,
// then both requantize (profiles/r03_wide_fine_trace.log) -- 24 % MFMA-busy.  Software-pipelining that kernel needs a second accumulator
// set of 80 registers (round 3: built, 4 x 32 patches, a tie).  Here the roles of the operands are turned round:
//   * a wave owns 32 output channels and keeps ALL their weights in registers: 9 taps x 64 channels x 32 rows = 18 KiB = 72 VGPRs,
//     loaded once per workgroup -- no weight stream at all (the L2 -> register stream was 20 % of the old K loop);
//   * the pixels stream: one 32-pixel row of the patch at a time, 18 MFMAs (9 taps x 2 K halves) into ONE 16-register accumulator whose
//     C input of the first MFMA is the channel's correction term;
//   * so the accumulators of a tile are 16 registers, two sets cost 32, and the epilogue of row t - 1 is woven into the 18 MFMAs of
//     row t at four-output granularity: the matrix pipe and the VALU of a SIMD both run all the time, whatever the partner wave does;
//   * the per-channel constants (scale, bias, weight-offset) sit in registers too (48): the epilogue reads no LDS but the window sums;
//   * with the accumulators independent of the patch height, a workgroup takes 10 x 32 patches (4 waves = 2 channel blocks x 2 row
//     groups of 5 rows): 12 x 34 / (10 x 32) = 1.28 halo bytes per pixel instead of 1.49.
// The halo tile, its planar LDS layout, the LDS-DMA and the per-pixel channel sums are those of conv_i8_wide.hip.
#include "common.h"

namespace qv2x {

namespace {

constexpr int TW = 32, HWD = TW + 2;

struct WsArgs {
    const int8_t* in; const int8_t* wt; const float* scale; const int* corr; const int* aw; const float* bias; int8_t* out;
    int n, hp, wp, cin_total, cin_off, cout, ho, wo, tiles_x, tiles_y;
    int out_ctotal, out_c0, relu;
    float out_delta, out_zp;
    int items;
};

template <int V> struct IC { static constexpr int value = V; };

#ifdef QV2X_WS_FINE         // dev build only (tools/ws_fine.py): s_memtime stamps of wave 0 during the workgroup's THIRD item
__device__ long long g_ws_fine[4096 * 16];
#define WSFINE(k) do { if (nit == 2 && threadIdx.x == 0 && blockIdx.x < 4096) g_ws_fine[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSFINE(k) do { } while (0)
#endif

// NCB 32-channel blocks x RG row groups = NW waves; ROWS rows per wave: a workgroup's patch is (RG * ROWS) x 32 pixels x (NCB * 32) channels.
// <2, 2, 5>: four waves, a 10 x 32 patch, 60 KB of LDS -- TWO workgroups per CU, so the two waves of a SIMD belong to different workgroups:
// the older wave of a SIMD gets most of its VALU issue slots and runs ahead of the younger one, which an eight-wave workgroup paid for at
// both of its barriers per item (1.4k + 2.2k of 16.7k cycles, tools/ws_fine.py); here no barrier joins two waves of one SIMD.
template <int NCB, int RG, int ROWS>
__global__ __launch_bounds__(NCB * RG * 64, NCB * RG == 4 ? 2 : 1) void conv3x3_i8_ws64_kernel(const WsArgs a) {
    constexpr int NW = NCB * RG;
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    constexpr int TH = RG * ROWS, HPIX = (TH + 2) * HWD, HPAD = (HPIX + 63) / 64 * 64, PLANE = HPAD * 16, HBUF = 4 * PLANE;
    constexpr int HBLK = 4 * (HPAD / 64), LH = (HBLK + NW - 1) / NW;   // 1 KiB DMA pieces per halo tile; per wave
    static_assert(2 * PLANE + (2 * HWD + 2) * 16 + (ROWS - 1) * HWD * 16 < 65536, "fragment reads: base register + 16-bit immediate");
    __shared__ __attribute__((aligned(16))) int8_t lds[2 * HBUF + 2 * HPAD * 4];
    int8_t* hbuf = lds;
    int* psum = (int*)(lds + 2 * HBUF);                                // [set][halo pixel]: per-pixel channel sums of the item's tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int cb = wave_u % NCB, rg = wave_u / NCB;                    // this wave's channel block and row group
    const int npatch = a.n * a.tiles_x * a.tiles_y;
    int item = blockIdx.x;
    if (item >= npatch) return;

    struct Where { int y0, x0, img; };
    auto place = [&](int it) __attribute__((always_inline)) {
        const int txi = it % a.tiles_x, tyi = (it / a.tiles_x) % a.tiles_y, img = it / (a.tiles_x * a.tiles_y);
        return Where{tyi * TH, txi * TW, img};
    };
    // ---- halo DMA: piece blk = (plane = blk / (HPAD / 64), pixel block = blk % (HPAD / 64)); lane l moves the 16 bytes `plane` of halo
    //      pixel 64 * block + l (LDS side lane-linear: 1 KiB of one plane per instruction).  Inline asm on purpose: see conv_i8_wide.hip.
    // (the lane's halo pixel of piece j -- (hy, hx) inside the tile -- never changes: kept packed, one register per piece)
    int hyx[LH];
#pragma unroll
    for (int j = 0; j < LH; ++j) {
        const int blk = wave_u + NW * j;
        int hpx = (blk % (HPAD / 64)) * 64 + lane;
        hpx = hpx < HPIX ? hpx : HPIX - 1;
        const int hy = hpx / HWD;
        hyx[j] = (hy << 16) | (hpx - hy * HWD);
    }
    auto issue_halo = [&](const Where& w, int buf) __attribute__((always_inline)) {
        const unsigned ldsb = (unsigned)(uintptr_t)((__attribute__((address_space(3))) int8_t*)(hbuf + buf * HBUF));
        const int rowb = w.img * a.hp;
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int blk = wave_u + NW * j;
            if (HBLK % NW != 0 && blk >= HBLK) break;
            const int yy = min(w.y0 + (hyx[j] >> 16), a.hp - 1), xx = min(w.x0 + (hyx[j] & 0xffff), a.wp - 1);
            const unsigned src = (unsigned)(((rowb + yy) * a.wp + xx) * a.cin_total + a.cin_off + (blk / (HPAD / 64)) * 16);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(ldsb + blk * 1024), "v"(src), "s"(a.in) : "memory", "m0");
        }
    };
    auto add_psum = [&](int buf, int set) __attribute__((always_inline)) {
        const int8_t* tile = hbuf + buf * HBUF;
        v4i v[LH];
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int blk = wave_u + NW * j;
            v[j] = *(const v4i*)(tile + (blk < HBLK ? blk : 0) * 1024 + lane * 16);
        }
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int blk = wave_u + NW * j;
            if (HBLK % NW != 0 && blk >= HBLK) break;
            int s = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) s = __builtin_amdgcn_sdot4(v[j][q], 0x01010101, s, false);
            __hip_atomic_fetch_add(psum + set * HPAD + (blk % (HPAD / 64)) * 64 + lane, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    // ---- once per workgroup: this wave's 72 weight registers and its channels' constants ---------------------------------------------
    // w_wide layout (qv2x_conv3x3_i8_pack_wide, one chunk): [tap][cout / 32][K half][lane][16 B]
    v4i wreg[9][2];
    {
        const int8_t* wp = a.wt + (size_t)cb * 2048 + lane * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) wreg[t][h] = *(const v4i*)(wp + (size_t)t * (a.cout / 32) * 2048 + h * 1024);
    }
    // register r of the 32 x 32 accumulator holds channel 32 cb + 8 (r >> 2) + 4 half + (r & 3) of pixel lane & 31
    v16i corr0;
    float sc[16], bs[16];
    int awr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = cb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        corr0[r] = a.corr[co]; sc[r] = a.scale[co]; bs[r] = a.bias[co]; awr[r] = a.aw[co];
    }
    for (int t = tid; t < 2 * HPAD; t += NW * 64) psum[t] = 0;
    Where cur = place(item);
    issue_halo(cur, 0);

    const float rd = 1.0f / a.out_delta, lowc = a.relu ? a.out_zp + 8388608.0f : 8388608.0f;
    const int rlane = half * PLANE + (lane & 31) * 16;
    v16i acc[2];
    // the tile whose epilogue is pending (its sums sit in acc[1 - P] while tile P multiplies)
    int p_tot = 0;
    long long p_off = -1;                                              // byte offset of this lane's 16 output bytes; < 0: nothing to store
    int pk[4];
    (void)lowc;
    const int lane_x = lane & 31, ch_off = a.out_c0 + cb * 32 + half * 16;

    // ONE output of the pending tile per call -- register R of its accumulator: channel 8 (R >> 2) + 4 half + (R & 3) of its pixel.  Every
    // fourth call closes a group of four (the sandwich of common.h), the sixteenth swaps half-waves and stores.  One output per MFMA gap:
    // the 18 MFMAs of a row are a dependent chain (~50 cycles from one to the next), and ten VALU instructions fit each gap; four outputs
    // behind every fourth MFMA left 14 gaps empty and made 4 too long (SQ counters: 28 % MFMA-busy, 42 % VALU-busy, 30 % issue stalls).
    const float qlow = lowc - 8388608.0f;
    unsigned qa = 0;
    float yq[4];
    auto epi_one = [&](auto p_c, auto r_c) __attribute__((always_inline)) {
        constexpr int P = decltype(p_c)::value, R = decltype(r_c)::value, G = R >> 2, E = R & 3;
#if defined(QV2X_WS_ABL) && QV2X_WS_ABL == 1     // dev ablation (timing only): no epilogue at all, the sums stay live
        if (R == 0) asm volatile("" :: "v"(acc[P]));
        return;
#endif
        const int T = __mul24(awr[R], p_tot) + acc[P][R];              // (the channel's correction term went in as the first MFMA's C operand)
        yq[E] = bs[R] + (float)T * sc[R];
        if (E == 0) qa = 0;
        q_add(yq[E], E, rd, a.out_zp, qlow, qa);                          // (common.h: fma + v_cvt_pk_u8_f32, two instructions per output)
        if (E == 3) pk[G] = (int)(qa ^ 0x80808080u);
        if (R == 15) {                                                 // half-wave exchange -> 16 contiguous channels per lane, one 16-byte store
            const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
            v4i ob;
            ob[0] = s02[0]; ob[1] = s02[1]; ob[2] = s13[0]; ob[3] = s13[1];
#if defined(QV2X_WS_ABL) && QV2X_WS_ABL == 5     // dev ablation: no stores
            asm volatile("" :: "v"(ob));
            if (false)
#else
            if (p_off >= 0)
#endif
                *(v4i*)(a.out + p_off) = ob;
        }
    };
    // One row of the patch: 18 MFMAs into acc[P]; the pending tile's epilogue (acc[1 - P]) in four pieces between them.
    // Fragment of (tap, K half): 16 bytes of halo pixel (lane & 31) + 34 (row + dy) + dx in plane 2 KS + half -- base register + immediate.
    auto do_row = [&](auto p_c, const int8_t* hb) __attribute__((always_inline)) {
        constexpr int P = decltype(p_c)::value;
        v4i fr[4];
        auto rd_frag = [&](auto s_c) __attribute__((always_inline)) {
            constexpr int S = decltype(s_c)::value, TAP = S >> 1, KS = S & 1;
            fr[S & 3] = *(const v4i*)(hb + KS * 2 * PLANE + (HWD * (TAP / 3) + TAP % 3) * 16);
        };
        auto step = [&](auto s_c) __attribute__((always_inline)) {
            constexpr int S = decltype(s_c)::value, TAP = S >> 1, KS = S & 1;
#if !defined(QV2X_WS_ABL) || QV2X_WS_ABL != 3     // (dev ablation 3: one fragment read per row instead of 18)
            if constexpr (S + 3 < 18) rd_frag(IC<S + 3>{});
#endif
#if defined(QV2X_WS_ABL) && QV2X_WS_ABL == 2     // dev ablation: no MFMAs, the operands stay live
            if constexpr (S == 0) acc[P] = corr0;
            acc[P][S & 15] += wreg[TAP][KS][0] ^ fr[S & 3][1];
            if constexpr (true) {} else
#endif
            if constexpr (S == 0) acc[P] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wreg[TAP][KS], fr[S & 3], corr0, 0, 0, 0);
            else acc[P] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wreg[TAP][KS], fr[S & 3], acc[P], 0, 0, 0);
            if constexpr (S >= 1 && S <= 16) epi_one(IC<1 - P>{}, IC<S - 1>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        rd_frag(IC<0>{}); rd_frag(IC<1>{}); rd_frag(IC<2>{});
        step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{});
        step(IC<6>{}); step(IC<7>{}); step(IC<8>{}); step(IC<9>{}); step(IC<10>{}); step(IC<11>{});
        step(IC<12>{}); step(IC<13>{}); step(IC<14>{}); step(IC<15>{}); step(IC<16>{}); step(IC<17>{});
    };

    // ---- the wave's rows as one stream: row -> (item, j); even rows of the stream multiply into acc[0], odd ones into acc[1] --------------
    int buf = 0, set = 0, j = 0;
    int nit = 0;                                                       // items finished (dev stamps)
    (void)nit;
    bool has_next = false, done = false;
    Where nxw = cur;
    const int8_t* hb0 = hbuf;
    int rs0 = 0, rs1 = 0;
    acc[1] = corr0;                                                    // (the first row has no pending tile: its woven "epilogue" stores nothing)
    auto item_start = [&]() __attribute__((always_inline)) {
        const int nx = item + (int)gridDim.x;
        has_next = nx < npatch;
        WSFINE(0);
        // the item's tile has landed (own pieces: vmcnt; everybody's: the barrier); every wave's reads of the other buffer are done
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        WSFINE(1);
        __builtin_amdgcn_s_barrier();
        WSFINE(2);
        add_psum(buf, set);
#if defined(QV2X_WS_ABL) && QV2X_WS_ABL == 4     // dev ablation: no halo DMA after the first tile
        if (has_next) nxw = place(nx);
#else
        if (has_next) { nxw = place(nx); issue_halo(nxw, buf ^ 1); }
#endif
        int* nz = psum + (set ^ 1) * HPAD;                             // the other set: read for the last time before the barrier above
        for (int t = tid; t < HPAD; t += NW * 64) nz[t] = 0;
        hb0 = hbuf + buf * HBUF + rlane + (rg * ROWS) * HWD * 16;
        WSFINE(3);
    };
    // row j of the current item has just been multiplied: it becomes the pending tile (window sum of its pixel, where it goes)
    auto after_row = [&]() __attribute__((always_inline)) {
        const int* ps = psum + set * HPAD + (rg * ROWS) * HWD + (lane & 31);
        auto rowsum = [&](int k) { return ps[k * HWD] + ps[k * HWD + 1] + ps[k * HWD + 2]; };
        if (j == 0) {
            WSFINE(4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // every wave's LDS atomics of this item are in
            WSFINE(5);
            rs0 = rowsum(0); rs1 = rowsum(1);
        }
        if (j >= 1) WSFINE(5 + j);                                     // rows 1 .. ROWS - 1 done: stamps 6 ..
        const int rs2 = rowsum(j + 2);
        p_tot = rs0 + rs1 + rs2; rs0 = rs1; rs1 = rs2;
        {
            const int row = cur.y0 + rg * ROWS + j, xo = cur.x0 + lane_x;
            const long long off = ((long long)(cur.img * (a.ho + 2) + row + 1) * (a.wo + 2) + xo + 1) * a.out_ctotal + ch_off;
            p_off = (row < a.ho && xo < a.wo) ? off : -1;
        }
        if (++j == ROWS) {
            j = 0; ++nit;
            if (!has_next) { done = true; return; }
            item += (int)gridDim.x; cur = nxw; buf ^= 1; set ^= 1;
        }
    };
    int last = 0;
    for (;;) {
        if (j == 0) item_start();
        do_row(IC<0>{}, hb0 + j * HWD * 16);
        after_row();
        if (done) { last = 0; break; }
        if (j == 0) item_start();
        do_row(IC<1>{}, hb0 + j * HWD * 16);
        after_row();
        if (done) { last = 1; break; }
    }
    // the last row's epilogue: nothing left to weave it into
    auto epi_all = [&](auto p_c) __attribute__((always_inline)) {
        epi_one(p_c, IC<0>{}); epi_one(p_c, IC<1>{}); epi_one(p_c, IC<2>{}); epi_one(p_c, IC<3>{});
        epi_one(p_c, IC<4>{}); epi_one(p_c, IC<5>{}); epi_one(p_c, IC<6>{}); epi_one(p_c, IC<7>{});
        epi_one(p_c, IC<8>{}); epi_one(p_c, IC<9>{}); epi_one(p_c, IC<10>{}); epi_one(p_c, IC<11>{});
        epi_one(p_c, IC<12>{}); epi_one(p_c, IC<13>{}); epi_one(p_c, IC<14>{}); epi_one(p_c, IC<15>{});
    };
    if (last == 0) epi_all(IC<0>{}); else epi_all(IC<1>{});
}

}  // namespace

#ifdef QV2X_WS_FINE
extern "C" __attribute__((visibility("default"))) int qv2x_debug_ws_fine(long long* host_out, int nblocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ws_fine), (size_t)nblocks * 16 * sizeof(long long));
}
extern "C" __attribute__((visibility("default"))) int qv2x_debug_ws_fine_clear() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_ws_fine)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(g_ws_fine));
}
#endif

// called by qv2x_conv3x3_i8_wide (conv_i8_wide.hip) for the layers this form takes; the arguments are already validated there
int launch_ws64(const qv2x_conv_desc* d, const int8_t* in, const int8_t* w_wide, const float* scale, const int32_t* corr, const int32_t* aw,
                const float* bias, int8_t* out, hipStream_t st) {
    WsArgs a{};
    a.in = in; a.wt = w_wide; a.scale = scale; a.corr = corr; a.aw = aw; a.bias = bias; a.out = out;
    a.n = d->n; a.hp = d->h + 2; a.wp = d->w + 2; a.cin_total = d->cin_total; a.cin_off = d->group_c0[0]; a.cout = d->cout;
    a.ho = d->h; a.wo = d->w;
    constexpr int TH = 10;
    a.tiles_x = (a.wo + TW - 1) / TW; a.tiles_y = (a.ho + TH - 1) / TH;
    a.out_ctotal = d->out_ctotal; a.out_c0 = d->out_c0; a.relu = d->relu; a.out_delta = d->out_delta; a.out_zp = d->out_zp;
    a.items = a.n * a.tiles_x * a.tiles_y;
    const int slots = 2 * 256;                                         // two four-wave workgroups per CU
    const dim3 grid(a.items < slots ? a.items : slots);
    conv3x3_i8_ws64_kernel<2, 2, 5><<<grid, 256, 0, st>>>(a);
    return hip_check(hipGetLastError(), "qv2x_conv3x3_i8_wide (weights-stationary form) launch");
}

bool ws64_takes(const qv2x_conv_desc* d) {
    if (d->stride != 1 || d->ngroups != 1 || d->group_c[0] != 64 || d->cout != 64) return false;
    const long long patches = (long long)d->n * ((d->h + 9) / 10) * ((d->w + TW - 1) / TW);
    return patches >= 1024;                                            // two items per workgroup at least: the pipeline's fill is one row of 5
}

}  // namespace qv2x
