// Weights-stationary 3x3 QuantModule convolution, STRIDE 2, 64 input channels: the ZeroPad2d + stride-2 first layer of backbone level 0
// (64 -> 64 over the pillar canvas) and of level 1 (64 -> 128) -- opencood/models/sub_modules/base_bev_backbone.py:60-66 under
// quant/quant_block.py:243-303.  Same arithmetic as conv_i8.hip / conv_i8_wide.hip / conv_i8_ws.hip (exact i32 sums on
// v_mfma_i32_32x32x32_i8, then the fp32 requantizer of quant_layer.py:132-133), bit-identical results.
//
// Round 5.  These two layers ran on the im2col LDS-DMA ring of conv_i8.hip at 0.14 / 0.18 of the int8 peak (123 + 42 us per batch of 32
// frames); a first port to the weights-stationary kernel with a PLANAR halo tile (one 16-byte K piece of 64 pixels per DMA instruction) ran
// 115 + 47 us, and its ablations (profiles/r05_ws2_ablations.log) said why: with the DMA's source made contiguous the level-0 layer takes
// 74 us -- at stride 2 the 64 lanes of a planar piece touch 64 different 128-byte lines (16 bytes each), 64 requests per instruction where
// a contiguous piece makes 8 -- and with ROWS = 2 output rows per wave and item the two workgroup barriers, the window-sum pass and the
// DMA issue of a 72 KB tile were paid every two rows.  This form:
//   * the halo tile is ARRAY-OF-PIXELS: a DMA instruction moves 16 WHOLE pixels (16 x 64 contiguous bytes at stride 1 x ..., 16 half-lines
//     at stride 2 instead of 64 quarter-segments), the source of a piece is a SCALAR base plus one per-lane constant (no VALU per piece);
//     the four 16-byte K pieces of a pixel are XOR-swizzled by (slot >> 2) & 3 -- on the per-lane SOURCE address, the LDS side of an
//     LDS-DMA being lane-linear -- so that the 16 lanes of a ds_read_b128 group hit 16 distinct bank groups (as conv_i8.hip:swz does);
//   * a tile row holds the 32 even input columns, then the 32 odd ones (64 slots = 4 pieces, never straddling rows), then ONE extra slot:
//     input column 64, which only output pixel 31's third tap reads -- seventeen pixels per tile, fetched by plain loads of one wave;
//   * fragment (tap dy, dx; K half) of output row j = slot (dx even: x + dx / 2; dx odd: 32 + x) of tile row 2 j + dy: six per-lane
//     address registers (three dx, two K halves) advanced once per output row, the row of the tap an immediate;
//   * FOUR waves per workgroup (channel blocks x row groups = 2 x 2 or 4 x 1), TH = 8 output rows per item, ONE 70 KB tile per workgroup
//     and TWO workgroups per CU: no barrier joins two waves of one SIMD (conv_i8_ws.hip), an item's fixed costs are paid per four or
//     eight rows of a wave, and one workgroup's tile fetch hides behind the other's rows;
//   * the per-pixel channel sums (for the weight zero-point term) are written, not accumulated: a quad of lanes holds one pixel.
#include "common.h"

namespace qv2x {

namespace {

constexpr int TW = 32, TH = 8, IR = 2 * TH + 1;                        // output tile 8 x 32; 17 input rows
constexpr int ROWB = 64 * 64 + 64;                                     // bytes of a tile row: 64 slots + the extra one
constexpr int TILE = IR * ROWB;                                        // 70 720
constexpr int PSW = 65;                                                // channel sums per tile row (slot order)
constexpr int PSUM = (IR * PSW + 15) / 16 * 16;

struct Ws2Args {
    const int8_t* in; const int8_t* wt; const float* scale; const int* corr; const int* aw; const float* bias; int8_t* out;
    int n, hp, wp, cin_total, cin_off, cout, ho, wo, tiles_x, tiles_y;
    int out_ctotal, out_c0, relu;
    float out_delta, out_zp;
    int items;
    int tapstep[9];                    // weight step of tap t inside w_wide (qv2x_conv3x3_i8_pack_wide stores a stride-2 layer's taps plane by plane)
};

template <int V> struct IC { static constexpr int value = V; };

// NCB 32-channel blocks x RG row groups = 4 waves; ROWS output rows per wave and item (RG * ROWS = TH)
template <int NCB, int RG, int ROWS>
__global__ __launch_bounds__(256, 2) void conv3x3_i8_ws2_kernel(const Ws2Args a) {
    static_assert(NCB * RG == 4 && RG * ROWS == TH, "four waves, eight output rows");
    __shared__ __attribute__((aligned(16))) int8_t lds[TILE + PSUM * 4];
    int* psum = (int*)(lds + TILE);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, x = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave_u % NCB, rg = wave_u / NCB;
    const int npatch = a.n * a.tiles_x * a.tiles_y;
    int item = blockIdx.x;
    if (item >= npatch) return;

    struct Where { int y0, x0, img; };
    auto place = [&](int it) __attribute__((always_inline)) {
        const int txi = it % a.tiles_x, tyi = (it / a.tiles_x) % a.tiles_y, img = it / (a.tiles_x * a.tiles_y);
        return Where{tyi * TH, txi * TW, img};
    };

    // ---- tile fetch.  Piece (row k, sub s) = slots 16 s .. 16 s + 15 of tile row k: input columns 2 (16 (s & 1) + p) + (s >> 1), p = lane >> 2;
    //      lane l lands at byte 16 l of the piece and fetches K piece (l & 3) ^ ((l >> 4) & 3) of its pixel (the swizzle: (slot >> 2) & 3 =
    //      (l >> 4) & 3 whatever s).  Wave w moves pieces w, w + 4, ... of the 68.  Inline asm on purpose: see conv_i8_wide.hip.
    const int qsw = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned ldsb = (unsigned)(uintptr_t)((__attribute__((address_space(3))) int8_t*)lds);
    int xtra[2][4];                                                    // wave 3: the extra slot of tile rows lane >> 2 (and of row 16: lanes 0-3)
    auto issue_tile = [&](const Where& w) __attribute__((always_inline)) {
        // the lane's byte offset inside an input row for this wave's sub (clamped to the row: the last tile column of a ragged map)
        const int s = wave_u;
#if defined(QV2X_WS2_ABL) && QV2X_WS2_ABL == 1      // dev ablation (timing only): every lane fetches the border pixel of its row -- the
        const unsigned voff = (unsigned)(a.cin_off + qsw * 16);      // floor of an occupancy-redirected fetch (profiles/r05_ws2_ablations.log)
#else
        const unsigned voff = (unsigned)(min(2 * w.x0 + 32 * (s & 1) + (s >> 1) + 2 * (lane >> 2), a.wp - 1) * a.cin_total + a.cin_off + qsw * 16);
#endif
        const int rowb = w.img * a.hp;
#if defined(QV2X_WS2_ABL) && QV2X_WS2_ABL == 2      // dev ablation: no halo DMA after the first tile
        if (w.img + w.y0 + w.x0 != 0) return;
#endif
#pragma unroll
        for (int k = 0; k < IR; ++k) {                                  // piece (row k, sub s) -> LDS k * ROWB + s * 1024
            const int yy = min(2 * w.y0 + k, a.hp - 1);
            const int8_t* base = a.in + (size_t)(rowb + yy) * a.wp * a.cin_total;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(ldsb + k * ROWB + s * 1024), "v"(voff), "s"(base) : "memory", "m0");
        }
        if (wave_u == 3) {
            const int col = min(2 * w.x0 + 64, a.wp - 1);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = min(16 * e + (lane >> 2), IR - 1);
                const int yy = min(2 * w.y0 + k, a.hp - 1);
                const v4i v = *(const v4i*)(a.in + ((size_t)(rowb + yy) * a.wp + col) * a.cin_total + a.cin_off + (lane & 3) * 16);
                xtra[e][0] = v[0]; xtra[e][1] = v[1]; xtra[e][2] = v[2]; xtra[e][3] = v[3];
            }
        }
    };
    // the tile has landed (own pieces: vmcnt(0)): wave 3 stores the extra slots (no swizzle: one lane reads them per fragment) and their sums
    auto store_extras = [&]() __attribute__((always_inline)) {
        if (wave_u == 3) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 16 * e + (lane >> 2);
                if (k < IR) {
                    v4i v; v[0] = xtra[e][0]; v[1] = xtra[e][1]; v[2] = xtra[e][2]; v[3] = xtra[e][3];
                    *(v4i*)(lds + k * ROWB + 4096 + (lane & 3) * 16) = v;
                }
                int s = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) s = __builtin_amdgcn_sdot4(xtra[e][q], 0x01010101, s, false);
                s += __builtin_amdgcn_mov_dpp(s, 0xB1, 0xf, 0xf, false);       // quad_perm [1,0,3,2]
                s += __builtin_amdgcn_mov_dpp(s, 0x4E, 0xf, 0xf, false);       // quad_perm [2,3,0,1]
                if (k < IR && (lane & 3) == 0) psum[k * PSW + 64] = s;
            }
        }
    };
    // per-pixel channel sums of the landed tile: the wave's own 17 pieces, a quad of lanes per pixel
    auto write_psum = [&]() __attribute__((always_inline)) {
        auto some = [&](auto j0_c, auto n_c) __attribute__((always_inline)) {      // (in three batches: 17 pieces at once are 68 registers)
            constexpr int J0 = decltype(j0_c)::value, N = decltype(n_c)::value;
            v4i v[N];
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = *(const v4i*)(lds + (J0 + j) * ROWB + wave_u * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                int s = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) s = __builtin_amdgcn_sdot4(v[j][q], 0x01010101, s, false);
                s += __builtin_amdgcn_mov_dpp(s, 0xB1, 0xf, 0xf, false);
                s += __builtin_amdgcn_mov_dpp(s, 0x4E, 0xf, 0xf, false);
                if ((lane & 3) == 0) psum[(J0 + j) * PSW + wave_u * 16 + (lane >> 2)] = s;
            }
        };
        some(IC<0>{}, IC<6>{}); some(IC<6>{}, IC<6>{}); some(IC<12>{}, IC<5>{});
    };

    // ---- once per workgroup: this wave's 72 weight registers and its channels' constants ---------------------------------------------
    // w_wide layout (qv2x_conv3x3_i8_pack_wide, one chunk): [step][cout / 32][K half][lane][16 B], step = a.tapstep[tap]
    v4i wreg[9][2];
    {
        const int8_t* wp = a.wt + (size_t)cb * 2048 + lane * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) wreg[t][h] = *(const v4i*)(wp + (size_t)a.tapstep[t] * (a.cout / 32) * 2048 + h * 1024);
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
    Where cur = place(item);
    issue_tile(cur);

    // fragment addresses: slot of (x, dx), K piece 2 ks + half at its swizzled position; pixel 31's third tap reads the extra slot
    unsigned fa[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = (dx & 1) ? 32 + x : x + (dx >> 1), q = 2 * ks + half;
            const bool ext = dx == 2 && x == 31;
            fa[dx][ks] = ldsb + (unsigned)((rg * ROWS) * 2 * ROWB) + (ext ? 4096u + q * 16 : (unsigned)(slot * 64 + ((q ^ ((slot >> 2) & 3)) << 4)));
        }

    const float rd = 1.0f / a.out_delta, lowc = a.relu ? a.out_zp + 8388608.0f : 8388608.0f;
    v16i acc[2];
    int p_tot = 0;
    long long p_off = -1;                                              // byte offset of this lane's 16 output bytes; < 0: nothing to store
    int pk[4];
    const int ch_off = a.out_c0 + cb * 32 + half * 16;
    const float qlow = lowc - 8388608.0f;
    unsigned qa = 0;
    float yq[4];
    // ONE output of the pending tile per call (conv_i8_ws.hip:epi_one): register R of its accumulator
    auto epi_one = [&](auto p_c, auto r_c) __attribute__((always_inline)) {
        constexpr int P = decltype(p_c)::value, R = decltype(r_c)::value, G = R >> 2, E = R & 3;
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
            if (p_off >= 0) *(v4i*)(a.out + p_off) = ob;
        }
    };
    // One output row: 18 MFMAs into acc[P]; the pending row's epilogue (acc[1 - P]) between them.  Fragment of (tap, K half): the address
    // register of (dx, K half) + the tap's tile row as an immediate.
    auto do_row = [&](auto p_c) __attribute__((always_inline)) {
        constexpr int P = decltype(p_c)::value;
        v4i fr[4];
        auto rd_frag = [&](auto s_c) __attribute__((always_inline)) {
            constexpr int S = decltype(s_c)::value, TAP = S >> 1, KS = S & 1;
            fr[S & 3] = *(const __attribute__((address_space(3))) v4i*)(uintptr_t)(fa[TAP % 3][KS] + (TAP / 3) * ROWB);
        };
        auto step = [&](auto s_c) __attribute__((always_inline)) {
            constexpr int S = decltype(s_c)::value, TAP = S >> 1, KS = S & 1;
            if constexpr (S + 3 < 18) rd_frag(IC<S + 3>{});
            if constexpr (S == 0) acc[P] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wreg[TAP][KS], fr[S & 3], corr0, 0, 0, 0);
            else acc[P] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wreg[TAP][KS], fr[S & 3], acc[P], 0, 0, 0);
            if constexpr (S >= 1 && S <= 16) epi_one(IC<1 - P>{}, IC<S - 1>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        rd_frag(IC<0>{}); rd_frag(IC<1>{}); rd_frag(IC<2>{});
        step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{});
        step(IC<6>{}); step(IC<7>{}); step(IC<8>{}); step(IC<9>{}); step(IC<10>{}); step(IC<11>{});
        step(IC<12>{}); step(IC<13>{}); step(IC<14>{}); step(IC<15>{}); step(IC<16>{}); step(IC<17>{});
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[dx][ks] += 2 * ROWB;     // the next output row: two tile rows down
    };

    // ---- the wave's rows as one stream: even rows of the stream multiply into acc[0], odd ones into acc[1] ---------------------------------
    int j = 0, rs0 = 0;
    bool has_next = false, done = false;
    Where nxw = cur;
    acc[1] = corr0;                                                    // (the first row has no pending tile: its woven "epilogue" stores nothing)
    auto item_start = [&]() __attribute__((always_inline)) {
        const int nx = item + (int)gridDim.x;
        has_next = nx < npatch;
        if (has_next) nxw = place(nx);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // own pieces (and wave 3's extra pixels) have landed
        store_extras();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // everybody's pieces are in
        write_psum();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // the channel sums are complete
    };
    // row j of the current item has just been multiplied: it becomes the pending tile (window sum of its pixel, where it goes)
    auto after_row = [&]() __attribute__((always_inline)) {
        // input columns 2 x, 2 x + 1, 2 x + 2 = slots x, 32 + x, x + 1 (pixel 31: the extra slot 64) of tile rows 2 j', 2 j' + 1, 2 j' + 2
        const int* ps = psum + (rg * ROWS) * 2 * PSW;
        const int third = x == 31 ? 64 : x + 1;
        auto rowsum = [&](int k) { return ps[k * PSW + x] + ps[k * PSW + 32 + x] + ps[k * PSW + third]; };
        if (j == 0) rs0 = rowsum(0);
        const int mid = rowsum(2 * j + 1), rs2 = rowsum(2 * j + 2);
        p_tot = rs0 + mid + rs2; rs0 = rs2;
        {
            const int row = cur.y0 + rg * ROWS + j, xo = cur.x0 + x;
            const long long off = ((long long)(cur.img * (a.ho + 2) + row + 1) * (a.wo + 2) + xo + 1) * a.out_ctotal + ch_off;
            p_off = (row < a.ho && xo < a.wo) ? off : -1;
        }
        if (++j == ROWS) {
            j = 0;
            // every wave's reads of the tile and of the sums are done once it has passed this barrier: the tile can be refilled
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[dx][ks] -= ROWS * 2 * ROWB;
            if (!has_next) { done = true; return; }
            item += (int)gridDim.x; cur = nxw;
            issue_tile(cur);
        }
    };
    int last = 0;
    for (;;) {
        if (j == 0) item_start();
        do_row(IC<0>{});
        after_row();
        if (done) { last = 0; break; }
        if (j == 0) item_start();
        do_row(IC<1>{});
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

// called by qv2x_conv3x3_i8_wide (conv_i8_wide.hip) for the layers this form takes; the arguments are already validated there
int launch_ws2(const qv2x_conv_desc* d, const int8_t* in, const int8_t* w_wide, const float* scale, const int32_t* corr, const int32_t* aw,
               const float* bias, int8_t* out, hipStream_t st) {
    Ws2Args a{};
    a.in = in; a.wt = w_wide; a.scale = scale; a.corr = corr; a.aw = aw; a.bias = bias; a.out = out;
    a.n = d->n; a.hp = d->h + 2; a.wp = d->w + 2; a.cin_total = d->cin_total; a.cin_off = d->group_c0[0]; a.cout = d->cout;
    a.ho = (d->h + 2 - 3) / 2 + 1; a.wo = (d->w + 2 - 3) / 2 + 1;
    a.tiles_x = (a.wo + TW - 1) / TW; a.tiles_y = (a.ho + TH - 1) / TH;
    a.out_ctotal = d->out_ctotal; a.out_c0 = d->out_c0; a.relu = d->relu; a.out_delta = d->out_delta; a.out_zp = d->out_zp;
    a.items = a.n * a.tiles_x * a.tiles_y;
    // pack_wide stores a stride-2 layer's nine weight steps plane by plane -- taps 0 2 6 8 | 1 7 | 3 5 | 4 (conv_i8_wide.hip)
    const int s2step[9] = {0, 4, 1, 6, 8, 7, 2, 5, 3};
    for (int t = 0; t < 9; ++t) a.tapstep[t] = s2step[t];
    const int slots = 2 * 256;                                         // two four-wave workgroups per CU
    const dim3 grid(a.items < slots ? a.items : slots);
    if (d->cout == 64) conv3x3_i8_ws2_kernel<2, 2, 4><<<grid, 256, 0, st>>>(a);
    else conv3x3_i8_ws2_kernel<4, 1, 8><<<grid, 256, 0, st>>>(a);
    return hip_check(hipGetLastError(), "qv2x_conv3x3_i8_wide (weights-stationary stride-2 form) launch");
}

bool ws2_takes(const qv2x_conv_desc* d) {
    if (d->stride != 2 || d->ngroups != 1 || d->group_c[0] != 64 || (d->cout != 64 && d->cout != 128)) return false;
    const int ho = (d->h - 1) / 2 + 1, wo = (d->w - 1) / 2 + 1;
    return (long long)d->n * ((ho + TH - 1) / TH) * ((wo + TW - 1) / TW) >= 1024;   // two items per workgroup at least
}

}  // namespace qv2x
