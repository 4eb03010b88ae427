// Wide 3x3 QuantModule convolution (cout % 256 == 0, stride 1): the two shrinker layers of the V2X-Real model
// (opencood/quant/quant_block.py:552-572 over DoubleConv, downsample_conv.py:17-31), i.e. 52 of the 77 GMAC of the
// integer convolution stack.  Same arithmetic as conv_i8.hip (exact i32 sums on the i8 MFMA, per-group fp32 fold),
// different data movement -- what bounds this layer on gfx950 is the L2 -> LDS fetch rate and the LDS fragment reads,
// not the MFMA pipe (tools/probes/mfma_i8_wave_tile_probe.hip):
//   * one workgroup = a 5 x 32 patch of output pixels (160 GEMM rows) x 256 output channels; 8 waves (two per SIMD,
//     which hides the per-step barrier better than one wave per SIMD with software-pipelined fragment reads did),
//     each wave a 160 x 32 register tile = 5 accumulators of v_mfma_i32_32x32x32_i8;
//   * the input is staged as a 7 x 34 pixel HALO tile per 64-channel chunk and the nine taps read it at shifted
//     pixel offsets: 15.2 KB fetched per nine K-steps instead of 9 x 10 KB;
//   * the weights are pre-tiled (qv2x_conv3x3_i8_pack_wide) so that the [wtile][64] tile of one K-step is contiguous, and go
//     straight from L2 to registers: a wave only ever reads ITS OWN 32 rows of a tile, so the LDS ring of round 1 (16 KB written
//     and 16 KB read per step, a workgroup barrier per step) bought nothing but asynchrony.  Two 16-byte loads per lane per
//     step, a ring of three steps; LDS holds the two halo tiles only and the waves meet once per 64-channel chunk (nine steps).
#include "common.h"

#include <cstdlib>

namespace qv2x {

// conv_i8_ws.hip: the weights-stationary form for layers with 64 input channels (same w_wide layout, same results)
bool ws64_takes(const qv2x_conv_desc* d);
// conv_i8_ws2.hip: the weights-stationary STRIDE-2 form for 64 input channels (cout 64 | 128)
bool ws2_takes(const qv2x_conv_desc* d);
int launch_ws2(const qv2x_conv_desc* d, const int8_t* in, const int8_t* w_wide, const float* scale, const int32_t* corr, const int32_t* aw,
               const float* bias, int8_t* out, hipStream_t st);
int launch_ws64(const qv2x_conv_desc* d, const int8_t* in, const int8_t* w_wide, const float* scale, const int32_t* corr, const int32_t* aw,
                const float* bias, int8_t* out, hipStream_t st);

namespace {

constexpr int TH = 5, TW = 32, HWD = TW + 2, HPIX = (TH + 2) * HWD;    // 238 halo pixels
constexpr int HPAD = 256;                                              // halo pixels per plane, padded to four 64-pixel DMA blocks
constexpr int PLANE = HPAD * 16;                                       // one 16-byte K piece of every halo pixel
constexpr int HBLK = 4 * (HPAD / 64);                                  // 16 DMA instructions of 1 KiB per halo tile
constexpr int HBUF = 4 * PLANE;                                        // 16 KiB per halo tile
constexpr int BM = TH * TW, WTILE = 256;                               // pixels per workgroup; rows of one pre-tiled weight tile
constexpr int MT = 5;
constexpr int MAX_CHUNKS = 24;

struct WideArgs {
    const int8_t* in; const int8_t* wt; const float* scale; const int* corr; const int* aw; const float* bias; int8_t* out;
    int n, hp, wp, cin_total, cout, ho, wo, tiles_x, tiles_y;
    int out_ctotal, out_c0, relu;
    float out_delta, out_zp;
    int nchunks, ngroups;
    int wtile;                        // rows of one pre-tiled weight tile: min(cout, 256)
    int cend[QV2X_MAX_GROUPS];        // first chunk index past group g
    int coff[MAX_CHUNKS];             // channel byte offset of each 64-channel chunk inside a pixel
    int items;                        // (patch, channel block) pairs, patches padded to a multiple of 8
    int nsteps;                       // weight steps of an item's K loop: 9 per 64 input channels
    int stride2;                      // 1: the stride-2 form (four parity-plane chunks per 64 input channels)
};

template <int V> struct IC { static constexpr int value = V; };

#if defined(QV2X_WABL) && QV2X_WABL == 1      // dev ablation: no MFMAs, the operands stay live
__device__ __forceinline__ v16i wabl_nomfma(v4i a, v4i b, v16i c) { c[0] += a[0] + b[0]; c[1] += a[3] ^ b[3]; return c; }
#define WMFMA(a, b, c) wabl_nomfma(a, b, c)
#else
#define WMFMA(a, b, c) __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0)
#endif

#ifdef QV2X_WIDE_TRACE      // dev build only (tools/wide_trace.py): s_memtime stamps of wave 0 of every workgroup
__device__ long long g_wide_trace[8192 * 8];
#define WTRACE(k) do { if (tid == 0 && blockIdx.x < 8192) g_wide_trace[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WTRACE(k) do { } while (0)
#endif
#ifdef QV2X_WIDE_FINE       // dev build only (tools/wide_fine.py): finer s_memtime stamps of wave 0 during the workgroup's THIRD item
__device__ long long g_wide_fine[4096 * 16];
#define WFINE(k) do { if (nit == 2 && threadIdx.x == 0 && blockIdx.x < 4096) g_wide_fine[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WFINE(k) do { } while (0)
#endif

// NW waves, each a 160 x 32 register tile (NT = 1), BN = NW * 32 output channels per workgroup:
//   <8, 1, 256>  the shrinker: one workgroup per 5 x 32 patch (two waves per SIMD);
//   <4, 1, 128>  the backbone's 128-channel layers, and 256-channel layers whose patches alone would leave CUs idle (25 x 88 x 8 frames =
//                120 patches): the two channel halves of a patch get block ids 8 apart (same XCD: one L2 serves the halo they both fetch);
//   <2, 1, 64>   the backbone's 64-channel layers.
//
// Round 2, second form.  The first form ran every K step as {request weights, 10 fragment reads, 10 MFMAs} behind one scheduling
// fence; the two waves of a SIMD fall into lockstep (both read, then both multiply), so a step cost the SUM of its MFMA time and of its
// address / LDS / issue time: 34 % MFMA-busy.  Now
//   * the halo tile is PLANAR in LDS -- [16-byte K piece][halo pixel][16 B] -- so a fragment read is one base register plus an
//     immediate (the tap shift, the M tile and the K half are compile-time constants): no swizzle arithmetic in the loop, and a
//     16-lane service group of ds_read_b128 covers 256 contiguous bytes (conflict-free for every tap);
//   * the K loop is software-pipelined by HALF steps inside each wave: the five fragments of K half 1 are requested before the
//     five MFMAs of K half 0 and the next step's K half 0 before this step's K half 1 -- in the same 40 fragment registers;
//   * the window sums (sum of the input codes under each output's 3 x 3 x C window, needed for the zero-point terms) no longer
//     come from dot4 instructions on the fragments inside the loop (with a branch: "whose turn is it"): each landed halo tile adds
//     its per-pixel channel sums into a 256-entry LDS table once (two 1 KiB pieces per wave) and an output's window sum is nine
//     table entries at the group fold.  The loop body is one basic block; the fold needs no barrier.
//
// (Round 4 tried the epilogue woven at HALF-output granularity behind every MFMA of the next item -- bit-exact, a tie again:
// profiles/r04_woven_conv_experiment.log.)
// (Round 3 also built a PING-PONG form -- groups of waves half an item apart, held there by workgroup barriers -- and a SOFTWARE-
// PIPELINED one -- the epilogue of item i woven into the K loop of item i + 1 over two accumulator sets.  Both were bit-exact and
// neither was faster (profiles/r03_wide_forms.log, DESIGN_HISTORY.md); round 4 took them out of the product library: they are in the
// history at commit 0601783.)
// Round 3: S2 = the ZeroPad2d + stride-2 first convolution of a backbone level (base_bev_backbone.py:60-66) on the same kernel.  The input
// is read as four PARITY PLANES (row parity, column parity of the padded input): tap (dy, dx) of output pixel (y, x) is pixel
// (y + dy / 2, x + dx / 2) of plane (dy & 1, dx & 1), so a plane's halo tile is an ordinary 6 x 33 tile and a "chunk" becomes
// (64 input channels, plane) with 4 / 2 / 2 / 1 taps -- the same nine K steps per 64 channels, the same fragment reads (base + immediate),
// four halo tiles instead of one.  Only the DMA's source addresses know about the stride (the global side of an LDS-DMA is per lane).
template <bool MULTI, int NW, int NT, int BN, int MTP = 5, bool S2 = false>
// (second bound = waves per SIMD the registers must allow: 8-wave workgroups and the three-group 4-wave form sit alone on a CU)
#define QV2X_WIDE_BOUNDS(MULTI, NW) (((NW) >= 8 || ((MULTI) && (NW) == 4)) ? 1 : 2)
__global__ __launch_bounds__(NW * 64, QV2X_WIDE_BOUNDS(MULTI, NW)) void conv3x3_i8_wide_kernel(const WideArgs a) {
    static_assert(NW * NT * 32 == BN && WTILE % BN == 0, "wave layout");
    static_assert(!S2 || !MULTI, "the stride-2 form: one input group");
    constexpr int NPL = S2 ? 4 : 1;                                    // window-sum tables per set: one per parity plane
    constexpr int MT = MTP, TH = MTP, HPIX = (TH + 2) * HWD;           // (shadow the file-scope values: patch height = M tiles per wave)
    static_assert(HPIX <= HPAD, "halo tile");
    static_assert(HBLK % NW == 0, "every wave moves the same number of halo pieces");
    constexpr int LH = HBLK / NW;                                      // halo DMA instructions per wave
    constexpr int NF = MULTI ? 16 : 1;
    static_assert(NT == 1, "the epilogue below is written for one 32-channel tile per wave");
    constexpr int NG = MULTI ? QV2X_MAX_GROUPS : 1;
    __shared__ __attribute__((aligned(16))) int8_t lds[2 * HBUF + 3 * NG * NPL * HPAD * 4 + NG * BN * 16];
    // window-sum tables [set][group][halo pixel]: item k of a workgroup uses set k % 3.  Its first tile is summed into the set by the
    // previous item's last tap 8, and item k's start clears set (k + 1) % 3 -- last read two items ago, i.e. before barriers every wave has
    // passed -- so an item starts without a barrier or a wait of its own.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int8_t* hbuf = lds;
    int* psum = (int*)(lds + 2 * HBUF);
    v4i* ctab = (v4i*)(psum + 3 * NG * NPL * HPAD);                    // [group][BN] {aw, corr_g, scale_g (bits), bias (bits)} per channel
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // PERSISTENT workgroups: item id = ((patch / 8) * nblk + cb) * 8 + patch % 8 -> (patch, channel block); a workgroup takes the items
    // blockIdx.x, + gridDim.x, ... -- the grid is a multiple of 8 * nblk, so its channel block (weights, constants) and its XCD never change.
    // While an item's epilogue runs, the first two halo tiles and the first two weight steps of the next item are already in flight.
    const int nblk = a.cout / BN, npatch = a.n * a.tiles_x * a.tiles_y;
    const int vstride = (int)gridDim.x;
    int item = (int)blockIdx.x;
    auto patch_of = [&](int it) { return (it / (8 * nblk)) * 8 + (it & 7); };
    auto valid = [&](int it) { return it < a.items && patch_of(it) < npatch; };
    if (!valid(item)) return;
    const int cb = (item >> 3) % nblk, n0 = cb * BN;
    const int total = a.nsteps;
    WTRACE(0);
#ifdef QV2X_WIDE_TRACE
    { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); if (tid == 0 && blockIdx.x < 8192) { g_wide_trace[blockIdx.x * 8 + 5] = hw; g_wide_trace[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime(); } }
#endif

    // ---- DMA: piece blk = (plane = blk >> 2, pixel block = blk & 3); lane l of the instruction moves the 16 bytes `plane` of halo
    //      pixel 64 (blk & 3) + l (the LDS side of global_load_lds is lane-linear: 1 KiB of ONE plane per instruction)
    struct Where { int y0, x0, img; };
    auto place = [&](int it) __attribute__((always_inline)) {
        const int patch = patch_of(it);
        const int txi = patch % a.tiles_x, tyi = (patch / a.tiles_x) % a.tiles_y, img = patch / (a.tiles_x * a.tiles_y);
        return Where{tyi * TH, txi * TW, img};
    };
    // byte offset (from a.in: the tensor is < 4 GiB) of this lane's 16 bytes of DMA piece j of an item's halo tile; recomputed at every
    // request (once per nine K steps) rather than kept: the three-group kernel has no registers to spare
    auto src_of = [&](const Where& w, int j, int plane) __attribute__((always_inline)) {
        const int blk = wave + NW * j;
        int hpx = (blk & 3) * 64 + lane;
        hpx = hpx < HPIX ? hpx : HPIX - 1;
        const int hy = hpx / HWD, hx = hpx - hy * HWD;
        const int yy = S2 ? min(2 * (w.y0 + hy) + (plane >> 1), a.hp - 1) : min(w.y0 + hy, a.hp - 1);
        const int xx = S2 ? min(2 * (w.x0 + hx) + (plane & 1), a.wp - 1) : min(w.x0 + hx, a.wp - 1);
        return (unsigned)(((w.img * a.hp + yy) * a.wp + xx) * a.cin_total + (blk >> 2) * 16);
    };
    Where cur = place(item), nxw = cur;
    bool has_next = false;
    int pb = 0;                                                        // halo buffer of the current item's chunk 0

    v16i acc[MT][NT];
    float facc[MT][NT][NF];
    // The WEIGHTS are the A operand of the MFMA (out^T = W x^T): lane l holds pixel (l & 31) of every M tile and the 16 channels
    //   cl(r) = 32 wave + 8 (r >> 2) + 4 (l >> 5) + (r & 3)
    // of this wave's tile -- the constants of a channel come as one ds_read_b128 from `ctab`, and four consecutive channels pack
    // into one dword of the output row (see conv_i8.hip).
    const int half = lane >> 5;

    // halo tile of chunk c of the current item (c < nchunks) or of chunk c - nchunks of the next one, into buffer (pb + c) & 1
    auto issue_halo = [&](int c) __attribute__((always_inline)) {
        const int k = c - a.nchunks;
        if (k >= 0 && (!has_next || k >= a.nchunks)) return;
        const int off = a.coff[k < 0 ? c : k];
        int8_t* buf = hbuf + ((pb + c) & 1) * HBUF;
        // Issued as inline asm ON PURPOSE: with a builtin LDS-DMA pending, hipcc turns the next vmcnt wait into vmcnt(0) (it treats the
        // counter as out of order once DMA and plain loads are mixed), which drains the weight ring and the DMA itself right after
        // issue.  Unseen by the compiler, its own counts for the weight loads are merely two too strict for one step after a tile
        // request; the waits that cover the DMAs are the explicit ones (item start, tap 8).
        const unsigned ldsb = (unsigned)(uintptr_t)((__attribute__((address_space(3))) int8_t*)buf) + wave_u * 1024;
#pragma unroll
        for (int j = 0; j < LH; ++j)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(ldsb + NW * j * 1024), "v"(src_of(k < 0 ? cur : nxw, j, (k < 0 ? c : k) & 3) + (unsigned)off), "s"(a.in) : "memory", "m0");
    };
    // per-pixel channel sums of a landed halo tile -> psum[slot] (four planes of a pixel arrive in four pieces: LDS atomics)
    int pset = 0;                                                      // this item's window-sum set
    auto add_psum = [&](int chunk, int slot) __attribute__((always_inline)) {   // slot = set * NG + group
        if (S2) slot = slot * NPL + ((chunk >= a.nchunks ? chunk - a.nchunks : chunk) & 3);     // one table per parity plane
        const int8_t* buf = hbuf + ((pb + chunk) & 1) * HBUF;
        // every piece is read before the first atomic is issued: the compiler cannot move a read above an LDS atomic it may alias, and one
        // read -> wait -> dot -> atomic round trip per piece was 1.3k cycles per item in the two-wave form (eight pieces per wave)
        v4i v[LH];
#pragma unroll
        for (int j = 0; j < LH; ++j) v[j] = *(const v4i*)(buf + (wave + NW * j) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int blk = wave + NW * j;
            int s = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) s = __builtin_amdgcn_sdot4(v[j][q], 0x01010101, s, false);
            __hip_atomic_fetch_add(psum + slot * HPAD + (blk & 3) * 64 + lane, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    // this lane's 2 x 16 bytes of the weight tile of step `st`: row = its output channel (wave, lane & 31), piece ks * 2 + half of the row's 64 bytes
    const int wstep = a.wtile * 64;                                    // bytes of one step's weight tile
    // (the tile is stored in fragment order -- [32-row block][K half][lane][16 B] -- so each load instruction reads 1 KiB contiguous)
    v4i wr[3][2];
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.wt + (size_t)(n0 / a.wtile) * total * wstep + ((n0 % a.wtile) / 32 + wave_u) * 2048), 0, total * wstep, 0x00020000);
    auto load_w = [&](auto slot_c, int st) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value;
#if defined(QV2X_WABL) && QV2X_WABL == 3
        if (st > 1) return;
#endif
        // Round 4: buffer loads -- the workgroup's weight slice as a resource, a SCALAR step offset, the lane's constant 16-byte offset: no
        // 64-bit per-lane address arithmetic in the loop (2 % on the 256-channel layers and the shrinker, bit-identical)
        const int so = (st < total ? st : st - total) * wstep;       // past the end: steps 0, 1 again -- the next item's
        wr[SLOT][0] = (v4i)__builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16, so, 0);
        wr[SLOT][1] = (v4i)__builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + 1024, so, 0);
    };

    // fragment of (M tile i, tap, K half ks): 16 bytes of halo pixel (lane & 31) + 34 (i + dy) + dx in plane ks * 2 + half:
    // lane part in one register, the rest an immediate
    // FA1 (the three-group form: 80 fp32 partial results per lane live beside the 80 accumulators): ONE fragment set, each half step's
    // reads issued after the MFMAs of the half before -- the SIMD's other wave multiplies meanwhile (measured on the one-group forms in
    // round 3: no difference in time); with two sets that kernel needed 275 registers and spilled 28 bytes per lane.
#ifdef QV2X_WIDE_NO_FA1
    constexpr bool FA1 = false;
#else
    constexpr bool FA1 = MULTI;
#endif
    v4i fa[FA1 ? 1 : 2][MT];
    const int rlane = half * PLANE + (lane & 31) * 16;
    // (TAP = 3 oy + ox names the tile offset (oy, ox) of the step's pixels: the tap itself at stride 1, the plane-local offset at stride 2)
    auto read_half = [&](auto ks_c, auto tap_c, int chunk) __attribute__((always_inline)) {
        constexpr int KS = decltype(ks_c)::value, TAP = decltype(tap_c)::value;
#if defined(QV2X_WABL) && QV2X_WABL == 2
        if (chunk > 0 || TAP > 0) return;
#endif
        const int8_t* hb = hbuf + ((pb + chunk) & 1) * HBUF + rlane;
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[FA1 ? 0 : KS][i] = *(const v4i*)(hb + KS * 2 * PLANE + (HWD * (i + TAP / 3) + TAP % 3) * 16);
    };

    int g = 0;
    int nit = 0;                                                       // items this group has finished (dev traces)
    (void)nit;
    auto window_sums = [&](int (&totv)[MT]) __attribute__((always_inline)) {
        // nine psum entries per output pixel (rows i .. i + 2 of the halo, columns x .. x + 2)
        if (S2) {                                                      // plane (0,0): four taps, (0,1): two rows, (1,0): two columns, (1,1): one
            const int* p0 = psum + (pset * NPL) * HPAD + (lane & 31);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int o = i * HWD;
                totv[i] = (p0[o] + p0[o + 1] + p0[o + HWD] + p0[o + HWD + 1]) + (p0[HPAD + o] + p0[HPAD + o + HWD])
                          + (p0[2 * HPAD + o] + p0[2 * HPAD + o + 1]) + p0[3 * HPAD + o];
            }
            return;
        }
        int rowsum[MT + 2];
        const int* ps = psum + (pset * NG + g) * HPAD + (lane & 31);
#pragma unroll
        for (int k = 0; k < MT + 2; ++k) rowsum[k] = ps[k * HWD] + ps[k * HWD + 1] + ps[k * HWD + 2];
#pragma unroll
        for (int i = 0; i < MT; ++i) totv[i] = rowsum[i] + rowsum[i + 1] + rowsum[i + 2];
    };
    auto fold_group = [&]() __attribute__((always_inline)) {
        int totv[MT];
        window_sums(totv);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const v4i c = ctab[g * BN + wave * 32 + 8 * (r >> 2) + 4 * half + (r & 3)];
            const int sci = c[2];                                      // (bit_cast straight from the vector element reads element 0)
            const float sc = __int_as_float(sci);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int T = acc[i][0][r] + __mul24(c[0], totv[i]) + c[1];
                facc[i][0][r] = facc[i][0][r] + (float)T * sc;
                acc[i][0][r] = 0;
            }
        }
    };

    // One K step = (chunk, tap), in two halves.  On entry the K-half-0 fragments of the step are in flight or landed.
    // Q: the step's index inside its nine-step period (weight ring slot Q % 3); OFF = 3 oy + ox: tile offset of its pixels; LAST: last step
    // of its chunk (the next chunk's tile must have landed: wait, barrier, window sums, refill); NOFF: offset of the chunk's next step;
    // VM2: the chunk is ONE step long (see below).
    auto gstep = [&](auto q_c, auto off_c, auto last_c, auto noff_c, auto vm2_c, int chunk, int step) __attribute__((always_inline)) {
        constexpr int Q = decltype(q_c)::value;
        constexpr bool LAST = decltype(last_c)::value != 0;
        load_w(IC<(Q + 2) % 3>{}, step + 2);                           // slot of step - 1, which is done
        if (!FA1) read_half(IC<1>{}, off_c, chunk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][0] = WMFMA(wr[Q % 3][0], fa[0][i], acc[i][0]);
        __builtin_amdgcn_sched_barrier(0);
        if (FA1) {                                                     // K half 1 into the same registers, multiplied before anything else is read
            read_half(IC<1>{}, off_c, chunk);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][0] = WMFMA(wr[Q % 3][1], fa[0][i], acc[i][0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!LAST) {
            read_half(IC<0>{}, noff_c, chunk);
        } else {
            // the next chunk's halo tile was requested before all but the youngest weight loads (four of them: steps + 1 and + 2; when this
            // chunk is a single step, the pair for step + 1 is OLDER than the request, so only two are younger); every wave's reads of
            // THIS chunk's tile are done once it has passed the wait below, so after the barrier the tile's buffer can be refilled
            WFINE(2);
            if (decltype(vm2_c)::value) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            WFINE(3);
            __builtin_amdgcn_s_barrier();
            WFINE(4);
            const int nxt = chunk + 1;                                 // (past the last chunk: the next item's first tile, into ITS set)
            add_psum(nxt, nxt < a.nchunks ? pset * NG + (MULTI ? (nxt >= a.cend[g] ? g + 1 : g) : 0) : (pset == 2 ? 0 : pset + 1) * NG);
#if !defined(QV2X_WABL) || QV2X_WABL != 5                             // (dev ablation 5: no halo DMA inside the K loop)
            issue_halo(chunk + 2);                                     // (after the LDS atomics above: an LDS write after a pending
                                                                       //  LDS-DMA makes the compiler drain vmcnt)
#endif
            WFINE(5);
            read_half(IC<0>{}, IC<0>{}, nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!FA1) {
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][0] = WMFMA(wr[Q % 3][1], fa[1][i], acc[i][0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // One K step = (chunk, tap) of a stride-1 layer.  On entry the K-half-0 fragments of the step are in flight or landed.
    auto one_step = [&](auto tap_c, int chunk) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tap_c)::value;
        gstep(tap_c, tap_c, IC<TAP == 8>{}, IC<(TAP + 1) % 9>{}, IC<0>{}, chunk, chunk * 9 + TAP);
    };
    // The nine steps of 64 input channels at stride 2: planes (0,0) | (0,1) | (1,0) | (1,1) = chunks 4 cc .. 4 cc + 3 with 4 | 2 | 2 | 1 steps
    // (taps 0 2 6 8 | 1 7 | 3 5 | 4: the order qv2x_conv3x3_i8_pack_wide stores the weight steps in)
    auto s2_period = [&](int cc) __attribute__((always_inline)) {
        const int c0 = 4 * cc, s0 = 9 * cc;
        gstep(IC<0>{}, IC<0>{}, IC<0>{}, IC<1>{}, IC<0>{}, c0, s0);
        gstep(IC<1>{}, IC<1>{}, IC<0>{}, IC<3>{}, IC<0>{}, c0, s0 + 1);
        gstep(IC<2>{}, IC<3>{}, IC<0>{}, IC<4>{}, IC<0>{}, c0, s0 + 2);
        gstep(IC<3>{}, IC<4>{}, IC<1>{}, IC<0>{}, IC<0>{}, c0, s0 + 3);
        gstep(IC<4>{}, IC<0>{}, IC<0>{}, IC<3>{}, IC<0>{}, c0 + 1, s0 + 4);
        gstep(IC<5>{}, IC<3>{}, IC<1>{}, IC<0>{}, IC<0>{}, c0 + 1, s0 + 5);
        gstep(IC<6>{}, IC<0>{}, IC<0>{}, IC<1>{}, IC<0>{}, c0 + 2, s0 + 6);
        gstep(IC<7>{}, IC<1>{}, IC<1>{}, IC<0>{}, IC<0>{}, c0 + 2, s0 + 7);
        gstep(IC<8>{}, IC<0>{}, IC<1>{}, IC<0>{}, IC<1>{}, c0 + 3, s0 + 8);
    };

    // ---- once per workgroup: the constants of its channel block, the first item's first tiles and weight steps ---------------
    {
        const int nx = item + vstride;
        has_next = valid(nx);
        if (has_next) nxw = place(nx);
    }
    issue_halo(0);
    load_w(IC<0>{}, 0);
    load_w(IC<1>{}, 1);
    if (tid < BN) {
        const int co = n0 + tid;
        const int awv = a.aw[co];
        const float bs = a.bias[co];
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) {
            if (gg < a.ngroups) {
                v4i c;
                c[0] = awv; c[1] = a.corr[gg * a.cout + co];
                c[2] = __float_as_int(a.scale[gg * a.cout + co]); c[3] = __float_as_int(bs);
                ctab[gg * BN + tid] = c;
            }
        }
    }

    for (int t = tid; t < 3 * NG * NPL * HPAD; t += NW * 64) psum[t] = 0;
    int corr0[16];                                                     // single-group layers: accumulator r starts at its channel's corr (one add per output less)
#pragma unroll
    for (int r = 0; r < 16; ++r) corr0[r] = MULTI ? 0 : a.corr[n0 + wave * 32 + 8 * (r >> 2) + 4 * half + (r & 3)];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // the second tile only now: when a launch is one round of workgroups they all fetch their prologue at once (~11 B / cycle / CU), and the
    // first K steps need tile 0 alone -- tile 1 has nine steps to land like every later one
    issue_halo(1);
    __builtin_amdgcn_s_barrier();
    add_psum(0, 0);

    for (;;) {
        WFINE(0);
        // ---- item start: no barrier, no wait -- tile 0 landed and was summed at the previous item's last tap 8 (or just above) ------------
        {
            int* nz = psum + (pset == 2 ? 0 : pset + 1) * NG * NPL * HPAD;   // the next item's set
            for (int t = tid; t < NG * NPL * HPAD; t += NW * 64) nz[t] = 0;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = MULTI ? 0 : corr0[r];     // the item's sums start AT the channel's correction term
        if (MULTI) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const v4f b = *(const v4f*)(a.bias + n0 + wave * 32 + 8 * g4 + 4 * half);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) facc[i][0][4 * g4 + e] = b[e];
            }
        }
        g = 0;
        read_half(IC<0>{}, IC<0>{}, 0);
        WTRACE(1);
        WFINE(1);

        if (S2) {
            for (int cc = 0; cc < a.nchunks / 4; ++cc) s2_period(cc);
        } else
        for (int chunk = 0; chunk < a.nchunks; ++chunk) {
            one_step(IC<0>{}, chunk); one_step(IC<1>{}, chunk); one_step(IC<2>{}, chunk);
            one_step(IC<3>{}, chunk); one_step(IC<4>{}, chunk); one_step(IC<5>{}, chunk);
            one_step(IC<6>{}, chunk); one_step(IC<7>{}, chunk); one_step(IC<8>{}, chunk);
            if (MULTI && chunk + 1 == a.cend[g]) { fold_group(); ++g; }
        }
        WTRACE(2);
        WFINE(6);
#if defined(QV2X_WABL) && QV2X_WABL == 4     // dev ablation: the K loop alone (no window sums, no epilogue, no stores)
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" :: "v"(acc[i][0]));
        if (!has_next) break;
        goto next_item;
#endif
        int totv[MT];
        if (!MULTI) window_sums(totv);
        WTRACE(3);
        WFINE(7);

        // ---- epilogue: a channel quad's constants are read once and applied to all MT tiles (MT independent chains in flight), the
        //      packed bytes stay in registers and leave as MT 16-byte stores per lane -------------------------------------------
        const float rd = 1.0f / a.out_delta, lowc = a.relu ? a.out_zp + 8388608.0f : 8388608.0f;   // the ReLU lives in the clamp (q_pack4)
        int pk[MT][4];                                                 // [tile][channel quad g4]: channels 8 g4 + 4 half + (0..3) of pixel lane & 31
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            v4i c[4];
            if (!MULTI) {
#pragma unroll
                for (int e = 0; e < 4; ++e) c[e] = ctab[wave * 32 + 8 * g4 + 4 * half + e];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    if (MULTI) {
                        y[e] = facc[i][0][r];
                    } else {
                        const int T = __mul24(c[e][0], totv[i]) + acc[i][0][r];                  // (corr is in the accumulator since the item's start)
                        const int sci = c[e][2], bsi = c[e][3];
                        y[e] = __int_as_float(bsi) + (float)T * __int_as_float(sci);
                    }
                }
                pk[i][g4] = q_pack4(y[0], y[1], y[2], y[3], a.out_delta, rd, a.out_zp, lowc);
                // A tile leaves as soon as its last quad is packed.  (Five stores back to back behind the whole epilogue showed as 2.9k cycles
                // of "store issue" per item in the phase stamps; spreading them over the last quad pass changes no layer's time -- the
                // epilogue is bound by the SIMD's VALU issue, the stores only took the blame.)  No LDS staging: the half-wave exchange
                // v_permlane32_swap turns the lane's four channel quads (0-3 | 8-11 | 16-19 | 24-27 in the lower half-wave, +4 in the upper)
                // into 16 contiguous channels -- lower half 0-15, upper half 16-31 of the same pixel.
                if (g4 == 3) {
                    const auto s02 = __builtin_amdgcn_permlane32_swap(pk[i][0], pk[i][2], false, false);
                    const auto s13 = __builtin_amdgcn_permlane32_swap(pk[i][1], pk[i][3], false, false);
                    v4i ob;
                    ob[0] = s02[0]; ob[1] = s02[1]; ob[2] = s13[0]; ob[3] = s13[1];
                    const int yo = cur.y0 + i, xo = cur.x0 + (lane & 31);
                    if (yo < a.ho && xo < a.wo)
                        *(v4i*)(a.out + ((size_t)(cur.img * (a.ho + 2) + yo + 1) * (a.wo + 2) + xo + 1) * a.out_ctotal + a.out_c0 + n0 + wave * 32 + half * 16) = ob;
                }
            }
        }
        WFINE(8);
        WTRACE(4);
        WFINE(9);
        WFINE(10);
        ++nit;
        if (!has_next) break;
#if defined(QV2X_WABL) && QV2X_WABL == 4
    next_item:
#endif
        // ---- rotate to the next item: its tiles 0 (and 1) and weight steps 0, 1 are in flight since the last taps of the K loop ----------
        item += vstride;
        cur = nxw;
        pb = (pb + a.nchunks) & 1;
        pset = pset == 2 ? 0 : pset + 1;
        {
            const int nx = item + vstride;
            has_next = valid(nx);
            if (has_next) nxw = place(nx);
        }
        if (a.nchunks == 1) issue_halo(1);                             // (a one-chunk layer's K loop only requested the next item's tile 0)
    }
#ifdef QV2X_WIDE_TRACE
    if (tid == 0 && blockIdx.x < 8192) g_wide_trace[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
}

// [Cout][G][3][3][C_g] -> [Cout/wtile][chunk = (g, cc)][tap][wtile/32][K half][lane][16], wtile = min(Cout, 256)
__global__ void pack_wide_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ wt, int cout, int ktot, int nchunks,
                                 WideArgs a) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte unit each
    const long long units = (long long)cout * ktot / 16;
    if (u >= units) return;
    // destination unit u = ((tile * (wtile / 32) + row block) * 2 + K half) * 64 + lane: the 16 bytes lane `lane` feeds to the MFMA as
    // its A operand (row = lane & 31 of the block, K piece = 2 * half + (lane >> 5))
    const int ln = u & 63, ksh = (int)((u >> 6) & 1);
    const int rb = (int)((u >> 7) % (a.wtile / 32));
    const int c16 = ksh * 2 + (ln >> 5);
    const int row = rb * 32 + (ln & 31);
    const long long tile = u / (a.wtile * 4);
    const int step = tile % a.nsteps;
    const int nb = tile / a.nsteps;
    // stride 1: chunk = 64 channels, nine taps each.  Stride 2 (one group): the nine steps of 64 channels run plane by plane --
    // taps 0 2 6 8 | 1 7 | 3 5 | 4 (conv3x3_i8_wide_kernel<.., S2>::s2_period)
    const int s2seq[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};
    const int chunk = step / 9, tap = a.stride2 ? s2seq[step % 9] : step % 9;
    // group of this chunk and its K offset in the row-major layout
    int g = 0, k0 = 0, cfirst = 0;
    const int per = a.stride2 ? 4 : 1;                                 // (stride 2: cend counts four plane chunks per 64 channels)
    while (chunk >= a.cend[g] / per) { const int gc = (a.cend[g] / per - cfirst) * 64; k0 += 9 * gc; cfirst = a.cend[g] / per; ++g; }
    const int gc = (a.cend[g] / per - cfirst) * 64;
    const int k = k0 + tap * gc + (chunk - cfirst) * 64 + c16 * 16;
    *(v4i*)(wt + u * 16) = *(const v4i*)(w + (size_t)(nb * a.wtile + row) * ktot + k);
}

int fill_args(const qv2x_conv_desc* d, WideArgs& a) {
    if (d->n <= 0 || d->h <= 0 || d->w <= 0) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_wide: bad shape n=%d h=%d w=%d", d->n, d->h, d->w);
    if ((d->stride != 1 && d->stride != 2) || (d->cout != 64 && d->cout != 128 && d->cout % 256))
        return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_wide: stride 1 | 2 and cout 64 | 128 | a multiple of 256");
    if (d->stride == 2 && d->ngroups != 1) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_wide: the stride-2 form takes one input group");
    if (d->cout < 256 && d->ngroups != 1) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_wide: the 64 / 128-channel form takes one input group");
    a.wtile = d->cout < 256 ? d->cout : 256;
    if (d->ngroups < 1 || d->ngroups > QV2X_MAX_GROUPS) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_wide: 1..%d input groups", QV2X_MAX_GROUPS);
    if (d->cin_total % 16 || d->out_ctotal % 16 || d->out_c0 % 16 || d->out_ctotal < d->out_c0 + d->cout)
        return fail(QV2X_EALIGN, "qv2x_conv3x3_i8_wide: cin_total, out_ctotal, out_c0 %% 16; out channel window");
    if (!(d->out_delta > 0.0f)) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_wide: out_delta must be positive");
    a.n = d->n; a.hp = d->h + 2; a.wp = d->w + 2; a.cin_total = d->cin_total; a.cout = d->cout;
    a.stride2 = d->stride == 2;
    a.ho = (d->h + 2 - 3) / d->stride + 1; a.wo = (d->w + 2 - 3) / d->stride + 1;     // (stride 1: h x w)
    a.tiles_x = (a.wo + TW - 1) / TW; a.tiles_y = (a.ho + TH - 1) / TH;
    a.out_ctotal = d->out_ctotal; a.out_c0 = d->out_c0; a.relu = d->relu; a.out_delta = d->out_delta; a.out_zp = d->out_zp;
    a.ngroups = d->ngroups;
    int n = 0;
    for (int g = 0; g < QV2X_MAX_GROUPS; ++g) {
        if (g < d->ngroups) {
            const int c0 = d->group_c0[g], c = d->group_c[g];
            if (c <= 0 || c % 64 || c0 % 16 || c0 + c > d->cin_total)
                return fail(QV2X_EALIGN, "qv2x_conv3x3_i8_wide: group %d (c0=%d, c=%d): channels must come in multiples of 64", g, c0, c);
            for (int cc = 0; cc < c / 64; ++cc) {
                if (n >= MAX_CHUNKS) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_wide: more than %d 64-channel chunks", MAX_CHUNKS);
                a.coff[n++] = c0 + cc * 64;
            }
            a.cend[g] = n;
        } else {
            a.cend[g] = 1 << 30;
        }
    }
    a.nchunks = n;
    a.nsteps = 9 * n;
    if (a.stride2) {                                                   // chunk = (64 channels, parity plane): four per 64 channels
        if (4 * n > MAX_CHUNKS) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_wide: stride 2 takes at most %d input channels", MAX_CHUNKS / 4 * 64);
        for (int c = 4 * n - 1; c >= 0; --c) a.coff[c] = a.coff[c / 4];
        a.nchunks = 4 * n;
        a.cend[0] = 4 * n;
    }
    return QV2X_OK;
}

}  // namespace

}  // namespace qv2x

// output channels per workgroup: 256 (8 waves) while that fills the chip, else 128 (4 waves), 64 for the 64-channel layers
static long long wide_patches(const qv2x_conv_desc* d) {
    const int ho = (d->h + 2 - 3) / d->stride + 1, wo = (d->w + 2 - 3) / d->stride + 1;      // output map (stride 1: h x w)
    return (long long)d->n * ((ho + qv2x::TH - 1) / qv2x::TH) * ((wo + qv2x::TW - 1) / qv2x::TW);
}

static int wide_bn(const qv2x_conv_desc* d) {
    const long long patches = wide_patches(d);
    if (d->cout % 256 == 0) return patches * (d->cout / 256) >= 192 ? 256 : 128;
    return d->cout;
}

#ifdef QV2X_WIDE_FINE
extern "C" __attribute__((visibility("default"))) int qv2x_debug_wide_fine(long long* host_out, int nblocks) {
    using namespace qv2x;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wide_fine), (size_t)nblocks * 16 * sizeof(long long));
}
extern "C" __attribute__((visibility("default"))) int qv2x_debug_wide_fine_clear() {
    using namespace qv2x;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wide_fine)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(g_wide_fine));
}
#endif

#ifdef QV2X_WIDE_TRACE
extern "C" __attribute__((visibility("default"))) int qv2x_debug_wide_trace(long long* host_out, int nblocks) {
    using namespace qv2x;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wide_trace), (size_t)nblocks * 8 * sizeof(long long));
}
extern "C" __attribute__((visibility("default"))) int qv2x_debug_wide_trace_clear() {
    using namespace qv2x;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wide_trace)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(g_wide_trace));
}
#endif

extern "C" int qv2x_conv3x3_i8_wide_ok(const qv2x_conv_desc* d) {
    if (!d || (d->stride != 1 && d->stride != 2) || (d->cout != 64 && d->cout != 128 && d->cout % 256) || d->ngroups < 1 || d->ngroups > QV2X_MAX_GROUPS) return 0;
    if ((d->cout < 256 || d->stride == 2) && d->ngroups != 1) return 0;
    int chunks = 0;
    for (int g = 0; g < d->ngroups; ++g) {
        if (d->group_c[g] <= 0 || d->group_c[g] % 64 || d->group_c0[g] % 16) return 0;
        chunks += d->group_c[g] / 64;
    }
    // enough workgroups (a 5 x 32 patch x 64 / 128 / 256 output channels each) to fill the chip: below that the 64 x 64-tile kernel
    // of qv2x_conv3x3_i8 spreads the layer over more CUs
    const long long patches = wide_patches(d);
    const long long wgs = patches * (d->cout / wide_bn(d));
    // Stride 2 (measured, batch of 32 / 8): four tiles, barriers and window-sum passes per nine K steps instead of one -- 256 output
    // channels amortise them (34 vs 39 us, 13.8 vs 17.6 us), 128 do not (46 vs 44 us) and 64 lose (149 vs 129 us on the 192-pixel im2col tiles)
    if (qv2x::ws2_takes(d)) return 1;                                  // stride 2, 64 input channels: the weights-stationary form (conv_i8_ws2.hip)
    if (d->stride == 2 && d->cout % 256) return 0;
    return chunks * (d->stride == 2 ? 4 : 1) <= qv2x::MAX_CHUNKS && patches * qv2x::TH * qv2x::TW >= 16384 && wgs >= (d->cout == 64 ? 1024 : 192);
}

extern "C" int qv2x_conv3x3_i8_pack_wide(const qv2x_conv_desc* d, const int8_t* w, int8_t* w_wide, void* stream) {
    using namespace qv2x;
    if (!d || !w || !w_wide) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_pack_wide: null pointer");
    if (((uintptr_t)w & 15) || ((uintptr_t)w_wide & 15)) return fail(QV2X_EALIGN, "qv2x_conv3x3_i8_pack_wide: 16-byte alignment");
    WideArgs a{};
    if (int rc = fill_args(d, a)) return rc;
    const int ktot = a.nsteps * 64;                                    // bytes of one output channel's weights
    const long long units = (long long)d->cout * ktot / 16;
    pack_wide_kernel<<<(unsigned)((units + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, w_wide, d->cout, ktot, a.nchunks, a);
    return hip_check(hipGetLastError(), "qv2x_conv3x3_i8_pack_wide launch");
}

extern "C" int qv2x_conv3x3_i8_wide(const qv2x_conv_desc* d, const int8_t* in, const int8_t* w_wide, const float* scale,
                                    const int32_t* corr, const int32_t* aw, const float* bias, int8_t* out, void* stream) {
    using namespace qv2x;
    if (!d || !in || !w_wide || !scale || !corr || !aw || !bias || !out) return fail(QV2X_EINVAL, "qv2x_conv3x3_i8_wide: null pointer");
    if (((uintptr_t)in & 15) || ((uintptr_t)w_wide & 15) || ((uintptr_t)out & 15)) return fail(QV2X_EALIGN, "qv2x_conv3x3_i8_wide: in / w / out must be 16-byte aligned");
    WideArgs a{};
    if (int rc = fill_args(d, a)) return rc;
    a.in = in; a.wt = w_wide; a.scale = scale; a.corr = corr; a.aw = aw; a.bias = bias; a.out = out;
    hipStream_t st = (hipStream_t)stream;
#ifndef QV2X_NO_WS64
    if (ws64_takes(d)) return launch_ws64(d, in, w_wide, scale, corr, aw, bias, out, st);
    if (ws2_takes(d)) return launch_ws2(d, in, w_wide, scale, corr, aw, bias, out, st);
#endif
    const int patches8 = (a.n * a.tiles_x * a.tiles_y + 7) / 8 * 8;   // block ids come in groups of 8 (one per XCD)
    const int bn = wide_bn(d);
    a.items = patches8 * (a.cout / bn);
    // persistent workgroups: one round of what a CU holds (two waves per SIMD: 174-256 VGPRs; LDS 36-60 KB per workgroup), a multiple
    // of 8 * (cout / bn)
    const int period = 8 * (a.cout / bn);                              // ids `period` apart share the channel block and the XCD
    // (stride 2: four window-sum tables per set -- 45 KB of LDS: three 64-channel workgroups per CU)
    const int slots = 256 * (bn == 256 ? 1 : (bn == 128 ? (d->ngroups > 1 ? 1 : 2) : (a.stride2 ? 3 : 4))) / period * period;
    const dim3 grid(a.items < slots ? a.items : slots);
    if (a.stride2) {
        if (bn == 256) conv3x3_i8_wide_kernel<false, 8, 1, 256, 5, true><<<grid, 512, 0, st>>>(a);
        else if (bn == 128) conv3x3_i8_wide_kernel<false, 4, 1, 128, 5, true><<<grid, 256, 0, st>>>(a);
        else conv3x3_i8_wide_kernel<false, 2, 1, 64, 5, true><<<grid, 128, 0, st>>>(a);
    } else if (d->ngroups > 1) {
        if (bn == 256) conv3x3_i8_wide_kernel<true, 8, 1, 256><<<grid, 512, 0, st>>>(a);
        else conv3x3_i8_wide_kernel<true, 4, 1, 128><<<grid, 256, 0, st>>>(a);
    } else {
        if (bn == 256) conv3x3_i8_wide_kernel<false, 8, 1, 256><<<grid, 512, 0, st>>>(a);
        else if (bn == 128) conv3x3_i8_wide_kernel<false, 4, 1, 128><<<grid, 256, 0, st>>>(a);
        else conv3x3_i8_wide_kernel<false, 2, 1, 64><<<grid, 128, 0, st>>>(a);
    }
    return hip_check(hipGetLastError(), "qv2x_conv3x3_i8_wide launch");
}
