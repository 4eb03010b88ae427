/*
 * qv2x.h -- C ABI of libqv2x.so: the MI355X (gfx950) implementation of QuantV2X's quantized
 * per-agent encode + intermediate-fusion hot path.
 *
 * The reference has no FFI on this path -- it is Python on torch ops.  Each entry
 * point below replaces the torch call sequence of one reference function; the Python host
 * (quantv2x_amd/engine.py) binds them with ctypes.  Conventions:
 *   - plain device pointers + sizes; no allocation inside; the caller owns every buffer
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only enqueue work
 *   - return 0 on success, a negative QV2X_E* code on an argument error, or -1000 - hipError_t;
 *     qv2x_last_error() returns a message for the calling thread's last failure
 *   - reentrant per stream, no global mutable state
 *
 * Data layout in HBM (the "i8 BEV" format used between kernels):
 *   activation tensor = signed int8, value = uint8 code - 128, NHWC with a one-pixel border:
 *       [N][H + 2][W + 2][C],  element (n, y, x, c) at ((n*(H+2) + y+1)*(W+2) + x+1)*C + c
 *   the border holds (zero_point - 128), i.e. the code of 0.0, so 3x3 windows need no bounds checks.
 */
#ifndef QV2X_H
#define QV2X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libqv2x.so is built with -fvisibility=hidden: the entry points declared in this header are its whole dynamic symbol table
 * (tests/test_cabi_cpu.py checks `nm -D` against it). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define QV2X_OK 0
#define QV2X_EINVAL (-1)     /* bad argument (shape not supported, null pointer ...) */
#define QV2X_EALIGN (-2)     /* pointer / channel count not aligned as required */
#define QV2X_MAX_GROUPS 4

const char* qv2x_last_error(void);
/* The ABI of this header.  A caller built against another value must not pass its structs: 5 = qv2x_encode_desc grew the trailing `segs`
 * field (round 5; a caller built before it passes garbage there unless its struct was zero-initialised -- ADVICE r5); 6 = round 6:
 * qv2x_codebook_encode_candidates_i8 / qv2x_codebook_encode_listed_f32 added, no struct changed. */
#define QV2X_ABI_VERSION 6
int qv2x_version(void);            /* == QV2X_ABI_VERSION of the library's own build */

/* Fill a padded i8 BEV tensor (border AND interior) with one byte value: the per-frame canvas clear and the
 * one-time border initialisation.  Replaces torch.zeros(...) in PointPillarScatter.forward
 * (opencood/models/sub_modules/point_pillar_scatter.py:45-49). */
int qv2x_fill_i8(int8_t* buf, int64_t bytes, int value, void* stream);

/* SURVEY.md §8(f) rank 1 ("next" row).  Points -> pillars, the pre-step the reference runs on the CPU with the un-vendored
 * spconv.utils.Point2VoxelCPU3d (opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:54-85).  Contract (not a parity
 * claim): voxels in order of first point appearance, first-come <= max_points points per voxel, <= max_voxels voxels,
 * coords (agent, z, y, x), zero padded; deterministic.
 *   points f32 [n_points][4] (x, y, z, intensity), n_points < 2^20; lidar_range [6], voxel_size [3] HOST floats
 *   outputs sized for max_voxels: voxel_features f32 [max_voxels][max_points][4] (zero filled), voxel_coords i32 [..][4],
 *   voxel_num_points i32 [..]; n_voxels: DEVICE int32 = number of voxels written.  Rows past that count are left defined --
 *   coords (-1, -1, -1, -1) (an agent index qv2x_pfn_scatter_i8 drops), zero points, zero features -- so a caller may hand all
 *   max_voxels rows on without reading the count back.  Every clear is a kernel: the call is HIP-graph capturable.
 *   workspace: qv2x_voxelize_workspace_bytes(n_points) bytes of device memory */
int64_t qv2x_voxelize_workspace_bytes(int n_points);
int qv2x_voxelize_f32(const float* points, int n_points, const float* lidar_range, const float* voxel_size, int agent,
                      int max_points, int max_voxels, void* workspace, int64_t workspace_bytes,
                      float* voxel_features, int32_t* voxel_coords, int32_t* voxel_num_points, int32_t* n_voxels, void* stream);

/* a1 + a2.  QuantPillarVFE/QuantPFNLayer + PointPillarScatter in one pass
 * (opencood/quant/quant_block.py:589-715, opencood/models/sub_modules/pillar_vfe.py:105-155,
 *  point_pillar_scatter.py:19-75).
 *   voxel_features f32 [M][32][4], voxel_coords i32 [M][4] = (agent, z, y, x), voxel_num_points i32 [M]
 *   w f32 [64][10] fake-quantized folded weight, b f32 [64]; (d1, z1) Linear output quantizer,
 *   (d2, z2) the second quantizer after the ReLU; vox/off = voxel size and (voxel/2 + range_min), xyz.
 *   canvas: padded i8 BEV [N][ny+2][nx+2][64], already filled with (z2 - 128). 
 * Slots past voxel_num_points must hold zeros (the voxel generator's padding; the reference's mean over all slots
 * relies on it too): only the filled slots are read. */
typedef struct {
    float w[64 * 10];
    float b[64];
    float d1, z1, d2, z2;
    float vox[3], off[3];
} qv2x_pfn_params;
int qv2x_pfn_scatter_i8(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points,
                        int M, int max_points, const qv2x_pfn_params* params /* host */,
                        int8_t* canvas, int N, int ny, int nx, void* stream);
/* Sets the cells the same pillars were scattered to back to `value` (the code of 0.0 minus 128): a canvas that was clean before
 * qv2x_pfn_scatter_i8 is clean again, without the full qv2x_fill_i8 per frame.  Same coordinate checks as the scatter. */
int qv2x_pfn_unscatter_i8(const int32_t* voxel_coords, int M, int value, int8_t* canvas, int N, int ny, int nx, void* stream);

/* a3 / a4 / a5.  One QuantModule 3x3 convolution (zero padding 1, stride 1 or 2) + folded BN bias + ReLU +
 * output activation quantizer (opencood/quant/quant_layer.py:391-410 on F.conv2d; blocks of
 * QuantBaseBEVBackbone quant_block.py:243-303 and QuantDoubleConv :552-572), on integer codes:
 *     T_g  = sum_{kh,kw,ci in group g} (x - zx_g) * (w - zw[co])                  exact int32 (MFMA i8)
 *     y    = bias[co] + sum_g float(T_g) * scale[g][co]                           fp32 mul, add
 *     out  = clamp(rint(max(y, 0) / out_delta) + out_zp, 0, 255) - 128
 *   in   : padded i8 BEV [N][H+2][W+2][cin_total]; input channel groups (the concat of the three deblocks
 *          carries three activation scales) described by group_c0/group_c/group_zx
 *   w    : i8 (code - 128), [Cout][G][3][3][C_g]  (K contiguous per output channel)
 *   scale: f32 [G][Cout] = delta_x[g] * delta_w[co];  corr: i32 [G][Cout] = ax_g * sum_k ws + K_g * ax_g * aw[co]
 *          with ax_g = 128 - zx_g, aw[co] = 128 - zw[co], ws = w code - 128;  aw: i32 [Cout]; bias f32 [Cout]
 *   out  : padded i8 BEV [N][Ho+2][Wo+2][out_ctotal], written at channel offset out_c0 (interior only). */
typedef struct {
    int32_t n, h, w, cin_total, stride, cout;
    int32_t ngroups;
    int32_t group_c0[QV2X_MAX_GROUPS], group_c[QV2X_MAX_GROUPS], group_zx[QV2X_MAX_GROUPS];
    int32_t out_ctotal, out_c0;
    int32_t relu;
    float out_delta, out_zp;
} qv2x_conv_desc;
int qv2x_conv3x3_i8(const qv2x_conv_desc* desc /* host */, const int8_t* in, const int8_t* w,
                    const float* scale, const int32_t* corr, const int32_t* aw, const float* bias,
                    int8_t* out, void* stream);

/* The same convolution for wide layers (stride 1, cout % 256 == 0: the DoubleConv shrinker,
 * opencood/models/sub_modules/downsample_conv.py:17-31 under quant_block.py:552-572) with the weights pre-tiled:
 *   w_wide : i8 [Cout/T][chunk][tap][T/32][K half][lane][16], T = min(Cout, 256), chunk = the 64-channel chunks of group 0, then
 *            of group 1, ...: the MFMA A-operand fragments of a K step in load order (T x 64 bytes contiguous per K step, 1 KiB per
 *            wave-instruction); opaque to the caller, made from the row-major layout by qv2x_conv3x3_i8_pack_wide, once.
 * Results are bit-identical to qv2x_conv3x3_i8.  qv2x_conv3x3_i8_wide_ok returns 1 when the layer qualifies
 * (and is large enough for the wide kernel to pay: N*H*W >= 16384), else 0.
 * Round 3: stride 2 too (the ZeroPad2d + stride-2 first convolution of a backbone level, base_bev_backbone.py:60-66): one input group; the
 * kernel reads the input as four parity planes, each an ordinary halo tile with 4 / 2 / 2 / 1 of the nine taps, and w_wide stores the nine
 * steps of 64 input channels in that order (taps 0 2 6 8 | 1 7 | 3 5 | 4).  _ok() answers 1 for it from 256 output channels up. */
int qv2x_conv3x3_i8_wide_ok(const qv2x_conv_desc* desc /* host */);
int qv2x_conv3x3_i8_pack_wide(const qv2x_conv_desc* desc /* host */, const int8_t* w, int8_t* w_wide, void* stream);
int qv2x_conv3x3_i8_wide(const qv2x_conv_desc* desc /* host */, const int8_t* in, const int8_t* w_wide,
                         const float* scale, const int32_t* corr, const int32_t* aw, const float* bias,
                         int8_t* out, void* stream);

/* a3, one backbone level in ONE launch.  A chain of `depth` (1..4) such convolutions with 64 input and 64 output channels
 * each (level 0 of BaseBEVBackbone, base_bev_backbone.py:96-119 / quant_block.py:243-303): layer 0 has stride `stride0`
 * (the ZeroPad2d + stride-2 convolution, or 1), the others stride 1; every layer has one input group.  Intermediate maps stay
 * in LDS; results are identical to `depth` qv2x_conv3x3_i8 calls.
 *   in     : padded i8 BEV [N][in_h+2][in_w+2][64];  out: padded i8 BEV [N][h+2][w+2][64] (interior written)
 *   w_chain: i8 (code - 128) [depth][2][9][2][64][16]: the B fragments of v_mfma_i32_32x32x32_i8 in load order --
 *            [layer][cout / 32][tap][cin / 32][lane][16 B], lane = 32 * ((cin / 16) & 1) + cout % 32, bytes = cin % 16
 *   scale, bias f32 [depth][64]; corr, aw i32 [depth][64] (as qv2x_conv3x3_i8, one group);
 *   out_delta / out_zp: the output quantizer of every layer (the pad code of layer l's map is out_zp[l] - 128). */
typedef struct {
    int32_t n, h, w;           /* output map of every layer */
    int32_t in_h, in_w;        /* input map of layer 0 */
    int32_t depth, stride0, relu;
    float out_delta[4], out_zp[4];
} qv2x_chain_desc;
int qv2x_conv3x3_i8_chain64(const qv2x_chain_desc* desc /* host */, const int8_t* in, const int8_t* w_chain,
                            const float* scale, const int32_t* corr, const int32_t* aw, const float* bias,
                            int8_t* out, void* stream);

/* a3 deblocks.  QuantModule over ConvTranspose2d with kernel == stride == s, + bias + ReLU + output quantizer.
 * The reference's per-dim-0 weight scales are per C_in here (quant_layer.py:192-195 on a [Cin,Cout,s,s]
 * weight), i.e. on the reduction axis, so the sum runs in fp32 on the f32 MFMA as an ascending-ci fmaf chain:
 *     acc = 0; acc = fma(float(x - zx) * dx, wdeq[ci][co][i][j], acc);  y = acc + bias[co]
 *   w: f32, [Cin/4][s*s*Cout][4] (column = (i*s + j)*Cout + co; four consecutive ci innermost, stored k0, k2, k1, k3). */
typedef struct {
    int32_t n, h, w, cin, cout, s;
    int32_t in_zx;
    float in_delta;
    int32_t out_ctotal, out_c0, relu;
    float out_delta, out_zp;
    int32_t out_h, out_w;      /* interior size of the padded destination [n][out_h+2][out_w+2][out_ctotal]: must equal
                                * (h*s, w*s), else QV2X_EINVAL (the reference's torch.cat raises on such a mismatch) */
} qv2x_deconv_desc;
int qv2x_deconv_i8(const qv2x_deconv_desc* desc /* host */, const int8_t* in, const float* w, const float* bias,
                   int8_t* out, void* stream);

/* Up to 4 such layers in one launch (the deblocks only feed the concat: one pool of wave tiles instead of one tail per
 * layer).  Arrays of `n` host-side entries; results identical to `n` qv2x_deconv_i8 calls. */
int qv2x_deconv_i8_batch(const qv2x_deconv_desc* descs /* host */, int n, const int8_t* const* ins /* host array */,
                         const float* const* ws, const float* const* biases, int8_t* const* outs, void* stream);

/* a6.  UMGMQuantizer.encode (opencood/models/sub_modules/codebook.py:330-337 -> :231-239 -> :106-131),
 * D = 256, up to 4 residual levels, seg_num m = 1 | 2 | 4 segments of D / m dims (codebook.py:115-131: x.reshape(n, m, d), a distance and an
 * argmin per segment), dict_size kc <= 256 codes per segment and level (a multiple of 32; of 64 when m > 1; m * kc <= 512), on the
 * dequantized shrinker output.  The reference's yamls: (m, kc) = (1, 128) and (2, 256).
 *   in: padded i8 BEV [N][H+2][W+2][256] with (in_delta, in_zx)
 *   the codebook comes EXTENDED: Kc = m * kc rows of 256 floats, row s * kc + j = C[s][j] in dims [s d, (s + 1) d), zeros elsewhere
 *   (quantv2x_amd/ptq_state.py:extended_codebook) -- a row's dot product over all 256 dims is then the segment's own ascending fma chain
 *   weights: one f32 blob per level, laid out by the host as
 *       stage [64][256][4] | stage_b [256] | qhead [64][256][4] | qhead_b [256] | lhead [64][256][4] | lhead_b [256]
 *       | cb_packed [64][Kc][4] | cb [Kc][256] | c2 [Kc]
 *     ([K/4][cols][4] = four consecutive k innermost, stored in the order k0, k2, k1, k3); lhead* unused on the last level;
 *     then the same four matrices once more in MFMA A-operand order for the wave-per-32-cells form that launches of many frames take,
 *       stage [4][32][2][64][4] | qhead [4][32][2][64][4] | cb [ceil(Kc/64)][32][2][64][4] | lhead [4][32][2][64][4] | 4096 floats of padding
 *     ([pair][group][tile][lane = 32 h + c][s] = W[64 pair + 32 tile + c][8 group + 2 s + h], rows past Kc zero: one linear stream of
 *     1 KiB groups).  seg_num m > 1: the cb part holds the DIAGONAL blocks of the extended codebook only -- a code of segment sg is zero
 *     outside dims [256 sg / m, 256 (sg + 1) / m), and the wave form walks just those: [m / 2 segment pairs][dict_size / 32][32 / m groups]
 *     [2 segments of the pair][lane][s] = cb[sg * dict_size + 32 tile + c][256 sg / m + 8 group + 2 s + h]; lhead follows it directly and
 *     the section is zero-filled to the size above (quantv2x_amd/engine.py:_pack_wave_seg is the reference packing)
 *   codes: u8 [levels * m][N*H*W] -- THE WIRE FORMAT: plane l * m + s holds segment s's index (0 .. kc - 1) of level l; the decode side
 *   (qv2x_fuse_att_f32, qv2x_decode_lut_f32, ...) takes `levels` = levels * m planes and a table [levels * m][kc][256], which is the
 *   per-level table over the extended codebook's rows [levels][m * kc][256] read plane by plane. */
typedef struct {
    int32_t n, h, w;
    int32_t levels, kc;      /* kc: dict_size = codes per segment and level */
    int32_t in_zx;
    float in_delta;
    int32_t segs;            /* seg_num (m): 1, 2 or 4; 0 is read as 1 */
} qv2x_encode_desc;
/* floats of one level blob; `kc` = rows of the extended codebook (m * dict_size) */
int64_t qv2x_codebook_level_floats(int kc);
/* |C_k|^2 of a [kc][256] codebook with the summation order the encode kernel assumes (four 64-wide ascending
 * fma chains, (s0 + s1) + (s2 + s3)); fills the c2 slot of a level blob at engine-build time. */
int qv2x_codebook_c2_f32(const float* codebook, int kc, float* c2, void* stream);
int qv2x_codebook_encode_f32(const qv2x_encode_desc* desc /* host */, const int8_t* in,
                             const float* const* level_weights /* host array of device pointers */,
                             uint8_t* codes, void* stream);
/* The same with the wave-per-32-cells form forced (qv2x_codebook_encode_f32 / _f32in pick it by launch size; the codes are identical).
 * Exactly one of `in` (padded i8 BEV) and `in_f32` (padded fp32 BEV, as qv2x_codebook_encode_f32in) is non-null. */
int qv2x_codebook_encode_wave_f32(const qv2x_encode_desc* desc /* host */, const int8_t* in, const float* in_f32,
                                  const float* const* level_weights /* host array */, uint8_t* codes, void* stream);

/* OPT-IN, not the parity configuration: qv2x_codebook_encode_f32 with the encoder's affine heads collapsed on the host
 * (quantv2x_amd/engine.py: collapse_encoder).  distance_l[k] - |q_l|^2 = G_l[k] . x + g_l[k] + sum_{j<l} T_lj[code_j][k] with x the shared
 * feature: ONE 256 -> levels*kc GEMM per cell plus an argmin chain over the tables, 6.3x fewer flops than the reference's op order.  The
 * argmin can differ from the reference's where the two best distances are closer than fp32 rounding error (measured: DESIGN.md §3); the
 * exact entry above stays the shipped default.  levels <= 3, levels * kc <= 384.
 *   g_packed f32 [levels*kc/32][128][64]: value (tile t, step i, lane l) = in_delta * G[32 t + l % 32][2 i + l / 32]
 *   bias f32 [levels*kc] = g + in_delta * (0 - in_zx) * rowsum(G)      (the kernel multiplies by the uint8 code itself)
 *   tables f32 [levels*(levels-1)/2][kc][kc]: table (l, j) at index l (l - 1) / 2 + j */
int qv2x_codebook_encode_collapsed_f32(const qv2x_encode_desc* desc /* host */, const int8_t* in, const float* g_packed, const float* bias,
                                       const float* tables, uint8_t* codes, void* stream);

/* The two-stage EXACT encode (round 6; quantv2x_amd/encode_two_stage.py has the derivation and builds the operands): the same indices as
 * qv2x_codebook_encode_f32 -- UMGMQuantizer.encode, codebook.py:330-337 -> :231-239 -> :106-131, first-argmin ties included -- for a
 * fraction of its work.
 *
 * Stage 1, qv2x_codebook_encode_candidates_i8: the collapsed scores s_l[k] = dist_l[k] - |q_l|^2 of every cell in EXACT integer arithmetic
 * (G on a fixed-point grid h as three balanced int8 limbs, v_mfma_i32_32x32x32_i8; limbs, bias and table rows combined on integers below
 * 2^53) -> codes [levels][n*h*w] for every cell, and the cells whose gap between the two best scores at ANY level does not exceed
 *     tau_l / h = tau[l][0] + tau[l][1] N0 + tau[l][2] N0^2 + sum |code - zx|,    N0 = in_delta * sqrt(sum (code - zx)^2)
 * appended (in no particular order) to `list`, their number in counters[0].  tau bounds twice the largest difference the fp32 chain's
 * rounding can make to a score plus this stage's own grid error: a cell NOT listed has the fp32 chain's strict minimum at every level.
 *   g_limbs  i8  [levels][kc/32][3][8][64][16]: A fragments, lane = 32 * half + score % 32, bytes = input channel 32 * step + 16 * half + 0..15
 *   bias_packed f64 [levels*kc] = 128 * (rint(g / h) + (128 - zx) * rowsum(G_int)) + k      (the kernel multiplies by the stored byte code - 128)
 *   tables   i32 [levels(levels-1)/2][kc][kc] = rint(T_lj / h), table (l, j) at l (l - 1) / 2 + j (>= one table's worth allocated)
 *   tau      f32 [levels][3], HOST (copied into the launch)
 *   list u32 [3][n*h*w]: list c = the cells first undecided at level c (their codes BELOW level c are proven); counters u32 [4]: [0] all
 *   listed cells, [1 + c] the length of list c -- DEVICE, zeroed by this call (a kernel node under stream capture)
 * Stage 2, qv2x_codebook_encode_listed_f32: qv2x_codebook_encode_f32's kernel (one wave per 32 cells, the reference's op order, bit-exact)
 * on the listed cells only -- for a cell of list c the quantization head, |q|^2 and the distances of the levels below c are skipped and the
 * code stage 1 stored is used (the latent chain stage -> lhead -> residual runs as always: the same x at level c); `list` and `list_count`
 * are stage 1's `list` and `counters` (DEVICE), the launch is a fixed number of persistent waves + a fixed number of workgroups for the
 * remainder: both stages are capturable in a HIP graph.  seg_num 1, dict_size 32 | 64 | 96 | 128, up to three levels. */
int qv2x_codebook_encode_candidates_i8(const qv2x_encode_desc* desc /* host */, const int8_t* in, const int8_t* g_limbs, const double* bias_packed,
                                       const int32_t* tables, const float* tau /* host */, uint8_t* codes, uint32_t* list, uint32_t* counters,
                                       void* stream);
int qv2x_codebook_encode_listed_f32(const qv2x_encode_desc* desc /* host */, const int8_t* in, const float* const* level_weights /* host array */,
                                    const uint32_t* list, const uint32_t* list_count, uint8_t* codes, void* stream);

/* a7 + a8 + a9 + a10.  UMGMQuantizer.decode as three table look-ups (codebook.py:339-343; all heads affine),
 * warp_affine_simple (torch_transformation_utils.py:323-332: affine_grid + bilinear grid_sample, zeros,
 * align_corners=False) of every agent into the ego frame and AttFusion's per-cell scaled-dot-product
 * attention, ego row (fusion_in_one.py:126-151).
 *   codes   : u8 code planes; plane (agent, level) starts at agent*code_agent_stride + level*code_level_stride
 *             ([levels][A][H*W] out of qv2x_codebook_encode_f32, [A][levels][H*W] after an all-gather)
 *   lut     : f32 [levels][kc][256], lut_bias f32 [256]
 *   feats   : optional f32 [A][H*W][256] instead of codes (models without the codebook); pass codes = NULL
 *   pairwise: DEVICE f64 [max_cav][max_cav][4][4], T[i][j] = T_j^-1 T_i; row `ego` is normalised in-kernel exactly as
 *             normalize_pairwise_tfm(T, H, W, discrete_ratio) does (transformation_utils.py:68-92)
 *   fused   : f32 [H*W][256] */
typedef struct {
    int32_t agents, h, w, levels, kc;
    int32_t max_cav, ego;
    int64_t code_agent_stride, code_level_stride;
    double h_metres, w_metres, discrete_ratio;
    int32_t fusion;            /* 0 = AttFusion (per-cell attention, ego row; fusion_in_one.py:126-151), 1 = MaxFusion (F-Cooper:
                                * elementwise max over the warped agents, fusion_in_one.py:87-123) */
} qv2x_fuse_desc;
int qv2x_fuse_att_f32(const qv2x_fuse_desc* desc /* host */, const uint8_t* codes, const float* lut, const float* lut_bias,
                      const float* feats, const double* pairwise, float* fused, void* stream);

/* The same for several scenes of ONE call in one launch (the reference's loop over record_len in AttFusion.forward,
 * fusion_in_one.py:131-151): scene s fuses `scene_agents[s]` agents whose first code plane starts `scene_offset[s]` BYTES into `codes`
 * (or whose first fp32 map starts `scene_offset[s]` FLOATS into `feats`), with pairwise + s * max_cav * max_cav * 16, into
 * fused + s * h * w * 256; every scene's ego is `desc->ego` of its own agents.  `desc->agents` is ignored.  1..64 scenes. */
int qv2x_fuse_att_batch_f32(const qv2x_fuse_desc* desc /* host */, int n_scenes, const int64_t* scene_offset /* host */,
                            const int32_t* scene_agents /* host */, const uint8_t* codes, const float* lut, const float* lut_bias,
                            const float* feats, const double* pairwise, float* fused, void* stream);

/* codes -> fp32 rows (decode only), used for the *_single heads: out f32 [R][256] */
int qv2x_decode_lut_f32(const uint8_t* codes, int R, int levels, int kc, const float* lut, const float* lut_bias,
                        float* out, void* stream);

/* Interior of a padded i8 BEV tensor -> fp32 rows [N*H*W][C], x = (code - zp) * delta: the shared feature of models
 * WITHOUT the codebook (HeterModelBaseline), i.e. the dequantized output activation of the shrinker's last QuantModule. */
int qv2x_dequant_i8_f32(const int8_t* in, int n, int h, int w, int c, int zp, float delta, float* out, void* stream);

/* a11.  1x1 QuantModule heads (heter_model_baseline.py:128-133,242-260) on fp32 rows:
 *     y = fma chain over ci (acc0 = bias[co]);  out = (clamp(rint(y / da[co]) + za[co], 0, 255) - za[co]) * da[co]
 *   x f32 [R][256]; w f32 [64][cout_pad][4] (k0, k2, k1, k3 order; cout_pad = cout rounded up to 32, zero filled); bias/da/za f32 [cout_pad]
 *   (da[co] <= 0 disables the output quantizer for that channel); out f32 NCHW [B][cout][hw] with R = B * hw. */
int qv2x_heads_f32(const float* x, int R, int hw, int cout, int cout_pad, const float* w, const float* bias,
                   const float* da, const float* za, float* out, void* stream);

/* a7-a11 in ONE launch (round 4): qv2x_fuse_att_batch_f32 followed by qv2x_heads_f32 on the fused rows, tile by tile in LDS -- the fused
 * fp32 map (36 MB per V2X-Real frame) is neither written nor read back (heter_model_baseline.py:237-260 over fusion_in_one.py:126-151).
 * Same arithmetic as the two calls, bit for bit.  out f32 NCHW [n_scenes][cout][hw]; `fused_tap` (optional, may be null): receives the
 * fused rows f32 [n_scenes][hw][256] as well (parity tests, debugging).  Pays from a few rounds of 32-cell tiles on (a batch of frames);
 * one frame alone is faster as two launches. */
int qv2x_fuse_heads_batch_f32(const qv2x_fuse_desc* desc /* host */, int n_scenes, const int64_t* scene_offset /* host */,
                              const int32_t* scene_agents /* host */, const uint8_t* codes, const float* lut, const float* lut_bias,
                              const float* feats, const double* pairwise, int cout, int cout_pad, const float* w, const float* bias,
                              const float* da, const float* za, float* out, float* fused_tap, void* stream);

/* Every 1x1 head of SINGLE-AGENT scenes by table look-up (round 4).  AttFusion over one agent is the identity (fusion_in_one.py:131-151 with
 * record_len 1, T[0][0] = I), decode is a sum of per-level table rows (qv2x_decode_lut_f32) and a 1x1 head is linear, so cls | reg | dir on
 * the fused map (heter_model_baseline.py:242-260) and the *_single heads (:224-230) are
 *     y[co] = bias[co] + sum_l tables[l][code_l][co],    out = (clamp(rint(y / da[co]) + za[co], 0, 255) - za[co]) * da[co]   (da <= 0: no quantizer)
 * for co < c0 + c1: the first c0 channels go to out0 f32 NCHW [R / hw][c0][hw], the next c1 to out1 [R / hw][c1][hw] (either set may be empty:
 * c = 0 and a null pointer).  codes u8 [levels][R], rows agent-major; tables f32 [levels][kc][c0 + c1] = decode table x head weights, made by
 * the caller in float64 (engine.py); `levels` = code planes (1..16).  Tables of up to four planes that fit the 160 KB of LDS are copied there;
 * larger ones (seg_num 2 x dict_size 256: six planes) are read row by row from global memory: c0 + c1 a multiple of 4, <= 128, the arrays
 * 16-byte aligned.  A different fp32 association than
 * decode-then-GEMM: equal to ~1e-6 relative before the output quantizer (bounded in the tests like every head). */
int qv2x_table_heads_f32(const uint8_t* codes, int R, int hw, int levels, int kc, int c0, int c1, const float* tables,
                         const float* bias, const float* da, const float* za, float* out0, float* out1, void* stream);

/* *_preds_single (heter_model_baseline.py:224-230) in one launch: qv2x_decode_lut_f32 followed by qv2x_heads_f32 on
 * every agent's own decoded feature.  codes u8 [levels][R], rows agent-major (R = agents * hw); out f32 [agents][cout][hw]. */
int qv2x_decode_heads_f32(const uint8_t* codes, int R, int hw, int levels, int kc, const float* lut, const float* lut_bias,
                          int cout, int cout_pad, const float* w, const float* bias, const float* da, const float* za,
                          float* out, void* stream);

/* *_preds_single (heter_model_baseline.py:224-230) without a GEMM: decode (codebook.py:339-343) is a sum of per-level table rows and a
 * 1x1 head is linear, so head(decode(codes)) = bias + sum_l tables[l][code_l] with tables f32 [levels][kc][cout] = decode table x W^T
 * made by the caller (engine.py: float64 on the host); then the head's output quantizer (da[co] <= 0: off).  codes u8 [levels][R],
 * rows agent-major (R = agents * hw); out f32 [agents][cout][hw].  A different fp32 association than qv2x_decode_heads_f32: equal up to
 * the rounding boundaries of the output quantizer. */
int qv2x_single_heads_lut_f32(const uint8_t* codes, int R, int hw, int levels, int kc, int cout, const float* tables,
                              const float* bias, const float* da, const float* za, float* out, void* stream);

/* qv2x_heads_f32 on the fused rows AND qv2x_decode_heads_f32 on the agents' own codes in one launch (same results): the
 * two 1x1-head passes of a frame are independent of each other. */
int qv2x_heads_pair_f32(const float* x, int R, int hw, int cout, int cout_pad, const float* w, const float* bias,
                        const float* da, const float* za, float* out,
                        const uint8_t* codes, int R1, int levels, int kc, const float* lut, const float* lut_bias,
                        int cout1, int cout_pad1, const float* w1, const float* bias1, const float* da1,
                        const float* za1, float* out1, void* stream);

/* f2.  Head maps -> boxes for one CAV.
 * num_classes == 1: VoxelPostprocessor.post_process (data_utils/post_processor/voxel_postprocessor.py:245-405):
 *   sigmoid + score threshold, delta_to_boxes3d (:408-453), direction-bin fix (:316-331), corners
 *   (utils/box_utils.py:152-204), projection by `transform` (:278-316), remove_large_pred_bbx / remove_bbx_abnormal_z
 *   (:916-966: max_extent 6, z in [-3, 1]), rotated NMS on the bottom faces over the `max_boxes` best scores (:769-814;
 *   the reference takes 1000), range mask on all corners (:384-421).
 * num_classes  > 1: VoxelPostprocessor3Heads.post_process (voxel_postprocessor_3heads.py:318-478): `anchors_per_cell`
 *   counts every anchor of a cell over all anchor sets, the score is the largest of the `num_classes` class
 *   probabilities of an anchor and the label its index + 1 (:362-373), no direction fix (num_bins = 0), the size / z
 *   limits of utils/box_utils_mc.py (max_extent 100, z in [-100, 100]) and its x-y-only range mask (:388-419,
 *   range_xy_only = 1, range = the reference's GT_RANGE).
 *   cls f32 [A*num_classes][H][W], reg f32 [7A][H][W], dir f32 [num_bins*A][H][W] (NULL with num_bins == 0),
 *   anchors f32 [H*W*A][7] = (x, y, z, h, w, l, yaw) in (h, w, a) order;
 *   out_corners f32 [max_boxes][8][3], out_scores f32 [max_boxes], out_labels i32 [max_boxes] (may be NULL),
 *   out_count i32 [1] (device): boxes in descending score order.
 * Deterministic: candidates keep the (h, w, a) order, equal scores keep it through the stable sort. */
typedef struct {
    int32_t h, w, anchors_per_cell, num_bins;
    float score_threshold, nms_threshold, dir_offset;
    float range[6];                 /* x0, y0, z0, x1, y1, z1 */
    float transform[16];            /* row-major 4x4, CAV -> ego */
    int32_t max_boxes;              /* 1..1024 */
    int32_t num_classes;            /* 1..8 */
    int32_t range_xy_only;
    float max_extent, z_min, z_max;
} qv2x_postprocess_desc;
int64_t qv2x_postprocess_workspace_bytes(const qv2x_postprocess_desc* desc /* host */);
int qv2x_postprocess_f32(const qv2x_postprocess_desc* desc /* host */, const float* cls, const float* reg, const float* dir,
                         const float* anchors, void* workspace, int64_t workspace_bytes, float* out_corners,
                         float* out_scores, int32_t* out_labels, int32_t* out_count, void* stream);

/* Late fusion (hypes_yaml/v2x_real/LiDAROnly/lidar_late_mc_fusion.yaml): the same pipeline over `ncav` (1..8) CAVs -- the loop over
 * `cav_content` of the reference's post_process (voxel_postprocessor.py:272-345, voxel_postprocessor_3heads.py:345-420): every CAV's head
 * maps decoded against ITS anchors and projected by ITS matrix, candidates concatenated in CAV order, then one filter / sort / NMS /
 * range mask over the union.  Host arrays of `ncav` device pointers; transforms = HOST float [ncav][16] (desc->transform is ignored);
 * workspace: qv2x_postprocess_late_workspace_bytes(desc, ncav). */
int64_t qv2x_postprocess_late_workspace_bytes(const qv2x_postprocess_desc* desc /* host */, int ncav);
int qv2x_postprocess_late_f32(const qv2x_postprocess_desc* desc /* host */, int ncav, const float* const* cls, const float* const* reg,
                              const float* const* dir, const float* const* anchors, const float* transforms /* host */, void* workspace,
                              int64_t workspace_bytes, float* out_corners, float* out_scores, int32_t* out_labels, int32_t* out_count,
                              void* stream);

/* ---- the un-quantized model (SURVEY.md §8(b) "fp32 fall-backs for un-quantized mode") ------------------------------------------
 * What the reference's plain opencood/tools/inference.py:106-170 flow runs in fp32 -- PillarVFE (pillar_vfe.py:105-155) +
 * PointPillarScatter, BaseBEVBackbone (base_bev_backbone.py:96-119), DownsampleConv (downsample_conv.py:26-51) -- as f32-MFMA
 * kernels; the codebook, the fusion and the heads are fp32 in both modes.  BatchNorm folded by the host (fold_bn.py:19-127).
 * fp32 activations: NHWC with a one-pixel zero border, [N][H+2][W+2][C].
 *
 * qv2x_conv3x3_f32: 3x3, padding 1, stride 1 | 2:  out[co] = relu(sum_k x_k w[co][k] + bias[co]) over K = 9 * cin, k = tap * cin + ci.
 * qv2x_deconv_f32 : ConvTranspose2d with kernel == stride == s: column (i*s + j) * cout + co over K = cin.
 * Both: ONE fp32 fma chain per output, K walked in groups of 8 in the order k0, k4, k1, k5, k2, k6, k3, k7, acc0 = 0, then + bias
 * (oracle/qv2x_oracle.c:orc_gemm_f32 is the same chain).
 *   w: f32 [K / 8][columns][2][4] -- element [g][col][half][e] = W[col][8 g + 4 half + e]; columns % 64 == 0, cin % 8 == 0
 *   input channel window [cin0, cin0 + cin) of a cin_total-channel tensor, output window [out_c0, out_c0 + cout) of out_ctotal. */
typedef struct {
    int32_t n, h, w;             /* input map */
    int32_t cin_total, cin0, cin;
    int32_t stride;              /* conv: 1 | 2; deconv: s */
    int32_t cout;
    int32_t out_ctotal, out_c0, relu;
} qv2x_f32conv_desc;
int qv2x_conv3x3_f32(const qv2x_f32conv_desc* desc /* host */, const float* in, const float* w, const float* bias, float* out, void* stream);
int qv2x_deconv_f32(const qv2x_f32conv_desc* desc /* host */, const float* in, const float* w, const float* bias, float* out, void* stream);
/* a1 + a2 in fp32: Linear(10 -> 64) with folded BN, ReLU, max over the points, scattered into the zero-filled fp32 canvas
 * [N][ny+2][nx+2][64].  w [64][10], b [64], vox / off [3]: HOST arrays (passed by value to the kernel). */
int qv2x_pfn_scatter_f32(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points, int M, int max_points,
                         const float* w, const float* b, const float* vox, const float* off, float* canvas, int N, int ny, int nx, void* stream);
/* a6 on fp32 rows: qv2x_codebook_encode_f32 with the shared feature given as fp32 [N][h+2][w+2][256] (in_zx / in_delta unused) */
int qv2x_codebook_encode_f32in(const qv2x_encode_desc* desc /* host */, const float* in, const float* const* level_weights /* host array */,
                               uint8_t* codes, void* stream);

/* f3, first kernel (the HEAL Pyramid fusion model).  weighted_fuse of one scale of one scene
 * (opencood/models/fuse_modules/pyramid_fuse.py:17-62): every agent's feature map and occupancy score map warped into the ego frame
 * (warp_affine_simple, as qv2x_fuse_att_f32), warped scores that are exactly 0 masked to -inf, softmax over the agents (NaN -> 0),
 * score-weighted sum of the warped features.  The descriptor's agents / h / w / max_cav / ego / metres fields are used.
 *   feats f32 [agents][h*w][channels] (channels last), score f32 [agents][h*w] (sigmoid(occupancy) + 1e-4, camera crop mask applied),
 *   pairwise f64 [max_cav][max_cav][4][4];  out f32 [h*w][channels]. */
int qv2x_pyramid_weighted_fuse_f32(const qv2x_fuse_desc* desc /* host */, int channels, const float* feats, const float* score,
                                   const double* pairwise, float* out, void* stream);

/* ---- f3, the rest of the HEAL Pyramid path (heter_pyramid_collab_codebook_mc_encdec.py:33-181 under QuantModel) -----------------
 * Maps of activation CODES are padded i8 BEV tensors as everywhere else; maps that are not on a quantizer grid (the decoded
 * feature, the 1x1 shortcut branch, the fused levels) are plain fp32 rows [N*H*W][C]. */

/* A 1x1 convolution (stride 1 | 2, no padding) on codes: conv1 / conv3 / the shortcut of QuantBottleneck and QuantBasicBlock
 * (quant_block.py:68-131; QuantModule quant_layer.py:391-410):
 *     T = sum_ci (x - zx) * (w - zw[co])  exact (MFMA i8);   y = bias[co] + float(T) * scale[co]
 *   mode 0: out = quant(relu ? max(y, 0) : y)                         -> padded i8 BEV [N][Ho+2][Wo+2][out_ctotal] at out_c0
 *   mode 1: out = y                  (disable_act_quant)              -> f32 [N*Ho*Wo][cout]
 *   mode 2: out = quant(max(y + res, 0)),  res f32 [N*Ho*Wo][cout]     (the block's fp32 shortcut: `out += residual`, ReLU, the block's
 *   mode 3: out = quant(max(y + (r - res_zx) * res_delta, 0)),         quantizer) -- res i8 = the block INPUT codes, padded i8 BEV
 *                                                                      [N][Ho+2][Wo+2][cout]
 *   in  : padded i8 BEV [N][H+2][W+2][cin];  Ho = (H - 1) / stride + 1
 *   w_frag: i8 (code - 128), the A fragments of v_mfma_i32_32x32x32_i8 in load order: [cout/32][cin/32][lane 0..63][16 B] with
 *           lane = 32 * ((ci / 16) & 1) + co % 32, bytes = ci % 16
 *   scale f32 [cout] = delta_x * delta_w[co]; corr i32 [cout] = ax * sum_ci ws + cin * ax * aw[co]; aw i32 [cout]; bias f32 [cout]
 *   (ax = 128 - zx, aw[co] = 128 - zw[co], ws = w code - 128: as qv2x_conv3x3_i8). */
typedef struct {
    int32_t n, h, w, cin, cout, stride;
    int32_t mode, relu;
    int32_t out_ctotal, out_c0;
    float out_delta, out_zp;
    int32_t res_zx;
    float res_delta;
} qv2x_conv1x1_desc;
int qv2x_conv1x1_i8(const qv2x_conv1x1_desc* desc /* host */, const int8_t* in, const int8_t* w_frag, const float* scale,
                    const int32_t* corr, const int32_t* aw, const float* bias, const void* res, void* out, void* stream);

/* The grouped 3x3 convolution of QuantBottleneck (ResNeXt: groups of cg = 4 | 8 | 16 channels, c inputs = c outputs, c % 32 == 0; zero
 * padding 1, stride 1 | 2) + bias + ReLU + output quantizer; the same integer arithmetic.  Every 32-channel slab runs as a dense 32 -> 32
 * convolution with a block-diagonal weight matrix on the MFMA (structural zeros add exactly 0).
 *   in padded i8 BEV [N][H+2][W+2][c] -> out padded i8 BEV [N][Ho+2][Wo+2][c]
 *   w_frag: i8 (code - 128, 0 outside the group) [c/32 slabs][9 taps][lane 0..63][16 B] = the A fragments of v_mfma_i32_32x32x32_i8:
 *           lane = 32 * ((ci / 16) & 1) + co % 32, bytes = ci % 16 (ci, co inside the slab; tap = 3 * kh + kw)
 *   scale / corr / aw / bias as above with K = 9 * cg (corr over the group's taps only). */
typedef struct {
    int32_t n, h, w, c, cg, stride, relu;
    float out_delta, out_zp;
} qv2x_gconv_desc;
int qv2x_gconv3x3_i8(const qv2x_gconv_desc* desc /* host */, const int8_t* in, const int8_t* w_frag, const float* scale,
                     const int32_t* corr, const int32_t* aw, const float* bias, int8_t* out, void* stream);

/* One whole QuantBottleneck (quant_block.py:100-131) in one launch, for the blocks with stride 1 and an identity shortcut:
 *     out = quant_b(relu(conv3(quant_2(relu(gconv3x3(quant_1(relu(conv1(x))))))) + (x - in_zx) * in_delta))
 * -- the three layers' integer arithmetic exactly as qv2x_conv1x1_i8 / qv2x_gconv3x3_i8 / qv2x_conv1x1_i8 (mode 3) compute it; the
 * intermediate maps stay in LDS (conv1 is recomputed on the 1-pixel halo of each 2 x 32 output patch).
 *   x, out: padded i8 BEV [N][H+2][W+2][planes];  w / scale / corr / aw / bias: HOST arrays of three device pointers (conv1, conv2, conv3)
 *   in the layouts of the separate entry points (w[0]: [width/32][planes/32][64][16], w[1]: [width/32][9][64][16], w[2]: [planes/32][width/32][64][16]). */
typedef struct {
    int32_t n, h, w, planes, width, cg;
    int32_t in_zx;
    float in_delta;
    float delta1, zp1, delta2, zp2, out_delta, out_zp;
} qv2x_bottleneck_desc;
int qv2x_bottleneck_i8(const qv2x_bottleneck_desc* desc /* host */, const int8_t* x, const int8_t* const* w, const float* const* scale,
                       const int32_t* const* corr, const int32_t* const* aw, const float* const* bias, int8_t* out, void* stream);

/* qv2x_conv3x3_i8 with a fused residual end (conv2 of QuantBasicBlock, quant_block.py:76-96): one input group, cout = 64;
 *   res_mode 2: res f32 [N*Ho*Wo][cout];  res_mode 3: res = padded i8 BEV [N][Ho+2][Wo+2][cout] with (res_zx, res_delta);
 *   out = quant(max(y + shortcut, 0)) with the BLOCK's quantizer (desc->out_delta / out_zp). */
int qv2x_conv3x3_i8_res(const qv2x_conv_desc* desc /* host */, const int8_t* in, const int8_t* w, const float* scale, const int32_t* corr,
                        const int32_t* aw, const float* bias, int res_mode, const void* res, int res_zx, float res_delta,
                        int8_t* out, void* stream);

/* qv2x_deconv_i8 on an fp32 map (QuantPyramidFusion's deblocks take the fused levels, quant_block.py:444-459; with s = 1 and the
 * weight transposed it is also conv1 of the first pyramid block, whose input is the decoded feature):  in f32 [N*H*W][cin]; the
 * descriptor's in_zx / in_delta are ignored; everything else as qv2x_deconv_i8. */
int qv2x_deconv_f32in(const qv2x_deconv_desc* desc /* host */, const float* in, const float* w, const float* bias, int8_t* out, void* stream);

/* UMGMQuantizer.decode (codebook.py:339-343) as table look-ups for any codebook width d (% 4): row r = (agent, cell),
 *   out[r][:] = ((bias + lut[0][c0]) + lut[1][c1]) + ...,  c_l = codes[agent * agent_stride + l * level_stride + cell]
 *   lut f32 [levels][kc][d], bias f32 [d], out f32 [agents * hw][d]. */
int qv2x_codebook_decode_f32(const uint8_t* codes, int64_t agent_stride, int64_t level_stride, int agents, int hw, int levels,
                             int kc, int d, const float* lut, const float* bias, float* out, void* stream);

/* qv2x_codebook_encode_f32 for the Pyramid model's 64-wide codebook (heter_pyramid_collab_codebook_mc.py:25-51): D = 64, the first 64
 * channels of a padded i8 BEV with cin_total channels.  One f32 blob per level:
 *       stage [16][64][4] | stage_b [64] | qhead [16][64][4] | qhead_b [64] | lhead [16][64][4] | lhead_b [64]
 *       | cb_packed [16][Kc][4] | cb [Kc][64] | c2 [Kc]          (qv2x_codebook64_level_floats(Kc) floats; layouts as the D = 256 entry)
 * c2 = |C_k|^2 as ONE 64-wide ascending fma chain (qv2x_codebook64_c2_f32). */
int64_t qv2x_codebook64_level_floats(int kc);
int qv2x_codebook64_c2_f32(const float* codebook, int kc, float* c2, void* stream);
int qv2x_codebook_encode64_f32(const qv2x_encode_desc* desc /* host */, int cin_total, const int8_t* in,
                               const float* const* level_weights /* host array of device pointers */, uint8_t* codes, void* stream);

/* The same from fp32 rows (the un-quantized Pyramid model): in f32, padded map [N][H+2][W+2][cin_total], first 64 channels. */
int qv2x_codebook_encode64_f32in(const qv2x_encode_desc* desc /* host */, int cin_total, const float* in,
                                 const float* const* level_weights /* host array of device pointers */, uint8_t* codes, void* stream);

/* The un-quantized Pyramid model (plain opencood/tools/inference.py flow) reuses qv2x_conv3x3_f32 / qv2x_deconv_f32 (1x1 = deconv with
 * stride 1; a grouped 3x3 = one dense launch per 64-channel slab through the cin / out windows) and adds:
 *   qv2x_add_relu_f32     out = max(a + b, 0) over `count` floats (the end of a residual block, resblock.py:58-66, :118-128)
 *   qv2x_occ_sigmoid_f32  channel 0 of a padded fp32 map [n][h+2][w+2][c_total] -> occ f32 [n*h*w], score = sigmoid(occ) + 1e-4
 *   qv2x_pyramid_weighted_fuse_f32p   qv2x_pyramid_weighted_fuse_f32 with features AND output as padded maps [..][h+2][w+2][channels] */
int qv2x_add_relu_f32(const float* a, const float* b, float* out, int64_t count, void* stream);
int qv2x_occ_sigmoid_f32(const float* in, int n, int h, int w, int c_total, float* occ, float* score, void* stream);
int qv2x_pyramid_weighted_fuse_f32p(const qv2x_fuse_desc* desc /* host */, int channels, const float* feats, const float* score,
                                    const double* pairwise, float* out, void* stream);

/* One level's 1x1 occupancy head (QuantModule c -> 1 with its output quantizer, quant_block.py:475-479, :507-509):
 *     T exact;  y = bias + float(T) * scale;  code = quant(y);  score = score_lut[code]
 *   (score_lut[k] = sigmoid((k - out_zp) * out_delta) + 1e-4, built once by the caller);  in padded i8 BEV [N][H+2][W+2][c],
 *   w i8 (code - 128) [c];  aw = 128 - zw, corr = ax * sum ws + c * ax * aw;  score f32 [N*H*W]; occ_code u8 [N*H*W] or NULL. */
typedef struct {
    int32_t n, h, w, c;
    int32_t aw, corr;
    float scale, bias, out_delta, out_zp;
} qv2x_occ_desc;
int qv2x_occ_score_i8(const qv2x_occ_desc* desc /* host */, const int8_t* in, const int8_t* w, const float* score_lut, float* score,
                      uint8_t* occ_code, void* stream);

/* qv2x_pyramid_weighted_fuse_f32 with the features as activation codes: feats padded i8 BEV [agents][h+2][w+2][channels] with
 * (in_zx, in_delta), dequantized at the taps. */
int qv2x_pyramid_weighted_fuse_i8(const qv2x_fuse_desc* desc /* host */, int channels, const int8_t* feats, int in_zx, float in_delta,
                                  const float* score, const double* pairwise, float* out, void* stream);

/* ---- a13: the quantized SECOND encoder (secondary) -----------------------------------------------------
 * SECOND.forward (opencood/models/heter_encoders.py:66-81) = MeanVFE (sub_modules/mean_vfe.py:13-32) -> VoxelBackBone8x
 * (sub_modules/sparse_backbone_3d.py:48-153: twelve SubMConv3d / SparseConv3d + BatchNorm1d + ReLU) -> HeightCompression
 * (sub_modules/height_compression.py:12-27), every convolution under QuantSpconvModule.forward (quant/quant_layer.py:460-490).
 * The reference evaluates the convolutions with spconv, a pip wheel that is NOT part of the reference tree: these entry points
 * restate its published semantics (oracle/spec_second.py; parity against spconv itself is unpinned).
 *
 * A sparse level on the device: coords i32 [cap][4] = (agent, z, y, x); features i8 [cap + 1][C] of (code - 128), C a multiple of 32,
 * row `cap` = the fill row (zp - 128: the real value 0); the number of rows is a DEVICE int32 (no host round trip anywhere);
 * a dense index volume i32 [agents][D][H][W] (-1 = no site).  A convolution is a rulebook nbr i32 [K][cap_out] (input row or the fill
 * row `cap_in` per output site and window offset, z-major) followed by a gather-GEMM. */
typedef struct {
    int32_t subm;                         /* 1: SubMConv3d (outputs = the input's sites), 0: SparseConv3d */
    int32_t k[3], s[3], p[3];             /* window, stride, padding along (z, y, x) */
    int32_t in_shape[3], out_shape[3];    /* (D, H, W) */
    int32_t agents, cin, cout;            /* cin / cout: PADDED channel counts of the i8 rows (the f32-in layer: cin = 4) */
    int32_t cap_in, cap_out;              /* row capacities (the fill rows sit at index cap) */
    float out_delta, out_zp;              /* the layer's output quantizer */
} qv2x_spconv_desc;
/* MeanVFE: voxel_features f32 [cap][max_points][4] (unused slots zero) -> out f32 [cap][4], summed slot by slot. */
int qv2x_mean_vfe_f32(const float* voxel_features, const int32_t* voxel_num_points, const int32_t* n_voxels /* device */, int cap,
                      int max_points, float* out, void* stream);
/* set != 0: volume[coords[row]] = row for row < *n_rows;  set == 0: back to -1 (how a frame leaves the volume clean). */
int qv2x_sp_index_scatter(const int32_t* coords, const int32_t* n_rows /* device */, int cap, int agents, int D, int H, int W,
                          int32_t* volume, int set, void* stream);
/* SparseConv3d's active outputs: every output position whose window holds an active input.  out_volume must be all -1 on entry and
 * holds the new rows on return; out_coords / *n_out are written, rows in raster order of (agent, z, y, x) (spconv's own order is hash
 * order; nothing downstream depends on it).  workspace: qv2x_sp_out_sites_workspace_bytes(desc) bytes of device memory. */
int64_t qv2x_sp_out_sites_workspace_bytes(const qv2x_spconv_desc* desc /* host */);
int qv2x_sp_out_sites(const qv2x_spconv_desc* desc /* host */, const int32_t* in_coords, const int32_t* n_in, int32_t* out_volume,
                      int32_t* out_coords, int32_t* n_out, void* workspace, int64_t workspace_bytes, void* stream);
int qv2x_sp_rulebook(const qv2x_spconv_desc* desc /* host */, const int32_t* out_coords, const int32_t* n_out, const int32_t* in_volume,
                     int32_t* nbr, void* stream);
/* conv_input (fp32 means in): w f32 [K][4][16] = (code - zp_w) * delta_w; y = acc * bn_g + bn_h -> ReLU -> quantize; out i8 [cap_out+1][32]. */
int qv2x_sp_conv_f32in(const qv2x_spconv_desc* desc /* host */, const float* feat, const int32_t* nbr, const int32_t* n_out, const float* w,
                       const float* bn_g, const float* bn_h, int8_t* out, void* stream);
/* the eleven integer layers: w_frag i8 [K][cin/32][cout/32][64][16] (lane l: output channel l % 32 of the tile, input bytes
 * 16 * (l / 32) .. + 15 of the 32-channel step; values code - 128), scale = delta_w * delta_x, aw = 128 - zp_w,
 * corr = (128 - zp_x) * sum ws + K * cin * (128 - zp_x) * aw;  y = (T * scale) * bn_g + bn_h -> ReLU -> quantize. */
int qv2x_sp_conv_i8(const qv2x_spconv_desc* desc /* host */, const int8_t* in, const int32_t* nbr, const int32_t* n_out, const int8_t* w_frag,
                    const float* scale, const int32_t* corr, const int32_t* aw, const float* bn_g, const float* bn_h, int8_t* out, void* stream);
/* HeightCompression into the padded i8 BEV layout of the 2-D path: bev i8 [agents][H+2][W+2][c*D], channel = ch * D + z, every cell
 * without a site (and the border) = fill. */
int qv2x_sp_to_bev_i8(const int8_t* feat, const int32_t* coords, const int32_t* n_rows /* device */, int cap, int c, int c_padded, int agents,
                      int D, int H, int W, int fill, int8_t* bev, void* stream);

/* ---- the V2X link: one agent per GPU --------------------------------------------------------------------
 * The reference simulates the link in-process: all agents are rows of one batch (heter_model_baseline.py:216) and
 * fusion_in_one.py:131-151 regroups them; get_pairwise_transformation (utils/transformation_utils.py:21-66) builds the
 * pairwise matrix from every agent's pose on the host.  Here each rank sends ONE fixed-size payload
 *     [ uint8 code planes, levels x frames x H*W ][ float64 world pose, 4 x 4 row-major = 128 B ]
 * and receives all of them (all-gather).  float64 because the reference's pose path is float64 end to end.
 *
 * qv2x_comm_*: an opt-in RCCL communicator owned by the caller -- the library's only state.  RCCL is bound at run time (dlopen),
 * libqv2x.so does not link against it.  qv2x_comm_unique_id on one rank, distribute the 128 bytes by any means, then
 * qv2x_comm_init on every rank (collective).  quantv2x_amd/dist.py can use torch.distributed's communicator instead. */
#define QV2X_COMM_ID_BYTES 128
int qv2x_comm_unique_id(void* id /* host, QV2X_COMM_ID_BYTES */);
int qv2x_comm_init(const void* id /* host */, int world, int rank, void** comm);
int qv2x_comm_destroy(void* comm);
/* ncclAllGather of bytes_per_rank bytes: recv = [world][bytes_per_rank], rank-major = agent-major.  Enqueued on `stream`. */
int qv2x_allgather_codes(void* comm, const uint8_t* send, uint8_t* recv, int64_t bytes_per_rank, void* stream);

/* The reference's pairwise matrix from the gathered poses: pairwise[i][j] = T_j^-1 T_i = solve(T_j, T_i) for i, j < world,
 * identity elsewhere (get_pairwise_transformation, transformation_utils.py:21-66; np.linalg.solve there, a fixed-order float64
 * elimination with partial pivoting here: oracle/geometry.py:solve4 is the same arithmetic, equal to LAPACK's to a few ulp).
 *   gathered: [world][agent_stride_bytes], the pose of agent a at a * agent_stride_bytes + pose_offset_bytes
 *   pairwise: f64 [max_cav][max_cav][4][4] (what qv2x_fuse_att_f32 takes) */
int qv2x_pairwise_from_poses_f64(const uint8_t* gathered, int world, int64_t agent_stride_bytes, int64_t pose_offset_bytes,
                                 int max_cav, double* pairwise, void* stream);
/* The same for `frames` scenes in one launch: frame f reads every agent's pose `frame_stride_bytes` * f further inside its payload and
 * writes pairwise + f * max_cav * max_cav * 16. */
int qv2x_pairwise_from_poses_batch_f64(const uint8_t* gathered, int world, int64_t agent_stride_bytes, int64_t pose_offset_bytes,
                                       int frames, int64_t frame_stride_bytes, int max_cav, double* pairwise, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* QV2X_H */
