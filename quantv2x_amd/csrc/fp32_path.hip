// The UN-QUANTIZED model on the same C ABI (SURVEY.md §8(b) "fp32 fall-backs for un-quantized mode"): what the reference's plain
// opencood/tools/inference.py:106-170 flow runs -- PillarVFE + scatter, BaseBEVBackbone, DownsampleConv in fp32 -- as HIP kernels
// on v_mfma_f32_32x32x2_f32.  (The codebook encode, the decode + warp + attention kernel and the heads are fp32 already and are
// shared with the W8A8 path.)  BatchNorm is folded into the preceding convolution by the host (fold_bn.py's algebra).
//
// Activations: fp32 NHWC with a one-pixel ZERO border, [N][H+2][W+2][C] -- the i8 BEV layout with 4-byte elements.
//
// One GEMM kernel serves the 3x3 convolutions (rows = output pixels, K = 9 taps x Cin, gathered from the padded input) and the
// k == s transposed convolutions (rows = input pixels, columns = (i*s + j)*Cout + co, K = Cin).  Every output is ONE fp32 fma
// chain in a fixed order, so the CPU oracle (oracle/qv2x_oracle.c: orc_gemm_f32) reproduces it bit for bit:
//     K is walked in groups of 8 consecutive k (k = tap * Cin + ci); inside a group the order is k0, k4, k1, k5, k2, k6, k3, k7
//     (lanes 0-31 of the wave hold k0..k3 of the group as one float4, lanes 32-63 hold k4..k7; MFMA step t multiplies element t of
//      both halves, lower half first);  acc starts at 0;  y = acc + bias;  ReLU.
#include "common.h"

namespace qv2x {
namespace {

struct GemmArgs {
    const float* in; const float* w; const float* bias; float* out;
    int n, h, w_, cin_total, cin0, cin, taps, stride, cout, ncols, relu;
    int ho, wo, M, out_ctotal, out_c0, deconv;
};

// wave tile: 32 rows x 64 columns (two accumulators share the A fragment)
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs a) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, par = lane >> 5;
    const int tiles_n = a.ncols / 64;
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int tiles_m = (a.M + 31) >> 5;
    if (tile >= tiles_m * tiles_n) return;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;

    // this lane's row (pixel): input base and output pixel index
    int m = tm * 32 + l31;
    const bool live = m < a.M;
    m = live ? m : a.M - 1;
    const int per_img = a.deconv ? a.h * a.w_ : a.ho * a.wo;
    const int img = m / per_img, rem = m - img * per_img;
    const int rw = a.deconv ? a.w_ : a.wo;
    const int ry = rem / rw, rx = rem - ry * rw;
    const float* src;
    int opix;                                                           // padded output pixel index of this row (sub-position 0, 0)
    if (a.deconv) {
        src = a.in + ((size_t)(img * (a.h + 2) + ry + 1) * (a.w_ + 2) + rx + 1) * a.cin_total + a.cin0;
        opix = (img * (a.h * a.stride + 2) + ry * a.stride + 1) * (a.w_ * a.stride + 2) + rx * a.stride + 1;
    } else {
        src = a.in + ((size_t)(img * (a.h + 2) + ry * a.stride) * (a.w_ + 2) + rx * a.stride) * a.cin_total + a.cin0;
        opix = (img * (a.ho + 2) + ry + 1) * (a.wo + 2) + rx + 1;
    }
    opix = live ? opix : -1;

    v16f acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // weights: [K / 8][ncols][2][4]; this lane: column tn * 64 + t * 32 + l31, half par
    const v4f* wq = (const v4f*)a.w + ((size_t)(tn * 64 + l31) * 2 + par);
    const int groups = a.cin / 8;
    int kg = 0;
    for (int tap = 0; tap < a.taps; ++tap) {
        const float* sp = src + (size_t)((tap / 3) * (a.w_ + 2) + (tap % 3)) * a.cin_total + par * 4;
        for (int g = 0; g < groups; ++g, ++kg) {
            const v4f av = *(const v4f*)(sp + g * 8);
            const v4f b0 = wq[(size_t)kg * a.ncols * 2], b1 = wq[((size_t)kg * a.ncols + 32) * 2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b0[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b1[e], acc[1], 0, 0, 0);
            }
        }
    }

    // epilogue: C-fragment row r of this lane is pixel tm * 32 + mfma32_row(r, lane); 32 lanes = 32 consecutive columns
    const int orow = a.w_ * a.stride + 2;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = tn * 64 + t * 32 + l31;
        int co = col, sub = 0;
        if (a.deconv) {
            const int ij = col / a.cout;
            co = col - ij * a.cout;
            const int di = ij / a.stride, dj = ij - di * a.stride;
            sub = di * orow + dj;
        }
        const float bias = a.bias[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int op = __shfl(opix, mfma32_row(r, lane));
            float y = acc[t][r] + bias;
            if (a.relu) y = fmaxf(y, 0.0f);
            if (op >= 0) a.out[(size_t)(op + sub) * a.out_ctotal + a.out_c0 + co] = y;
        }
    }
}

struct PfnF32 { float w[640]; float b[64]; float vox[3], off[3]; };

// a1 + a2 in fp32: the kernel of pfn_scatter.hip without the two quantizers (Linear with folded BN, ReLU, max over the points;
// a zero-masked slot contributes relu(bias)), written as floats into the zero-filled fp32 canvas
__global__ __launch_bounds__(256) void pfn_scatter_f32_kernel(const float4* __restrict__ vf, const int4* __restrict__ coords, const int* __restrict__ npts,
                                                              int M, int P, const PfnF32 prm, float* __restrict__ canvas, int N, int ny, int nx) {
    const int lane = threadIdx.x & 63;
    int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    m = __builtin_amdgcn_readfirstlane(m);
    if (m >= M) return;
    const float4* pts = vf + (size_t)m * P;
    const int4 c = coords[m];
    const int np = npts[m];
    const int real = np < P ? np : P;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int p = 0; p < real; ++p) { const float4 q = pts[p]; sx += q.x; sy += q.y; sz += q.z; }
    const float n = (float)np;
    const float mx = sx / n, my = sy / n, mz = sz / n;
    const float cx = (float)c.w * prm.vox[0] + prm.off[0], cy = (float)c.z * prm.vox[1] + prm.off[1], cz = (float)c.y * prm.vox[2] + prm.off[2];
    float w[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) w[k] = prm.w[lane * 10 + k];
    const float b = prm.b[lane];
    float ymax = -INFINITY;
    for (int p = 0; p < real; ++p) {
        const float4 q = pts[p];
        const float f[10] = {q.x, q.y, q.z, q.w, q.x - mx, q.y - my, q.z - mz, q.x - cx, q.y - cy, q.z - cz};
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 10; ++k) acc = fmaf(f[k], w[k], acc);
        ymax = fmaxf(ymax, acc + b);
    }
    if (real < P) ymax = fmaxf(ymax, b);
    ymax = fmaxf(ymax, 0.0f);
    if (c.x < 0 || c.x >= N || c.z < 0 || c.z >= ny || (c.y + c.w) < 0 || (c.y + c.w) >= nx) return;
    const size_t cell = ((size_t)c.x * (ny + 2) + (c.z + 1)) * (nx + 2) + (size_t)(c.y + c.w + 1);
    canvas[cell * 64 + lane] = ymax;
}

int gemm_launch(const qv2x_f32conv_desc* d, const float* in, const float* w, const float* bias, float* out, void* stream, int deconv, const char* who) {
    if (!d || !in || !w || !bias || !out) return fail(QV2X_EINVAL, "%s: null pointer", who);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->cin % 8 || d->cin0 % 4 || d->cin_total % 4 || d->cin0 + d->cin > d->cin_total)
        return fail(QV2X_EALIGN, "%s: input channels in multiples of 8 (window %d + %d of %d)", who, d->cin0, d->cin, d->cin_total);
    if (((uintptr_t)in & 15) || ((uintptr_t)w & 15)) return fail(QV2X_EALIGN, "%s: in / w must be 16-byte aligned", who);
    GemmArgs a;
    a.in = in; a.w = w; a.bias = bias; a.out = out;
    a.n = d->n; a.h = d->h; a.w_ = d->w; a.cin_total = d->cin_total; a.cin0 = d->cin0; a.cin = d->cin; a.cout = d->cout; a.relu = d->relu;
    a.out_ctotal = d->out_ctotal; a.out_c0 = d->out_c0; a.deconv = deconv; a.stride = d->stride;
    if (deconv) {
        if (d->stride < 1 || d->stride > 8) return fail(QV2X_EINVAL, "%s: stride 1..8", who);
        a.taps = 1; a.ncols = d->stride * d->stride * d->cout; a.ho = d->h * d->stride; a.wo = d->w * d->stride; a.M = d->n * d->h * d->w;
    } else {
        if (d->stride != 1 && d->stride != 2) return fail(QV2X_EINVAL, "%s: stride 1 or 2", who);
        a.taps = 9; a.ncols = d->cout; a.ho = (d->h + 2 - 3) / d->stride + 1; a.wo = (d->w + 2 - 3) / d->stride + 1; a.M = d->n * a.ho * a.wo;
    }
    if (a.ncols % 64 || d->cout <= 0 || d->out_c0 < 0 || d->out_ctotal < d->out_c0 + d->cout) return fail(QV2X_EALIGN, "%s: columns in multiples of 64, output channel window", who);
    if ((long long)a.M * 1 <= 0) return fail(QV2X_EINVAL, "%s: empty output", who);
    const int tiles = ((a.M + 31) / 32) * (a.ncols / 64);
    gemm_f32_kernel<<<(tiles + 3) / 4, 256, 0, (hipStream_t)stream>>>(a);
    return hip_check(hipGetLastError(), who);
}

// the end of a residual block in fp32 (resblock.py:58-66, :118-128): out = relu(a + b), elementwise over whole (padded) maps
__global__ __launch_bounds__(256) void add_relu_f32_kernel(const v4f* __restrict__ a, const v4f* __restrict__ b, v4f* __restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const v4f x = a[i], y = b[i];
    v4f o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(x[e] + y[e], 0.0f);
    out[i] = o;
}

// channel 0 of a padded fp32 map [n][h+2][w+2][ct] -> occupancy [n*h*w] and score = sigmoid(occupancy) + 1e-4 (pyramid_fuse.py:150-152)
__global__ __launch_bounds__(256) void occ_sigmoid_f32_kernel(const float* __restrict__ in, int n, int h, int w, int ct, float* __restrict__ occ,
                                                              float* __restrict__ score) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n * h * w) return;
    const int img = m / (h * w), rem = m - img * (h * w), y = rem / w, x = rem - y * w;
    const float v = in[((size_t)(img * (h + 2) + y + 1) * (w + 2) + x + 1) * ct];
    occ[m] = v;
    score[m] = 1.0f / (1.0f + expf(-v)) + 1e-4f;
}

}  // namespace
}  // namespace qv2x

extern "C" int qv2x_add_relu_f32(const float* a, const float* b, float* out, int64_t count, void* stream) {
    using namespace qv2x;
    if (!a || !b || !out) return fail(QV2X_EINVAL, "qv2x_add_relu_f32: null pointer");
    if (count <= 0 || count % 4 || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)out & 15)) return fail(QV2X_EALIGN, "qv2x_add_relu_f32: count %% 4, 16-byte aligned pointers");
    add_relu_f32_kernel<<<(unsigned)((count / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>((const v4f*)a, (const v4f*)b, (v4f*)out, count / 4);
    return hip_check(hipGetLastError(), "qv2x_add_relu_f32 launch");
}

extern "C" int qv2x_occ_sigmoid_f32(const float* in, int n, int h, int w, int c_total, float* occ, float* score, void* stream) {
    using namespace qv2x;
    if (!in || !occ || !score) return fail(QV2X_EINVAL, "qv2x_occ_sigmoid_f32: null pointer");
    if (n <= 0 || h <= 0 || w <= 0 || c_total <= 0) return fail(QV2X_EINVAL, "qv2x_occ_sigmoid_f32: bad shape");
    occ_sigmoid_f32_kernel<<<(n * h * w + 255) / 256, 256, 0, (hipStream_t)stream>>>(in, n, h, w, c_total, occ, score);
    return hip_check(hipGetLastError(), "qv2x_occ_sigmoid_f32 launch");
}

extern "C" int qv2x_conv3x3_f32(const qv2x_f32conv_desc* d, const float* in, const float* w, const float* bias, float* out, void* stream) {
    return qv2x::gemm_launch(d, in, w, bias, out, stream, 0, "qv2x_conv3x3_f32");
}

extern "C" int qv2x_deconv_f32(const qv2x_f32conv_desc* d, const float* in, const float* w, const float* bias, float* out, void* stream) {
    return qv2x::gemm_launch(d, in, w, bias, out, stream, 1, "qv2x_deconv_f32");
}

extern "C" int qv2x_pfn_scatter_f32(const float* voxel_features, const int32_t* voxel_coords, const int32_t* voxel_num_points, int M, int max_points,
                                    const float* w /* host [64][10] */, const float* b /* host [64] */, const float* vox /* host [3] */,
                                    const float* off /* host [3] */, float* canvas, int N, int ny, int nx, void* stream) {
    using namespace qv2x;
    if (M == 0) return QV2X_OK;
    if (!voxel_features || !voxel_coords || !voxel_num_points || !w || !b || !vox || !off || !canvas) return fail(QV2X_EINVAL, "qv2x_pfn_scatter_f32: null pointer");
    if (M < 0 || max_points <= 0 || N <= 0 || ny <= 0 || nx <= 0) return fail(QV2X_EINVAL, "qv2x_pfn_scatter_f32: bad sizes");
    if (((uintptr_t)voxel_features & 15) || ((uintptr_t)voxel_coords & 15)) return fail(QV2X_EALIGN, "qv2x_pfn_scatter_f32: inputs must be 16-byte aligned");
    PfnF32 p;
    for (int i = 0; i < 640; ++i) p.w[i] = w[i];
    for (int i = 0; i < 64; ++i) p.b[i] = b[i];
    for (int i = 0; i < 3; ++i) { p.vox[i] = vox[i]; p.off[i] = off[i]; }
    pfn_scatter_f32_kernel<<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>((const float4*)voxel_features, (const int4*)voxel_coords, voxel_num_points, M,
                                                                         max_points, p, canvas, N, ny, nx);
    return hip_check(hipGetLastError(), "qv2x_pfn_scatter_f32 launch");
}
