// f2: detection-head maps -> 3-D boxes on the GPU for the anchor heads; one CAV per call (intermediate /
// early fusion) or several (late fusion: every CAV's candidates decoded and projected with ITS matrix, concatenated in CAV order,
// then ONE filter / sort / NMS over the union -- the loop over `cav_content` of the reference's post_process):
// VoxelPostprocessor.post_process (opencood/data_utils/post_processor/voxel_postprocessor.py:245-405; line numbers below)
// and, with num_classes > 1, VoxelPostprocessor3Heads.post_process (voxel_postprocessor_3heads.py:318-478: score = largest
// class probability + label, no direction fix, the limits of box_utils_mc.py, x-y range mask).
//   1. score = sigmoid(cls) per anchor, flag = score > threshold                                    (:289-304)
//   2. exclusive scan of the flags: candidates keep the reference's (h, w, anchor) order -- no atomics decide an order
//   3. per candidate: delta_to_boxes3d (:408-453), direction-bin fix (:316-331), 8 corners (box_utils.py:152-204),
//      projection by the CAV -> ego matrix (:278-316), remove_large_pred_bbx / remove_bbx_abnormal_z (:916-966; the
//      former's "z extent" is the y extent tested for != 0 -- kept)
//   4. stable radix sort by score, descending; top-k (1000 in the reference)
//   5. rotated NMS on the bottom-face quadrilaterals: 1 bit per pair (convex clipping in fp64), then one greedy sweep
//      over the bit matrix in LDS                                                                    (:769-814)
//   6. range mask on all eight corners (box_utils.py:384-421), outputs compacted in score order.
// The reference's polygon IoU is shapely's (un-vendored): oracle/postprocess.py restates it with the same clipping as
// here; everything else is pinned against the reference by tests/golden/postprocess.npz and postprocess_mc.npz.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace qv2x {

namespace {

constexpr int TOPK_MAX = 1024;                       // bit-matrix row = 16 x u64

constexpr int MAX_CAVS = 8;

struct PPArgs {
    const float* cls[MAX_CAVS]; const float* reg[MAX_CAVS]; const float* dir[MAX_CAVS]; const float* anchors[MAX_CAVS];
    int h, w, a, na, na1, num_bins, topk, ncls, xy_only;          // na1 anchors per CAV, na = ncav * na1 (index = cav * na1 + local)
    float thr, nms_thr, dir_offset, max_extent, z_lo, z_hi;
    float range[6], t[MAX_CAVS][16];
    // workspace
    float* prob; int* flag; int* pos; float* cand_corners; float* cand_score; unsigned* key_in; unsigned* key_out;
    int* idx_in; int* idx_out; unsigned long long* mask;
    int* label; int* cand_label;
    float* out_corners; float* out_scores; int* out_labels; int* out_count;
};

__global__ void pp_score_kernel(const PPArgs p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // anchor index in (cav, h, w, a) order
    if (i >= p.na) return;
    const int cav = i / p.na1, li = i - cav * p.na1;
    const int a = li % p.a, cell = li / p.a;
    const size_t hw = (size_t)p.h * p.w;
    const float* cls = p.cls[cav];
    // one score class: the anchor's logit; several (voxel_postprocessor_3heads.py:362-373): channel a * ncls + k, the score is
    // the largest class probability (first one on ties), the label its index + 1
    float s = 1.0f / (1.0f + expf(-cls[(size_t)(a * p.ncls) * hw + cell]));
    int lab = 1;
    for (int k = 1; k < p.ncls; ++k) {
        const float v = 1.0f / (1.0f + expf(-cls[(size_t)(a * p.ncls + k) * hw + cell]));
        if (v > s) { s = v; lab = k + 1; }
    }
    p.label[i] = lab;
    p.prob[i] = s;
    p.flag[i] = s > p.thr ? 1 : 0;
    p.key_in[i] = 0u;                                              // slots past the candidates sort last
    p.idx_in[i] = i;                                               // candidate c is named by idx_in[c] == c
}

__device__ __forceinline__ float limit_period(float v, float offset, float period) { return v - floorf(v / period + offset) * period; }

__global__ void pp_decode_kernel(const PPArgs p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.na) return;
    if (!p.flag[i]) return;
    const int c = p.pos[i];                                        // candidate number, reference order
    const int cav = i / p.na1, li = i - cav * p.na1;
    const int a = li % p.a, cell = li / p.a;
    const size_t hw = (size_t)p.h * p.w;
    const float* reg = p.reg[cav];
    const float* dirm = p.dir[cav];
    const float* anchors = p.anchors[cav];
    const float* T = p.t[cav];
    float d[7], an[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) { d[k] = reg[(size_t)(a * 7 + k) * hw + cell]; an[k] = anchors[(size_t)li * 7 + k]; }
    const float diag = sqrtf(an[4] * an[4] + an[5] * an[5]);
    float bx = d[0] * diag + an[0], by = d[1] * diag + an[1], bz = d[2] * an[3] + an[2];
    const float bh = expf(d[3]) * an[3], bw = expf(d[4]) * an[4], bl = expf(d[5]) * an[5];
    float yaw = d[6] + an[6];
    if (dirm) {
        int label = 0;
        float best = dirm[(size_t)(a * p.num_bins) * hw + cell];
        for (int b = 1; b < p.num_bins; ++b) {
            const float v = dirm[(size_t)(a * p.num_bins + b) * hw + cell];
            if (v > best) { best = v; label = b; }                 // first maximum, as torch.max
        }
        const float period = (float)(2.0 * 3.141592653589793 / p.num_bins);
        const float rot = limit_period(yaw - p.dir_offset, 0.0f, period);
        yaw = rot + p.dir_offset + period * (float)label;
        yaw = limit_period(yaw, 0.5f, (float)(2.0 * 3.141592653589793));
    }
    // corners of the 'hwl' box: dims (l, w, h) along (x, y, z), rotated about z, translated, projected
    const float cosa = cosf(yaw), sina = sinf(yaw);
    const float sx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sy[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
    float* out = p.cand_corners + (size_t)c * 24;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float cx = bl * (sx[k] / 2.0f), cy = bw * (sy[k] / 2.0f), cz = bh * (sz[k] / 2.0f);
        const float rx = cx * cosa + cy * (-sina) + bx;
        const float ry = cx * sina + cy * cosa + by;
        const float rz = cz + bz;
        const float px = T[0] * rx + T[1] * ry + T[2] * rz + T[3];
        const float py = T[4] * rx + T[5] * ry + T[6] * rz + T[7];
        const float pz = T[8] * rx + T[9] * ry + T[10] * rz + T[11];
        out[k * 3 + 0] = px; out[k * 3 + 1] = py; out[k * 3 + 2] = pz;
        xmin = fminf(xmin, px); xmax = fmaxf(xmax, px); ymin = fminf(ymin, py); ymax = fmaxf(ymax, py);
        zmin = fminf(zmin, pz); zmax = fmaxf(zmax, pz);
    }
    const float xl = xmax - xmin, yl = ymax - ymin;
    const bool keep = xl <= p.max_extent && yl <= p.max_extent && yl != 0.0f && zmin >= p.z_lo && zmax <= p.z_hi;
    const float s = p.prob[i];
    p.cand_score[c] = s;
    p.cand_label[c] = p.label[i];
    p.key_in[c] = keep ? __builtin_bit_cast(unsigned, s) : 0u;     // positive floats order like their bit patterns
}

// area of the intersection of two convex quadrilaterals, Sutherland-Hodgman in fp64 (oracle/postprocess.py)
__device__ double quad_inter_area(const double (&p)[4][2], const double (&q)[4][2]) {
    double qa = 0.0;
    for (int i = 0; i < 4; ++i) qa += q[i][0] * q[(i + 1) & 3][1] - q[(i + 1) & 3][0] * q[i][1];
    const bool rev = qa < 0.0;                                     // clip polygon walked counter-clockwise
    double cur[8][2], nxt[8][2];
    int n = 4;
    for (int i = 0; i < 4; ++i) { cur[i][0] = p[i][0]; cur[i][1] = p[i][1]; }
    for (int e = 0; e < 4 && n > 0; ++e) {
        const int i0 = rev ? 3 - e : e, i1 = rev ? (2 - e) & 3 : (e + 1) & 3;      // reversed polygon: q3 q2 q1 q0
        const double ax = q[i0][0], ay = q[i0][1], bx = q[i1][0], by = q[i1][1];
        int m = 0;
        for (int j = 0; j < n; ++j) {
            const double cx = cur[j][0], cy = cur[j][1], dx = cur[(j + 1) % n][0], dy = cur[(j + 1) % n][1];
            const double sc = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
            const double sd = (bx - ax) * (dy - ay) - (by - ay) * (dx - ax);
            if (sc >= 0.0) { nxt[m][0] = cx; nxt[m][1] = cy; ++m; }
            if ((sc >= 0.0) != (sd >= 0.0)) {
                const double t = sc / (sc - sd);
                nxt[m][0] = cx + t * (dx - cx); nxt[m][1] = cy + t * (dy - cy); ++m;
            }
        }
        n = m;
        for (int j = 0; j < n; ++j) { cur[j][0] = nxt[j][0]; cur[j][1] = nxt[j][1]; }
    }
    if (n < 3) return 0.0;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += cur[j][0] * cur[(j + 1) % n][1] - cur[(j + 1) % n][0] * cur[j][1];
    return fabs(0.5 * s);
}

__device__ __forceinline__ void load_quad(const PPArgs& p, int sorted_i, double (&q)[4][2]) {
    const float* c = p.cand_corners + (size_t)p.idx_out[sorted_i] * 24;
#pragma unroll
    for (int k = 0; k < 4; ++k) { q[k][0] = (double)c[k * 3]; q[k][1] = (double)c[k * 3 + 1]; }
}

__device__ __forceinline__ double quad_area(const double (&q)[4][2]) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += q[i][0] * q[(i + 1) & 3][1] - q[(i + 1) & 3][0] * q[i][1];
    return fabs(0.5 * s);
}

// row i of the suppression matrix: bit j (j > i) = IoU(i, j) > threshold.  One workgroup per row, 64 threads per word.
__global__ __launch_bounds__(256) void pp_iou_kernel(const PPArgs p) {
    const int i = blockIdx.x;
    const int n = p.topk;
    double qi[4][2];
    const bool vi = p.key_out[i] != 0u;
    load_quad(p, i, qi);
    const double ai = quad_area(qi);
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        bool sup = false;
        if (vi && j > i && j < n && p.key_out[j] != 0u) {
            double qj[4][2];
            load_quad(p, j, qj);
            double ix0 = qi[0][0], ix1 = qi[0][0], iy0 = qi[0][1], iy1 = qi[0][1], jx0 = qj[0][0], jx1 = qj[0][0], jy0 = qj[0][1], jy1 = qj[0][1];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                ix0 = fmin(ix0, qi[k][0]); ix1 = fmax(ix1, qi[k][0]); iy0 = fmin(iy0, qi[k][1]); iy1 = fmax(iy1, qi[k][1]);
                jx0 = fmin(jx0, qj[k][0]); jx1 = fmax(jx1, qj[k][0]); jy0 = fmin(jy0, qj[k][1]); jy1 = fmax(jy1, qj[k][1]);
            }
            const bool apart = ix1 < jx0 || jx1 < ix0 || iy1 < jy0 || jy1 < iy0;     // disjoint bounding boxes: intersection 0
            const double inter = apart ? 0.0 : quad_inter_area(qi, qj);
            const double uni = ai + quad_area(qj) - inter;
            sup = uni > 0.0 && inter / uni > (double)p.nms_thr;
        }
        const unsigned long long bits = __builtin_amdgcn_ballot_w64(sup);
        if ((threadIdx.x & 63) == 0 && j < ((n + 63) & ~63)) p.mask[(size_t)i * (TOPK_MAX / 64) + (j >> 6)] = bits;
    }
}

// greedy sweep in score order + range mask + compaction.  One workgroup; the bit matrix (<= 128 KB) sits in LDS.
__global__ __launch_bounds__(256) void pp_sweep_kernel(const PPArgs p) {
    __shared__ unsigned long long m[TOPK_MAX * (TOPK_MAX / 64)];
    __shared__ int picked[TOPK_MAX];
    __shared__ int npick;
    const int n = p.topk, words = (n + 63) >> 6;
    for (int t = threadIdx.x; t < n * (TOPK_MAX / 64); t += blockDim.x) m[t] = (t % (TOPK_MAX / 64)) < words ? p.mask[t] : 0ull;
    __syncthreads();
    if (threadIdx.x < 64) {                                        // one wave; lane w < 16 keeps word w of the removed set
        unsigned long long mine = 0ull;
        int np = 0;
        for (int i = 0; i < n; ++i) {
            if (p.key_out[i] == 0u) break;                          // sorted: nothing valid after the first empty slot
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, i >> 6);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), i >> 6);
            const unsigned long long r = ((unsigned long long)hi << 32) | lo;
            if (!((r >> (i & 63)) & 1ull)) {
                if (threadIdx.x == 0) picked[np] = i;
                ++np;
                if (threadIdx.x < TOPK_MAX / 64) mine |= m[i * (TOPK_MAX / 64) + threadIdx.x];
            }
        }
        if (threadIdx.x == 0) npick = np;
    }
    __syncthreads();
    // range mask on all eight corners, then compaction in pick order (serial prefix: <= 1000 entries)
    __shared__ int inside[TOPK_MAX];
    for (int t = threadIdx.x; t < npick; t += blockDim.x) {
        const float* c = p.cand_corners + (size_t)p.idx_out[picked[t]] * 24;
        bool in = true;
        for (int k = 0; k < 8; ++k)
            in = in && c[k * 3] >= p.range[0] && c[k * 3 + 1] >= p.range[1] && c[k * 3] <= p.range[3] && c[k * 3 + 1] <= p.range[4] &&
                 (p.xy_only || (c[k * 3 + 2] >= p.range[2] && c[k * 3 + 2] <= p.range[5]));
        inside[t] = in ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int o = 0;
        for (int t = 0; t < npick; ++t) { const int f = inside[t]; inside[t] = f ? o : -1; o += f; }
        *p.out_count = o;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < npick; t += blockDim.x) {
        const int o = inside[t];
        if (o < 0) continue;
        const int cand = p.idx_out[picked[t]];
        for (int k = 0; k < 24; ++k) p.out_corners[(size_t)o * 24 + k] = p.cand_corners[(size_t)cand * 24 + k];
        p.out_scores[o] = p.cand_score[cand];
        if (p.out_labels) p.out_labels[o] = p.cand_label[cand];
    }
}

struct Layout {
    size_t prob, flag, pos, corners, score, key_in, key_out, idx_in, idx_out, mask, label, cand_label, cub, cub_bytes, total;
};

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

Layout layout(int na) {
    Layout l{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = align256(o + bytes); return at; };
    l.prob = take((size_t)na * 4); l.flag = take((size_t)na * 4); l.pos = take((size_t)na * 4);
    l.corners = take((size_t)na * 96); l.score = take((size_t)na * 4);
    l.key_in = take((size_t)na * 4); l.key_out = take((size_t)na * 4); l.idx_in = take((size_t)na * 4); l.idx_out = take((size_t)na * 4);
    l.mask = take((size_t)TOPK_MAX * (TOPK_MAX / 64) * 8);
    l.label = take((size_t)na * 4); l.cand_label = take((size_t)na * 4);
    size_t a = 0, b = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, a, (int*)nullptr, (int*)nullptr, na);
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, b, (unsigned*)nullptr, (unsigned*)nullptr, (int*)nullptr, (int*)nullptr, na, 0, 32);
    l.cub_bytes = a > b ? a : b;
    l.cub = take(l.cub_bytes);
    l.total = o;
    return l;
}

int check_desc(const qv2x_postprocess_desc* d, const char* who) {
    if (!d) return fail(QV2X_EINVAL, "%s: null descriptor", who);
    if (d->h <= 0 || d->w <= 0 || d->anchors_per_cell <= 0 || d->anchors_per_cell > 16) return fail(QV2X_EINVAL, "%s: bad head map shape", who);
    if (d->num_classes < 1 || d->num_classes > 8) return fail(QV2X_EINVAL, "%s: num_classes in 1..8", who);
    if (!(d->max_extent > 0.0f) || !(d->z_max > d->z_min)) return fail(QV2X_EINVAL, "%s: max_extent > 0 and z_max > z_min", who);
    if (d->num_bins < 0 || d->num_bins > 8) return fail(QV2X_EINVAL, "%s: num_bins in 0..8", who);
    if (d->max_boxes < 1 || d->max_boxes > TOPK_MAX) return fail(QV2X_EINVAL, "%s: max_boxes in 1..%d", who, TOPK_MAX);
    if (!(d->score_threshold > 0.0f)) return fail(QV2X_EINVAL, "%s: score_threshold must be positive (scores are ordered by their bit patterns)", who);
    return QV2X_OK;
}

}  // namespace

}  // namespace qv2x

static int postprocess_run(const qv2x_postprocess_desc* d, int ncav, const float* const* cls, const float* const* reg, const float* const* dir,
                           const float* const* anchors, const float* transforms, void* workspace, int64_t workspace_bytes, float* out_corners,
                           float* out_scores, int32_t* out_labels, int32_t* out_count, void* stream, const char* who) {
    using namespace qv2x;
    if (int rc = check_desc(d, who)) return rc;
    if (ncav < 1 || ncav > MAX_CAVS) return fail(QV2X_EINVAL, "%s: 1..%d CAVs", who, MAX_CAVS);
    if (!cls || !reg || !anchors || !transforms || !workspace || !out_corners || !out_scores || !out_count) return fail(QV2X_EINVAL, "%s: null pointer", who);
    const int na1 = d->h * d->w * d->anchors_per_cell, na = na1 * ncav;
    const Layout l = layout(na);
    if (workspace_bytes < (int64_t)l.total) return fail(QV2X_EINVAL, "%s: workspace of %lld bytes, need %lld", who, (long long)workspace_bytes, (long long)l.total);
    if ((uintptr_t)workspace & 255) return fail(QV2X_EALIGN, "%s: workspace must be 256-byte aligned", who);
    char* ws = (char*)workspace;
    PPArgs p{};
    for (int c = 0; c < ncav; ++c) {
        if (!cls[c] || !reg[c] || !anchors[c]) return fail(QV2X_EINVAL, "%s: null map of CAV %d", who, c);
        if (d->num_bins > 0 && (!dir || !dir[c])) return fail(QV2X_EINVAL, "%s: num_bins > 0 needs the direction map", who);
        p.cls[c] = cls[c]; p.reg[c] = reg[c]; p.dir[c] = d->num_bins > 0 ? dir[c] : nullptr; p.anchors[c] = anchors[c];
        for (int i = 0; i < 16; ++i) p.t[c][i] = transforms[c * 16 + i];
    }
    p.h = d->h; p.w = d->w; p.a = d->anchors_per_cell; p.na = na; p.na1 = na1; p.num_bins = d->num_bins;
    p.topk = d->max_boxes < na ? d->max_boxes : na;
    p.thr = d->score_threshold; p.nms_thr = d->nms_threshold; p.dir_offset = d->dir_offset;
    p.ncls = d->num_classes; p.xy_only = d->range_xy_only; p.max_extent = d->max_extent; p.z_lo = d->z_min; p.z_hi = d->z_max;
    for (int i = 0; i < 6; ++i) p.range[i] = d->range[i];
    p.prob = (float*)(ws + l.prob); p.flag = (int*)(ws + l.flag); p.pos = (int*)(ws + l.pos);
    p.cand_corners = (float*)(ws + l.corners); p.cand_score = (float*)(ws + l.score);
    p.key_in = (unsigned*)(ws + l.key_in); p.key_out = (unsigned*)(ws + l.key_out);
    p.idx_in = (int*)(ws + l.idx_in); p.idx_out = (int*)(ws + l.idx_out); p.mask = (unsigned long long*)(ws + l.mask);
    p.label = (int*)(ws + l.label); p.cand_label = (int*)(ws + l.cand_label);
    p.out_corners = out_corners; p.out_scores = out_scores; p.out_labels = out_labels; p.out_count = out_count;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (na + 255) / 256;
    int rc;
    pp_score_kernel<<<blocks, 256, 0, st>>>(p);
    size_t tb = l.cub_bytes;
    if ((rc = hip_check(hipcub::DeviceScan::ExclusiveSum(ws + l.cub, tb, p.flag, p.pos, na, st), "postprocess scan"))) return rc;
    pp_decode_kernel<<<blocks, 256, 0, st>>>(p);
    tb = l.cub_bytes;
    if ((rc = hip_check(hipcub::DeviceRadixSort::SortPairsDescending(ws + l.cub, tb, p.key_in, p.key_out, p.idx_in, p.idx_out, na, 0, 32, st),
                        "postprocess sort"))) return rc;
    pp_iou_kernel<<<p.topk, 256, 0, st>>>(p);
    pp_sweep_kernel<<<1, 256, 0, st>>>(p);
    return hip_check(hipGetLastError(), who);
}

extern "C" int64_t qv2x_postprocess_workspace_bytes(const qv2x_postprocess_desc* d) {
    using namespace qv2x;
    if (check_desc(d, "qv2x_postprocess_workspace_bytes")) return -1;
    return (int64_t)layout(d->h * d->w * d->anchors_per_cell).total;
}

extern "C" int64_t qv2x_postprocess_late_workspace_bytes(const qv2x_postprocess_desc* d, int ncav) {
    using namespace qv2x;
    if (check_desc(d, "qv2x_postprocess_late_workspace_bytes") || ncav < 1 || ncav > MAX_CAVS) return -1;
    return (int64_t)layout(d->h * d->w * d->anchors_per_cell * ncav).total;
}

extern "C" int qv2x_postprocess_f32(const qv2x_postprocess_desc* d, const float* cls, const float* reg, const float* dir,
                                    const float* anchors, void* workspace, int64_t workspace_bytes, float* out_corners,
                                    float* out_scores, int32_t* out_labels, int32_t* out_count, void* stream) {
    if (!d) return qv2x::fail(QV2X_EINVAL, "qv2x_postprocess_f32: null descriptor");
    return postprocess_run(d, 1, &cls, &reg, &dir, &anchors, d->transform, workspace, workspace_bytes, out_corners, out_scores, out_labels,
                           out_count, stream, "qv2x_postprocess_f32");
}

extern "C" int qv2x_postprocess_late_f32(const qv2x_postprocess_desc* d, int ncav, const float* const* cls, const float* const* reg,
                                         const float* const* dir, const float* const* anchors, const float* transforms, void* workspace,
                                         int64_t workspace_bytes, float* out_corners, float* out_scores, int32_t* out_labels,
                                         int32_t* out_count, void* stream) {
    return postprocess_run(d, ncav, cls, reg, dir, anchors, transforms, workspace, workspace_bytes, out_corners, out_scores, out_labels,
                           out_count, stream, "qv2x_postprocess_late_f32");
}
