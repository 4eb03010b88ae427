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
// Step 4, round 5: the sweep only ever reads the first `topk` entries of the order, so the stable descending sort of ALL `na` slots (hipCUB
// fell back to a merge sort of 22 launches, 140 us per frame for 70 400 slots) became a radix SELECT of the top-k + one workgroup's bitonic sort:
// two 12-bit histogram passes over the score bits find the 24-bit prefix below which nothing can make the top-k; everything at or above it
// (top-k + what shares the last prefix: a handful) is compacted as 64-bit keys (score bits, ~slot) -- distinct, so their descending order IS
// the stable sort's -- and sorted in LDS.  More than SEL_CAP survivors means hundreds of (near-)EQUAL scores at the cut -- a quantized head
// has at most 256 different logits per class -- and the sort kernel finishes the selection itself: the low byte of the score, then the slot
// index (ties enter in slot order) by three more histogram passes of its one workgroup, exact.
constexpr int SEL_BINS = 4096, SEL_CAP = 2048, SEL_WORDS = 16;
enum { SEL_B1 = 0, SEL_ABOVE1, SEL_THR24, SEL_COUNT, SEL_ABOVE2 };

struct PPArgs {
    const float* cls[MAX_CAVS]; const float* reg[MAX_CAVS]; const float* dir[MAX_CAVS]; const float* anchors[MAX_CAVS];
    int h, w, a, na, na1, num_bins, topk, ncls, xy_only;          // na1 anchors per CAV, na = ncav * na1 (index = cav * na1 + local)
    float thr, nms_thr, dir_offset, max_extent, z_lo, z_hi;
    float range[6], t[MAX_CAVS][16];
    // workspace
    float* prob; int* flag; int* pos; float* cand_corners; float* cand_score; unsigned* key_in; unsigned* key_out;
    int* idx_in; int* idx_out; unsigned long long* mask;
    int* sel; unsigned long long* cand;                              // top-k selection: 2 x 4096 histogram bins + SEL_WORDS ints; the candidates' composite keys
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

__global__ void pp_sel_clear_kernel(const PPArgs p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * SEL_BINS + SEL_WORDS) p.sel[i] = 0;
}

// histogram of the 12 score bits of pass PASS (0: bits 31..20 of every non-empty slot; 1: bits 19..8 of the slots whose bits 31..20 = b1)
template <int PASS>
__global__ __launch_bounds__(256) void pp_sel_hist_kernel(const PPArgs p) {
    __shared__ int h[SEL_BINS];
    for (int t = threadIdx.x; t < SEL_BINS; t += 256) h[t] = 0;
    __syncthreads();
    const int b1 = PASS ? p.sel[2 * SEL_BINS + SEL_B1] : 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.na; i += gridDim.x * 256) {
        const unsigned k = p.key_in[i];
        if (k == 0u) continue;
        if (PASS == 0) atomicAdd(&h[k >> 20], 1);
        else if ((int)(k >> 20) == b1) atomicAdd(&h[(k >> 8) & 0xfffu], 1);
    }
    __syncthreads();
    int* g = p.sel + PASS * SEL_BINS;
    for (int t = threadIdx.x; t < SEL_BINS; t += 256)
        if (h[t]) atomicAdd(&g[t], h[t]);
}

// the bin in which the running count FROM THE TOP reaches `want` (the last non-empty bin when the slots hold fewer): one workgroup of 1024
template <int PASS>
__global__ __launch_bounds__(1024) void pp_sel_pick_kernel(const PPArgs p) {
    __shared__ int part[1024];
    __shared__ int found[2];
    const int* g = p.sel + PASS * SEL_BINS;
    int* sel = p.sel + 2 * SEL_BINS;
    const int want = PASS ? p.topk - sel[SEL_ABOVE1] : p.topk;
    const int t = threadIdx.x;                                     // thread t owns bins 4 t .. 4 t + 3; suffix sums from the top
    const int c0 = g[4 * t], c1 = g[4 * t + 1], c2 = g[4 * t + 2], c3 = g[4 * t + 3];
    part[t] = c0 + c1 + c2 + c3;
    if (t < 2) found[t] = t ? 0 : -1;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                     // inclusive suffix scan: part[t] = sum of the bins >= 4 t
        const int v = t + off < 1024 ? part[t + off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const int above_me = part[t] - (c0 + c1 + c2 + c3);           // count in the bins above this thread's four
    int above = above_me;
    const int cs[4] = {c3, c2, c1, c0};
#pragma unroll
    for (int e = 0; e < 4; ++e) {                                  // bins 4 t + 3 down to 4 t
        if (above < want && above + cs[e] >= want) { found[0] = 4 * t + 3 - e; found[1] = above; }
        above += cs[e];
    }
    __syncthreads();
    if (t == 0) {
        int b = found[0], ab = found[1];
        if (b < 0) { b = 0; ab = 0; }                              // fewer than `want` in all: everything non-empty survives
        if (PASS == 0) { sel[SEL_B1] = b; sel[SEL_ABOVE1] = ab; }
        else { sel[SEL_THR24] = (sel[SEL_B1] << 12) | b; sel[SEL_ABOVE2] = ab; }
    }
}

__global__ __launch_bounds__(256) void pp_sel_compact_kernel(const PPArgs p) {
    __shared__ int n_here, base;
    int* sel = p.sel + 2 * SEL_BINS;
    const unsigned thr24 = (unsigned)sel[SEL_THR24];
    if (threadIdx.x == 0) n_here = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned k = i < p.na ? p.key_in[i] : 0u;
    const bool take = k != 0u && (k >> 8) >= thr24;
    int at = 0;
    if (take) at = atomicAdd(&n_here, 1);                          // (the order inside the list is free: the keys are distinct)
    __syncthreads();
    if (threadIdx.x == 0 && n_here) base = atomicAdd(&sel[SEL_COUNT], n_here);   // one device-scope atomic per workgroup that holds a survivor
    __syncthreads();
    if (take && base + at < SEL_CAP) p.cand[base + at] = ((unsigned long long)k << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
}

// the survivors in descending (score, -slot) order = the first entries of the stable descending sort; entries past them: empty
__global__ __launch_bounds__(1024) void pp_sel_sort_kernel(const PPArgs p) {
    __shared__ unsigned long long v[SEL_CAP];
    const int t = threadIdx.x;
    const int count = p.sel[2 * SEL_BINS + SEL_COUNT];
    if (count <= SEL_CAP) {
        int N = 64;
        while (N < count) N <<= 1;                                 // (top-k + the few that share its last 24-bit prefix: 1024 or 2048)
        for (int e = t; e < SEL_CAP; e += 1024) v[e] = e < count ? p.cand[e] : 0ull;
        __syncthreads();
        for (int k = 2; k <= N; k <<= 1)                           // bitonic sort, descending
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int e = t; e < N; e += 1024) {
                    const int o = e ^ j;
                    if (o > e) {
                        const unsigned long long a = v[e], b = v[o];
                        const bool desc = (e & k) == 0;
                        if (desc ? a < b : a > b) { v[e] = b; v[o] = a; }
                    }
                }
                __syncthreads();
            }
        for (int e = t; e < p.topk; e += 1024) {
            const unsigned long long c = v[e];
            p.key_out[e] = (unsigned)(c >> 32);
            p.idx_out[e] = c ? (int)(0xffffffffu - (unsigned)c) : 0;
        }
        return;
    }
    // More survivors than the LDS sort holds: finish the radix select here.  Entries above the boundary prefix all make the cut (fewer than
    // top-k of them); of the prefix's own entries the best r do -- by the score's low byte, ties by ascending slot.  A thread's slots are
    // t, t + 1024, ...: slot >> 10 = its trip count, slot & 1023 = t.
    __shared__ int hb[1024];
    __shared__ int pick[4];
    const unsigned thr24 = (unsigned)p.sel[2 * SEL_BINS + SEL_THR24];
    const int r = p.topk - p.sel[2 * SEL_BINS + SEL_ABOVE1] - p.sel[2 * SEL_BINS + SEL_ABOVE2];     // >= 1
    auto pass = [&](auto&& bin_of, const int want, const bool from_top) __attribute__((always_inline)) {
        // histogram of bin_of(slot, key) (-1: not a member) into hb[0..1023]; pick[0] = the bin where the running count reaches `want`
        // (from the top bin down, or from bin 0 up), pick[1] = the count before it
        hb[t] = 0;
        __syncthreads();
        for (int i0 = t; i0 < p.na; i0 += 8 * 1024) {                // (eight keys per thread requested together: the pass is latency-bound)
            unsigned kk[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) kk[q] = i0 + q * 1024 < p.na ? p.key_in[i0 + q * 1024] : 0u;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int bn = i0 + q * 1024 < p.na ? bin_of(i0 + q * 1024, kk[q]) : -1;
                if (bn >= 0) atomicAdd(&hb[bn], 1);
            }
        }
        __syncthreads();
        if (t == 0) {
            int acc = 0, b = from_top ? 1023 : 0;
            for (int s2 = 0; s2 < 1024; ++s2) {
                b = from_top ? 1023 - s2 : s2;
                if (acc + hb[b] >= want) break;
                acc += hb[b];
            }
            pick[0] = b; pick[1] = acc;
        }
        __syncthreads();
    };
    pass([&](int, unsigned k) { return (k != 0u && (k >> 8) == thr24) ? (int)(k & 0xffu) : -1; }, r, true);
    const unsigned T = (thr24 << 8) | (unsigned)pick[0];           // the exact score at the cut
    const int ties = r - pick[1];                                   // how many slots with that score enter, lowest slots first
    __syncthreads();
    pass([&](int i, unsigned k) { return k == T ? (i >> 10) : -1; }, ties, false);
    const int g = pick[0], ties2 = ties - pick[1];
    __syncthreads();
    pass([&](int i, unsigned k) { return (k == T && (i >> 10) == g) ? (i & 1023) : -1; }, ties2, false);
    const int last_slot = (g << 10) | pick[0];                      // ties with slot <= last_slot enter: exactly `ties` of them
    __syncthreads();
    if (t == 0) pick[2] = 0;
    for (int e = t; e < SEL_CAP; e += 1024) v[e] = 0ull;
    __syncthreads();
    for (int i0 = t; i0 < p.na; i0 += 8 * 1024) {
        unsigned kk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) kk[q] = i0 + q * 1024 < p.na ? p.key_in[i0 + q * 1024] : 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = i0 + q * 1024;
            const unsigned k = kk[q];
            if (i < p.na && (k > T || (k == T && i <= last_slot))) {
                const int at = atomicAdd(&pick[2], 1);              // (exactly top-k entries, or all there are)
                if (at < SEL_CAP) v[at] = ((unsigned long long)k << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
            }
        }
    }
    __syncthreads();
    for (int k = 2; k <= 1024; k <<= 1)                             // bitonic sort of 1024, descending
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int e = t, o = e ^ j;
            if (o > e) {
                const unsigned long long a = v[e], b = v[o];
                const bool desc = (e & k) == 0;
                if (desc ? a < b : a > b) { v[e] = b; v[o] = a; }
            }
            __syncthreads();
        }
    for (int e = t; e < p.topk; e += 1024) {
        const unsigned long long c = v[e];
        p.key_out[e] = (unsigned)(c >> 32);
        p.idx_out[e] = c ? (int)(0xffffffffu - (unsigned)c) : 0;
    }
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

// OR over the wave of a 64-bit value (every lane gets it): quad, half-row and row steps by DPP, the four rows by readlane
__device__ __forceinline__ unsigned wave_or32(unsigned x) {
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xf, 0xf, true);     // row_half_mirror
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xf, 0xf, true);     // row_mirror
    return (unsigned)__builtin_amdgcn_readlane((int)x, 0) | (unsigned)__builtin_amdgcn_readlane((int)x, 16) |
           (unsigned)__builtin_amdgcn_readlane((int)x, 32) | (unsigned)__builtin_amdgcn_readlane((int)x, 48);
}
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long x) {
    return ((unsigned long long)wave_or32((unsigned)(x >> 32)) << 32) | wave_or32((unsigned)x);
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long x, int l) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), l) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, l);
}

#ifdef QV2X_PP_FINE                            // dev build (tools/bench_post.py <tag>): s_memtime stamps of the sweep kernel's phases
__device__ long long g_pp_fine[8];
#define PFINE(k) do { if (threadIdx.x == 0) g_pp_fine[k] = __builtin_readcyclecounter(); } while (0)
#else
#define PFINE(k) do { } while (0)
#endif
// greedy sweep in score order + range mask + compaction.  One workgroup; the bit matrix (<= 136 KB, row pitch 17 words) sits in LDS.
// Round 5: the sweep walked the candidates one by one (two readlanes, a branch and an LDS read per candidate behind the previous one's
// result: 177 us per frame of 1000 candidates).  Now 64 candidates at a time: lane i holds candidate 64 c + i; inside the chunk only the
// kept ones cost a step (scalar bit scan over "not yet removed", the kept one's own-chunk word by readlane); what the chunk's kept rows
// remove further down is one wave-wide OR per later word.  Same greedy order, same result.
__global__ __launch_bounds__(1024) void pp_sweep_kernel(const PPArgs p) {
    constexpr int W = TOPK_MAX / 64, WP = W + 1;
    __shared__ unsigned long long m[(TOPK_MAX + 1) * WP];
    __shared__ unsigned long long pickw[W], insidew[W];
    __shared__ int base[W + 1];
    const int n = p.topk, words = (n + 63) >> 6;
    PFINE(0);
    {   // the matrix into LDS: every thread's loads requested together (sixteen 8-byte words per thread at 1024 threads and 1024 rows)
        unsigned long long tmp[W];
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const int t = threadIdx.x + q * 1024;
            tmp[q] = (t < n * W && (t & (W - 1)) < words) ? p.mask[t] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const int t = threadIdx.x + q * 1024;
            if (t < n * W) m[(t / W) * WP + (t & (W - 1))] = tmp[q];
        }
    }
    if (threadIdx.x < W) { pickw[threadIdx.x] = 0ull; insidew[threadIdx.x] = 0ull; }
    if (threadIdx.x < WP) m[TOPK_MAX * WP + threadIdx.x] = 0ull;
    __syncthreads();
    PFINE(1);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned long long mine = 0ull;                            // lane w: word w of the removed set
        // how many slots hold a candidate (sorted: the first `nvalid`): all the keys requested at once -- one global round trip, not one per chunk
        int nvalid = 0;
        {
            unsigned kk[W];
#pragma unroll
            for (int q = 0; q < W; ++q) kk[q] = 64 * q + lane < n ? p.key_out[64 * q + lane] : 0u;
#pragma unroll
            for (int q = 0; q < W; ++q) nvalid += __builtin_popcountll(__builtin_amdgcn_ballot_w64(kk[q] != 0u));
        }
        for (int c = 0; c < words; ++c) {
            const int i = 64 * c + lane;
            if (64 * c >= nvalid) break;                           // sorted: nothing valid after the first empty slot
            const unsigned long long vmask = nvalid - 64 * c >= 64 ? ~0ull : (1ull << (nvalid - 64 * c)) - 1ull;
            const unsigned long long own = i < n ? m[i * WP + c] : 0ull;          // bit j: IoU(i, 64 c + j) > threshold, j > lane only
            unsigned long long r = readlane64(mine, c);
            unsigned long long picked = 0ull;
            unsigned long long cand = vmask & ~r;
            while (cand) {                                          // (wave-uniform)
                const int b = __builtin_ctzll(cand);
                picked |= 1ull << b;
                r |= readlane64(own, b);
                cand = vmask & ~r & ~((2ull << b) - 1ull);         // (b = 63: 2 << 63 = 0, the mask is all ones: nothing left)
            }
            if (lane == 0) pickw[c] = picked;
            // what the chunk's kept rows remove further down: lane w ORs word w of every kept row (four rows requested at a time; row
            // TOPK_MAX is zeros)
            unsigned long long acc = 0ull, pp = picked;
            const int col = lane < W ? lane : 0;
            while (pp) {                                            // (wave-uniform)
                int rows[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rows[q] = pp ? 64 * c + __builtin_ctzll(pp) : TOPK_MAX;
                    pp &= pp - 1ull;                                // (0 stays 0)
                }
                unsigned long long x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = m[rows[q] * WP + col];
                acc |= (x[0] | x[1]) | (x[2] | x[3]);
            }
            if (lane > c && lane < words) mine |= acc;
        }
    }
    PFINE(2);
    __syncthreads();
    PFINE(3);
    // range mask on all eight corners of the kept boxes, then compaction in score order (prefix of popcounts)
    for (int i = threadIdx.x; i < words * 64; i += blockDim.x) {
        bool in = false;
        if ((pickw[i >> 6] >> (i & 63)) & 1ull) {
            const v4f* cp = (const v4f*)(p.cand_corners + (size_t)p.idx_out[i] * 24);     // (96-byte records: six aligned float4, requested together)
            float c[24];
#pragma unroll
            for (int q = 0; q < 6; ++q) { const v4f v = cp[q]; c[4 * q] = v[0]; c[4 * q + 1] = v[1]; c[4 * q + 2] = v[2]; c[4 * q + 3] = v[3]; }
            in = true;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                in = in & (c[k * 3] >= p.range[0]) & (c[k * 3 + 1] >= p.range[1]) & (c[k * 3] <= p.range[3]) & (c[k * 3 + 1] <= p.range[4]) &
                     (p.xy_only || (c[k * 3 + 2] >= p.range[2] && c[k * 3 + 2] <= p.range[5]));
        }
        const unsigned long long bits = __builtin_amdgcn_ballot_w64(in);          // (a wave = one word)
        if ((threadIdx.x & 63) == 0) insidew[i >> 6] = bits;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int o = 0;
        for (int w = 0; w < words; ++w) { base[w] = o; o += __builtin_popcountll(insidew[w]); }
        *p.out_count = o;
    }
    __syncthreads();
    PFINE(4);
    for (int i = threadIdx.x; i < words * 64; i += blockDim.x) {
        const unsigned long long wbits = insidew[i >> 6];
        if (!((wbits >> (i & 63)) & 1ull)) continue;
        const int o = base[i >> 6] + __builtin_popcountll(wbits & ((1ull << (i & 63)) - 1ull));
        const int cand = p.idx_out[i];
        const v4f* cp = (const v4f*)(p.cand_corners + (size_t)cand * 24);
        v4f cv[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) cv[q] = cp[q];
#pragma unroll
        for (int q = 0; q < 6; ++q)                                 // (the caller's array: no alignment promised -- dword stores)
#pragma unroll
            for (int e = 0; e < 4; ++e) p.out_corners[(size_t)o * 24 + 4 * q + e] = cv[q][e];
        p.out_scores[o] = p.cand_score[cand];
        if (p.out_labels) p.out_labels[o] = p.cand_label[cand];
    }
    PFINE(5);
}

struct Layout {
    size_t prob, flag, pos, corners, score, key_in, key_out, idx_in, idx_out, mask, label, cand_label, sel, cand, cub, cub_bytes, total;
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
    l.sel = take((size_t)(2 * SEL_BINS + SEL_WORDS) * 4); l.cand = take((size_t)SEL_CAP * 8);
    size_t a = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, a, (int*)nullptr, (int*)nullptr, na);
    l.cub_bytes = a;
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
    if ((long long)na1 * ncav > (1ll << 20)) return fail(QV2X_EINVAL, "%s: %lld anchors in all; the top-k selection indexes 2^20", who, (long long)na1 * ncav);
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
    p.sel = (int*)(ws + l.sel); p.cand = (unsigned long long*)(ws + l.cand);
    p.out_corners = out_corners; p.out_scores = out_scores; p.out_labels = out_labels; p.out_count = out_count;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (na + 255) / 256;
    int rc;
    pp_score_kernel<<<blocks, 256, 0, st>>>(p);
    pp_sel_clear_kernel<<<(2 * SEL_BINS + SEL_WORDS + 255) / 256, 256, 0, st>>>(p);
    size_t tb = l.cub_bytes;
    if ((rc = hip_check(hipcub::DeviceScan::ExclusiveSum(ws + l.cub, tb, p.flag, p.pos, na, st), "postprocess scan"))) return rc;
    pp_decode_kernel<<<blocks, 256, 0, st>>>(p);
    // the first `topk` entries of the stable descending order (see SEL_* above)
    const int hblocks = blocks < 64 ? blocks : 64;                 // (4096 LDS bins to clear and flush per workgroup)
    pp_sel_hist_kernel<0><<<hblocks, 256, 0, st>>>(p);
    pp_sel_pick_kernel<0><<<1, 1024, 0, st>>>(p);
    pp_sel_hist_kernel<1><<<hblocks, 256, 0, st>>>(p);
    pp_sel_pick_kernel<1><<<1, 1024, 0, st>>>(p);
    pp_sel_compact_kernel<<<blocks, 256, 0, st>>>(p);
    pp_sel_sort_kernel<<<1, 1024, 0, st>>>(p);
    pp_iou_kernel<<<p.topk, 256, 0, st>>>(p);
    pp_sweep_kernel<<<1, 1024, 0, st>>>(p);
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

#ifdef QV2X_PP_FINE
extern "C" __attribute__((visibility("default"))) int qv2x_debug_pp_fine(long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(qv2x::g_pp_fine), sizeof(long long) * 8) == hipSuccess ? 0 : -1;
}
#endif
