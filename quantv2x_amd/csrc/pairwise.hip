// The V2X link carries every agent's 4 x 4 world pose next to its code planes (quantv2x_amd/dist.py); the receiving GPU
// builds the reference's pairwise matrix from them:
//     pairwise[i][j] = T_j^-1 T_i = solve(T_j, T_i)        get_pairwise_transformation, opencood/utils/transformation_utils.py:21-66
// identity for i == j and for the padding rows / columns up to max_cav.  One thread per (i, j); float64 Gaussian elimination
// with partial pivoting in a fixed operation order (separate multiply and subtract): bit-identical to oracle/geometry.py:solve4.
#include "common.h"

namespace qv2x {

// blockIdx.y = frame: its poses lie `frame_stride` bytes further inside every agent's payload, its matrix L * L * 16 doubles further in `out`
__global__ void pairwise_from_poses_kernel(const uint8_t* __restrict__ gathered, int world, long long agent_stride, long long pose_offset,
                                           long long frame_stride, int L, double* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L * L) return;
    const int i = idx / L, j = idx - i * L;
    pose_offset += (long long)blockIdx.y * frame_stride;
    double* o = out + ((size_t)blockIdx.y * L * L + idx) * 16;
    if (i == j || i >= world || j >= world) {
        for (int e = 0; e < 16; ++e) o[e] = (e % 5 == 0) ? 1.0 : 0.0;
        return;
    }
    double a[4][4], b[4][4], x[4][4];
    const double* pj = (const double*)(gathered + (size_t)j * agent_stride + pose_offset);
    const double* pi = (const double*)(gathered + (size_t)i * agent_stride + pose_offset);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { a[r][c] = pj[r * 4 + c]; b[r][c] = pi[r * 4 + c]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int p = k;
#pragma unroll
        for (int r = k + 1; r < 4; ++r)
            if (fabs(a[r][k]) > fabs(a[p][k])) p = r;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double ta = a[k][c], tb = b[k][c];
#pragma unroll
            for (int r = k + 1; r < 4; ++r)
                if (r == p) { a[k][c] = a[r][c]; a[r][c] = ta; b[k][c] = b[r][c]; b[r][c] = tb; }
        }
#pragma unroll
        for (int r = k + 1; r < 4; ++r) {
            const double m = a[r][k] / a[k][k];
#pragma unroll
            for (int c = k + 1; c < 4; ++c) a[r][c] = a[r][c] - m * a[k][c];
#pragma unroll
            for (int c = 0; c < 4; ++c) b[r][c] = b[r][c] - m * b[k][c];
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 3; r >= 0; --r) {
            double s = b[r][c];
#pragma unroll
            for (int q = r + 1; q < 4; ++q) s = s - a[r][q] * x[q][c];
            x[r][c] = s / a[r][r];
        }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) o[r * 4 + c] = x[r][c];
}

}  // namespace qv2x

extern "C" int qv2x_pairwise_from_poses_f64(const uint8_t* gathered, int world, int64_t agent_stride_bytes, int64_t pose_offset_bytes,
                                            int max_cav, double* pairwise, void* stream) {
    using namespace qv2x;
    if (!gathered || !pairwise) return fail(QV2X_EINVAL, "qv2x_pairwise_from_poses_f64: null pointer");
    if (world < 1 || max_cav < world || max_cav > 64) return fail(QV2X_EINVAL, "qv2x_pairwise_from_poses_f64: 1 <= world <= max_cav <= 64");
    if (((uintptr_t)gathered & 7) || (agent_stride_bytes & 7) || (pose_offset_bytes & 7) || pose_offset_bytes < 0 || agent_stride_bytes < pose_offset_bytes + 128)
        return fail(QV2X_EALIGN, "qv2x_pairwise_from_poses_f64: the pose block is 16 float64, 8-byte aligned, inside the agent stride");
    const int n = max_cav * max_cav;
    pairwise_from_poses_kernel<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(gathered, world, agent_stride_bytes, pose_offset_bytes, 0, max_cav, pairwise);
    return hip_check(hipGetLastError(), "qv2x_pairwise_from_poses_f64 launch");
}

extern "C" int qv2x_pairwise_from_poses_batch_f64(const uint8_t* gathered, int world, int64_t agent_stride_bytes, int64_t pose_offset_bytes,
                                                  int frames, int64_t frame_stride_bytes, int max_cav, double* pairwise, void* stream) {
    using namespace qv2x;
    if (!gathered || !pairwise) return fail(QV2X_EINVAL, "qv2x_pairwise_from_poses_batch_f64: null pointer");
    if (world < 1 || max_cav < world || max_cav > 64 || frames < 1 || frames > 65535)
        return fail(QV2X_EINVAL, "qv2x_pairwise_from_poses_batch_f64: 1 <= world <= max_cav <= 64, 1 <= frames <= 65535");
    if (((uintptr_t)gathered & 7) || (agent_stride_bytes & 7) || (pose_offset_bytes & 7) || (frame_stride_bytes & 7) || pose_offset_bytes < 0 ||
        frame_stride_bytes < 128 || agent_stride_bytes < pose_offset_bytes + (int64_t)(frames - 1) * frame_stride_bytes + 128)
        return fail(QV2X_EALIGN, "qv2x_pairwise_from_poses_batch_f64: the pose blocks are 16 float64 each, 8-byte aligned, inside the agent stride");
    const int n = max_cav * max_cav;
    pairwise_from_poses_kernel<<<dim3((n + 63) / 64, frames), 64, 0, (hipStream_t)stream>>>(gathered, world, agent_stride_bytes, pose_offset_bytes,
                                                                                           frame_stride_bytes, max_cav, pairwise);
    return hip_check(hipGetLastError(), "qv2x_pairwise_from_poses_batch_f64 launch");
}
