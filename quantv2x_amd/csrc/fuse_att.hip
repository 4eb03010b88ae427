// a7 + a8 + a9 + a10: decode (three LUT gathers) + bilinear warp into the ego frame + per-cell attention.
//
// One wavefront per ego BEV cell, lane = 4 of the 256 channels.  Per agent the four bilinear taps are decoded
// straight from the u8 code planes -- the only large HBM stream is the fp32 output; the 3 x Kc x 1 KiB tables
// (384 KiB at Kc = 128) are shared by every wave and stay L2 resident.  The sampling grid follows torch:
//   affine_grid(theta, align_corners=False): x_j = (2j + 1)/W - 1, grid = theta . [x, y, 1]  (float64 theta ->
//   float64 grid, cast to fp32 as the reference's `.to(src)` does), grid_sample bilinear / zeros.
// Attention (fusion_in_one.py:41-45, 145-147): score_j = <f_0, f_j> / sqrt(C); softmax; out = sum_j p_j f_j.
#include <cstdlib>

#include "fuse_att.h"

namespace qv2x {

template <int NA, bool B3 = false>
__global__ __launch_bounds__(256) void fuse_att_kernel(const FuseArgs a) {
    const int lane = threadIdx.x & 63;
    int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    cell = __builtin_amdgcn_readfirstlane(cell);
    if (cell >= a.hw) return;
    a.fused[(size_t)cell * 64 + lane] = B3 ? fuse_cell_b3<NA>(a, cell, lane) : fuse_cell_n<NA>(a, cell, lane);
}

// several scenes in one launch: blockIdx.y = scene (SceneList: fuse_att.h)

template <int NA, bool B3 = false>
__global__ __launch_bounds__(256) void fuse_att_batch_kernel(FuseArgs a, const SceneList sl) {
    const int lane = threadIdx.x & 63;
    int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    cell = __builtin_amdgcn_readfirstlane(cell);
    if (cell >= a.hw) return;
    const int sc = blockIdx.y;
    a.agents = sl.agents[sc];
    if (a.feats) a.feats = (const float4*)((const float*)a.feats + sl.off[sc]);
    else a.codes += sl.off[sc];
    a.pairwise += (size_t)sc * a.L * a.L * 16;
    a.fused[((size_t)sc * a.hw + cell) * 64 + lane] = B3 ? fuse_cell_b3<NA>(a, cell, lane) : fuse_cell_n<NA>(a, cell, lane);
}

int fuse_args_from_desc(const qv2x_fuse_desc* d, const uint8_t* codes, const float* lut, const float* lut_bias, const float* feats,
                        const double* pairwise, const char* who, FuseArgs& a) {
    if (!d || !pairwise) return fail(QV2X_EINVAL, "%s: null pointer", who);
    if (!feats && (!codes || !lut || !lut_bias)) return fail(QV2X_EINVAL, "%s: need codes + lut + lut_bias, or feats", who);
    if (d->agents < 1 || d->agents > MAXA) return fail(QV2X_EINVAL, "%s: 1..%d agents, got %d", who, MAXA, d->agents);
    if (d->max_cav < d->agents || d->ego < 0 || d->ego >= d->agents) return fail(QV2X_EINVAL, "%s: max_cav / ego out of range", who);
    if (d->h <= 0 || d->w <= 0 || (!feats && (d->levels < 1 || d->levels > 16 || d->kc < 1 || d->kc > 256)))
        return fail(QV2X_EINVAL, "%s: bad sizes (1..16 code planes = levels * seg_num, dict_size <= 256)", who);
    if (!(d->h_metres > 0) || !(d->w_metres > 0) || !(d->discrete_ratio > 0)) return fail(QV2X_EINVAL, "%s: map extent must be positive", who);
    a.codes = codes; a.lut = (const float4*)lut; a.lut_bias = (const float4*)lut_bias; a.feats = (const float4*)feats;
    a.fused = nullptr; a.pairwise = pairwise;
    a.agents = d->agents; a.h = d->h; a.w = d->w; a.levels = d->levels; a.kc = d->kc; a.hw = d->h * d->w; a.L = d->max_cav; a.ego = d->ego;
    a.code_agent_stride = d->code_agent_stride; a.code_level_stride = d->code_level_stride;
    a.hm = d->h_metres; a.wm = d->w_metres; a.ratio = d->discrete_ratio;
    if (d->fusion != 0 && d->fusion != 1) return fail(QV2X_EINVAL, "%s: fusion 0 (attention) or 1 (max)", who);
    a.fusion = d->fusion;
    return QV2X_OK;
}

}  // namespace qv2x

extern "C" int qv2x_fuse_att_f32(const qv2x_fuse_desc* d, const uint8_t* codes, const float* lut, const float* lut_bias,
                                 const float* feats, const double* pairwise, float* fused, void* stream) {
    using namespace qv2x;
    if (!fused) return fail(QV2X_EINVAL, "qv2x_fuse_att_f32: null pointer");
    FuseArgs a;
    if (int rc = fuse_args_from_desc(d, codes, lut, lut_bias, feats, pairwise, "qv2x_fuse_att_f32", a)) return rc;
    a.fused = (float4*)fused;
    const dim3 grid((a.hw + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (!feats && a.levels == 3 && a.agents > 1) {                     // batched round trips (fuse_att.h:fuse_cell_b3), same bits
        switch (fuse_bound(a.agents)) {
            case 2: fuse_att_kernel<2, true><<<grid, 256, 0, st>>>(a); break;
            case 4: fuse_att_kernel<4, true><<<grid, 256, 0, st>>>(a); break;
            default: fuse_att_kernel<MAXA, true><<<grid, 256, 0, st>>>(a); break;
        }
        return hip_check(hipGetLastError(), "qv2x_fuse_att_f32 launch");
    }
    switch (fuse_bound(a.agents)) {
        case 1: fuse_att_kernel<1><<<grid, 256, 0, st>>>(a); break;
        case 2: fuse_att_kernel<2><<<grid, 256, 0, st>>>(a); break;
        case 4: fuse_att_kernel<4><<<grid, 256, 0, st>>>(a); break;
        default: fuse_att_kernel<MAXA><<<grid, 256, 0, st>>>(a); break;
    }
    return hip_check(hipGetLastError(), "qv2x_fuse_att_f32 launch");
}

extern "C" int qv2x_fuse_att_batch_f32(const qv2x_fuse_desc* d, int n_scenes, const int64_t* scene_offset, const int32_t* scene_agents,
                                       const uint8_t* codes, const float* lut, const float* lut_bias, const float* feats,
                                       const double* pairwise, float* fused, void* stream) {
    using namespace qv2x;
    if (!fused || !scene_offset || !scene_agents || !d) return fail(QV2X_EINVAL, "qv2x_fuse_att_batch_f32: null pointer");
    if (n_scenes < 1 || n_scenes > MAX_SCENES) return fail(QV2X_EINVAL, "qv2x_fuse_att_batch_f32: 1..%d scenes, got %d", MAX_SCENES, n_scenes);
    SceneList sl{};
    int most = 1;
    for (int s = 0; s < n_scenes; ++s) {
        if (scene_offset[s] < 0 || (feats && scene_offset[s] % 4) || scene_agents[s] < 1 || scene_agents[s] > MAXA || scene_agents[s] > d->max_cav || d->ego >= scene_agents[s])
            return fail(QV2X_EINVAL, "qv2x_fuse_att_batch_f32: scene %d: offset %lld, agents %d (1..%d, <= max_cav, > ego)", s, (long long)scene_offset[s], scene_agents[s], MAXA);
        sl.off[s] = scene_offset[s]; sl.agents[s] = scene_agents[s];
        most = scene_agents[s] > most ? scene_agents[s] : most;
    }
    qv2x_fuse_desc d1 = *d;
    d1.agents = most;
    FuseArgs a;
    if (int rc = fuse_args_from_desc(&d1, codes, lut, lut_bias, feats, pairwise, "qv2x_fuse_att_batch_f32", a)) return rc;
    a.fused = (float4*)fused;
    hipStream_t st = (hipStream_t)stream;
    // Scenes of two and more agents from ~four V2X-Real scenes on: persistent sixteen-wave workgroups with the table's first rows in LDS
    // (profiles/r05_fuse_by_agents.log).  Below that the 128 KB table fill per workgroup is not paid back; single-agent scenes gather
    // one tap per cell and are not bound by the table rows.
    // Three code planes and scenes of 2+ agents: the form with batched round trips (fuse_cell_b3; 88 against 105 us per scene of eight
    // V2X-Real agents, profiles/r05_fuse_by_agents.log).  QV2X_FUSE_MODE=0 (dev A/B switch, tools/bench_fuse.py): the round-4 walk.
    int mode_env = -1;
#ifdef QV2X_DEV_KNOBS
    static const int mode_knob = [] { const char* e = getenv("QV2X_FUSE_MODE"); return e ? atoi(e) : -1; }();
    mode_env = mode_knob;
#endif
    const bool b3 = !feats && a.levels == 3 && most > 1 && mode_env != 0;   // (single-agent scenes: one tap per cell, nothing to batch)
    const dim3 grid((a.hw + 3) / 4, n_scenes);
    if (b3) {
        switch (fuse_bound(most)) {
            case 2: fuse_att_batch_kernel<2, true><<<grid, 256, 0, st>>>(a, sl); break;
            case 4: fuse_att_batch_kernel<4, true><<<grid, 256, 0, st>>>(a, sl); break;
            default: fuse_att_batch_kernel<MAXA, true><<<grid, 256, 0, st>>>(a, sl); break;
        }
        return hip_check(hipGetLastError(), "qv2x_fuse_att_batch_f32 launch");
    }
    switch (fuse_bound(most)) {
        case 1: fuse_att_batch_kernel<1><<<grid, 256, 0, st>>>(a, sl); break;
        case 2: fuse_att_batch_kernel<2><<<grid, 256, 0, st>>>(a, sl); break;
        case 4: fuse_att_batch_kernel<4><<<grid, 256, 0, st>>>(a, sl); break;
        default: fuse_att_batch_kernel<MAXA><<<grid, 256, 0, st>>>(a, sl); break;
    }
    return hip_check(hipGetLastError(), "qv2x_fuse_att_batch_f32 launch");
}
