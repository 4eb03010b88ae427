// Shared by the two forms of a6 (codebook_encode.hip: a workgroup per 32 / 64 cells; codebook_encode_wave.hip: a wave per 32 cells).
#pragma once
#include "common.h"

namespace qv2x {

constexpr int ENC_D = 256;
constexpr int ENC_WAVE_PAD = 16 * 256;      // floats behind the wave-form section: its weight ring reads 16 groups (1 KiB each) ahead

struct EncArgs {
    const int8_t* in; const float* in_f32; uint8_t* codes;      // in_f32 != null: the rows come as fp32 (un-quantized model)
    const float* lvl[4];
    int n, h, w, levels, kc, ax, M;     // kc: codes per SEGMENT and level
    int segs, ke;                       // seg_num (m) = 1 | 2 | 4 segments of 256 / segs dims; ke = segs * kc rows of the extended codebook
    float dx;
    int m_lo, m_hi;             // the rows [m_lo, m_hi) of the M = n h w cells this launch encodes (a launch may be split between the two forms)
    int list_slots = 1, list_tail_max = 0;                 // the split of the listed cells between the two forms (codebook_encode.hip:list_full_tiles)
    const unsigned* list = nullptr; const unsigned* list_count = nullptr;      // LIST forms: the cells to encode (three lists) and their DEVICE-side counts (list_plan below)
};

// A level blob (include/qv2x.h): the workgroup form's section, then the wave form's.  `kc` here = rows of the (extended) codebook.
__device__ __host__ __forceinline__ int64_t level_floats_wg(int kc) { return 3LL * (ENC_D * ENC_D + ENC_D) + (int64_t)ENC_D * kc + (int64_t)kc * ENC_D + kc; }
__device__ __host__ __forceinline__ int64_t level_floats(int kc) { return level_floats_wg(kc) + 3LL * ENC_D * ENC_D + (int64_t)((kc + 63) / 64 * 64) * ENC_D + ENC_WAVE_PAD; }

// ---- stage 2 of the two-stage encode: the listed cells (codebook_encode_cand.hip writes them) --------------------------------------------------
// THREE lists by the level a cell is first undecided at: list c at a.list + c * a.M, its length at a.list_count[1 + c] (a.list_count[0] = all of
// them).  Tiles of 32 cells are numbered list by list; whole rounds of the wave form's `list_slots` persistent waves take tiles [0, full), the
// workgroup form the rest (up to `list_tail_max` tiles; a larger remainder is one more round of waves).  Both kernels derive the same plan from
// the DEVICE-side counts.
struct ListPlan { int n[3], first[3], total, full; };
__device__ __forceinline__ ListPlan list_plan(const EncArgs& a) {
    ListPlan p;
    int t = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { p.n[c] = (int)a.list_count[1 + c]; p.first[c] = t; t += (p.n[c] + 31) >> 5; }
    p.total = t;
    const int full = t / a.list_slots * a.list_slots;
    p.full = t - full > a.list_tail_max ? t : full;
    return p;
}
// tile -> (list, first entry of the tile inside it)
__device__ __forceinline__ void list_tile(const ListPlan& p, int tile, int& cls, int& i0) {
    cls = tile >= p.first[2] ? 2 : (tile >= p.first[1] ? 1 : 0);
    i0 = (tile - p.first[cls]) * 32;
}

// codebook_encode_wave.hip
int encode_wave_launch(const EncArgs& a, hipStream_t st);      // rows [a.m_lo, a.m_hi): one wave per 32
int encode_wave_list_launch(const EncArgs& a, int waves, hipStream_t st);   // the cells a.list[0 .. *a.list_count): persistent waves

}  // namespace qv2x
