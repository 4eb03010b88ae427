// Shared helpers of libqv2x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qv2x.h"

namespace qv2x {

int fail(int code, const char* fmt, ...);          // records the message for qv2x_last_error(), returns code
int hip_check(hipError_t e, const char* what);      // 0 or -1000 - e

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// UniformAffineQuantizer.forward on one value (quant_layer.py:132-133): code in [0, 255] as a float, equal bit for
// bit to clamp(rintf(y / delta) + zp, 0, 255) with IEEE division (what the CPU oracle computes) -- without paying for
// the division on every element.  t = y * fl(1/delta) is within |y/delta| * 1.8e-7 of fl(y/delta); below 1100 that is
// < 2.0e-4, so both round to the same integer unless t sits within 3e-4 of a half-integer; above 1100 either value
// clamps to 0 or 255 (0 <= zp <= 255).  The rare wave with a lane in the doubtful band redoes the division.
__device__ __forceinline__ float q_code(float y, float delta, float zp) {
    const float rdelta = 1.0f / delta;                  // uniform: hoisted out of every loop
    const float t = y * rdelta;
    float k = rintf(t);
    const bool doubtful = (fabsf(t - k) > 0.4997f) & (fabsf(t) < 1100.0f);
    if (__builtin_amdgcn_ballot_w64(doubtful) != 0) k = rintf(y / delta);
    float c = k + zp;
    c = fmaxf(c, 0.0f);
    return fminf(c, 255.0f);
}

// ---- the output quantizer of the CONVOLUTION epilogues, round 5: code = clamp(rint(fma(y, fl(1 / delta), zp)), 0, 255) -------------------
// oracle/qv2x_oracle.c:q_code_mul states it and says why: UniformAffineQuantizer.forward (quant_layer.py:132-133) with the division replaced
// by one fused multiply-add with the fp32 reciprocal.  v_fma_f32 + v_cvt_pk_u8_f32 (round to nearest even, saturate to 0..255, insert byte e:
// tools/probes/cvt_pk_u8_probe.hip) evaluate exactly that in TWO instructions per output -- the division-exact sandwich of round 4 took 5.25
// (below, q_pack4_div: kept for the sparse 3-D convolutions, whose CPU restatement is numpy), and the 64- / 128-channel layers are bound by
// VALU issue (profiles/r05_ws64_ablations.log).  `rdelta` = 1.0f / delta, hoisted by the caller.
// `low` = lowest code: 0 for the plain quantizer; zp folds a ReLU in front of the quantizer into a clamp -- q(max(y, 0)) =
// max(rint(fma(y, rd, zp)), zp): fma(0, rd, zp) = zp exactly and fma, rint are monotone.  With zp = 0 (every post-ReLU quantizer the
// reference's observers produce: the observed minimum is 0) the conversion's saturation IS that clamp.
// NaN / Inf contract: the callers' y are finite by construction (ptq_state.check_finite); an infinite y saturates, a NaN y converts to 0.
__device__ __forceinline__ void q_add(float y, int e, float rdelta, float zp, float low, unsigned& c) {
    float t = __builtin_fmaf(y, rdelta, zp);
    t = fmaxf(t, low);                                  // (an integer bound: rint(max(t, low)) = max(rint(t), low); with low = 0 a no-op in front of
                                                        //  the saturating conversion -- but ONE v_max: `if (low > 0)` compiled to v_max + v_cndmask)
    c = __builtin_amdgcn_cvt_pk_u8_f32(t, e, c);
}
// four outputs -> one dword of (code - 128) bytes (byte e = element e).  `lowc`: lowest code + 2^23 (the callers' convention since round 3).
__device__ __forceinline__ int q_pack4(float y0, float y1, float y2, float y3, float delta, float rdelta, float zp, float lowc = 8388608.0f) {
    (void)delta;
    const float low = lowc - 8388608.0f;
    unsigned c = 0;
    q_add(y0, 0, rdelta, zp, low, c);
    q_add(y1, 1, rdelta, zp, low, c);
    q_add(y2, 2, rdelta, zp, low, c);
    q_add(y3, 3, rdelta, zp, low, c);
    return (int)(c ^ 0x80808080u);
}
// one value, the same arithmetic (the 1x1 occupancy head of the Pyramid model): the code as a float
__device__ __forceinline__ float q_code_mul(float y, float delta, float zp) {
    const float rdelta = 1.0f / delta;                  // uniform: hoisted out of every loop
    return fminf(fmaxf(rintf(__builtin_fmaf(y, rdelta, zp)), 0.0f), 255.0f);
}

// ---- the DIVISION-EXACT form of round 4 (clamp(rint(y / delta) + zp) bit for bit, without paying for the division on every element) ------
// Four outputs -> one dword of (code - 128) bytes (byte e = element e), the same values q_code gives.  `rdelta` = 1.0f / delta,
// hoisted by the caller.  The product p = y * rdelta is within 1.2e-7 |p| of y / delta and fl(y / delta) within 6e-8 more; a code only
// depends on rint(.) for |p| < 256.5 (0 <= zp <= 255: beyond that both values clamp to 0 or 255 even when they differ by one), where the
// gap is < 4.7e-5.  Round 4: a SANDWICH instead of a residual test --
//   tA = fma(y, rdelta, zp + 1e-4),  tB = fma(y, rdelta, zp - 1e-4)      (fp32: zp +- 1e-4 keeps >= 9.2e-5 of the offset at zp <= 255)
//   cA = v_cvt_pk_u8_f32(tA), cB = v_cvt_pk_u8_f32(tB)                   (round to nearest even, saturate to 0..255, insert byte e:
//                                                                          tools/probes/cvt_pk_u8_probe.hip, profiles/r04_cvt_pk_u8_probe.log)
// fl, rint and the clamp are monotone and tB_exact < fl(y / delta) + zp < tA_exact, so cA == cB pins the exact code between two equal
// values; where they differ (y / delta within ~1e-4 of a half-integer, exact ties included: 5 % of the groups of 256 outputs) the wave
// takes the exact divisions.  6.25 -> 4.5 instructions per output (no v_med3, no v_perm: the conversion clamps and packs) -- the epilogues
// are bound by the SIMD's VALU issue, DESIGN.md 3.
// `lowc`: lowest code + 2^23 (the callers' convention since round 3): 8388608.0f (code 0) for the plain quantizer; zp + 2^23 folds a ReLU
// in front of the quantizer into the clamp -- rint is monotone and rint(0) = 0, so q(max(y, 0)) = max(rint(y / delta), 0) + zp.  With
// zp = 0 (every post-ReLU quantizer the reference's observers produce: the observed minimum is 0) the saturation IS that clamp.
// NaN / Inf contract: the callers' y are finite by construction (an exact i32 sum times a finite per-channel scale plus a finite bias; fp32
// dot products of finite codes and finite weights) -- deploy refuses non-finite scales, biases and deltas (ptq_state.check_finite).  An
// infinite y saturates like any out-of-range value; a NaN y converts to code 0 on both sides of the sandwich (v_cvt_pk_u8_f32(NaN) = 0), the
// lowest code, as q_code's fminf(fmaxf()) gave (ReLU layers with zp > 0: zp, through the v_max below).
// The sandwich in pieces, for epilogues that requantize ONE output between two MFMAs (conv_i8_ws.hip): add element e of a group of four ...
__device__ __forceinline__ void q_sandwich_add(float y, int e, float rdelta, float za, float zb, unsigned& ca, unsigned& cb) {
    ca = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(y, rdelta, za), e, ca);
    cb = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(y, rdelta, zb), e, cb);
}
// ... and close the group: the four (code - 128) bytes.  `low` = lowest code (0, or zp under a folded ReLU: uniform).  A ReLU folded in
// front of a quantizer with zp > 0 needs a lower clamp the conversion does not have: such a layer (none of the reference's post-ReLU
// observers produces one: their minimum is 0) takes the exact path for every group -- correct, slower.
__device__ __forceinline__ int q_sandwich_finish(unsigned ca, unsigned cb, float y0, float y1, float y2, float y3, float delta, float zp, float low) {
    if ((__builtin_amdgcn_ballot_w64(ca != cb) != 0) | (low > 0.0f)) {
        const float y[4] = {y0, y1, y2, y3};
        ca = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) ca = __builtin_amdgcn_cvt_pk_u8_f32(fmaxf(rintf(y[e] / delta) + zp, low), e, ca);   // (an integer: converts exactly)
    }
    return (int)(ca ^ 0x80808080u);
}
__device__ __forceinline__ int q_pack4_div(float y0, float y1, float y2, float y3, float delta, float rdelta, float zp, float lowc = 8388608.0f) {
    const float za = zp + 1.0e-4f, zb = zp - 1.0e-4f;
    unsigned ca = 0, cb = 0;
    q_sandwich_add(y0, 0, rdelta, za, zb, ca, cb);
    q_sandwich_add(y1, 1, rdelta, za, zb, ca, cb);
    q_sandwich_add(y2, 2, rdelta, za, zb, ca, cb);
    q_sandwich_add(y3, 3, rdelta, za, zb, ca, cb);
    return q_sandwich_finish(ca, cb, y0, y1, y2, y3, delta, zp, lowc - 8388608.0f);
}

// C/D fragment row of register r for the 32x32 MFMA forms (cdna guide §3): row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace qv2x
