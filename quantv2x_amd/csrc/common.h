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

// Four outputs -> one dword of (code - 128) bytes (byte e = element e), the same values q_code gives.  `rdelta` = 1.0f / delta,
// hoisted by the caller.  The product p = y * rdelta is within 1.2e-7 |p| of y / delta and fl(y / delta) within 6e-8 more; a code only
// depends on rint(.) for |p| < 256.5 (0 <= zp <= 255: beyond that both values clamp to 0 or 255 even when they differ by one), where the
// gap is < 4.7e-5, so the two round alike unless p lies within 1e-4 of a half-integer.
//   r = fma(y, rdelta, zp + 2^23)    ONE rounding of the exact p + zp + 2^23 to an integer (ulp 1 in [2^23, 2^24)): r = code + 2^23
//   e = fma(y, rdelta, (zp + 2^23) - r) = fl(p - k), k = r - zp - 2^23 exactly: how far p is from the integer it was rounded to
// NaN / Inf contract: the callers' y are finite by construction (an exact i32 sum times a finite per-channel scale plus a finite bias; fp32
// dot products of finite codes and finite weights) -- deploy refuses non-finite scales, biases and deltas (ptq_state.py).  An infinite y
// clamps like any out-of-range value; a NaN y yields an UNSPECIFIED code (v_med3_f32 with a NaN operand), where q_code's fminf(fmaxf())
// gave code 0: nothing on the parity path depends on it (tools/probes/q_pack4_probe.hip lists the behaviour).
// One test per group of four: max |e| > 0.4999 on any lane sends the wave through the exact divisions (5 % of the groups).  Below code 0
// (p + zp < 0) r falls under 2^23 where its ulp is 1/2 and k may be a half-integer: the clamp sets those to the lowest code, which is what
// they are.  The clamped code is read from the low mantissa byte of r.  (Round 3: 5.25 instead of 7.25 instructions per output -- the
// epilogues are bound by the SIMD's VALU issue)
// `lowc`: lowest code + 2^23.  8388608.0f (code 0) for the plain quantizer; zp + 2^23 folds a ReLU in front of the quantizer into the
// clamp -- rint is monotone and rint(0) = 0, so q(max(y, 0)) = max(rint(y / delta), 0) + zp -- and saves the caller one fmaxf per output.
__device__ __forceinline__ int q_pack4(float y0, float y1, float y2, float y3, float delta, float rdelta, float zp, float lowc = 8388608.0f) {
    const float y[4] = {y0, y1, y2, y3};
    const float zm = zp + 8388608.0f;
    float r[4], dmax = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = __builtin_fmaf(y[e], rdelta, zm);
        dmax = fmaxf(dmax, fabsf(__builtin_fmaf(y[e], rdelta, zm - r[e])));
    }
    if (__builtin_amdgcn_ballot_w64(dmax > 0.4999f) != 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = rintf(y[e] / delta) + zm;
    }
    unsigned b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_fmed3f(r[e], lowc, 8388863.0f));   // ONE v_med3_f32 (fminf(fmaxf()) is two: NaN rules); r is never NaN here
    const unsigned lo = __builtin_amdgcn_perm(b[1], b[0], 0x0c0c0400u);      // byte 0 of b0, byte 0 of b1
    const unsigned hi = __builtin_amdgcn_perm(b[3], b[2], 0x0c0c0400u);
    return (int)(__builtin_amdgcn_perm(hi, lo, 0x05040100u) ^ 0x80808080u);
}

// C/D fragment row of register r for the 32x32 MFMA forms (cdna guide §3): row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace qv2x
