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
// hoisted by the caller.  The |t| < 1100 gate of q_code is dropped: beyond it a doubtful lane only takes the division path
// needlessly (both results clamp).  The clamped code is read from the low mantissa byte of (code + 2^23).
__device__ __forceinline__ int q_pack4(float y0, float y1, float y2, float y3, float delta, float rdelta, float zp) {
    const float y[4] = {y0, y1, y2, y3};
    unsigned b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = y[e] * rdelta;
        float k = rintf(t);
        if (__builtin_amdgcn_ballot_w64(fabsf(t - k) > 0.4997f) != 0) k = rintf(y[e] / delta);
        const float c = fminf(fmaxf(k + zp, 0.0f), 255.0f) + 8388608.0f;
        b[e] = __builtin_bit_cast(unsigned, c);
    }
    const unsigned lo = __builtin_amdgcn_perm(b[1], b[0], 0x0c0c0400u);      // byte 0 of b0, byte 0 of b1
    const unsigned hi = __builtin_amdgcn_perm(b[3], b[2], 0x0c0c0400u);
    return (int)(__builtin_amdgcn_perm(hi, lo, 0x05040100u) ^ 0x80808080u);
}

// C/D fragment row of register r for the 32x32 MFMA forms (cdna guide §3): row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace qv2x
