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

// UniformAffineQuantizer.forward on one value (quant_layer.py:132-133): code in [0, 255] as a float.
// IEEE division and round-half-even, so it matches the CPU oracle bit for bit.
__device__ __forceinline__ float q_code(float y, float delta, float zp) {
    float t = rintf(y / delta) + zp;
    t = fmaxf(t, 0.0f);
    return fminf(t, 255.0f);
}

// C/D fragment row of register r for the 32x32 MFMA forms (cdna guide §3): row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace qv2x
