// Error plumbing + trivial entry points of the C ABI.
#include <cstdarg>
#include <cstdio>

#include "common.h"

namespace qv2x {
static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return QV2X_OK;
    return fail(-1000 - (int)e, "%s: %s", what, hipGetErrorString(e));
}
}  // namespace qv2x

extern "C" {
const char* qv2x_last_error(void) { return qv2x::g_err; }
int qv2x_version(void) { return 1; }

int qv2x_fill_i8(int8_t* buf, int64_t bytes, int value, void* stream) {
    if (!buf || bytes < 0) return qv2x::fail(QV2X_EINVAL, "qv2x_fill_i8: null buffer or negative size");
    if (bytes == 0) return QV2X_OK;
    return qv2x::hip_check(hipMemsetAsync(buf, value & 0xFF, (size_t)bytes, (hipStream_t)stream), "qv2x_fill_i8");
}
}
