// Host-side helpers of libd3p_hip.so: error reporting and launch checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <cmath>

#include "../../include/d3p_hip.h"

namespace d3p {

char* last_error_buf();  // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(D3P_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return D3P_OK;
}

#define D3P_HIP_TRY(expr)                                                                   \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) return d3p::fail(D3P_E_HIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

#define D3P_REQUIRE(cond, msg)                                         \
    do {                                                               \
        if (!(cond)) return d3p::fail(D3P_E_INVALID_ARG, "%s", msg);   \
    } while (0)

inline unsigned cdiv(unsigned long long a, unsigned long long b) { return (unsigned)((a + b - 1) / b); }

}  // namespace d3p
