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

// model spec checks shared by every entry point that takes a d3p_logreg_model; labels are only read by
// the Bernoulli family
inline int validate_model(const d3p_logreg_model* m, const void* y_dev, const char* what)
{
    if (!m) return fail(D3P_E_INVALID_ARG, "%s: null model", what);
    if (!(m->d >= 1 && m->prior_w > 0.f && m->prior_b > 0.f && m->inv_obs > 0.f))
        return fail(D3P_E_INVALID_ARG, "%s: bad model (d >= 1, prior scales > 0 and inv_obs > 0 are required)", what);
    if (m->guide_transform != D3P_GUIDE_SOFTPLUS && m->guide_transform != D3P_GUIDE_EXP)
        return fail(D3P_E_INVALID_ARG, "%s: unknown guide transform %d", what, m->guide_transform);
    if (m->family == D3P_FAMILY_LOGREG) {
        if (!y_dev) return fail(D3P_E_INVALID_ARG, "%s: null label pointer", what);
    } else if (m->family == D3P_FAMILY_GAUSS_MEAN) {
        if (m->intercept) return fail(D3P_E_INVALID_ARG, "%s: the Gaussian-mean family has no intercept", what);
        if (!(m->lik_sigma > 0.f)) return fail(D3P_E_INVALID_ARG, "%s: lik_sigma must be > 0", what);
    } else {
        return fail(D3P_E_INVALID_ARG, "%s: unknown likelihood family %d", what, m->family);
    }
    return D3P_OK;
}

inline unsigned cdiv(unsigned long long a, unsigned long long b) { return (unsigned)((a + b - 1) / b); }

// in-place sum-all-reduce of `count` floats over the ranks of a d3p_comm_* communicator (RCCL, resolved at run time: d3p_dpvi.hip)
int rccl_allreduce_f32(void* comm, float* buf, size_t count, hipStream_t s);

// in-place sum-all-reduce of the n floats a d3p_fmesh_* mesh was created for (full-mesh reduce-scatter + all-gather over the peers'
// hipIpc-mapped inboxes: d3p_fmesh.hip)
int fmesh_enqueue_allreduce(hipStream_t s, void* fmesh, float* buf, uint64_t n);

}  // namespace d3p
