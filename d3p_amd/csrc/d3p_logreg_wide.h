// Wide rows (2048 < d <= 4096, i.e. the NK = 8 geometry of k_logreg_main): that instantiation keeps 64 column pairs per lane
// in registers, spills 2 KB per lane and needs 5 D + W P floats of LDS, which forces one-wave workgroups: 2.2 ms per step at
// d = 4096.  This kernel is the clip-and-accumulate stage (MODE 0: one partial row of P + 2 floats per workgroup, consumed
// by k_finalize) in COLUMN CHUNKS: a wavefront walks its example three times in chunks of 256 column pairs --
//   pass 1: z = loc + s eps, logit x . z and the latent part of the loss               (svi.py:238-281)
//   pass 2: gradient entries and the squared joint norm -> clip factor                  (svi.py:68-124)
//   pass 3: clip-scaled gradient added to the wavefront's accumulator row in LDS        (svi.py:343-346)
// -- regenerating the guide noise in every pass (threefry + erf_inv: cheaper than keeping 8192 values per example anywhere)
// and reading the derived columns from the global pack (L2-resident, 5 D floats).  LDS holds only the W accumulator rows
// (W = 4: 128 KB at d = 4096).  Same formulas as k_logreg_main; sums are taken in a different order, so results agree with it
// (and the oracle) to rounding, not bit for bit.  Both likelihood families: for the Gaussian mean the residual x - z takes the place of
// the feature (as in k_logreg_main), the chunk hands it out instead of x.
#pragma once
#include "d3p_logreg_kernel.h"

namespace d3p {

#define D3P_WIDE_W 4

static inline size_t wide_lds_bytes(int P) { return (size_t)(D3P_WIDE_W * P + 2 * D3P_WIDE_W) * sizeof(float); }

// PXG: the materialising stage instead (MODE 1 of k_logreg_main, d3p_logreg_px_grads): pass 1, then one pass that writes
// the example's unclipped gradient row and loss (zeros for masked-out positions); no accumulator rows, no partial rows.
template <bool PXG>
__global__ void __launch_bounds__(64 * D3P_WIDE_W) k_logreg_wide(MainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int W = D3P_WIDE_W;
    const int D = a.D, half = a.half, P = 2 * D, d = a.d;
    float* acc = lds;                 // W x P
    float* tail = lds + (size_t)W * P;
    if (!PXG) {
        for (int i = threadIdx.x; i < W * P; i += blockDim.x) acc[i] = 0.f;
        __syncthreads();
    }
    float* mine = acc + (size_t)wave * P;
    const float* pk = a.pack;         // [loc | s | sg | q | lc] x D
    const bool eps_from_mem = a.eps_ext != nullptr;
    const bool gauss = a.family == D3P_FAMILY_GAUSS_MEAN;
    const bool vec_ok = !a.icpt && (d & 3) == 0 && (half & 3) == 0 && D == 2 * half &&
                        ((reinterpret_cast<uintptr_t>(a.X) | reinterpret_cast<uintptr_t>(pk) |
                          reinterpret_cast<uintptr_t>(a.eps_ext)) & 15u) == 0;
    const uint32_t n_valid = a.counts ? a.counts[1] : a.B;
    const uint32_t n_items = a.plist ? *a.n_list : a.B;
    const uint32_t total_waves = gridDim.x * W;
    float loss_acc = 0.f, n_acc = 0.f;
    // (materialising stage) loss * mask is NaN * 0 = NaN for a masked example once a parameter is not finite (svi.py:281): the
    // register-tiled kernel gets there by computing the row and multiplying by 0, this one writes the rows it skips
    float skipped = 0.f;
    if (PXG) {
        int p_bad = 0;
        for (int c = threadIdx.x; c < 5 * D; c += blockDim.x) p_bad |= !(fabsf(pk[c]) <= 3.402823466e38f);
        if (__syncthreads_or(p_bad)) skipped = __builtin_nanf("");   // (this instantiation runs without dynamic LDS)
    }

    for (uint32_t item = blockIdx.x * W + wave; item < n_items; item += total_waves) {
        const uint32_t pp = a.plist ? a.plist[item] : item;
        const uint32_t row_g = a.idx ? a.idx[pp] : pp;
        const bool valid = (pp < n_valid) && (a.mask ? a.mask[pp] != 0 : true);
        const bool own = (uint64_t)row_g >= a.row_lo && (uint64_t)row_g < a.row_hi;
        if (!(valid && own)) {  // wave-uniform
            if (PXG) {  // loss * mask => zero loss and gradient (svi.py:281)
                float* gr = a.px_grads + (size_t)pp * P;
                for (int c = lane; c < P; c += 64) gr[c] = skipped;
                if (lane == 0) a.px_loss[pp] = skipped;
            }
            continue;
        }
        const size_t row = (size_t)((uint64_t)row_g - a.row_lo);
        const float* xrow = a.X + row * (size_t)d;
        const float yv = a.y ? a.y[row] : 0.f;
        uint32_t k0 = 0u, k1 = 0u;
        if (!eps_from_mem) { k0 = a.skeys[2 * pp]; k1 = a.skeys[2 * pp + 1]; }
        const float* er = eps_from_mem ? a.eps_ext + (size_t)pp * D : nullptr;

        // 4 consecutive entries base[c .. c + 3]: one 16-byte load on the aligned fast path (d and D / 2 multiples of 4, no
        // intercept, chunk inside the row), guarded scalar loads otherwise
        auto ld4 = [&](const float* base, int c, int limit, bool fast, float (&v)[4]) {
            if (fast) {
                const float4 t = *reinterpret_cast<const float4*>(base + c);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (c + i < limit) ? base[c + i] : 0.f;
            }
        };
        // x, eps and z of the 4 column pairs (c, c + half), c = cb + 4 lane + i, of one chunk
        auto chunk = [&](int cb, float (&x0)[4], float (&x1)[4], float (&e0)[4], float (&e1)[4], float (&z0)[4], float (&z1)[4],
                         bool (&ok0)[4], bool (&ok1)[4]) {
            const int cl = cb + 4 * lane;
            const bool fast = vec_ok && cl + 3 < half;  // then c1 + 3 < D as well (D == 2 half)
            float l0[4], l1[4], s0[4], s1[4];
            if (fast) {
                ld4(xrow, cl, d, true, x0);
                ld4(xrow, cl + half, d, true, x1);
                if (eps_from_mem) { ld4(er, cl, D, true, e0); ld4(er, cl + half, D, true, e1); }
                ld4(pk, cl, D, true, l0);
                ld4(pk, cl + half, D, true, l1);
                ld4(pk + D, cl, D, true, s0);
                ld4(pk + D, cl + half, D, true, s1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c0 = cl + i, c1 = c0 + half;
                ok0[i] = c0 < half;
                ok1[i] = ok0[i] && c1 < D;
                if (!fast) {
                    x0[i] = ok0[i] ? (c0 < d ? xrow[c0] : 1.0f) : 0.f;  // column d = intercept
                    x1[i] = ok1[i] ? (c1 < d ? xrow[c1] : 1.0f) : 0.f;
                    l0[i] = ok0[i] ? pk[c0] : 0.f;
                    l1[i] = ok1[i] ? pk[c1] : 0.f;
                    s0[i] = ok0[i] ? pk[D + c0] : 0.f;
                    s1[i] = ok1[i] ? pk[D + c1] : 0.f;
                }
                if (eps_from_mem) {
                    if (!fast) {
                        e0[i] = ok0[i] ? er[c0] : 0.f;
                        e1[i] = ok1[i] ? er[c1] : 0.f;
                    }
                } else {
                    uint32_t b0, b1;
                    threefry2x32(k0, k1, ok0[i] ? (uint32_t)c0 : 0u, ok1[i] ? (uint32_t)c1 : 0u, b0, b1);
                    const float v0 = bits_to_normal_wu(b0), v1 = bits_to_normal_wu(b1);
                    e0[i] = ok0[i] ? v0 : 0.f;
                    e1[i] = ok1[i] ? v1 : 0.f;
                }
                z0[i] = __fmaf_rn(s0[i], e0[i], l0[i]);
                z1[i] = __fmaf_rn(s1[i], e1[i], l1[i]);
                if (gauss) {  // residuals take the place of the features: dloglik/dz = (x - z) / sigma^2
                    x0[i] = ok0[i] ? x0[i] - z0[i] : 0.f;
                    x1[i] = ok1[i] ? x1[i] - z1[i] : 0.f;
                }
            }
        };
        // 4 consecutive entries of derived-column array `arr` (0 .. 4) for both halves of a chunk
        auto pk4 = [&](int arr, int cb, float (&v0)[4], float (&v1)[4]) {
            const int cl = cb + 4 * lane;
            const bool fast = vec_ok && cl + 3 < half;
            ld4(pk + (size_t)arr * D, cl, half, fast, v0);
            ld4(pk + (size_t)arr * D, cl + half, D, fast, v1);
            if (!fast) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (!(cl + i < half)) v1[i] = 0.f;
            }
        };
        auto col_c1 = [&](int c) { return (a.icpt && c == d) ? a.c1_b : a.c1_w; };
        auto col_hz = [&](int c) { return (a.icpt && c == d) ? a.hz_b : a.hz_w; };

        // ---- pass 1: logit and the latent part of the loss
        float tp = 0.f, lp = 0.f;
        for (int cb = 0; cb < half; cb += 256) {
            float x0[4], x1[4], e0[4], e1[4], z0[4], z1[4];
            bool ok0[4], ok1[4];
            chunk(cb, x0, x1, e0, e1, z0, z1, ok0, ok1);
            float lc0[4], lc1[4];
            pk4(4, cb, lc0, lc1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c0 = cb + 4 * lane + i, c1 = c0 + half;
                tp = __fmaf_rn(x0[i], gauss ? x0[i] : z0[i], tp);   // logit x . z, or the squared residual norm
                tp = __fmaf_rn(x1[i], gauss ? x1[i] : z1[i], tp);
                lp += ok0[i] ? __fmaf_rn(col_hz(c0) * z0[i], z0[i], __fmaf_rn(-0.5f * e0[i], e0[i], lc0[i])) : 0.f;
                lp += ok1[i] ? __fmaf_rn(col_hz(c1) * z1[i], z1[i], __fmaf_rn(-0.5f * e1[i], e1[i], lc1[i])) : 0.f;
            }
        }
        const float t = wave_sum(tp);
        lp = wave_sum(lp);
        const float A = gauss ? 2.0f * a.A_scale * a.nh_inv_var : a.A_scale * (sigmoid_f(t) - yv);
        const float loglik = gauss ? __fmaf_rn(a.nh_inv_var, t, -a.ll_const) : yv * t - softplus_f(t);
        const float L = a.inv_obs * (lp - a.lik_scale * loglik);  // svi.py:278-281

        if (PXG) {
            // ---- the example's gradient row as it is (svi.py:291-306; clipping is a later stage)
            float* gr = a.px_grads + (size_t)pp * P;
            for (int cb = 0; cb < half; cb += 256) {
                float x0[4], x1[4], e0[4], e1[4], z0[4], z1[4];
                bool ok0[4], ok1[4];
                chunk(cb, x0, x1, e0, e1, z0, z1, ok0, ok1);
                float sgv0[4], sgv1[4], qv0[4], qv1[4];
                pk4(2, cb, sgv0, sgv1);
                pk4(3, cb, qv0, qv1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c0 = cb + 4 * lane + i, c1 = c0 + half;
                    const float sg0 = ok0[i] ? sgv0[i] : 0.f, sg1 = ok1[i] ? sgv1[i] : 0.f;
                    const float q0 = ok0[i] ? qv0[i] : 0.f, q1 = ok1[i] ? qv1[i] : 0.f;
                    const float g0 = __fmaf_rn(col_c1(c0), z0[i], A * x0[i]), g1 = __fmaf_rn(col_c1(c1), z1[i], A * x1[i]);
                    const float h0 = __fmaf_rn(g0 * e0[i], sg0, -q0), h1 = __fmaf_rn(g1 * e1[i], sg1, -q1);
                    if (ok0[i]) { gr[c0] = g0; gr[D + c0] = h0; }
                    if (ok1[i]) { gr[c1] = g1; gr[D + c1] = h1; }
                }
            }
            if (lane == 0) a.px_loss[pp] = L * a.obs_scale * a.meta[1];  // svi.py:306
            continue;
        }

        // ---- pass 2: squared joint norm of the example's gradient
        float n2 = 0.f;
        for (int cb = 0; cb < half; cb += 256) {
            float x0[4], x1[4], e0[4], e1[4], z0[4], z1[4];
            bool ok0[4], ok1[4];
            chunk(cb, x0, x1, e0, e1, z0, z1, ok0, ok1);
            float sgv0[4], sgv1[4], qv0[4], qv1[4];
            pk4(2, cb, sgv0, sgv1);
            pk4(3, cb, qv0, qv1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c0 = cb + 4 * lane + i, c1 = c0 + half;
                const float sg0 = ok0[i] ? sgv0[i] : 0.f, sg1 = ok1[i] ? sgv1[i] : 0.f;
                const float q0 = ok0[i] ? qv0[i] : 0.f, q1 = ok1[i] ? qv1[i] : 0.f;
                const float g0 = __fmaf_rn(col_c1(c0), z0[i], A * x0[i]), g1 = __fmaf_rn(col_c1(c1), z1[i], A * x1[i]);
                const float h0 = __fmaf_rn(g0 * e0[i], sg0, -q0), h1 = __fmaf_rn(g1 * e1[i], sg1, -q1);
                n2 = __fmaf_rn(g0, g0, n2);
                n2 = __fmaf_rn(h0, h0, n2);
                n2 = __fmaf_rn(g1, g1, n2);
                n2 = __fmaf_rn(h1, h1, n2);
            }
        }
        n2 = wave_sum(n2);
        const float cf = fminf(1.0f, a.clip * __builtin_amdgcn_rsqf(n2));  // svi.py:121-122

        // ---- pass 3: clipped gradient into this wavefront's accumulator row (each lane owns its columns: no conflicts)
        for (int cb = 0; cb < half; cb += 256) {
            float x0[4], x1[4], e0[4], e1[4], z0[4], z1[4];
            bool ok0[4], ok1[4];
            chunk(cb, x0, x1, e0, e1, z0, z1, ok0, ok1);
            float sgv0[4], sgv1[4], qv0[4], qv1[4];
            pk4(2, cb, sgv0, sgv1);
            pk4(3, cb, qv0, qv1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c0 = cb + 4 * lane + i, c1 = c0 + half;
                const float sg0 = ok0[i] ? sgv0[i] : 0.f, sg1 = ok1[i] ? sgv1[i] : 0.f;
                const float q0 = ok0[i] ? qv0[i] : 0.f, q1 = ok1[i] ? qv1[i] : 0.f;
                const float g0 = __fmaf_rn(col_c1(c0), z0[i], A * x0[i]), g1 = __fmaf_rn(col_c1(c1), z1[i], A * x1[i]);
                const float h0 = __fmaf_rn(g0 * e0[i], sg0, -q0), h1 = __fmaf_rn(g1 * e1[i], sg1, -q1);
                if (ok0[i]) { mine[c0] = __fmaf_rn(cf, g0, mine[c0]); mine[D + c0] = __fmaf_rn(cf, h0, mine[D + c0]); }
                if (ok1[i]) { mine[c1] = __fmaf_rn(cf, g1, mine[c1]); mine[D + c1] = __fmaf_rn(cf, h1, mine[D + c1]); }
            }
        }
        loss_acc += L;
        n_acc += 1.0f;
    }

    if (PXG) return;
    if (lane == 0) { tail[2 * wave] = loss_acc; tail[2 * wave + 1] = n_acc; }
    __syncthreads();
    float* out = a.partials + (size_t)blockIdx.x * (P + 2);
    for (int c = threadIdx.x; c < P; c += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < W; ++w) s += acc[(size_t)w * P + c];
        out[c] = s;
    }
    // (workgroup 0: a step parameter that is not finite makes the loss NaN even when no example is valid, as in k_logreg_main)
    // (the vote goes through the accumulator rows, read out by now: the dynamic LDS may be all of the CU's 160 KB, no room for the
    //  static word of __syncthreads_or)
    int p_bad = 0;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < 5 * D; c += blockDim.x) p_bad |= !(fabsf(pk[c]) <= 3.402823466e38f);
        const bool wave_bad = __any(p_bad);
        __syncthreads();
        if (lane == 0) acc[wave] = wave_bad ? 1.0f : 0.0f;
        __syncthreads();
        p_bad = 0;
        if (threadIdx.x == 0)
            for (int w = 0; w < W; ++w) p_bad |= acc[w] != 0.0f;
    }
    if (threadIdx.x < 2) {
        float s = 0.f;
        for (int w = 0; w < W; ++w) s += tail[2 * w + threadIdx.x];
        if (threadIdx.x == 0 && p_bad) s = __builtin_nanf("");
        out[P + threadIdx.x] = s;
    }
}

}  // namespace d3p
