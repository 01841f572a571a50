// Stage-level DP-VI API on materialised tensors (the five stages the reference's tests call
// directly), optimiser steps and the synthetic-table generator.
#include "d3p_logreg_kernel.h"
#include "d3p_logreg_wide.h"

namespace d3p {

__device__ __forceinline__ void adam_update(float& x, float& m, float& v, float g, int i, const d3p_dpsvi_hyper& h)
{
    // jax.example_libraries.optimizers.adam as wrapped by numpyro.optim.Adam
    m = (1.0f - h.b1) * g + h.b1 * m;
    v = (1.0f - h.b2) * g * g + h.b2 * v;
    const float mhat = m / (1.0f - powf(h.b1, (float)(i + 1)));
    const float vhat = v / (1.0f - powf(h.b2, (float)(i + 1)));
    x = x - h.lr * mhat / (sqrtf(vhat) + h.adam_eps);
}

// ------------------------------------------------------------------------------------------
// stage-level kernels on materialised tensors (API parity with the reference's five stages)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mask_meta(const uint8_t* __restrict__ mask, uint32_t B, float* __restrict__ meta)
{
    __shared__ float lds[256];
    float s = 0.f;
    for (uint32_t i = threadIdx.x; i < B; i += 256) s += mask ? (mask[i] != 0 ? 1.f : 0.f) : 1.f;
    lds[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) lds[threadIdx.x] += lds[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n = lds[0];
        meta[0] = n;
        meta[1] = (n == 0.f) ? 0.f : (float)B / n;
    }
}

__global__ void k_clip_rows(float* __restrict__ g, uint32_t B, uint32_t P, float c)
{
    const uint32_t row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    float* r = g + (size_t)row * P;
    float ss = 0.f;
    for (uint32_t j = lane; j < P; j += 64) ss = __fmaf_rn(r[j], r[j], ss);
    ss = wave_sum(ss);
    // svi.py:121-122; jnp.maximum propagates NaN (fmaxf returns the other operand): a row with a NaN entry is NaN throughout, as there
    const float ratio = __fsqrt_rn(ss) / c;
    const float scale = 1.0f / (ratio != ratio ? ratio : fmaxf(1.0f, ratio));
    for (uint32_t j = lane; j < P; j += 64) r[j] *= scale;
}

__global__ void __launch_bounds__(256)
k_combine(const float* __restrict__ g, const float* __restrict__ px_loss, uint32_t B, uint32_t P, float* __restrict__ avg,
          float* __restrict__ loss)
{
    __shared__ float lds[256];
    const int c = threadIdx.x % D3P_FIN_COLS, rg = threadIdx.x / D3P_FIN_COLS;
    const uint32_t col = blockIdx.x * D3P_FIN_COLS + c;
    const float tot = column_sum(g, B, P, col, col < P, c, rg, lds);
    if (rg == 0 && col < P) avg[col] = tot / (float)B;
    if (blockIdx.x == 0 && px_loss && loss) {
        const float l = block_sum_column(px_loss, B, 1, 0, lds);
        if (threadIdx.x == 0) *loss = l / (float)B;
    }
}

__global__ void k_full_norm(const float* __restrict__ v, uint64_t n, float* __restrict__ out)
{
    __shared__ float lds[256];
    float ss = 0.f;
    for (uint64_t j = threadIdx.x; j < n; j += 256) ss = __fmaf_rn(v[j], v[j], ss);
    lds[threadIdx.x] = ss;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) lds[threadIdx.x] += lds[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = __fsqrt_rn(lds[0]);
}

// numpy.linalg.norm(v, ord) of a vector for the orders other than 2 (d3p/svi.py:68-87 hands `ord` through): kind 0: the
// number of non-zero entries; 1: sum |x|; 2: (sum |x|^p)^(1/p); 3: max |x|; 4: min |x|
__global__ void k_full_norm_ord(const float* __restrict__ v, uint64_t n, int kind, float p, float* __restrict__ out)
{
    __shared__ float lds[256];
    float acc = kind == 4 ? __builtin_inff() : 0.f;
    for (uint64_t j = threadIdx.x; j < n; j += 256) {
        const float a = fabsf(v[j]);
        if (kind == 0) acc += (v[j] != 0.f) ? 1.f : 0.f;
        else if (kind == 1) acc += a;
        else if (kind == 2) acc += powf(a, p);
        else if (kind == 3) acc = (a > acc || a != a) ? a : acc;   // (a NaN entry makes the norm NaN, as numpy's max does)
        else acc = (a < acc || a != a) ? a : acc;
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float x = lds[threadIdx.x], y = lds[threadIdx.x + off];
            lds[threadIdx.x] = kind <= 2 ? x + y : kind == 3 ? ((y > x || y != y) ? y : x) : ((y < x || y != y) ? y : x);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = kind == 2 ? powf(lds[0], 1.f / p) : lds[0];
}

__global__ void k_perturb_site(const uint32_t* __restrict__ site_key, const float* __restrict__ avg, uint32_t n_site,
                               float dp_scale, float c, const float* __restrict__ meta, float obs_scale,
                               float* __restrict__ out)
{
    // one thread = one ChaCha block = 16 consecutive elements of the site
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (16ull * b >= n_site) return;
    uint32_t k[16], o[16];
    load_key(site_key, k);
    keystream_block(k, b, o);
    const float n = meta[0], factor = meta[1];
    const float scale = dp_scale * (c / n);
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t e = 16u * b + w;
        if (e < n_site) out[e] = (avg[e] + bits_to_normal(o[w]) * scale) * obs_scale * factor;
    }
}

__global__ void k_perturb_apply(const float* __restrict__ avg, const float* __restrict__ noise, uint64_t n,
                                float dp_scale, float c, const float* __restrict__ meta, float obs_scale,
                                float* __restrict__ out)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float scale = dp_scale * (c / meta[0]);
    out[e] = (avg[e] + noise[e] * scale) * obs_scale * meta[1];
}

__global__ void k_adam(float* __restrict__ x, float* __restrict__ m, float* __restrict__ v, const int32_t* __restrict__ step,
                       const float* __restrict__ g, uint32_t P, d3p_dpsvi_hyper h)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P) return;
    float xx = x[j], mm = m[j], vv = v[j];
    adam_update(xx, mm, vv, g[j], *step, h);
    x[j] = xx;
    m[j] = mm;
    v[j] = vv;
}

// ---- ADADP (d3p/optimizers.py:29-112).  Odd steps need err = ||(x_stepped - new_x) / max(1, x_stepped)||_2 over
// the whole vector before any element can be accepted or rejected: k_adadp_err leaves one partial sum of squares per
// workgroup (fixed grid -> fixed summation order), k_adadp_apply re-reduces them in every workgroup.
#define D3P_ADADP_BLOCKS 64

__global__ void __launch_bounds__(256) k_adadp_err(const float* __restrict__ x, const float* __restrict__ lr,
                                                   const float* __restrict__ x_stepped, const int32_t* __restrict__ step,
                                                   const float* __restrict__ g, uint32_t P, float* __restrict__ partials)
{
    __shared__ float red[256];
    if ((*step & 1) == 0) return;  // even steps take no error estimate (optimizers.py:62-69)
    const float half_lr = 0.5f * *lr;
    float s = 0.f;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < P; j += D3P_ADADP_BLOCKS * 256) {
        const float nx = x[j] - half_lr * g[j];
        const float xs = x_stepped[j];
        const float e = (xs - nx) / fmaxf(1.0f, xs);  // max(1, x), as the reference (no absolute value)
        s = __fmaf_rn(e, e, s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) k_adadp_apply(float* __restrict__ x, const float* __restrict__ lr,
                                                     float* __restrict__ x_stepped, float* __restrict__ x_prev,
                                                     const int32_t* __restrict__ step, const float* __restrict__ g, uint32_t P,
                                                     float tol, int stability_check, const float* __restrict__ partials,
                                                     float* __restrict__ lr_next)
{
    const float l = *lr;
    const bool odd = (*step & 1) != 0;
    bool reject = false;
    if (odd) {
        float ss = 0.f;
        for (int b = 0; b < D3P_ADADP_BLOCKS; ++b) ss += partials[b];
        const float err = sqrtf(ss);
        reject = stability_check && err > tol;
        // the literals 0.9 / 1.1 are the reference's (optimizers.py:89-91 ignores alpha_min / alpha_max)
        if (blockIdx.x == 0 && threadIdx.x == 0) *lr_next = l * fminf(fmaxf(sqrtf(tol / err), 0.9f), 1.1f);
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        *lr_next = l;
    }
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < P; j += gridDim.x * 256) {
        const float xx = x[j], gg = g[j];
        const float nx = xx - (0.5f * l) * gg;
        if (odd) {
            x[j] = reject ? x_prev[j] : nx;
        } else {
            x_prev[j] = xx;
            x_stepped[j] = xx - l * gg;
            x[j] = nx;
        }
    }
}

__global__ void k_adadp_commit(float* lr, const float* lr_next, int32_t* step)
{
    *lr = *lr_next;
    *step += 1;
}

// ---- DPSVI.evaluate (svi.py:436-449): -ELBO of a batch with ONE guide draw
// k_eval_latent: key plumbing, eps, z = loc + softplus(u) * eps, latent[D] = logq - logp summed -> lat[0]
// sites: the guide's sample sites in program order (AutoDiagonalNormal / one-site guides: ONE site of D elements; the example's own
// guide, examples/logistic_regression.py:67-86: 'w' (d) then 'intercept' (1)) -- numpyro's seed handler advances
// rng, site_key = split(rng) per site, and a site's eps is normal(site_key, (size,)).
struct EvalSites {
    int n;
    int size[8];
};
__global__ void __launch_bounds__(256) k_eval_latent(d3p_logreg_model m, const float* __restrict__ params,
                                                     const uint32_t* __restrict__ jax_key, float* __restrict__ z,
                                                     float* __restrict__ lat, EvalSites sites)
{
    __shared__ float red[256];
    const int D = m.d + (m.intercept ? 1 : 0);
    // rng_key_eval = split(key)[1]; guide_seed = split(.)[1]; then per site: rng, sample key = split(rng)  (jax split(k, 2)[1] =
    // (threefry(k, (0, 2))[1], threefry(k, (1, 3))[1]), [0] = the first outputs)
    uint32_t r0 = jax_key[0], r1 = jax_key[1];
#pragma unroll
    for (int lvl = 0; lvl < 2; ++lvl) {
        uint32_t a, b0, b1;
        threefry2x32(r0, r1, 0u, 2u, a, b0);
        threefry2x32(r0, r1, 1u, 3u, a, b1);
        r0 = b0;
        r1 = b1;
    }
    float acc = 0.f;
    int off = 0;
    for (int s = 0; s < sites.n; ++s) {
        uint32_t c0, c1, k0, k1;
        threefry2x32(r0, r1, 0u, 2u, c0, k0);
        threefry2x32(r0, r1, 1u, 3u, c1, k1);
        r0 = c0;
        r1 = c1;
        const int n = sites.size[s];
        for (int j = threadIdx.x; j < n; j += 256) {
            const int e = off + j;
            const float eps = bits_to_normal(tf_iota_word(k0, k1, (uint64_t)n, (uint64_t)j));
            float sc, dsc;
            guide_scale(m.guide_transform, params[D + e], sc, dsc);
            const float ps = (e < m.d) ? m.prior_w : m.prior_b;
            const float zz = __fmaf_rn(sc, eps, params[e]);
            z[e] = zz;
            acc += (-0.5f * eps * eps - logf(sc)) - (-0.5f * (zz / ps) * (zz / ps) - logf(ps));
        }
        off += n;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off2 = 128; off2 > 0; off2 >>= 1) {
        if ((int)threadIdx.x < off2) red[threadIdx.x] += red[threadIdx.x + off2];
        __syncthreads();
    }
    if (threadIdx.x == 0) lat[0] = red[0];
}

// one wavefront per example: loglik_i = y t - softplus(t), t = x . z (+ intercept)
__global__ void k_eval_loglik(d3p_logreg_model m, const float* __restrict__ X, const float* __restrict__ y,
                              const float* __restrict__ z, uint32_t B, float* __restrict__ ll)
{
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= B) return;
    float tp = 0.f;
    if (m.family == D3P_FAMILY_GAUSS_MEAN) {  // log N(x_i; z, sigma) summed over the event dimension
        for (int c = lane; c < m.d; c += 64) {
            const float r = X[(size_t)i * m.d + c] - z[c];
            tp = __fmaf_rn(r, r, tp);
        }
        const float t = wave_sum(tp);
        if (lane == 0)
            ll[i] = -0.5f * t / (m.lik_sigma * m.lik_sigma) - (float)m.d * (logf(m.lik_sigma) + 0.91893853320467267f);
        return;
    }
    for (int c = lane; c < m.d; c += 64) tp = __fmaf_rn(X[(size_t)i * m.d + c], z[c], tp);
    float t = wave_sum(tp);
    if (m.intercept) t += z[m.d];
    if (lane == 0) ll[i] = y[i] * t - softplus_f(t);
}

__global__ void __launch_bounds__(256) k_eval_finish(d3p_logreg_model m, const float* __restrict__ ll, uint32_t B,
                                                     const float* __restrict__ lat, float* __restrict__ loss)
{
    __shared__ float red[256];
    float s = 0.f;
    for (uint32_t i = threadIdx.x; i < B; i += 256) s += ll[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    // -ELBO = (logq - logp) - (N / B) * sum_i loglik_i
    if (threadIdx.x == 0) *loss = lat[0] - (m.lik_scale / (float)B) * red[0];
}

// jax.random.randint layout for the debug suite (d3p/random/debug.py:39)
__global__ void k_tf_randint(const uint32_t* __restrict__ key, uint64_t n, uint32_t minval, uint32_t span, uint32_t mult,
                             int32_t* __restrict__ out)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t a, k10, k11, k20, k21;
    threefry2x32(key[0], key[1], 0u, 2u, k10, k20);  // split(key, 2): counts [0,1 | 2,3]
    threefry2x32(key[0], key[1], 1u, 3u, k11, k21);
    (void)a;
    const uint32_t hi = tf_iota_word(k10, k11, n, j), lo = tf_iota_word(k20, k21, n, j);
    const uint32_t off = ((hi % span) * mult + (lo % span)) % span;
    out[j] = (int32_t)(minval + off);
}

// GaussianMixture.log_prob (d3p/gmm.py:71-86): one wavefront per row; lanes stride the event dimension,
// the K component sums go through the fixed-order wave reduction, logsumexp over components is wave-uniform.
template <int KMAX>
__global__ void k_gmm_log_prob(const float* __restrict__ x, uint32_t B, int d, const float* __restrict__ locs,
                               const float* __restrict__ scales, const float* __restrict__ pis, int K,
                               float* __restrict__ out)
{
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= B) return;
    float comp[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) comp[k] = 0.f;
    for (int j = lane; j < d; j += 64) {
        const float xv = x[(size_t)i * d + j];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                const float sc = scales[(size_t)k * d + j];
                const float z = (xv - locs[(size_t)k * d + j]) / sc;
                comp[k] += -0.5f * z * z - logf(sc) - D3P_HALF_LOG_2PI;
            }
        }
    }
    float best = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            comp[k] = wave_sum(comp[k]) + logf(pis[k]);
            best = fmaxf(best, comp[k]);
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) acc += expf(comp[k] - best);
    if (lane == 0) out[i] = best + logf(acc);
}

__global__ void k_incr_i32(int32_t* p) { *p += 1; }

__global__ void k_sgd(float* __restrict__ x, const float* __restrict__ g, uint32_t P, float lr)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < P) x[j] = x[j] - lr * g[j];
}

// synthetic table: element (r, c) = f(seed, r, c)   
__global__ void k_synth_logreg(uint32_t seed, uint64_t row0, uint64_t n_rows, int d, float* __restrict__ X,
                               float* __restrict__ y)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= n_rows) return;
    const uint64_t r = row0 + i;
    float tp = 0.f;
    uint32_t a, b;
    for (int c = lane; c < d; c += 64) {
        threefry2x32(seed, 0x58u, (uint32_t)r, (uint32_t)c, a, b);
        const float xv = bits_to_normal(a);
        X[i * (uint64_t)d + c] = xv;
        threefry2x32(seed, 0x57u, (uint32_t)c, 0u, a, b);
        tp = __fmaf_rn(xv, bits_to_normal(a), tp);
    }
    float t = wave_sum(tp);
    if (lane == 0) {
        threefry2x32(seed, 0x57u, (uint32_t)d, 0u, a, b);
        t += bits_to_normal(a);
        threefry2x32(seed, 0x59u, (uint32_t)r, 0u, a, b);
        const float u = bits_to_uniform(a, 0.0f, 1.0f);
        y[i] = (u < sigmoid_f(t)) ? 1.0f : 0.0f;
    }
}

__global__ void k_px_keys(const uint32_t* __restrict__ jax_key, uint32_t B, uint32_t* __restrict__ skeys)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    uint32_t s0, s1;
    px_sample_key(jax_key[0], jax_key[1], B, p, s0, s1);
    skeys[2 * p] = s0;
    skeys[2 * p + 1] = s1;
}

}  // namespace d3p

using namespace d3p;

extern "C" {

static size_t px_ws_bytes(const d3p_logreg_model* model, uint32_t B)
{
    const size_t D = (size_t)model->d + (model->intercept ? 1 : 0);
    return align_up(5 * D * sizeof(float), 256) + align_up(2 * (size_t)B * sizeof(uint32_t), 256);
}

size_t d3p_logreg_px_grads_workspace(const d3p_logreg_model* model, uint32_t B)
{
    return model ? px_ws_bytes(model, B) : 0;
}

int d3p_logreg_px_grads(void* stream, const d3p_logreg_model* model, const float* params_dev, const float* X_dev,
                        const float* y_dev, const uint8_t* mask_dev, uint32_t B, const float* eps_dev,
                        const uint32_t* jax_key_dev, float* px_loss_dev, float* px_grads_dev, float* meta_dev,
                        void* workspace_dev, size_t workspace_bytes)
{
    D3P_REQUIRE(model && params_dev && X_dev && px_loss_dev && px_grads_dev && meta_dev && workspace_dev,
                "d3p_logreg_px_grads: null pointer");
    if (int rc = validate_model(model, y_dev, "d3p_logreg_px_grads")) return rc;
    D3P_REQUIRE(eps_dev || jax_key_dev, "d3p_logreg_px_grads: either eps_dev or jax_key_dev must be given");
    D3P_REQUIRE(B >= 1, "d3p_logreg_px_grads: B must be >= 1");
    if (workspace_bytes < px_ws_bytes(model, B))
        return fail(D3P_E_WORKSPACE, "d3p_logreg_px_grads: workspace too small (%zu < %zu)", workspace_bytes,
                    px_ws_bytes(model, B));
    hipStream_t s = (hipStream_t)stream;
    const int D = model->d + (model->intercept ? 1 : 0);
    float* pack = (float*)workspace_dev;
    uint32_t* skeys = (uint32_t*)((char*)workspace_dev + align_up(5 * (size_t)D * sizeof(float), 256));
    MainGeom g;
    int rc = main_geometry(model, B, &g, true, false);  // (rows too wide for the register-tiled kernel: the column-chunked one)
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack, dim3(cdiv(D, 256)), dim3(256), 0, s, *model, params_dev, pack);
    hipLaunchKernelGGL(k_mask_meta, dim3(1), dim3(256), 0, s, mask_dev, B, meta_dev);
    if (!eps_dev) hipLaunchKernelGGL(k_px_keys, dim3(cdiv(B, 256)), dim3(256), 0, s, jax_key_dev, B, skeys);
    MainArgs a;
    memset(&a, 0, sizeof(a));
    fill_model_scalars(model, &a);
    a.X = X_dev;
    a.y = y_dev;
    a.mask = mask_dev;
    a.skeys = skeys;
    a.eps_ext = eps_dev;
    a.pack = pack;
    a.px_grads = px_grads_dev;
    a.px_loss = px_loss_dev;
    a.meta = meta_dev;
    a.B = B;
    a.row_lo = 0;
    a.row_hi = B;
    a.clip = 1.0f;
    if (g.wide) {
        hipLaunchKernelGGL(k_logreg_wide<true>, dim3(g.blocks), dim3(64 * D3P_WIDE_W), 0, s, a);
        return check_launch("k_logreg_wide");
    }
    return launch_main<1>(s, g, a);
}

int d3p_clip_rows(void* stream, float* px_grads_dev, uint32_t B, uint32_t P, float c)
{
    D3P_REQUIRE(c != 0.0f, "The clipping threshold must be greater than 0.");  // svi.py:119-120
    D3P_REQUIRE(px_grads_dev || B == 0 || P == 0, "d3p_clip_rows: null pointer");
    if (B == 0 || P == 0) return D3P_OK;
    hipLaunchKernelGGL(k_clip_rows, dim3(cdiv((uint64_t)B * 64, 256)), dim3(256), 0, (hipStream_t)stream, px_grads_dev,
                       B, P, c);
    return check_launch("d3p_clip_rows");
}

int d3p_full_norm(void* stream, const float* v_dev, uint64_t n, float* out_dev, void* workspace_dev,
                  size_t workspace_bytes)
{
    (void)workspace_dev;
    (void)workspace_bytes;
    D3P_REQUIRE(out_dev && (v_dev || n == 0), "d3p_full_norm: null pointer");
    hipLaunchKernelGGL(k_full_norm, dim3(1), dim3(256), 0, (hipStream_t)stream, v_dev, n, out_dev);
    return check_launch("d3p_full_norm");
}

int d3p_full_norm_ord(void* stream, const float* v_dev, uint64_t n, double ord, float* out_dev)
{
    D3P_REQUIRE(out_dev && v_dev && n >= 1, "d3p_full_norm_ord: null pointer or empty vector");
    D3P_REQUIRE(ord == ord, "d3p_full_norm_ord: ord is NaN");
    if (ord == 2.0) return d3p_full_norm(stream, v_dev, n, out_dev, nullptr, 0);
    const int kind = ord == 0.0 ? 0 : ord == 1.0 ? 1 : std::isinf(ord) ? (ord > 0 ? 3 : 4) : 2;
    hipLaunchKernelGGL(k_full_norm_ord, dim3(1), dim3(256), 0, (hipStream_t)stream, v_dev, n, kind, (float)ord, out_dev);
    return check_launch("d3p_full_norm_ord");
}

int d3p_combine(void* stream, const float* px_grads_dev, const float* px_loss_dev, uint32_t B, uint32_t P,
                float* avg_dev, float* loss_dev)
{
    D3P_REQUIRE(px_grads_dev && avg_dev, "d3p_combine: null pointer");
    D3P_REQUIRE(B >= 1 && P >= 1, "d3p_combine: empty input");
    hipLaunchKernelGGL(k_combine, dim3(cdiv(P, D3P_FIN_COLS)), dim3(256), 0, (hipStream_t)stream, px_grads_dev,
                       px_loss_dev, B, P, avg_dev, loss_dev);
    return check_launch("d3p_combine");
}

int d3p_perturb(void* stream, const uint32_t* key_dev, const float* avg_dev, const int32_t* site_sizes_host,
                int n_sites, float dp_scale, float c, const float* meta_dev, float obs_scale, float* out_dev,
                uint32_t* site_keys_dev)
{
    D3P_REQUIRE(key_dev && avg_dev && site_sizes_host && meta_dev && out_dev && site_keys_dev,
                "d3p_perturb: null pointer");
    D3P_REQUIRE(n_sites >= 1, "d3p_perturb: need at least one site");
    int rc = d3p_rng_split(stream, key_dev, n_sites, site_keys_dev);  // svi.py:491
    if (rc) return rc;
    size_t off = 0;
    for (int k = 0; k < n_sites; ++k) {
        const int32_t n = site_sizes_host[k];
        D3P_REQUIRE(n >= 0, "d3p_perturb: negative site size");
        if (n > 0)
            hipLaunchKernelGGL(k_perturb_site, dim3(cdiv(cdiv(n, 16), 128)), dim3(128), 0, (hipStream_t)stream,
                               (const uint32_t*)(site_keys_dev + 16 * k), avg_dev + off, (uint32_t)n, dp_scale, c,
                               meta_dev, obs_scale, out_dev + off);
        off += (size_t)n;
    }
    return check_launch("d3p_perturb");
}

int d3p_perturb_apply(void* stream, const float* avg_dev, const float* noise_dev, uint64_t n, float dp_scale, float c,
                      const float* meta_dev, float obs_scale, float* out_dev)
{
    D3P_REQUIRE((avg_dev && noise_dev && out_dev) || n == 0, "d3p_perturb_apply: null pointer");
    D3P_REQUIRE(meta_dev, "d3p_perturb_apply: null meta");
    if (n == 0) return D3P_OK;
    hipLaunchKernelGGL(k_perturb_apply, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, avg_dev, noise_dev, n,
                       dp_scale, c, meta_dev, obs_scale, out_dev);
    return check_launch("d3p_perturb_apply");
}

int d3p_adam_step(void* stream, float* params_dev, float* m_dev, float* v_dev, int32_t* step_dev,
                  const float* grads_dev, uint32_t P, float lr, float b1, float b2, float eps)
{
    D3P_REQUIRE(params_dev && m_dev && v_dev && step_dev && grads_dev, "d3p_adam_step: null pointer");
    d3p_dpsvi_hyper h;
    h.clip = 1.f; h.dp_scale = 0.f; h.lr = lr; h.b1 = b1; h.b2 = b2; h.adam_eps = eps;
    if (P > 0)
        hipLaunchKernelGGL(k_adam, dim3(cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, params_dev, m_dev, v_dev,
                           (const int32_t*)step_dev, grads_dev, P, h);
    hipLaunchKernelGGL(k_incr_i32, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    return check_launch("d3p_adam_step");
}

int d3p_sgd_step(void* stream, float* params_dev, int32_t* step_dev, const float* grads_dev, uint32_t P, float lr)
{
    D3P_REQUIRE(params_dev && step_dev && grads_dev, "d3p_sgd_step: null pointer");
    if (P > 0)
        hipLaunchKernelGGL(k_sgd, dim3(cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, params_dev, grads_dev, P, lr);
    hipLaunchKernelGGL(k_incr_i32, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    return check_launch("d3p_sgd_step");
}

size_t d3p_adadp_workspace(void) { return (D3P_ADADP_BLOCKS + 1) * sizeof(float); }

int d3p_adadp_step(void* stream, float* params_dev, float* lr_dev, float* x_stepped_dev, float* x_prev_dev, int32_t* step_dev,
                   const float* grads_dev, uint32_t P, float tol, int stability_check, void* workspace_dev,
                   size_t workspace_bytes)
{
    D3P_REQUIRE(params_dev && lr_dev && x_stepped_dev && x_prev_dev && step_dev && grads_dev && workspace_dev,
                "d3p_adadp_step: null pointer");
    if (workspace_bytes < d3p_adadp_workspace()) return fail(D3P_E_WORKSPACE, "d3p_adadp_step: workspace too small");
    float* partials = (float*)workspace_dev;
    float* lr_next = partials + D3P_ADADP_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_adadp_err, dim3(D3P_ADADP_BLOCKS), dim3(256), 0, s, (const float*)params_dev, (const float*)lr_dev,
                       (const float*)x_stepped_dev, (const int32_t*)step_dev, grads_dev, P, partials);
    const unsigned blocks = P == 0 ? 1u : (cdiv(P, 256) < 1024u ? cdiv(P, 256) : 1024u);
    hipLaunchKernelGGL(k_adadp_apply, dim3(blocks), dim3(256), 0, s, params_dev, (const float*)lr_dev, x_stepped_dev, x_prev_dev,
                       (const int32_t*)step_dev, grads_dev, P, tol, stability_check, (const float*)partials, lr_next);
    hipLaunchKernelGGL(k_adadp_commit, dim3(1), dim3(1), 0, s, lr_dev, (const float*)lr_next, step_dev);
    return check_launch("d3p_adadp_step");
}

size_t d3p_logreg_evaluate_workspace(const d3p_logreg_model* model, uint32_t B)
{
    if (!model) return 0;
    const size_t D = (size_t)model->d + (model->intercept ? 1 : 0);
    return align_up(D * sizeof(float), 256) + align_up((size_t)B * sizeof(float), 256) + 256;
}

int d3p_logreg_evaluate(void* stream, const d3p_logreg_model* model, const float* params_dev, const float* X_dev,
                        const float* y_dev, uint32_t B, const uint32_t* jax_key_dev, float* loss_dev, void* workspace_dev,
                        size_t workspace_bytes)
{
    return d3p_logreg_evaluate_sites(stream, model, params_dev, X_dev, y_dev, B, jax_key_dev, nullptr, 1, loss_dev, workspace_dev, workspace_bytes);
}

int d3p_logreg_evaluate_sites(void* stream, const d3p_logreg_model* model, const float* params_dev, const float* X_dev,
                              const float* y_dev, uint32_t B, const uint32_t* jax_key_dev, const int32_t* site_sizes_host, int32_t n_sites,
                              float* loss_dev, void* workspace_dev, size_t workspace_bytes)
{
    D3P_REQUIRE(model && params_dev && X_dev && jax_key_dev && loss_dev && workspace_dev,
                "d3p_logreg_evaluate: null pointer");
    D3P_REQUIRE(B >= 1, "d3p_logreg_evaluate: B must be >= 1");
    if (int rc = validate_model(model, y_dev, "d3p_logreg_evaluate")) return rc;
    if (workspace_bytes < d3p_logreg_evaluate_workspace(model, B))
        return fail(D3P_E_WORKSPACE, "d3p_logreg_evaluate: workspace too small");
    const size_t D = (size_t)model->d + (model->intercept ? 1 : 0);
    EvalSites sites;
    memset(&sites, 0, sizeof(sites));
    if (!site_sizes_host) {
        D3P_REQUIRE(n_sites == 1, "d3p_logreg_evaluate_sites: site sizes are needed for more than one site");
        sites.n = 1;
        sites.size[0] = (int)D;
    } else {
        D3P_REQUIRE(n_sites >= 1 && n_sites <= 8, "d3p_logreg_evaluate_sites: 1 <= n_sites <= 8");
        size_t total = 0;
        for (int i = 0; i < n_sites; ++i) {
            D3P_REQUIRE(site_sizes_host[i] >= 1, "d3p_logreg_evaluate_sites: a site has at least one element");
            sites.size[i] = site_sizes_host[i];
            total += (size_t)site_sizes_host[i];
        }
        D3P_REQUIRE(total == D, "d3p_logreg_evaluate_sites: the site sizes must add up to the latent dimension (d + intercept)");
        sites.n = n_sites;
    }
    float* z = (float*)workspace_dev;
    float* ll = (float*)((char*)workspace_dev + align_up(D * sizeof(float), 256));
    float* lat = (float*)((char*)ll + align_up((size_t)B * sizeof(float), 256));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_eval_latent, dim3(1), dim3(256), 0, s, *model, params_dev, jax_key_dev, z, lat, sites);
    hipLaunchKernelGGL(k_eval_loglik, dim3(cdiv((uint64_t)B * 64, 256)), dim3(256), 0, s, *model, X_dev, y_dev,
                       (const float*)z, B, ll);
    hipLaunchKernelGGL(k_eval_finish, dim3(1), dim3(256), 0, s, *model, (const float*)ll, B, (const float*)lat, loss_dev);
    return check_launch("d3p_logreg_evaluate");
}

int d3p_tf_randint(void* stream, const uint32_t* key_dev, uint64_t n, int32_t minval, int32_t maxval, int32_t* out_dev)
{
    D3P_REQUIRE(key_dev && (out_dev || n == 0), "d3p_tf_randint: null pointer");
    if (n == 0) return D3P_OK;
    const uint32_t span = (maxval <= minval) ? 1u : (uint32_t)maxval - (uint32_t)minval;
    uint32_t mult = 65536u % span;
    mult = (mult * mult) % span;
    hipLaunchKernelGGL(k_tf_randint, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, key_dev, n, (uint32_t)minval,
                       span, mult, out_dev);
    return check_launch("d3p_tf_randint");
}

int d3p_gmm_log_prob(void* stream, const float* x_dev, uint32_t B, int32_t d, const float* locs_dev,
                     const float* scales_dev, const float* pis_dev, int32_t K, float* out_dev)
{
    D3P_REQUIRE(x_dev && locs_dev && scales_dev && pis_dev && (out_dev || B == 0), "d3p_gmm_log_prob: null pointer");
    D3P_REQUIRE(d >= 1 && K >= 1, "d3p_gmm_log_prob: d and K must be >= 1");
    if (K > 64) return fail(D3P_E_UNSUPPORTED, "d3p_gmm_log_prob: at most 64 mixture components are supported (K = %d)", K);
    if (B == 0) return D3P_OK;
    const dim3 grid(cdiv((uint64_t)B * 64, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (K <= 4)
        hipLaunchKernelGGL(k_gmm_log_prob<4>, grid, block, 0, s, x_dev, B, d, locs_dev, scales_dev, pis_dev, K, out_dev);
    else if (K <= 16)
        hipLaunchKernelGGL(k_gmm_log_prob<16>, grid, block, 0, s, x_dev, B, d, locs_dev, scales_dev, pis_dev, K, out_dev);
    else
        hipLaunchKernelGGL(k_gmm_log_prob<64>, grid, block, 0, s, x_dev, B, d, locs_dev, scales_dev, pis_dev, K, out_dev);
    return check_launch("d3p_gmm_log_prob");
}

// Self-test of the cross-lane sums every step kernel is built on (d3p_device.h: wave_sum, wave_sum2 -- DPP adds issued from inline
// assembly, whose hazards the compiler cannot see): one wave per 64 inputs; the sums are taken directly behind a DIVERGENT
// branch (odd lanes do extra dependent arithmetic first, then all lanes reconverge), the situation in which an EXEC write sits
// closest in front of the DPP block.  out[3 w + {0, 1, 2}] = wave_sum(x), and the two sums of wave_sum2(x, 2 x).
__global__ void __launch_bounds__(256) k_selftest_wave_sums(const float* __restrict__ in, uint32_t n_waves, float* __restrict__ out)
{
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n_waves) return;
    float x = in[(size_t)w * 64 + lane];
    if (lane & 1) {   // divergent: a few dependent operations on half of the lanes, then undone exactly
        float t = x;
#pragma unroll
        for (int i = 0; i < 4; ++i) t = t * 2.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) t = t * 0.5f;
        x = t;
    }
    const float s = wave_sum(x);
    float sa, sb;
    if (lane & 2) x = x + 0.0f;   // (a second divergent region directly in front of wave_sum2)
    wave_sum2(x, 2.0f * x, sa, sb);
    if (lane == 0) {
        out[3 * (size_t)w] = s;
        out[3 * (size_t)w + 1] = sa;
        out[3 * (size_t)w + 2] = sb;
    }
}

int d3p_selftest_wave_sums(void* stream, const float* in_dev, uint32_t n_waves, float* out_dev)
{
    D3P_REQUIRE(in_dev && out_dev, "d3p_selftest_wave_sums: null pointer");
    if (n_waves == 0) return D3P_OK;
    hipLaunchKernelGGL(k_selftest_wave_sums, dim3(cdiv((uint64_t)n_waves * 64, 256)), dim3(256), 0, (hipStream_t)stream, in_dev, n_waves, out_dev);
    return check_launch("d3p_selftest_wave_sums");
}

int d3p_synth_logreg(void* stream, uint32_t seed, uint64_t row0, uint64_t n_rows, int32_t d, float* X_dev,
                     float* y_dev)
{
    D3P_REQUIRE(X_dev && y_dev && d >= 1, "d3p_synth_logreg: bad arguments");
    if (n_rows == 0) return D3P_OK;
    // at most 2^31 threads per launch: chunk the rows
    const uint64_t chunk = 1ull << 22;
    for (uint64_t r = 0; r < n_rows; r += chunk) {
        const uint64_t n = (n_rows - r < chunk) ? n_rows - r : chunk;
        hipLaunchKernelGGL(k_synth_logreg, dim3(cdiv(n * 64, 256)), dim3(256), 0, (hipStream_t)stream, seed, row0 + r, n,
                           d, X_dev + r * (uint64_t)d, y_dev + r);
    }
    return check_launch("d3p_synth_logreg");
}

// ---- measurement aid ("also measure a device-to-device copy peak on the box and report both fractions")
// A plain streaming copy: thread <-> W bytes per load, U loads in flight before the first store, grid-stride.  W = 16 is the
// figure the guide quotes as the achievable HBM rate (float4 copy); W = 4 / 8 exist so that the FETCH_SIZE / WRITE_SIZE
// counters can be calibrated per access width on a known byte count (tools/probes/fetch_calibration.py).
extern "C++" {
typedef unsigned int d3p_copy16 __attribute__((ext_vector_type(4)));
template <typename T, int U, bool NT>
__global__ void __launch_bounds__(256) k_hbm_copy(T* __restrict__ dst, const T* __restrict__ src, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
}  // extern "C++"

int d3p_hbm_copy(void* stream, void* dst_dev, const void* src_dev, uint64_t bytes, int32_t bytes_per_lane)
{
    D3P_REQUIRE(dst_dev && src_dev, "d3p_hbm_copy: null pointer");
    D3P_REQUIRE(bytes_per_lane == 4 || bytes_per_lane == 8 || bytes_per_lane == 16, "d3p_hbm_copy: bytes_per_lane must be 4, 8 or 16");
    D3P_REQUIRE(bytes % 16 == 0 && ((uintptr_t)dst_dev % 16) == 0 && ((uintptr_t)src_dev % 16) == 0, "d3p_hbm_copy: 16-byte aligned buffers and size");
    if (bytes == 0) return D3P_OK;
    hipStream_t s = (hipStream_t)stream;
    // developer switches (probing the box's copy rate: tools/probes/copy_peak_probe.py), read per call: workgroups of the grid-stride
    // launch (default 32 per CU) and plain instead of nontemporal loads / stores
    const char* eg = getenv("D3P_COPY_GRID");
    const unsigned grid = eg && atoi(eg) > 0 ? (unsigned)atoi(eg) : 256u * 32u;   // (8192: the best of the grids probed, profiles/r06_copy_peak_probe.json)
    const bool nt = getenv("D3P_COPY_TEMPORAL") == nullptr;
#define D3P_COPY_LAUNCH(T_, n_)                                                                                                   \
    do {                                                                                                                          \
        if (nt) hipLaunchKernelGGL((k_hbm_copy<T_, 4, true>), dim3(grid), dim3(256), 0, s, (T_*)dst_dev, (const T_*)src_dev, n_); \
        else hipLaunchKernelGGL((k_hbm_copy<T_, 4, false>), dim3(grid), dim3(256), 0, s, (T_*)dst_dev, (const T_*)src_dev, n_);   \
    } while (0)
    if (bytes_per_lane == 16) D3P_COPY_LAUNCH(d3p_copy16, bytes / 16);
    else if (bytes_per_lane == 8) D3P_COPY_LAUNCH(unsigned long long, bytes / 8);
    else D3P_COPY_LAUNCH(uint32_t, bytes / 4);
#undef D3P_COPY_LAUNCH
    return check_launch("d3p_hbm_copy");
}


// ---- DPSVI.update for a parameter dict with SEVERAL leaves (the example's own guide: four leaves) around the fused clipped sums:
// d3p_dpvi_leaves_begin (the step's keys + the parameters in the kernels' column order) -> d3p_px_eps_sites -> d3p_dpvi_logreg_local_sums
// -> d3p_dpvi_leaves_finalize (mean, one noise key per leaf, rescaling, Adam).  Nine launches instead of the ~95 of the stage-wise composition.
#define D3P_MAX_LEAVES 16
extern "C++" {
struct LeavesBeginArgs {
    const uint32_t* key;
    int n_leaves;
    const float* params_tree;
    const int32_t* col_of;
    uint32_t P;
    uint32_t* next_key;
    uint32_t* jax_key;
    uint32_t* leaf_keys;
    float* params_kernel;
};

// Workgroup 0: (next key, gradient key, perturbation key) = split(state key, 3) (svi.py:413-415), the gradient key's two threefry
// words (random/__init__.py:149-155) and split(perturbation key, n_leaves) (svi.py:491).  The others: kernel column col_of[j] <- leaf element j.
__global__ void __launch_bounds__(64) k_leaves_begin(LeavesBeginArgs a)
{
    const int t = threadIdx.x;
    if (blockIdx.x != 0) {
        const uint32_t j = (blockIdx.x - 1) * 64u + t;
        if (j < a.P) a.params_kernel[a.col_of[j]] = a.params_tree[j];
        return;
    }
    __shared__ uint32_t kids[3][16];
    if (t < 3) {
        uint32_t parent[16], child[16];
        load_key(a.key, parent);
        derive_child(parent, (uint32_t)t, 0u, D3P_TAG_SPLIT, child);
#pragma unroll
        for (int w = 0; w < 16; ++w) kids[t][w] = child[w];
        if (t == 0) {
#pragma unroll
            for (int w = 0; w < 16; ++w) a.next_key[w] = child[w];
        }
    }
    __syncthreads();
    if (t == 0) {
        uint32_t k[16], o[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) k[w] = kids[1][w];
        keystream_block(k, 0u, o);
        a.jax_key[0] = o[0];
        a.jax_key[1] = o[1];
    } else if (t - 1 < a.n_leaves) {
        uint32_t parent[16], child[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) parent[w] = kids[2][w];
        derive_child(parent, (uint32_t)(t - 1), 0u, D3P_TAG_SPLIT, child);
#pragma unroll
        for (int w = 0; w < 16; ++w) a.leaf_keys[16 * (t - 1) + w] = child[w];
    }
}

struct LeavesFinalArgs {
    const float* sums;      // [P clipped sums, kernel column order | loss sum | n]
    const int32_t* col_of;
    const uint32_t* leaf_keys;
    uint32_t leaf_off[D3P_MAX_LEAVES + 1];
    int n_leaves;
    uint32_t P, B;
    float obs_scale;
    d3p_dpsvi_hyper h;
    const float *x_in, *m_in, *v_in;
    const int32_t* step_in;
    float *x_out, *m_out, *v_out;
    int32_t* step_out;
    float* loss_out;
    float* grad_out;  // nullable
};

// One thread per leaf element j (tree order): svi.py:343-346 (mean), :365-366 + :487-488 (leaf += normal(leaf key) * scale: word e of the
// leaf key's stream, as d3p_rng_normal draws it), :375 (rescaling), :379-393 (Adam).
__global__ void __launch_bounds__(64) k_leaves_finalize(LeavesFinalArgs a)
{
    const uint32_t j = blockIdx.x * 64u + threadIdx.x;
    const float n = a.sums[a.P + 1], Bf = (float)a.B;
    const float factor = (n == 0.0f) ? 0.0f : Bf / n;  // svi.py:305
    if (j < a.P) {
        int leaf = 0;
        while (leaf + 1 < a.n_leaves && j >= a.leaf_off[leaf + 1]) ++leaf;
        const uint32_t e = j - a.leaf_off[leaf];
        uint32_t k[16], o[16];
        load_key(a.leaf_keys + 16 * leaf, k);
        keystream_block(k, e >> 4, o);
        uint32_t bits = o[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) bits = ((e & 15u) == (uint32_t)w) ? o[w] : bits;
        const float z = bits_to_normal(bits);
        const float avg = a.sums[a.col_of[j]] / Bf;
        const float scale = a.h.dp_scale * (a.h.clip / n);  // (n == 0 -> inf, as the reference)
        const float g = (avg + z * scale) * a.obs_scale * factor;
        if (a.grad_out) a.grad_out[j] = g;
        float x = a.x_in[j], m = a.m_in[j], v = a.v_in[j];
        adam_update(x, m, v, g, *a.step_in, a.h);
        a.x_out[j] = x;
        a.m_out[j] = m;
        a.v_out[j] = v;
    }
    if (j == 0) {
        const float* x_in = a.x_in;
        *a.loss_out = (n == 0.0f) ? empty_batch_loss((int)a.P, [x_in](int c) { return x_in[c]; }) : (a.sums[a.P] / Bf) * a.obs_scale * factor;
        *a.step_out = *a.step_in + 1;
    }
}
}  // extern "C++"

int d3p_dpvi_leaves_begin(void* stream, const uint32_t* state_key_dev, int32_t n_leaves, const float* params_tree_dev,
                          const int32_t* col_of_dev, uint32_t P, uint32_t* next_key_dev, uint32_t* jax_key_dev, uint32_t* leaf_keys_dev,
                          float* params_kernel_dev)
{
    D3P_REQUIRE(state_key_dev && next_key_dev && jax_key_dev && leaf_keys_dev, "d3p_dpvi_leaves_begin: null key pointer");
    D3P_REQUIRE(n_leaves >= 1 && n_leaves <= D3P_MAX_LEAVES, "d3p_dpvi_leaves_begin: 1 <= n_leaves <= 16");
    D3P_REQUIRE(P == 0 || (params_tree_dev && col_of_dev && params_kernel_dev), "d3p_dpvi_leaves_begin: null parameter pointer");
    LeavesBeginArgs a{state_key_dev, n_leaves, params_tree_dev, col_of_dev, P, next_key_dev, jax_key_dev, leaf_keys_dev, params_kernel_dev};
    hipLaunchKernelGGL(k_leaves_begin, dim3(1 + cdiv(P, 64)), dim3(64), 0, (hipStream_t)stream, a);
    return check_launch("d3p_dpvi_leaves_begin");
}

int d3p_dpvi_leaves_finalize(void* stream, const d3p_dpsvi_hyper* hyper, const float* sums_dev, const int32_t* col_of_dev,
                             const uint32_t* leaf_keys_dev, const int32_t* leaf_sizes_host, int32_t n_leaves, uint32_t B, float obs_scale,
                             const float* params_in_dev, const float* m_in_dev, const float* v_in_dev, const int32_t* step_in_dev,
                             float* params_out_dev, float* m_out_dev, float* v_out_dev, int32_t* step_out_dev, float* loss_dev,
                             float* grad_out_dev)
{
    D3P_REQUIRE(hyper && sums_dev && col_of_dev && leaf_keys_dev && leaf_sizes_host, "d3p_dpvi_leaves_finalize: null pointer");
    D3P_REQUIRE(n_leaves >= 1 && n_leaves <= D3P_MAX_LEAVES, "d3p_dpvi_leaves_finalize: 1 <= n_leaves <= 16");
    D3P_REQUIRE(params_in_dev && m_in_dev && v_in_dev && step_in_dev && params_out_dev && m_out_dev && v_out_dev && step_out_dev && loss_dev,
                "d3p_dpvi_leaves_finalize: null state pointer");
    D3P_REQUIRE(B >= 1, "d3p_dpvi_leaves_finalize: batch size must be >= 1");
    D3P_REQUIRE(hyper->clip > 0.f, "The clipping threshold must be greater than 0.");
    LeavesFinalArgs a{};
    a.sums = sums_dev; a.col_of = col_of_dev; a.leaf_keys = leaf_keys_dev; a.n_leaves = n_leaves;
    uint64_t off = 0;
    for (int k = 0; k < n_leaves; ++k) {
        D3P_REQUIRE(leaf_sizes_host[k] >= 0, "d3p_dpvi_leaves_finalize: negative leaf size");
        a.leaf_off[k] = (uint32_t)off;
        off += (uint64_t)leaf_sizes_host[k];
    }
    D3P_REQUIRE(off >= 1 && off <= 0x7FFFFFFFull, "d3p_dpvi_leaves_finalize: the leaves must hold 1 .. 2^31 - 1 elements");
    for (int k = n_leaves; k <= D3P_MAX_LEAVES; ++k) a.leaf_off[k] = (uint32_t)off;
    a.P = (uint32_t)off; a.B = B; a.obs_scale = obs_scale; a.h = *hyper;
    a.x_in = params_in_dev; a.m_in = m_in_dev; a.v_in = v_in_dev; a.step_in = step_in_dev;
    a.x_out = params_out_dev; a.m_out = m_out_dev; a.v_out = v_out_dev; a.step_out = step_out_dev;
    a.loss_out = loss_dev; a.grad_out = grad_out_dev;
    hipLaunchKernelGGL(k_leaves_finalize, dim3(cdiv(a.P, 64)), dim3(64), 0, (hipStream_t)stream, a);
    return check_launch("d3p_dpvi_leaves_finalize");
}

}  // extern "C"
