// Per-example ELBO gradients of the Gaussian-mixture MODEL of BASELINE config 3
// (examples/gaussian_mixture_model.py:51-85; density d3p/gmm.py:71-86): stage 1 of DPSVI.update
// (svi.py:238-308) for  params = {alpha_log (K), mus_loc (K x d)},  P = K + K d.
//
// One wavefront per example.  Lane l owns the feature dimensions dd = l + 64 s (s < DS) of every
// component, so the per-component log-densities are wave reductions and the jax iota layout of the
// K x d normal / uniform draws pairs components k and k + K/2 of the same lane (word j = k d + dd with
// word j + K d / 2).  Lanes < K additionally carry the Dirichlet part in float64: the Gamma(alpha_k)
// draw (Marsaglia-Tsang), its implicit-reparametrisation derivative (series of the regularised
// incomplete gamma function) and the digamma terms -- a few dozen double operations on 16 lanes, small
// next to the 2 K d / 64 threefry calls per lane.  Formulas and stream layout: oracle/d3p_oracle.c
// (d3po_gmm_*), which also documents what is unpinned against jax.random.gamma.
#include "d3p_device.h"
#include "d3p_host.h"

namespace d3p {

static inline size_t align_up_g(size_t v, size_t a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ double digamma_d(double x)
{
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x
           - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760))))));
}

// d/dalpha of the Gamma(alpha, 1) quantile at fixed CDF value (see d3po_gamma_grad)
__device__ __forceinline__ double gamma_grad_d(double alpha, double x)
{
    if (!(x > 0.0)) return 0.0;
    double t = 1.0, h = 0.0, S = 1.0, Sp = 0.0;
    for (int n = 1; n < 2000; ++n) {
        t *= x / (alpha + n);
        h += 1.0 / (alpha + n);
        S += t;
        Sp -= t * h;
        if (t < 1e-18 * S && n > x) break;
    }
    return -(x / alpha) * (S * (log(x) - digamma_d(alpha + 1.0)) + Sp);
}

__device__ __forceinline__ double open_unit_d(uint32_t b) { return ((double)b + 0.5) * (1.0 / 4294967296.0); }

__device__ __forceinline__ double gamma_sample_d(uint32_t k0, uint32_t k1, uint32_t comp, double alpha)
{
    const double a = alpha < 1.0 ? alpha + 1.0 : alpha;
    const double dd = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
    double g = 0.0;
    for (uint32_t attempt = 0; attempt < 0x7fffffffu; ++attempt) {
        uint32_t b0, b1;
        threefry2x32(k0, k1, comp, attempt, b0, b1);
        const double x = (double)bits_to_normal(b0);
        const double U = open_unit_d(b1);
        const double v1 = 1.0 + c * x;
        if (v1 <= 0.0) continue;
        const double v = v1 * v1 * v1;
        if (log(U) < 0.5 * x * x + dd - dd * v + dd * log(v)) { g = dd * v; break; }
    }
    if (alpha < 1.0) {
        uint32_t b0, b1;
        threefry2x32(k0, k1, comp, 0x80000000u, b0, b1);
        g *= pow(open_unit_d(b0), 1.0 / alpha);
    }
    return g;
}

__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// jax.random.split(key, 2): words (0, 2) -> child 0, (1, 3) -> child 1 of the 4-word stream
__device__ __forceinline__ void tf_split2(uint32_t k0, uint32_t k1, uint32_t& a0, uint32_t& a1, uint32_t& b0, uint32_t& b1)
{
    threefry2x32(k0, k1, 0u, 2u, a0, b0);
    threefry2x32(k0, k1, 1u, 3u, a1, b1);
}

// pack (doubles): [alpha_k (K) | psi(alpha_k) (K) | psi(A0), A0, lgamma(A0) - sum lgamma(alpha_k) - lgamma(K)]
__global__ void __launch_bounds__(64) k_gmm_pack(const float* __restrict__ params, int K, double* __restrict__ pack)
{
    const int k = threadIdx.x;
    double alpha = 0.0, lg = 0.0;
    if (k < K) {
        alpha = exp((double)params[k]);
        lg = lgamma(alpha);
        pack[k] = alpha;
        pack[K + k] = digamma_d(alpha);
    }
    double A0 = 0.0, LG = 0.0;
    for (int j = 0; j < K; ++j) {  // fixed order
        A0 += readlane_d(alpha, j);
        LG += readlane_d(lg, j);
    }
    if (k == 0) {
        pack[2 * K] = digamma_d(A0);
        pack[2 * K + 1] = A0;
        pack[2 * K + 2] = lgamma(A0) - LG - lgamma((double)K);
    }
}

__global__ void __launch_bounds__(256) k_gmm_mask_meta(const uint8_t* __restrict__ mask, uint32_t B, float* __restrict__ meta)
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
        meta[1] = (n == 0.f) ? 0.f : (float)B / n;  // svi.py:305
    }
}

struct GmmArgs {
    const float* params;
    const double* pack;
    const float* X;
    const uint8_t* mask;
    const uint32_t* jax_key;
    const float* meta;
    float* px_loss;
    float* px_grads;
    float* latents_out;  // nullable: B x (K + 2 K d): g, eps, sigs of every example (tests)
    uint32_t B;
    int K, d;
    float inv_ps2, log_ps, lik_scale, inv_obs, obs_scale;
};

template <int KH, int DS>
__global__ void __launch_bounds__(256) k_gmm_px(GmmArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // wave-uniform
    if (p >= a.B) return;
    // components are handled in pairs (kk, kk + Kh), Kh = ceil(K / 2); for even K the pair shares its threefry calls
    // (words j and j + K d / 2 of jax's iota layout), for odd K every entry takes its own call
    const int K = a.K, d = a.d, Kh = (K + 1) >> 1, P = K + K * d;
    const uint32_t n_lat = (uint32_t)(K * d), half = n_lat >> 1;
    const bool paired = (K & 1) == 0;
    const float live = (a.mask ? a.mask[p] != 0 : true) ? 1.0f : 0.0f;

    // ---- keys of the guide's sample sites (numpyro.handlers.seed over pis, mus, sigs)
    uint32_t px0 = tf_iota_word(a.jax_key[0], a.jax_key[1], 2ull * a.B, 2ull * p);
    uint32_t px1 = tf_iota_word(a.jax_key[0], a.jax_key[1], 2ull * a.B, 2ull * p + 1);
    uint32_t r0, r1, t0, t1, kp0, kp1, km0, km1, ks0, ks1;
    tf_split2(px0, px1, t0, t1, r0, r1);   // guide_seed = child 1
    tf_split2(r0, r1, t0, t1, kp0, kp1);   // k_pis = child 1, next state = child 0
    tf_split2(t0, t1, r0, r1, km0, km1);
    tf_split2(r0, r1, t0, t1, ks0, ks1);

    // ---- Dirichlet part on lanes < K (float64)
    double alpha = 1.0, g = 0.0, gp = 0.0;
    if (lane < K) {
        alpha = a.pack[lane];
        g = gamma_sample_d(kp0, kp1, (uint32_t)lane, alpha);
        gp = gamma_grad_d(alpha, g);
    }
    double S = 0.0;
    for (int k = 0; k < K; ++k) S += readlane_d(g, k);
    const double pis = lane < K ? g / S : 1.0;
    const float logpis = (float)log(pis);

    // ---- mus, sigs and the per-component log-densities (lanes <-> feature dimensions)
    float xs[DS];
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        const int dd = lane + 64 * s;
        xs[s] = dd < d ? a.X[(size_t)p * d + dd] : 0.f;
    }
    float wv[2 * KH * DS], muv[2 * KH * DS];
    float acomp[2 * KH];
    float lmu = 0.f;  // sum of  -eps^2/2 + (mu/ps)^2/2 + log ps  over this lane's entries
    float* lat = a.latents_out ? a.latents_out + (size_t)p * (K + 2 * n_lat) : nullptr;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        float ll0 = 0.f, ll1 = 0.f;
        if (kk < Kh) {
            const bool has1 = kk + Kh < K;
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                const int dd = lane + 64 * s;
                const bool ok0 = dd < d, ok1 = ok0 && has1;
                const uint32_t j0 = (uint32_t)(kk * d + dd), j1 = (uint32_t)((kk + Kh) * d + dd);
                uint32_t b0, b1, u0, u1;
                if (paired) {  // j1 == j0 + half
                    threefry2x32(km0, km1, ok0 ? j0 : 0u, ok0 ? j1 : 0u, b0, b1);
                    threefry2x32(ks0, ks1, ok0 ? j0 : 0u, ok0 ? j1 : 0u, u0, u1);
                } else {
                    b0 = tf_iota_word(km0, km1, n_lat, ok0 ? j0 : 0u);
                    u0 = tf_iota_word(ks0, ks1, n_lat, ok0 ? j0 : 0u);
                    b1 = tf_iota_word(km0, km1, n_lat, ok1 ? j1 : 0u);
                    u1 = tf_iota_word(ks0, ks1, n_lat, ok1 ? j1 : 0u);
                }
                const float e0 = bits_to_normal(b0), e1 = bits_to_normal(b1);
                // Exponential(1) by inversion; sigs = 1 / ex, so 1 / sig = ex and -log sig = log ex
                const float ex0 = -logf(((float)(u0 >> 9) + 0.5f) * 1.1920928955078125e-07f);
                const float ex1 = -logf(((float)(u1 >> 9) + 0.5f) * 1.1920928955078125e-07f);
                const float mu0 = ok0 ? a.params[K + j0] + e0 : 0.f, mu1 = ok1 ? a.params[K + j1] + e1 : 0.f;
                const float z0 = (xs[s] - mu0) * ex0, z1 = (xs[s] - mu1) * ex1;
                const int i0 = kk * DS + s, i1 = (KH + kk) * DS + s;
                wv[i0] = ok0 ? z0 * ex0 : 0.f;
                wv[i1] = ok1 ? z1 * ex1 : 0.f;
                muv[i0] = mu0;
                muv[i1] = mu1;
                if (ok0) {
                    ll0 += __fmaf_rn(-0.5f * z0, z0, logf(ex0) - D3P_HALF_LOG_2PI);
                    lmu += __fmaf_rn(-0.5f * e0, e0, __fmaf_rn(0.5f * a.inv_ps2 * mu0, mu0, a.log_ps));
                    if (lat) {
                        lat[K + j0] = e0;
                        lat[K + n_lat + j0] = 1.0f / ex0;
                    }
                }
                if (ok1) {
                    ll1 += __fmaf_rn(-0.5f * z1, z1, logf(ex1) - D3P_HALF_LOG_2PI);
                    lmu += __fmaf_rn(-0.5f * e1, e1, __fmaf_rn(0.5f * a.inv_ps2 * mu1, mu1, a.log_ps));
                    if (lat) {
                        lat[K + j1] = e1;
                        lat[K + n_lat + j1] = 1.0f / ex1;
                    }
                }
            }
        }
        acomp[kk] = wave_sum(ll0);
        acomp[KH + kk] = wave_sum(ll1);
    }
    lmu = wave_sum(lmu);
    if (lat && lane < K) lat[lane] = (float)g;

    // ---- mixture: a_k = log pis_k + ll_k, responsibilities r_k, loglik = logsumexp_k a_k  (gmm.py:71-86)
    float best = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        if (kk < Kh) {
            const bool has1 = kk + Kh < K;
            acomp[kk] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(logpis), kk));
            acomp[KH + kk] = has1 ? acomp[KH + kk] + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(logpis), has1 ? kk + Kh : 0))
                                  : -INFINITY;
            best = fmaxf(best, fmaxf(acomp[kk], acomp[KH + kk]));
        }
    }
    float se = 0.f;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        if (kk < Kh) {
            acomp[kk] = expf(acomp[kk] - best);
            acomp[KH + kk] = expf(acomp[KH + kk] - best);  // exp(-inf) = 0 for the missing partner of an odd K
            se += acomp[kk] + acomp[KH + kk];
        }
    }
    const float loglik = best + logf(se);
    const float inv_se = 1.0f / se;

    // ---- gradient wrt mus_loc: inv_obs * (mu / ps^2 - N r_k w)
    float* gr = a.px_grads + (size_t)p * P;
    float my_r = 0.f;  // r_k of this lane's own component (lanes < K)
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        if (kk < Kh) {
            const bool has1 = kk + Kh < K;
            const float ra = acomp[kk] * inv_se, rb = acomp[KH + kk] * inv_se;
            if (lane == kk) my_r = ra;
            if (lane == kk + Kh) my_r = rb;
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                const int dd = lane + 64 * s;
                if (dd < d) {
                    const uint32_t j0 = (uint32_t)(kk * d + dd), j1 = (uint32_t)((kk + Kh) * d + dd);
                    const int i0 = kk * DS + s, i1 = (KH + kk) * DS + s;
                    gr[K + j0] = a.inv_obs * __fmaf_rn(-a.lik_scale * ra, wv[i0], a.inv_ps2 * muv[i0]) * live;
                    if (has1) gr[K + j1] = a.inv_obs * __fmaf_rn(-a.lik_scale * rb, wv[i1], a.inv_ps2 * muv[i1]) * live;
                }
            }
        }
    }

    // ---- gradient wrt alpha_log (lanes < K, float64) and the Dirichlet part of log q - log p
    const double A0 = a.pack[2 * K + 1];
    double lq_term = 0.0;
    if (lane < K) {
        const double psi0 = a.pack[2 * K], psik = a.pack[K + lane];
        const double gs = gp / S;
        const double dq = psi0 - psik + log(pis) + gs * ((alpha - 1.0) / pis - (A0 - (double)K));
        const double dl = gs * ((double)my_r / pis - 1.0);
        gr[lane] = (float)(alpha * (double)a.inv_obs * (dq - (double)a.lik_scale * dl)) * live;
        lq_term = (alpha - 1.0) * log(pis);
    }
    double lq = a.pack[2 * K + 2];
    for (int k = 0; k < K; ++k) lq += readlane_d(lq_term, k);
    if (lane == 0) {
        const float L = a.inv_obs * (((float)lq + lmu) - a.lik_scale * loglik);
        a.px_loss[p] = L * live * a.obs_scale * a.meta[1];  // svi.py:281, :306
    }
}

}  // namespace d3p

using namespace d3p;

extern "C" {

size_t d3p_gmm_px_grads_workspace(int32_t K)
{
    return align_up_g((size_t)(2 * (K > 0 ? K : 0) + 3) * sizeof(double), 256);
}

int d3p_gmm_px_grads(void* stream, const d3p_gmm_model* model, const float* params_dev, const float* X_dev,
                     const uint8_t* mask_dev, uint32_t B, const uint32_t* jax_key_dev, float* px_loss_dev, float* px_grads_dev,
                     float* meta_dev, float* latents_out_dev, void* workspace_dev, size_t workspace_bytes)
{
    D3P_REQUIRE(model && params_dev && X_dev && jax_key_dev && px_loss_dev && px_grads_dev && meta_dev && workspace_dev,
                "d3p_gmm_px_grads: null pointer");
    D3P_REQUIRE(B >= 1, "d3p_gmm_px_grads: B must be >= 1");
    D3P_REQUIRE(model->K >= 1 && model->d >= 1, "d3p_gmm_px_grads: K and d must be >= 1");
    D3P_REQUIRE(model->prior_mu_scale > 0.f && model->inv_obs > 0.f, "d3p_gmm_px_grads: bad model");
    if (model->K > 32 || model->d > 256 || (model->K > 16 && model->d > 128))
        return fail(D3P_E_UNSUPPORTED, "d3p_gmm_px_grads: supported shapes are K <= 16 with d <= 256 and K <= 32 with d <= 128 "
                                       "(K = %d, d = %d)", model->K, model->d);
    if (workspace_bytes < d3p_gmm_px_grads_workspace(model->K)) return fail(D3P_E_WORKSPACE, "d3p_gmm_px_grads: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double* pack = (double*)workspace_dev;
    hipLaunchKernelGGL(k_gmm_pack, dim3(1), dim3(64), 0, s, params_dev, model->K, pack);
    hipLaunchKernelGGL(k_gmm_mask_meta, dim3(1), dim3(256), 0, s, mask_dev, B, meta_dev);
    GmmArgs a;
    a.params = params_dev;
    a.pack = pack;
    a.X = X_dev;
    a.mask = mask_dev;
    a.jax_key = jax_key_dev;
    a.meta = meta_dev;
    a.px_loss = px_loss_dev;
    a.px_grads = px_grads_dev;
    a.latents_out = latents_out_dev;
    a.B = B;
    a.K = model->K;
    a.d = model->d;
    a.inv_ps2 = 1.0f / (model->prior_mu_scale * model->prior_mu_scale);
    a.log_ps = logf(model->prior_mu_scale);
    a.lik_scale = model->lik_scale;
    a.inv_obs = model->inv_obs;
    a.obs_scale = 1.0f / model->inv_obs;
    const dim3 grid(cdiv((uint64_t)B * 64, 256)), block(256);
    const int KH = (model->K + 1) / 2 <= 8 ? 8 : 16, DS = (model->d + 63) / 64;
#define D3P_GMM_LAUNCH(KH_, DS_) hipLaunchKernelGGL((k_gmm_px<KH_, DS_>), grid, block, 0, s, a)
    if (KH == 8) {
        switch (DS) {
        case 1: D3P_GMM_LAUNCH(8, 1); break;
        case 2: D3P_GMM_LAUNCH(8, 2); break;
        default: D3P_GMM_LAUNCH(8, 4); break;
        }
    } else {
        switch (DS) {
        case 1: D3P_GMM_LAUNCH(16, 1); break;
        default: D3P_GMM_LAUNCH(16, 2); break;
        }
    }
#undef D3P_GMM_LAUNCH
    return check_launch("d3p_gmm_px_grads");
}

}  // extern "C"
