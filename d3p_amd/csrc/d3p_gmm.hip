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
        const double rcp = 1.0 / (alpha + n);
        t *= x * rcp;
        h += rcp;
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

// guide_seed -> the three site keys of example p
__device__ __forceinline__ void gmm_site_keys(const uint32_t* jax_key, uint32_t B, uint32_t p, uint32_t& kp0, uint32_t& kp1,
                                              uint32_t& km0, uint32_t& km1, uint32_t& ks0, uint32_t& ks1)
{
    const uint32_t px0 = tf_iota_word(jax_key[0], jax_key[1], 2ull * B, 2ull * p);
    const uint32_t px1 = tf_iota_word(jax_key[0], jax_key[1], 2ull * B, 2ull * p + 1);
    uint32_t r0, r1, t0, t1;
    tf_split2(px0, px1, t0, t1, r0, r1);   // guide_seed = child 1
    tf_split2(r0, r1, t0, t1, kp0, kp1);   // k_pis = child 1, next state = child 0
    tf_split2(t0, t1, r0, r1, km0, km1);
    tf_split2(r0, r1, t0, t1, ks0, ks1);
}

// Dirichlet part, one THREAD per (example, component): Gamma(alpha_k) draw and its derivative wrt alpha_k.
// dir[(p K + k) * 2 + {0, 1}] = {g, dg/dalpha}
// (data-parallel: the B examples are positions pos0 .. pos0 + B - 1 of a global batch of B_total; the keys of an example are
// functions of its GLOBAL position)
__global__ void __launch_bounds__(256) k_gmm_dirichlet(const double* __restrict__ pack, const uint32_t* __restrict__ jax_key,
                                                       uint32_t B, uint32_t B_total, uint32_t pos0, int K, double* __restrict__ dir)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)B * K) return;
    const uint32_t p = (uint32_t)(i / K), k = (uint32_t)(i % K);
    uint32_t kp0, kp1, km0, km1, ks0, ks1;
    gmm_site_keys(jax_key, B_total, pos0 + p, kp0, kp1, km0, km1, ks0, ks1);
    const double alpha = pack[k];
    const double g = gamma_sample_d(kp0, kp1, k, alpha);
    dir[2 * i] = g;
    dir[2 * i + 1] = gamma_grad_d(alpha, g);
}

struct GmmArgs {
    const float* params;
    const double* pack;
    const double* dir;       // B x K x 2 from k_gmm_dirichlet
    const float* X;
    const uint32_t* idx;     // nullable: row of example p is X[idx[p]] (minibatch of a resident table)
    const uint8_t* mask;
    const uint32_t* jax_key;
    const float* meta;
    float* px_loss;
    float* px_grads;
    float* latents_out;  // nullable: B x (K + 2 K d): g, eps, sigs of every example (tests)
    float* partials;     // SUM mode: one row of P + 2 per wavefront: [sum_i c_i g_i | sum_i loss_i | n]
    uint32_t B;
    uint32_t B_total, pos0;  // key derivation: example p is position pos0 + p of a global batch of B_total (= B, 0 on one device)
    int K, d;
    float inv_ps2, log_ps, lik_scale, inv_obs, obs_scale, clip;
};

// SUM = false: materialise px_loss / px_grads (stage API).  SUM = true: clip each example's gradient by its joint
// L2 norm and accumulate (svi.py:310-348 fused into stage 1); every wavefront strides over the batch and leaves one
// partial row, summed in fixed order by k_gmm_finalize.
// The unrolled component loop holds 4 KH DS values per lane; asking for 4 (2) resident waves per SIMD keeps the
// scheduler from interleaving all threefry chains at once (which drove the small shapes to 256 VGPRs, occupancy 1).
template <int KH, int DS, bool SUM, bool PAIRED>
__global__ void __launch_bounds__(256, (KH * DS <= 8 ? 3 : KH * DS <= 16 ? 2 : 1)) k_gmm_px(GmmArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // wave-uniform
    const uint32_t total_waves = (gridDim.x * blockDim.x) >> 6;
    float accg[2 * KH * DS], acca = 0.f, loss_acc = 0.f, n_acc = 0.f;
    float locv[2 * KH * DS];  // this lane's entries of mus_loc, loaded once (slot layout as wv / muv below)
    {
        const int K = a.K, d = a.d, Kh = (K + 1) >> 1;
#pragma unroll
        for (int kk = 0; kk < KH; ++kk)
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                const int dd = lane + 64 * s;
                const bool ok0 = kk < Kh && dd < d, ok1 = ok0 && kk + Kh < K;
                accg[kk * DS + s] = accg[(KH + kk) * DS + s] = 0.f;
                locv[kk * DS + s] = ok0 ? a.params[K + kk * d + dd] : 0.f;
                locv[(KH + kk) * DS + s] = ok1 ? a.params[K + (kk + Kh) * d + dd] : 0.f;
            }
    }
    for (uint32_t p = gw; p < a.B; p += total_waves) {
    // components are handled in pairs (kk, kk + Kh), Kh = ceil(K / 2); for even K the pair shares its threefry calls
    // (words j and j + K d / 2 of jax's iota layout), for odd K every entry takes its own call
    const int K = a.K, d = a.d, Kh = (K + 1) >> 1, P = K + K * d;
    const uint32_t n_lat = (uint32_t)(K * d), half = n_lat >> 1;
    const float live = (a.mask ? a.mask[p] != 0 : true) ? 1.0f : 0.0f;
    if (SUM && live == 0.0f) continue;  // masked examples contribute nothing (svi.py:281)
    const size_t row = a.idx ? a.idx[p] : p;

    // ---- keys of the guide's sample sites (numpyro.handlers.seed over pis, mus, sigs)
    uint32_t kp0, kp1, km0, km1, ks0, ks1;
    gmm_site_keys(a.jax_key, a.B_total, a.pos0 + p, kp0, kp1, km0, km1, ks0, ks1);

    // ---- Dirichlet part on lanes < K (float64)
    double alpha = 1.0, g = 0.0, gp = 0.0;
    if (lane < K) {
        alpha = a.pack[lane];
        g = a.dir[((size_t)p * K + lane) * 2];
        gp = a.dir[((size_t)p * K + lane) * 2 + 1];
    }
    double S = 0.0;
    for (int k = 0; k < K; ++k) S += readlane_d(g, k);
    const double pis = lane < K ? g / S : 1.0;
    const double logpis_d = log(pis);
    const float logpis = (float)logpis_d;

    // ---- mus, sigs and the per-component log-densities (lanes <-> feature dimensions)
    float xs[DS];
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        const int dd = lane + 64 * s;
        xs[s] = dd < d ? a.X[(size_t)row * d + dd] : 0.f;
    }
    float wv[2 * KH * DS], muv[2 * KH * DS];
    float acomp[2 * KH];
    float lmu = 0.f;  // sum of  -eps^2/2 + (mu/ps)^2/2 + log ps  over this lane's entries
    float* lat = (!SUM && a.latents_out) ? a.latents_out + (size_t)p * (K + 2 * n_lat) : nullptr;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        float ll0 = 0.f, ll1 = 0.f;
        if (kk < Kh) {
            const bool has1 = kk + Kh < K;
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                const int dd = lane + 64 * s;
                const bool ok0 = dd < d, ok1 = ok0 && has1;
                const uint32_t j0 = (uint32_t)(kk * d + dd), j1 = (uint32_t)((kk + Kh) * d + dd);
                uint32_t b0, b1, u0, u1;
                if (PAIRED) {  // K even: j1 == j0 + half
                    threefry2x32(km0, km1, ok0 ? j0 : 0u, ok0 ? j1 : 0u, b0, b1);
                    threefry2x32(ks0, ks1, ok0 ? j0 : 0u, ok0 ? j1 : 0u, u0, u1);
                } else {
                    b0 = tf_iota_word(km0, km1, n_lat, ok0 ? j0 : 0u);
                    u0 = tf_iota_word(ks0, ks1, n_lat, ok0 ? j0 : 0u);
                    b1 = tf_iota_word(km0, km1, n_lat, ok1 ? j1 : 0u);
                    u1 = tf_iota_word(ks0, ks1, n_lat, ok1 ? j1 : 0u);
                }
                const float e0 = bits_to_normal_wu(b0), e1 = bits_to_normal_wu(b1);
                // Exponential(1) by inversion; sigs = 1 / ex, so 1 / sig = ex and -log sig = log ex
                // (the argument is a normal float in (0, 1): the hardware log2 path is 1 ulp and a sixth of the instructions)
                const float ex0 = -__logf(((float)(u0 >> 9) + 0.5f) * 1.1920928955078125e-07f);
                const float ex1 = -__logf(((float)(u1 >> 9) + 0.5f) * 1.1920928955078125e-07f);
                const int i0 = kk * DS + s, i1 = (KH + kk) * DS + s;
                const float mu0 = ok0 ? locv[i0] + e0 : 0.f, mu1 = ok1 ? locv[i1] + e1 : 0.f;
                const float z0 = (xs[s] - mu0) * ex0, z1 = (xs[s] - mu1) * ex1;
                wv[i0] = ok0 ? z0 * ex0 : 0.f;
                wv[i1] = ok1 ? z1 * ex1 : 0.f;
                muv[i0] = mu0;
                muv[i1] = mu1;
                if (ok0) {
                    ll0 += __fmaf_rn(-0.5f * z0, z0, __logf(ex0) - D3P_HALF_LOG_2PI);
                    lmu += __fmaf_rn(-0.5f * e0, e0, __fmaf_rn(0.5f * a.inv_ps2 * mu0, mu0, a.log_ps));
                    if (lat) {
                        lat[K + j0] = e0;
                        lat[K + n_lat + j0] = 1.0f / ex0;
                    }
                }
                if (ok1) {
                    ll1 += __fmaf_rn(-0.5f * z1, z1, __logf(ex1) - D3P_HALF_LOG_2PI);
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
        __builtin_amdgcn_sched_barrier(0);  // one component pair at a time: keeps the live threefry chains (and VGPRs) bounded
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
            acomp[kk] = __expf(acomp[kk] - best);
            acomp[KH + kk] = __expf(acomp[KH + kk] - best);  // exp(-inf) = 0 for the missing partner of an odd K
            se += acomp[kk] + acomp[KH + kk];
        }
    }
    const float loglik = best + logf(se);
    const float inv_se = 1.0f / se;

    // ---- gradient wrt mus_loc: inv_obs * (mu / ps^2 - N r_k w)
    float* gr = SUM ? nullptr : a.px_grads + (size_t)p * P;
    float my_r = 0.f;  // r_k of this lane's own component (lanes < K)
    float n2 = 0.f;
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
                    const float g0 = a.inv_obs * __fmaf_rn(-a.lik_scale * ra, wv[i0], a.inv_ps2 * muv[i0]);
                    const float g1 = has1 ? a.inv_obs * __fmaf_rn(-a.lik_scale * rb, wv[i1], a.inv_ps2 * muv[i1]) : 0.f;
                    if (SUM) {  // keep the values for the clipped accumulation below
                        wv[i0] = g0;
                        wv[i1] = g1;
                        n2 = __fmaf_rn(g0, g0, __fmaf_rn(g1, g1, n2));
                    } else {
                        gr[K + j0] = g0 * live;
                        if (has1) gr[K + j1] = g1 * live;
                    }
                } else if (SUM) {
                    wv[kk * DS + s] = 0.f;
                    wv[(KH + kk) * DS + s] = 0.f;
                }
            }
        }
    }

    // ---- gradient wrt alpha_log (lanes < K, float64) and the Dirichlet part of log q - log p
    const double A0 = a.pack[2 * K + 1];
    double lq_term = 0.0;
    float ga = 0.f;
    if (lane < K) {
        const double psi0 = a.pack[2 * K], psik = a.pack[K + lane];
        const double gs = gp / S;
        const double dq = psi0 - psik + logpis_d + gs * ((alpha - 1.0) / pis - (A0 - (double)K));
        const double dl = gs * ((double)my_r / pis - 1.0);
        ga = (float)(alpha * (double)a.inv_obs * (dq - (double)a.lik_scale * dl));
        if (!SUM) gr[lane] = ga * live;
        lq_term = (alpha - 1.0) * logpis_d;
    }
    double lq = a.pack[2 * K + 2];
    for (int k = 0; k < K; ++k) lq += readlane_d(lq_term, k);
    const float L = a.inv_obs * (((float)lq + lmu) - a.lik_scale * loglik);
    if (SUM) {
        n2 = wave_sum(__fmaf_rn(ga, ga, n2));
        const float cf = 1.0f / fmaxf(1.0f, sqrtf(n2) / a.clip);  // svi.py:121-122
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            if (kk < Kh) {
#pragma unroll
                for (int s = 0; s < DS; ++s) {
                    accg[kk * DS + s] = __fmaf_rn(cf, wv[kk * DS + s], accg[kk * DS + s]);
                    accg[(KH + kk) * DS + s] = __fmaf_rn(cf, wv[(KH + kk) * DS + s], accg[(KH + kk) * DS + s]);
                }
            }
        }
        acca = __fmaf_rn(cf, ga, acca);
        loss_acc += L;
        n_acc += 1.0f;
    } else if (lane == 0) {
        a.px_loss[p] = L * live * a.obs_scale * a.meta[1];  // svi.py:281, :306
    }
    }  // examples of this wavefront
    if (SUM) {
        const int K = a.K, d = a.d, Kh = (K + 1) >> 1, P = K + K * d;
        float* out = a.partials + (size_t)gw * (P + 2);
        if (lane < K) out[lane] = acca;
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            if (kk < Kh) {
#pragma unroll
                for (int s = 0; s < DS; ++s) {
                    const int dd = lane + 64 * s;
                    if (dd < d) {
                        out[K + kk * d + dd] = accg[kk * DS + s];
                        if (kk + Kh < K) out[K + (kk + Kh) * d + dd] = accg[(KH + kk) * DS + s];
                    }
                }
            }
        }
        if (lane == 0) {
            out[P] = loss_acc;
            out[P + 1] = n_acc;
        }
    }
}

// ---- DPSVI.evaluate for the mixture model: one guide draw for the whole batch (oracle: d3po_gmm_evaluate).
// Writes mus / sigs (K d each) and pis (K) for d3p_gmm_log_prob, and lat[0] = (log q - log p)(latents).
__global__ void __launch_bounds__(256) k_gmm_eval_latents(const float* __restrict__ params, const uint32_t* __restrict__ jax_key,
                                                          int K, int d, float inv_ps2, float log_ps, float* __restrict__ mus,
                                                          float* __restrict__ sigs, float* __restrict__ pis, float* __restrict__ lat)
{
    __shared__ double sh_g[64], sh_t[64];
    __shared__ float red[256];
    uint32_t a0, a1, r0, r1, t0, t1, kp0, kp1, km0, km1, ks0, ks1;
    tf_split2(jax_key[0], jax_key[1], a0, a1, r0, r1);  // rng_key_eval = child 1
    tf_split2(r0, r1, a0, a1, t0, t1);                  // guide_seed = child 1
    tf_split2(t0, t1, a0, a1, kp0, kp1);
    tf_split2(a0, a1, r0, r1, km0, km1);
    tf_split2(r0, r1, a0, a1, ks0, ks1);
    const int tid = threadIdx.x;
    double alpha = 1.0;
    if (tid < K) {
        alpha = exp((double)params[tid]);
        sh_g[tid] = gamma_sample_d(kp0, kp1, (uint32_t)tid, alpha);
    }
    __syncthreads();
    double S = 0.0;
    for (int k = 0; k < K; ++k) S += sh_g[k];
    if (tid < K) {
        const double p = sh_g[tid] / S;
        pis[tid] = (float)p;
        sh_t[tid] = -lgamma(alpha) + (alpha - 1.0) * log(p);
        sh_g[tid] = alpha;
    }
    const uint32_t n = (uint32_t)(K * d);
    float acc = 0.f;
    for (uint32_t j = tid; j < n; j += 256) {
        const float e = bits_to_normal(tf_iota_word(km0, km1, n, j));
        const uint32_t b = tf_iota_word(ks0, ks1, n, j);
        const float ex = -logf(((float)(b >> 9) + 0.5f) * 1.1920928955078125e-07f);
        const float mu = params[K + j] + e;
        mus[j] = mu;
        sigs[j] = 1.0f / ex;
        acc += __fmaf_rn(-0.5f * e, e, __fmaf_rn(0.5f * inv_ps2 * mu, mu, log_ps));
    }
    red[tid] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        double A0 = 0.0, t = 0.0;
        for (int k = 0; k < K; ++k) { A0 += sh_g[k]; t += sh_t[k]; }
        lat[0] = (float)(lgamma(A0) - lgamma((double)K) + t) + red[0];
    }
}

__global__ void __launch_bounds__(256) k_gmm_eval_finish(const float* __restrict__ ll, uint32_t B, const float* __restrict__ lat,
                                                         float lik_scale, float* __restrict__ loss)
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
    if (threadIdx.x == 0) *loss = lat[0] - (lik_scale / (float)B) * red[0];  // plate(N, B): likelihood scaled by N / B
}

// Replicated tail of the step: fixed-order column sums of the per-wavefront partial rows, mean over the padded batch
// (svi.py:343-346), Gaussian mechanism with per-site noise (svi.py:365-375, :487-491), numpyro Adam (svi.py:379-393).
// The partial rows (<= D3P_GMM_MAX_WAVES) are summed by 8 row groups per column.
struct GmmFinalArgs {
    const float* partials;
    uint32_t n_rows;     // partial rows
    uint32_t B;
    int P;
    const float* noise;  // P standard normals: site alpha_log (K) then site mus_loc (K d); unused when site_keys is given
    const uint32_t* site_keys;  // nullable: split(perturbation_key, 2) -- the noise is then generated inside the kernel
    int K;
    float* params;
    float* adam_m;
    float* adam_v;
    int32_t* step;
    float* loss_out;     // nullable
    float* grad_out;     // nullable
    d3p_dpsvi_hyper h;
    float obs_scale;
};

// Everything of one update that is a function of the state alone, in ONE launch (each of these was a launch of ~5 us: three
// key derivations, a one-block keystream, the parameter pack, the step counter and a 64-byte copy made up a third of the
// 140 us step):  wave 0: [next | gradient | perturbation] = split(state_key, 3) (svi.py:208-211), the next state key written
// straight into the other key slot, jax_key = convert_to_jax_rng_key(gradient_key) (svi.py:259), site keys =
// split(perturbation_key, 2) (svi.py:491); wave 1: the double-precision pack of the Dirichlet parameters; the optimiser step
// index is saved for k_gmm_finalize and advanced.
struct GmmPreArgs {
    const uint32_t* cur_key;
    uint32_t* next_slot;
    uint32_t* keys;        // workspace: split3 (48) | site keys (32) | folded batch key (16) | jax key (2)
    const float* params;
    int K;
    double* pack;
    int32_t* step;
    int32_t* step_saved;
    const uint32_t* batch_key;  // nullable (run loop): keys[80..95] = fold_in(batch_key, batch_index), minibatch.py:226-230
    uint32_t batch_index;
    int advance;                // 0: keys and pack only (the local-sums half of a data-parallel update leaves the state alone)
};

// (key derivations with the 4-lane ChaCha block of d3p_device.h: quad j of wave 0 derives one child, a third of the serial
// instruction count of the one-lane block -- the launch is pure latency)
__device__ __forceinline__ void gmm_store_child_quad(uint32_t* dst, const uint32_t* parent, int q, uint32_t a, uint32_t b)
{
    dst[q] = parent[q];  // constants row
    dst[4 + q] = a;
    dst[8 + q] = b;
    dst[12 + q] = 0u;
}

__global__ void __launch_bounds__(128) k_gmm_pre(GmmPreArgs a)
{
    __shared__ uint32_t sk[3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int quad = lane >> 2, q = lane & 3;
    if (wave == 0) {
        // quads 0..2: split(state_key, 3)[quad]; quad 3: fold_in(batch_key, batch_index) (idle copy of the split otherwise)
        const bool fold = quad == 3 && a.batch_key != nullptr;
        const uint32_t* parent = fold ? a.batch_key : a.cur_key;
        uint32_t ka, kb;
        derive_child_quad(parent, (quad < 3) ? (uint32_t)quad : 0u, fold ? D3P_TAG_FOLD : D3P_TAG_SPLIT, fold ? a.batch_index : 0u, ka, kb);
        if (quad < 3) {
            gmm_store_child_quad(a.keys + 16 * quad, parent, q, ka, kb);
            gmm_store_child_quad(sk[quad], parent, q, ka, kb);
            if (quad == 0 && a.advance) gmm_store_child_quad(a.next_slot, parent, q, ka, kb);
        } else if (fold) {
            gmm_store_child_quad(a.keys + 80, parent, q, ka, kb);
        }
        if (lane == 63 && a.advance) {
            const int32_t i = *a.step;
            *a.step_saved = i;
            *a.step = i + 1;
        }
    } else {
        // pack (doubles): [alpha_k (K) | psi(alpha_k) (K) | psi(A0), A0, lgamma(A0) - sum lgamma(alpha_k) - lgamma(K)]
        const int k = lane, K = a.K;
        double alpha = 0.0, lg = 0.0;
        if (k < K) {
            alpha = exp((double)a.params[k]);
            lg = lgamma(alpha);
            a.pack[k] = alpha;
            a.pack[K + k] = digamma_d(alpha);
        }
        double A0 = 0.0, LG = 0.0;
        for (int j = 0; j < K; ++j) {  // fixed order
            A0 += readlane_d(alpha, j);
            LG += readlane_d(lg, j);
        }
        if (k == 0) {
            a.pack[2 * K] = digamma_d(A0);
            a.pack[2 * K + 1] = A0;
            a.pack[2 * K + 2] = lgamma(A0) - LG - lgamma((double)K);
        }
    }
    __syncthreads();
    if (wave == 0) {
        // quad 0: block 0 of the gradient key's stream -> jax key (random_bits(gradient_key, 32, (2,)));
        // quads 1, 2: split(perturbation_key, 2)
        const uint32_t* parent = quad == 0 ? sk[1] : sk[2];
        uint32_t ka, kb;
        derive_child_quad(parent, (quad == 1 || quad == 2) ? (uint32_t)(quad - 1) : 0u, quad == 0 ? 0u : D3P_TAG_SPLIT, 0u, ka, kb);
        if (quad == 0 && q < 2) a.keys[96 + q] = ka;
        if (quad == 1 || quad == 2) gmm_store_child_quad(a.keys + 48 + 16 * (quad - 1), parent, q, ka, kb);
    }
}

// First reduction level: the per-wavefront partial rows are cut into D3P_GMM_CHUNKS chunks of consecutive rows; workgroup
// (column tile of 64, chunk) sums its rows with 4 row subgroups and leaves one row per chunk.  Fixed order throughout.
#define D3P_GMM_CHUNKS 64u

__global__ void __launch_bounds__(256) k_gmm_reduce(const float* __restrict__ parts, uint32_t n_rows, int width,
                                                    float* __restrict__ reduced)
{
    __shared__ float lds[256];
    const int c = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const uint32_t per = (n_rows + D3P_GMM_CHUNKS - 1) / D3P_GMM_CHUNKS;
    const uint32_t r0 = blockIdx.y * per, r1 = (r0 + per < n_rows) ? r0 + per : n_rows;
    float s = 0.f;
    if (col < width)
        for (uint32_t r = r0 + sg; r < r1; r += 4) s += parts[(size_t)r * width + col];
    lds[threadIdx.x] = s;
    __syncthreads();
    if (sg == 0 && col < width)
        reduced[(size_t)blockIdx.y * width + col] = (lds[c] + lds[64 + c]) + (lds[128 + c] + lds[192 + c]);
}

// fixed-order sum of one column of the partial rows by the whole workgroup; every thread returns the total
__device__ __forceinline__ float gmm_block_column_sum(const float* __restrict__ parts, uint32_t n_rows, size_t stride, int col,
                                                      float* lds)
{
    float s = 0.f;
    for (uint32_t r = threadIdx.x; r < n_rows; r += 256) s += parts[r * stride + col];
    lds[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) lds[threadIdx.x] += lds[threadIdx.x + off];
        __syncthreads();
    }
    const float tot = lds[0];
    __syncthreads();
    return tot;
}

// 32 columns per workgroup x 8 row groups; the row groups are combined through LDS in fixed order
__global__ void __launch_bounds__(256) k_gmm_finalize(GmmFinalArgs a)
{
    __shared__ float lds[256];
    const size_t stride = (size_t)a.P + 2;
    const float n = gmm_block_column_sum(a.partials, a.n_rows, stride, a.P + 1, lds);
    const float ls = gmm_block_column_sum(a.partials, a.n_rows, stride, a.P, lds);
    const float Bf = (float)a.B;
    const float factor = (n == 0.f) ? 0.f : Bf / n;  // svi.py:305
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    float s = 0.f;
    if (col < a.P)
        for (uint32_t r = rg; r < a.n_rows; r += 8) s += a.partials[r * stride + col];
    lds[rg * 32 + c] = s;
    __syncthreads();
    if (rg == 0 && col < a.P) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) tot += lds[g * 32 + c];
        float z;
        if (a.site_keys) {  // normal(site_key, leaf shape)[e] generated here (d3p.random.normal: word e of the key's stream)
            const int site = col >= a.K ? 1 : 0;
            const uint32_t e = (uint32_t)(col - (site ? a.K : 0));
            uint32_t key[16], o[16];
            load_key(a.site_keys + 16 * site, key);
            keystream_block(key, e >> 4, o);
            uint32_t word = o[0];
#pragma unroll
            for (int w = 1; w < 16; ++w) word = ((e & 15u) == (uint32_t)w) ? o[w] : word;
            z = bits_to_normal(word);
        } else {
            z = a.noise[col];
        }
        const float g = (tot / Bf + z * (a.h.dp_scale * (a.h.clip / n))) * a.obs_scale * factor;
        if (a.grad_out) a.grad_out[col] = g;
        const int i = *a.step;
        float x = a.params[col], m = a.adam_m[col], v = a.adam_v[col];
        m = (1.0f - a.h.b1) * g + a.h.b1 * m;
        v = (1.0f - a.h.b2) * g * g + a.h.b2 * v;
        const float mhat = m / (1.0f - powf(a.h.b1, (float)(i + 1)));
        const float vhat = v / (1.0f - powf(a.h.b2, (float)(i + 1)));
        a.params[col] = x - a.h.lr * mhat / (sqrtf(vhat) + a.h.adam_eps);
        a.adam_m[col] = m;
        a.adam_v[col] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.loss_out) *a.loss_out = (ls / Bf) * a.obs_scale * factor;  // svi.py:342, :306
}

#define D3P_GMM_MAX_WAVES 4096u

struct GmmWorkspace {
    double* pack;
    double* dir;
    float* partials;
    float* reduced;   // D3P_GMM_CHUNKS x (P + 2)
    float* noise;
    float* meta;
    uint32_t* keys;   // 3 x 16 (split of the state key) + 2 x 16 (site keys) + 16 (folded batch key) + jax key (2)
    uint32_t* idx;    // B
    int32_t* step_saved;  // optimiser step index of the update in flight (k_gmm_pre -> k_gmm_finalize)
};

static size_t gmm_carve(const d3p_gmm_model* m, uint32_t B, char* base, GmmWorkspace* ws)
{
    const size_t K = (size_t)m->K, P = K + K * (size_t)m->d;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up_g(bytes, 256); return base ? base + o : nullptr; };
    char* q;
    q = take((2 * K + 3) * sizeof(double)); if (ws) ws->pack = (double*)q;
    q = take((size_t)B * K * 2 * sizeof(double)); if (ws) ws->dir = (double*)q;
    q = take((size_t)D3P_GMM_MAX_WAVES * (P + 2) * sizeof(float)); if (ws) ws->partials = (float*)q;
    q = take((size_t)D3P_GMM_CHUNKS * (P + 2) * sizeof(float)); if (ws) ws->reduced = (float*)q;
    q = take(P * sizeof(float)); if (ws) ws->noise = (float*)q;
    q = take(2 * sizeof(float)); if (ws) ws->meta = (float*)q;
    q = take((6 * 16 + 2) * sizeof(uint32_t)); if (ws) ws->keys = (uint32_t*)q;
    q = take((size_t)B * sizeof(uint32_t)); if (ws) ws->idx = (uint32_t*)q;
    q = take(sizeof(int32_t)); if (ws) ws->step_saved = (int32_t*)q;
    return off;
}

static int gmm_validate(const d3p_gmm_model* model, const char* what)
{
    if (!model) return fail(D3P_E_INVALID_ARG, "%s: null model", what);
    if (!(model->K >= 1 && model->d >= 1 && model->prior_mu_scale > 0.f && model->inv_obs > 0.f))
        return fail(D3P_E_INVALID_ARG, "%s: bad model (K, d >= 1, prior_mu_scale > 0, inv_obs > 0)", what);
    if (model->K > 32 || model->d > 256 || (model->K > 16 && model->d > 128))
        return fail(D3P_E_UNSUPPORTED, "%s: supported shapes are K <= 16 with d <= 256 and K <= 32 with d <= 128 (K = %d, d = %d)",
                    what, model->K, model->d);
    return D3P_OK;
}

static void gmm_fill(GmmArgs* a, const d3p_gmm_model* model, const float* params, const float* X, const uint32_t* idx,
                     const uint8_t* mask, uint32_t B, const uint32_t* jax_key, float clip)
{
    memset(a, 0, sizeof(*a));
    a->params = params;
    a->X = X;
    a->idx = idx;
    a->mask = mask;
    a->jax_key = jax_key;
    a->B = B;
    a->B_total = B;
    a->pos0 = 0;
    a->K = model->K;
    a->d = model->d;
    a->inv_ps2 = 1.0f / (model->prior_mu_scale * model->prior_mu_scale);
    a->log_ps = logf(model->prior_mu_scale);
    a->lik_scale = model->lik_scale;
    a->inv_obs = model->inv_obs;
    a->obs_scale = 1.0f / model->inv_obs;
    a->clip = clip;
}

template <bool SUM>
static int gmm_launch_px(hipStream_t s, const d3p_gmm_model* model, const GmmArgs& a, uint32_t n_waves)
{
    const dim3 grid(cdiv((uint64_t)n_waves * 64, 256)), block(256);
    const int KH = (model->K + 1) / 2 <= 8 ? 8 : 16, DS = (model->d + 63) / 64;
#define D3P_GMM_LAUNCH(KH_, DS_)                                                                 \
    if (model->K % 2 == 0)                                                                       \
        hipLaunchKernelGGL((k_gmm_px<KH_, DS_, SUM, true>), grid, block, 0, s, a);               \
    else                                                                                         \
        hipLaunchKernelGGL((k_gmm_px<KH_, DS_, SUM, false>), grid, block, 0, s, a)
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
    return check_launch("k_gmm_px");
}

}  // namespace d3p

using namespace d3p;

extern "C" {

size_t d3p_gmm_px_grads_workspace(int32_t K, uint32_t B)
{
    const size_t k = (size_t)(K > 0 ? K : 0);
    return align_up_g((2 * k + 3) * sizeof(double), 256) + align_up_g((size_t)B * k * 2 * sizeof(double), 256);
}

int d3p_gmm_px_grads(void* stream, const d3p_gmm_model* model, const float* params_dev, const float* X_dev,
                     const uint8_t* mask_dev, uint32_t B, const uint32_t* jax_key_dev, float* px_loss_dev, float* px_grads_dev,
                     float* meta_dev, float* latents_out_dev, void* workspace_dev, size_t workspace_bytes)
{
    D3P_REQUIRE(model && params_dev && X_dev && jax_key_dev && px_loss_dev && px_grads_dev && meta_dev && workspace_dev,
                "d3p_gmm_px_grads: null pointer");
    D3P_REQUIRE(B >= 1, "d3p_gmm_px_grads: B must be >= 1");
    if (int rc = gmm_validate(model, "d3p_gmm_px_grads")) return rc;
    if (workspace_bytes < d3p_gmm_px_grads_workspace(model->K, B)) return fail(D3P_E_WORKSPACE, "d3p_gmm_px_grads: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double* pack = (double*)workspace_dev;
    double* dir = (double*)((char*)workspace_dev + align_up_g((2 * (size_t)model->K + 3) * sizeof(double), 256));
    hipLaunchKernelGGL(k_gmm_pack, dim3(1), dim3(64), 0, s, params_dev, model->K, pack);
    hipLaunchKernelGGL(k_gmm_mask_meta, dim3(1), dim3(256), 0, s, mask_dev, B, meta_dev);
    hipLaunchKernelGGL(k_gmm_dirichlet, dim3(cdiv((uint64_t)B * model->K, 256)), dim3(256), 0, s, (const double*)pack, jax_key_dev, B, B, 0u,
                       model->K, dir);
    GmmArgs a;
    gmm_fill(&a, model, params_dev, X_dev, nullptr, mask_dev, B, jax_key_dev, 1.0f);
    a.pack = pack;
    a.dir = dir;
    a.meta = meta_dev;
    a.px_loss = px_loss_dev;
    a.px_grads = px_grads_dev;
    a.latents_out = latents_out_dev;
    return gmm_launch_px<false>(s, model, a, B);
}

size_t d3p_gmm_evaluate_workspace(const d3p_gmm_model* model, uint32_t B)
{
    if (!model || model->K < 1 || model->d < 1) return 0;
    const size_t n = (size_t)model->K * model->d;
    return 2 * align_up_g(n * sizeof(float), 256) + align_up_g((size_t)model->K * sizeof(float), 256) +
           align_up_g((size_t)B * sizeof(float), 256) + 256;
}

int d3p_gmm_evaluate(void* stream, const d3p_gmm_model* model, const float* params_dev, const float* X_dev, uint32_t B,
                     const uint32_t* jax_key_dev, float* loss_dev, void* workspace_dev, size_t workspace_bytes)
{
    D3P_REQUIRE(params_dev && X_dev && jax_key_dev && loss_dev && workspace_dev, "d3p_gmm_evaluate: null pointer");
    D3P_REQUIRE(B >= 1, "d3p_gmm_evaluate: B must be >= 1");
    if (int rc = gmm_validate(model, "d3p_gmm_evaluate")) return rc;
    if (workspace_bytes < d3p_gmm_evaluate_workspace(model, B)) return fail(D3P_E_WORKSPACE, "d3p_gmm_evaluate: workspace too small");
    const size_t n = (size_t)model->K * model->d;
    char* q = (char*)workspace_dev;
    float* mus = (float*)q; q += align_up_g(n * sizeof(float), 256);
    float* sigs = (float*)q; q += align_up_g(n * sizeof(float), 256);
    float* pis = (float*)q; q += align_up_g((size_t)model->K * sizeof(float), 256);
    float* ll = (float*)q; q += align_up_g((size_t)B * sizeof(float), 256);
    float* lat = (float*)q;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gmm_eval_latents, dim3(1), dim3(256), 0, s, params_dev, jax_key_dev, model->K, model->d,
                       1.0f / (model->prior_mu_scale * model->prior_mu_scale), logf(model->prior_mu_scale), mus, sigs, pis, lat);
    if (int rc = d3p_gmm_log_prob(stream, X_dev, B, model->d, mus, sigs, pis, model->K, ll)) return rc;
    hipLaunchKernelGGL(k_gmm_eval_finish, dim3(1), dim3(256), 0, s, (const float*)ll, B, (const float*)lat, model->lik_scale, loss_dev);
    return check_launch("d3p_gmm_evaluate");
}

size_t d3p_dpvi_gmm_workspace(const d3p_gmm_model* model, uint32_t B)
{
    if (!model || model->K < 1 || model->d < 1) return 0;
    return gmm_carve(model, B, nullptr, nullptr);
}

// out[c] = sum over the D3P_GMM_CHUNKS reduced rows (fixed order): the rank's [clipped sums | loss sum | n]
__global__ void __launch_bounds__(256) k_gmm_fold(const float* __restrict__ reduced, int width, float* __restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= width) return;
    float s = 0.f;
    for (uint32_t r = 0; r < D3P_GMM_CHUNKS; ++r) s += reduced[(size_t)r * width + c];
    out[c] = s;
}

// One DPSVI.update (svi.py:395-434) for the mixture model, enqueued on `stream` without host synchronisation.
// stage: 0 = the whole update; 1 = local sums only (keys and pack without touching the state, the rank's P + 2 sums folded
// into sums_io); 2 = apply only (sums_io holds the reduced sums of the whole batch of B_total examples).  B_total / pos0:
// the B examples are positions pos0 .. of a global batch of B_total (0 / 0: B is the whole batch).
// batch_key_dev != nullptr (run loop): the batch of this step is get_batch(batch_index, batch_key) of
// subsample_batchify_data (minibatch.py:226-237) over n_rows rows -- fold_in in the pre kernel, Feistel indices into ws.idx
static int gmm_enqueue_update(hipStream_t s, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                              int slot, const float* X_dev, const uint32_t* idx_dev, const uint8_t* mask_dev, uint32_t B,
                              float* loss_dev, float* grad_out_dev, const GmmWorkspace& ws,
                              const uint32_t* batch_key_dev = nullptr, uint32_t batch_index = 0, uint32_t n_rows = 0,
                              int stage = 0, uint32_t B_total = 0, uint32_t pos0 = 0, float* sums_io = nullptr)
{
    int rc;
    const int K = model->K, P = K + K * model->d;
    uint32_t* cur_key = state->rng_key + 16 * (slot & 1);
    uint32_t* split3 = ws.keys;             // [next | gradient | perturbation]   (svi.py:208-211, :413-414)
    uint32_t* site_keys = ws.keys + 48;     // split(perturbation_key, 2)          (svi.py:491)
    uint32_t* jax_key = ws.keys + 96;       // convert_to_jax_rng_key(gradient_key) (svi.py:259)
    {
        GmmPreArgs pa;
        pa.cur_key = cur_key;
        pa.next_slot = state->rng_key + 16 * ((slot + 1) & 1);
        pa.keys = ws.keys;
        pa.params = state->params;
        pa.K = K;
        pa.pack = ws.pack;
        pa.step = state->step;
        pa.step_saved = ws.step_saved;
        pa.batch_key = batch_key_dev;
        pa.batch_index = batch_index;
        pa.advance = stage == 1 ? 0 : 1;
        hipLaunchKernelGGL(k_gmm_pre, dim3(1), dim3(128), 0, s, pa);
    }
    if (stage != 2) {
    if (batch_key_dev) {
        if ((rc = d3p_feistel_sample(s, ws.keys + 80, n_rows, B, ws.idx))) return rc;
        idx_dev = ws.idx;
    }
    (void)split3;
    hipLaunchKernelGGL(k_gmm_dirichlet, dim3(cdiv((uint64_t)B * K, 256)), dim3(256), 0, s, (const double*)ws.pack,
                       (const uint32_t*)jax_key, B, B_total ? B_total : B, pos0, K, ws.dir);
    GmmArgs a;
    gmm_fill(&a, model, state->params, X_dev, idx_dev, mask_dev, B, jax_key, hyper->clip);
    if (B_total) { a.B_total = B_total; a.pos0 = pos0; }
    a.pack = ws.pack;
    a.dir = ws.dir;
    a.partials = ws.partials;
    const uint32_t n_waves = B < D3P_GMM_MAX_WAVES ? B : D3P_GMM_MAX_WAVES;
    if ((rc = gmm_launch_px<true>(s, model, a, n_waves))) return rc;
    const uint32_t rows = cdiv((uint64_t)n_waves * 64, 256) * 4;  // every launched wavefront wrote a row
    hipLaunchKernelGGL(k_gmm_reduce, dim3(cdiv(P + 2, 64), D3P_GMM_CHUNKS), dim3(256), 0, s, (const float*)ws.partials, rows, P + 2,
                       ws.reduced);
    }
    if (stage == 1) {
        hipLaunchKernelGGL(k_gmm_fold, dim3(cdiv(P + 2, 256)), dim3(256), 0, s, (const float*)ws.reduced, P + 2, sums_io);
        return check_launch("d3p_dpvi_gmm_local_sums");
    }
    GmmFinalArgs f;
    f.partials = stage == 2 ? sums_io : ws.reduced;
    f.n_rows = stage == 2 ? 1u : D3P_GMM_CHUNKS;
    f.B = (stage == 2 && B_total) ? B_total : B;
    f.P = P;
    f.noise = nullptr;
    f.site_keys = site_keys;
    f.K = K;
    f.params = state->params;
    f.adam_m = state->adam_m;
    f.adam_v = state->adam_v;
    f.step = ws.step_saved;
    f.loss_out = loss_dev;
    f.grad_out = grad_out_dev;
    f.h = *hyper;
    f.obs_scale = 1.0f / model->inv_obs;
    hipLaunchKernelGGL(k_gmm_finalize, dim3(cdiv(P, 32)), dim3(256), 0, s, f);
    return check_launch("d3p_dpvi_gmm_update");
}

static int gmm_check_common(const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state, uint32_t B,
                            void* workspace_dev, size_t workspace_bytes, const char* what)
{
    if (int rc = gmm_validate(model, what)) return rc;
    D3P_REQUIRE(hyper && state && state->rng_key && state->params && state->adam_m && state->adam_v && state->step && workspace_dev,
                "null pointer");
    if (!(hyper->clip > 0.f) || !std::isfinite(hyper->clip))
        return fail(D3P_E_INVALID_ARG, "%s: the clipping threshold must be finite and greater than 0", what);
    D3P_REQUIRE(B >= 1, "B must be >= 1");
    if (workspace_bytes < d3p_dpvi_gmm_workspace(model, B)) return fail(D3P_E_WORKSPACE, "%s: workspace too small", what);
    return D3P_OK;
}

int d3p_dpvi_gmm_update(void* stream, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                        const float* X_dev, const uint8_t* mask_dev, uint32_t B, float* loss_dev, float* grad_out_dev,
                        void* workspace_dev, size_t workspace_bytes)
{
    if (int rc = gmm_check_common(model, hyper, state, B, workspace_dev, workspace_bytes, "d3p_dpvi_gmm_update")) return rc;
    D3P_REQUIRE(X_dev, "d3p_dpvi_gmm_update: null data pointer");
    GmmWorkspace ws;
    gmm_carve(model, B, (char*)workspace_dev, &ws);
    return gmm_enqueue_update((hipStream_t)stream, model, hyper, state, state->key_slot, X_dev, nullptr, mask_dev, B, loss_dev,
                              grad_out_dev, ws);
}

int d3p_dpvi_gmm_local_sums(void* stream, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                            const float* X_dev, const uint8_t* mask_dev, uint32_t B_local, uint32_t B_total, uint32_t pos0,
                            float* sums_dev, void* workspace_dev, size_t workspace_bytes)
{
    if (int rc = gmm_check_common(model, hyper, state, B_local, workspace_dev, workspace_bytes, "d3p_dpvi_gmm_local_sums")) return rc;
    D3P_REQUIRE(X_dev && sums_dev, "d3p_dpvi_gmm_local_sums: null pointer");
    D3P_REQUIRE((uint64_t)pos0 + B_local <= B_total, "d3p_dpvi_gmm_local_sums: pos0 + B_local must not exceed B_total");
    GmmWorkspace ws;
    gmm_carve(model, B_local, (char*)workspace_dev, &ws);
    return gmm_enqueue_update((hipStream_t)stream, model, hyper, state, state->key_slot, X_dev, nullptr, mask_dev, B_local, nullptr,
                              nullptr, ws, nullptr, 0, 0, 1, B_total, pos0, sums_dev);
}

int d3p_dpvi_gmm_apply(void* stream, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                       float* sums_dev, uint32_t B_total, uint32_t B_local, float* loss_dev, float* grad_out_dev,
                       void* workspace_dev, size_t workspace_bytes)
{
    if (int rc = gmm_check_common(model, hyper, state, B_local, workspace_dev, workspace_bytes, "d3p_dpvi_gmm_apply")) return rc;
    D3P_REQUIRE(sums_dev && B_total >= 1, "d3p_dpvi_gmm_apply: null pointer or empty batch");
    GmmWorkspace ws;
    gmm_carve(model, B_local, (char*)workspace_dev, &ws);
    return gmm_enqueue_update((hipStream_t)stream, model, hyper, state, state->key_slot, nullptr, nullptr, nullptr, B_local, loss_dev,
                              grad_out_dev, ws, nullptr, 0, 0, 2, B_total, 0, sums_dev);
}

int d3p_dpvi_gmm_run(void* stream, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                     const uint32_t* batch_key_dev, uint32_t first_batch, const float* X_dev, uint32_t n_rows, uint32_t B,
                     uint32_t num_steps, float* losses_dev, void* workspace_dev, size_t workspace_bytes)
{
    if (int rc = gmm_check_common(model, hyper, state, B, workspace_dev, workspace_bytes, "d3p_dpvi_gmm_run")) return rc;
    D3P_REQUIRE(X_dev && batch_key_dev, "d3p_dpvi_gmm_run: null pointer");
    D3P_REQUIRE(B <= n_rows, "d3p_dpvi_gmm_run: batch larger than the table");
    GmmWorkspace ws;
    gmm_carve(model, B, (char*)workspace_dev, &ws);
    for (uint32_t t = 0; t < num_steps; ++t) {
        int rc;
        if ((rc = gmm_enqueue_update((hipStream_t)stream, model, hyper, state, state->key_slot + (int)t, X_dev, nullptr, nullptr, B,
                                     losses_dev ? losses_dev + t : nullptr, nullptr, ws, batch_key_dev, first_batch + t, n_rows)))
            return rc;
    }
    return D3P_OK;
}

}  // extern "C"
