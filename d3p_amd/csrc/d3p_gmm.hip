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

#define D3P_GMM_ACC_R 4  // replicas of the fixed-point accumulator (workgroup b adds to replica b % R)

static inline size_t align_up_g(size_t v, size_t a) { return (v + a - 1) / a * a; }

// 1 / x for finite positive normal x: v_rcp_f64 (about 2^-24) and two Newton steps, to within an ulp or two of the division --
// a third of its instructions.  Used where a reciprocal sits inside a serial double-precision chain (the series below).
__device__ __forceinline__ double rcp_d(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

__device__ __forceinline__ double digamma_d(double x)
{
    double r = 0.0;
    while (x < 10.0) { r -= rcp_d(x); x += 1.0; }
    const double f = rcp_d(x * x);
    return r + log(x) - 0.5 / x
           - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760))))));
}

// d/dalpha of the Gamma(alpha, 1) quantile at fixed CDF value (see d3po_gamma_grad); psi1 = digamma(alpha + 1).
// The series is cut where its terms fall below 1e-11 of the sum (the oracle goes on to 1e-18; the result is used in float32, four
// orders of magnitude coarser: the cut at 1e-16 of round 3 ran about ten terms longer on the head kernel's serial chain).
__device__ __forceinline__ double gamma_grad_d(double alpha, double x, double psi1)
{
    if (!(x > 0.0)) return 0.0;
    double t = 1.0, h = 0.0, S = 1.0, Sp = 0.0;
    for (int n = 1; n < 2000; ++n) {
        const double rcp = rcp_d(alpha + n);
        t *= x * rcp;
        h += rcp;
        S += t;
        Sp -= t * h;
        if (t < 1e-11 * S && n > x) break;
    }
    return -(x * rcp_d(alpha)) * (S * (log(x) - psi1) + Sp);
}

__device__ __forceinline__ double open_unit_d(uint32_t b) { return ((double)b + 0.5) * (1.0 / 4294967296.0); }

__device__ __forceinline__ double gamma_sample_d(uint32_t k0, uint32_t k1, uint32_t comp, double alpha)
{
    // A concentration that is NaN or +inf (a diverged state: exp(alpha_log)) makes every acceptance test false: jax.random.gamma's
    // while_loop condition is false on NaN too and returns NaN at once -- so does this; and the loop is bounded whatever comes in
    // (acceptance >= 0.95 per attempt: 1024 misses do not happen), a kernel must not spin on a parameter value.
    if (!(alpha < 1.7976931348623157e308)) return alpha;
    const double a = alpha < 1.0 ? alpha + 1.0 : alpha;
    const double dd = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
    double g = 0.0;
    for (uint32_t attempt = 0; attempt < 1024u; ++attempt) {
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

// Dirichlet part of a step, one THREAD per (example, component), whole examples per workgroup (256 / K of them): the
// Gamma(alpha_k) draw g_k (Marsaglia-Tsang), its derivative wrt alpha_k at fixed CDF value, and -- after normalising over the
// example's components through LDS -- everything of the example's gradient that depends on the Dirichlet site alone:
//   dir[(p K + k) * 4 + {0, 1, 2, 3}] = { log pis_k, gs_k = g'_k / S, (alpha_k - 1) / pis_k, 1 / pis_k },   S = sum_k g_k
//   lqs[p] = sum_k (alpha_k - 1) log pis_k        gdraw[p K + k] = g_k (float; nullable, for latents_out)
// k_gmm_px finishes d/d alpha_log_k from these and the responsibilities with a dozen double operations on K lanes.
// sh: 2 x 256 doubles of LDS.  All 256 threads of the workgroup call it (p_ok: this thread has an example).
#define D3P_GMM_DIR_ROW 4
__device__ __forceinline__ void gmm_dirichlet_thread(bool p_ok, uint32_t p, uint32_t k, int K, uint32_t kp0, uint32_t kp1, double alpha,
                                                     double psi1, double* sh, double* __restrict__ dir, double* __restrict__ lqs,
                                                     float* __restrict__ gdraw)
{
    const int tid = threadIdx.x, e0 = tid - (int)k;  // first thread of this example
    double g = 1.0, gp = 0.0;
    if (p_ok) {
        g = gamma_sample_d(kp0, kp1, k, alpha);
        gp = gamma_grad_d(alpha, g, psi1);
    }
    sh[tid] = g;
    __syncthreads();
    double S = 0.0;
    if (p_ok)
        for (int j = 0; j < K; ++j) S += sh[e0 + j];  // fixed order
    const double rS = p_ok ? 1.0 / S : 1.0;
    const double pis = g * rS, logpis = log(pis), rp = 1.0 / pis;
    sh[256 + tid] = (alpha - 1.0) * logpis;
    __syncthreads();
    if (!p_ok) return;
    double* row = dir + ((size_t)p * K + k) * D3P_GMM_DIR_ROW;
    row[0] = logpis;
    row[1] = gp * rS;
    row[2] = (alpha - 1.0) * rp;
    row[3] = rp;
    if (gdraw) gdraw[(size_t)p * K + k] = (float)g;
    if (k == 0) {
        double q = 0.0;
        for (int j = 0; j < K; ++j) q += sh[256 + e0 + j];
        lqs[p] = q;
    }
}

// stage API (d3p_gmm_px_grads): keys derived here; data-parallel: the B examples are positions pos0 .. pos0 + B - 1 of a global
// batch of B_total, the keys of an example are functions of its GLOBAL position
__global__ void __launch_bounds__(256) k_gmm_dirichlet(const double* __restrict__ pack, const uint32_t* __restrict__ jax_key,
                                                       uint32_t B, uint32_t B_total, uint32_t pos0, int K, double* __restrict__ dir,
                                                       double* __restrict__ lqs, float* __restrict__ gdraw)
{
    __shared__ double sh[512];
    const uint32_t epw = 256u / (uint32_t)K, pl = threadIdx.x / (uint32_t)K, k = threadIdx.x % (uint32_t)K;
    const uint32_t p = blockIdx.x * epw + pl;
    const bool p_ok = pl < epw && p < B;
    uint32_t kp0 = 0, kp1 = 0, km0, km1, ks0, ks1;
    double alpha = 1.0, psi1 = 0.0;
    if (p_ok) {
        gmm_site_keys(jax_key, B_total, pos0 + p, kp0, kp1, km0, km1, ks0, ks1);
        alpha = pack[k];
        psi1 = pack[K + k] + 1.0 / alpha;  // psi(alpha + 1) = psi(alpha) + 1 / alpha
    }
    gmm_dirichlet_thread(p_ok, p, k, K, kp0, kp1, alpha, psi1, sh, dir, lqs, gdraw);
}

struct GmmArgs {
    const float* params;
    const double* pack;
    const double* dir;       // B x K x D3P_GMM_DIR_ROW and
    const double* lqs;       // B from gmm_dirichlet_thread
    const float* gdraw;      // B x K Gamma draws (latents_out only)
    const float* X;
    const uint32_t* idx;     // nullable: row of example p is X[idx[p]] (minibatch of a resident table)
    const uint8_t* mask;
    const uint32_t* jax_key;
    const float* meta;
    float* px_loss;
    float* px_grads;
    float* latents_out;  // nullable: B x (K + 2 K d): g, eps, sigs of every example (tests)
    const uint32_t* skeys;  // nullable: B x 6 words [k_pis | k_mus | k_sigs] of every example, prepared once per batch of steps
    long long* acc;      // SUM mode: D3P_GMM_ACC_R x D3P_ACC_COLS(P) fixed-point sums [sum_i c_i g_i | loss lo | n | loss hi | -]
    double sg;           // fixed-point scale of the gradient columns (2^40 / clip)
    uint32_t B;
    uint32_t B_total, pos0;  // key derivation: example p is position pos0 + p of a global batch of B_total (= B, 0 on one device)
    int K, d;
    float inv_ps2, log_ps, lik_scale, inv_obs, obs_scale, clip;
};

// SUM = false: materialise px_loss / px_grads (stage API).  SUM = true: clip each example's gradient by its joint
// L2 norm and accumulate (svi.py:310-348 fused into stage 1); every wavefront strides over the batch, the four wavefronts of
// a workgroup are summed through LDS in fixed order and the workgroup adds its P + 3 partials to one of D3P_GMM_ACC_R
// replicas of a 64-bit fixed-point accumulator (exact, hence independent of the order the workgroups arrive in).
// The example index is wave-uniform (readfirstlane), so the key loads / derivations run on the scalar unit.
// The unrolled component loop holds 4 KH DS values per lane; asking for 4 (2) resident waves per SIMD keeps the
// scheduler from interleaving all threefry chains at once (which drove the small shapes to 256 VGPRs, occupancy 1).
// FULLT: K == 2 KH and d == 64 DS exactly (BASELINE config 3: K = 16, d = 64) -- every lane / slot predicate is true and folds
// away (about a tenth of the instructions of the component loop).
template <int KH, int DS, bool SUM, bool PAIRED, bool FULLT = false>
__global__ void __launch_bounds__(256, (KH * DS <= 8 ? 3 : KH * DS <= 16 ? 2 : 1)) k_gmm_px(GmmArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t gw = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const uint32_t total_waves = (gridDim.x * blockDim.x) >> 6;
    const int K = FULLT ? 2 * KH : a.K, d = FULLT ? 64 * DS : a.d, Kh = FULLT ? KH : (K + 1) >> 1, P = K + K * d;
    float accg[2 * KH * DS], acca = 0.f, loss_acc = 0.f, n_acc = 0.f;
    float locv[2 * KH * DS];  // this lane's entries of mus_loc, loaded once (slot layout as wv / muv below)
#pragma unroll
    for (int kk = 0; kk < KH; ++kk)
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const int dd = lane + 64 * s;
            const bool ok0 = FULLT || (kk < Kh && dd < d), ok1 = FULLT || (ok0 && kk + Kh < K);
            accg[kk * DS + s] = accg[(KH + kk) * DS + s] = 0.f;
            locv[kk * DS + s] = ok0 ? a.params[K + kk * d + dd] : 0.f;
            locv[(KH + kk) * DS + s] = ok1 ? a.params[K + (kk + Kh) * d + dd] : 0.f;
        }
    // Dirichlet constants of this lane's component (lanes < K)
    double alpha = 1.0, psi_d = 0.0;  // psi(A0) - psi(alpha_k)
    if (lane < K) {
        alpha = a.pack[lane];
        psi_d = a.pack[2 * K] - a.pack[K + lane];
    }
    const double A0mK = a.pack[2 * K + 1] - (double)K, lq0 = a.pack[2 * K + 2];
    for (uint32_t p = gw; p < a.B; p += total_waves) {
    // components are handled in pairs (kk, kk + Kh), Kh = ceil(K / 2); for even K the pair shares its threefry calls
    // (words j and j + K d / 2 of jax's iota layout), for odd K every entry takes its own call
    const uint32_t n_lat = (uint32_t)(K * d);
    const float live = (a.mask ? a.mask[p] != 0 : true) ? 1.0f : 0.0f;
    if (SUM && live == 0.0f) continue;  // masked examples contribute nothing (svi.py:281)
    const size_t row = a.idx ? a.idx[p] : p;

    // ---- keys of the guide's sample sites (numpyro.handlers.seed over pis, mus, sigs)
    uint32_t kp0, kp1, km0, km1, ks0, ks1;
    if (a.skeys) {
        const uint32_t* sk = a.skeys + (size_t)p * 6;
        km0 = sk[2]; km1 = sk[3]; ks0 = sk[4]; ks1 = sk[5];
    } else {
        gmm_site_keys(a.jax_key, a.B_total, a.pos0 + p, kp0, kp1, km0, km1, ks0, ks1);
    }

    // ---- Dirichlet part (lanes < K, float64): finished per (example, component) by gmm_dirichlet_thread
    double logpis_d = 0.0, gs = 0.0, apm = 0.0, rp = 1.0;
    if (lane < K) {
        const double* drow = a.dir + ((size_t)p * K + lane) * D3P_GMM_DIR_ROW;
        logpis_d = drow[0];
        gs = drow[1];
        apm = drow[2];
        rp = drow[3];
    }
    const float logpis = (float)logpis_d;

    // ---- mus, sigs and the per-component log-densities (lanes <-> feature dimensions)
    float xs[DS];
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        const int dd = lane + 64 * s;
        xs[s] = (FULLT || dd < d) ? a.X[(size_t)row * d + dd] : 0.f;
    }
    float wv[2 * KH * DS], muv[2 * KH * DS];
    float acomp[2 * KH];
    float lmu = 0.f;  // sum of  -eps^2/2 + (mu/ps)^2/2 + log ps  over this lane's entries
    float* lat = (!SUM && a.latents_out) ? a.latents_out + (size_t)p * (K + 2 * n_lat) : nullptr;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        float ll0 = 0.f, ll1 = 0.f;
        if (FULLT || kk < Kh) {
            const bool has1 = FULLT || kk + Kh < K;
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                const int dd = lane + 64 * s;
                const bool ok0 = FULLT || dd < d, ok1 = ok0 && has1;
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
                // (both logarithms below take normal arguments -- the uniform is in [2^-24, 1), ex in [6e-8, 17] --, so the raw
                // v_log_f32 needs none of __logf's denormal scaling and fix-up: ten instructions less per logarithm, four per entry)
                const float ex0 = -0.693147182f * __builtin_amdgcn_logf(((float)(u0 >> 9) + 0.5f) * 1.1920928955078125e-07f);
                const float ex1 = -0.693147182f * __builtin_amdgcn_logf(((float)(u1 >> 9) + 0.5f) * 1.1920928955078125e-07f);
                const int i0 = kk * DS + s, i1 = (KH + kk) * DS + s;
                const float mu0 = ok0 ? locv[i0] + e0 : 0.f, mu1 = ok1 ? locv[i1] + e1 : 0.f;
                const float z0 = (xs[s] - mu0) * ex0, z1 = (xs[s] - mu1) * ex1;
                wv[i0] = ok0 ? z0 * ex0 : 0.f;
                wv[i1] = ok1 ? z1 * ex1 : 0.f;
                muv[i0] = mu0;
                muv[i1] = mu1;
                if (ok0) {
                    ll0 += __fmaf_rn(-0.5f * z0, z0, __fmaf_rn(0.693147182f, __builtin_amdgcn_logf(ex0), -D3P_HALF_LOG_2PI));
                    lmu += __fmaf_rn(-0.5f * e0, e0, __fmaf_rn(0.5f * a.inv_ps2 * mu0, mu0, a.log_ps));
                    if (lat) {
                        lat[K + j0] = e0;
                        lat[K + n_lat + j0] = 1.0f / ex0;
                    }
                }
                if (ok1) {
                    ll1 += __fmaf_rn(-0.5f * z1, z1, __fmaf_rn(0.693147182f, __builtin_amdgcn_logf(ex1), -D3P_HALF_LOG_2PI));
                    lmu += __fmaf_rn(-0.5f * e1, e1, __fmaf_rn(0.5f * a.inv_ps2 * mu1, mu1, a.log_ps));
                    if (lat) {
                        lat[K + j1] = e1;
                        lat[K + n_lat + j1] = 1.0f / ex1;
                    }
                }
            }
        }
        wave_sum2(ll0, ll1, acomp[kk], acomp[KH + kk]);   // (the pair's two sums in nine instructions instead of 2 x 10)
        __builtin_amdgcn_sched_barrier(0);  // one component pair at a time: keeps the live threefry chains (and VGPRs) bounded
    }
    lmu = wave_sum(lmu);
    if (lat && lane < K) lat[lane] = a.gdraw[(size_t)p * K + lane];

    // ---- mixture: a_k = log pis_k + ll_k, responsibilities r_k, loglik = logsumexp_k a_k  (gmm.py:71-86)
    float best = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        if (FULLT || kk < Kh) {
            const bool has1 = FULLT || kk + Kh < K;
            acomp[kk] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(logpis), kk));
            acomp[KH + kk] = has1 ? acomp[KH + kk] + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(logpis), has1 ? kk + Kh : 0))
                                  : -INFINITY;
            best = fmaxf(best, fmaxf(acomp[kk], acomp[KH + kk]));
        }
    }
    float se = 0.f;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        if (FULLT || kk < Kh) {
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
        if (FULLT || kk < Kh) {
            const bool has1 = FULLT || kk + Kh < K;
            const float ra = acomp[kk] * inv_se, rb = acomp[KH + kk] * inv_se;
            if (lane == kk) my_r = ra;
            if (lane == kk + Kh) my_r = rb;
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                const int dd = lane + 64 * s;
                if (FULLT || dd < d) {
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
    float ga = 0.f;
    if (lane < K) {
        const double dq = psi_d + logpis_d + gs * (apm - A0mK);
        const double dl = gs * ((double)my_r * rp - 1.0);
        ga = (float)(alpha * (double)a.inv_obs * (dq - (double)a.lik_scale * dl));
        if (!SUM) gr[lane] = ga * live;
    }
    const double lq = lq0 + a.lqs[p];
    const float L = a.inv_obs * (((float)lq + lmu) - a.lik_scale * loglik);
    if (SUM) {
        n2 = wave_sum(__fmaf_rn(ga, ga, n2));
        const float cf = 1.0f / fmaxf(1.0f, sqrtf(n2) / a.clip);  // svi.py:121-122
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            if (FULLT || kk < Kh) {
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
        extern __shared__ float red[];  // P + 2 floats: [gradient columns | loss | n], the workgroup's four wavefronts in turn
        const int wave = threadIdx.x >> 6;
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
                if (lane < K) red[lane] = (w ? red[lane] : 0.f) + acca;
#pragma unroll
                for (int kk = 0; kk < KH; ++kk) {
                    if (FULLT || kk < Kh) {
#pragma unroll
                        for (int s = 0; s < DS; ++s) {
                            const int dd = lane + 64 * s;
                            if (FULLT || dd < d) {
                                const int c0 = K + kk * d + dd, c1 = K + (kk + Kh) * d + dd;
                                red[c0] = (w ? red[c0] : 0.f) + accg[kk * DS + s];
                                if (FULLT || kk + Kh < K) red[c1] = (w ? red[c1] : 0.f) + accg[(KH + kk) * DS + s];
                            }
                        }
                    }
                }
                if (lane == 0) {
                    red[P] = (w ? red[P] : 0.f) + loss_acc;
                    red[P + 1] = (w ? red[P + 1] : 0.f) + n_acc;
                }
            }
            __syncthreads();
        }
        long long* out = a.acc + (size_t)(blockIdx.x % D3P_GMM_ACC_R) * D3P_ACC_COLS(P);
        const int tid = threadIdx.x;
        bool bad = false;
        for (int c = tid; c < P; c += 256) {
            const float v = red[c];
            bad |= !(fabsf(v) <= 3.0e38f);
            if (v != 0.f) atomicAdd(reinterpret_cast<unsigned long long*>(out + c), (unsigned long long)__double2ll_rn((double)v * a.sg));
        }
        if (tid < 3) {  // thread 0: loss, fine part; 1: example count; 2: loss, coarse part
            long long hi, lo;
            const bool ok = loss_split(red[P], hi, lo);
            if (tid != 1) bad |= !ok;
            const long long v = tid == 0 ? lo : tid == 1 ? (long long)red[P + 1] : hi;
            if (v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(out + P + tid), (unsigned long long)v);
        }
        // (workgroup 0) so does a parameter that is not finite, even when no example is valid: the reference's masked sum is NaN * 0 = NaN
        // there (svi.py:271-281), while this kernel evaluates no masked example
        if (blockIdx.x == 0)
            for (int c = tid; c < P; c += 256) bad |= !(fabsf(a.params[c]) <= 3.402823466e38f);
        // a non-finite partial poisons the count column: the update that reads it yields NaN like the reference's float sums
        if (bad) atomicAdd(reinterpret_cast<unsigned long long*>(out + P + 1), 1ull << 44);
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

// ------------------------------------------------------------------------------------------
// The update loop of the mixture model (DPSVI.update, svi.py:395-434) as launches:
//   per batch of <= gmm_step_batch(B) steps (nothing here depends on the parameters):
//     k_gmm_prep_chain : the serial key chain -- (next, gradient, perturbation) = split(state_key, 3) per step
//                        (svi.py:208-211, :413-414), 4-lane ChaCha block, one workgroup
//     k_gmm_prep_steps : per step, in parallel: jax key of the gradient key (svi.py:259), the batchifier's fold_in and Feistel
//                        rows (minibatch.py:226-237), the three site keys of every example (numpyro seed handler over pis,
//                        mus, sigs -- one thread per example instead of once per lane of the example's wavefront), the
//                        per-site keys split(perturbation_key, 2) and their P normals (svi.py:487-491), Adam's bias terms
//   per step, two launches:
//     k_gmm_head : applies the PREVIOUS step's update (mean, Gaussian mechanism, Adam; svi.py:343-393) from the fixed-point
//                  sums -- every workgroup for the K Dirichlet columns it needs, the first ones for all columns into the other
//                  state buffer --, zeroes this step's accumulator, packs the Dirichlet constants and draws the Gamma variates
//                  with their implicit-reparametrisation derivatives, one thread per (example, component)
//     k_gmm_px   : per-example gradients, clip, sum into the accumulator (above)
//   k_gmm_flush : the update still pending after the last step, into the caller's arrays.
// The optimiser state ping-pongs between the caller's arrays (even steps of a run) and a workspace copy (odd steps), so no
// kernel reads a column another workgroup of the same launch writes.
// ------------------------------------------------------------------------------------------
struct GmmSlot {
    uint32_t grad_key[16];
    uint32_t pert_key[16];
    int32_t adam_i;
    uint32_t batch_i;
    float bc1, bc2;  // 1 - b1^(i+1), 1 - b2^(i+1)
    uint32_t pad[12];
};

// steps prepared per launch pair: bounded so that the prepared rows / site keys (28 bytes per example and step) stay small
static inline uint32_t gmm_step_batch(uint32_t B)
{
    const uint32_t n = (1u << 20) / (B ? B : 1u);
    return n < 1u ? 1u : n > 64u ? 64u : n;
}

struct GmmChainArgs {
    const uint32_t* in_key;
    uint32_t* out_key;      // nullable: the key after the K steps
    const int32_t* step;    // optimiser step counter of the state (not advanced here)
    int32_t step_add;       // steps of this run already prepared
    uint32_t batch0;        // batch index of the first step
    GmmSlot* slots;
    int K;
};

__global__ void __launch_bounds__(64) k_gmm_prep_chain(GmmChainArgs a)
{
    const int lane = threadIdx.x, q = lane & 3, child = (lane >> 2) < 3 ? (lane >> 2) : 0;
    const int32_t adam0 = *a.step + a.step_add;
    const uint32_t p0 = a.in_key[q];
    uint32_t p1 = a.in_key[4 + q], p2 = a.in_key[8 + q], p3 = a.in_key[12 + q];
    for (int t = 0; t < a.K; ++t) {
        uint32_t ka, kb;
        derive_child_quad_regs(p0, p1, p2, p3, (uint32_t)child, D3P_TAG_SPLIT, 0u, ka, kb);
        if (lane >= 4 && lane < 12) {  // gradient key (child 1), perturbation key (child 2)
            uint32_t* dst = lane < 8 ? a.slots[t].grad_key : a.slots[t].pert_key;
            dst[q] = p0;
            dst[4 + q] = ka;
            dst[8 + q] = kb;
            dst[12 + q] = 0u;
        } else if (lane == 12) {
            a.slots[t].adam_i = adam0 + t;
            a.slots[t].batch_i = a.batch0 + (uint32_t)t;
        }
        p1 = __shfl(ka, q);  // the next state key is child 0 (lanes 0..3): every quad continues from it
        p2 = __shfl(kb, q);
        p3 = 0u;
    }
    if (lane < 4 && a.out_key) {
        a.out_key[q] = p0;
        a.out_key[4 + q] = p1;
        a.out_key[8 + q] = p2;
        a.out_key[12 + q] = p3;
    }
}

struct GmmPrepArgs {
    GmmSlot* slots;
    const uint32_t* batch_key;  // nullable: the step's examples are the rows the caller passes
    uint32_t* idx;              // K x B rows of the Feistel batches
    uint32_t* skeys;            // K x B x 6
    float* noise;               // K x P
    uint32_t B, B_total, pos0;  // example p of this rank is position pos0 + p of a batch of B_total
    uint32_t capacity;
    int bits_lower, bits_upper;
    int Kc, P;
    float b1, b2;
};

// grid (ceil(B / 256) + 1, K): blockIdx.y = step of the batch; the last x-block of a step produces the step's noise and scalars
__global__ void __launch_bounds__(256) k_gmm_prep_steps(GmmPrepArgs a)
{
    __shared__ uint32_t sh_key[2][16], sh_jax[2], sh_rc[32];
    const int tid = threadIdx.x, t = blockIdx.y;
    GmmSlot* slot = a.slots + t;
    if (blockIdx.x + 1 < gridDim.x) {
        if (tid == 0) {  // convert_to_jax_rng_key(gradient_key) (svi.py:259; random/__init__.py:155)
            uint32_t k[16], o[16];
            load_key(slot->grad_key, k);
            keystream_block(k, 0u, o);
            sh_jax[0] = o[0];
            sh_jax[1] = o[1];
        } else if (tid == 64 && a.batch_key) {  // fold_in(batchifier_state, i) (minibatch.py:230)
            uint32_t k[16], c[16];
            load_key(a.batch_key, k);
            derive_child(k, 0u, slot->batch_i, D3P_TAG_FOLD, c);
#pragma unroll
            for (int w = 0; w < 16; ++w) sh_key[0][w] = c[w];
        }
        __syncthreads();
        if ((tid == 64 || tid == 128) && a.batch_key) {  // round constants (util.py:240-246)
            const uint32_t b = tid == 64 ? 0u : 1u;
            uint32_t k[16], o[16];
#pragma unroll
            for (int w = 0; w < 16; ++w) k[w] = sh_key[0][w];
            keystream_block(k, b, o);
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const int g = 16 * (int)b + w;
                if (g < 30) sh_rc[g] = (g % 3 == 0) ? (o[w] | 1u) : o[w];
            }
        }
        __syncthreads();
        const uint32_t p = blockIdx.x * 256u + (uint32_t)tid;
        if (p < a.B) {
            if (a.batch_key) a.idx[(size_t)t * a.B + p] = feistel_permute_dev(sh_rc, a.capacity, a.bits_lower, a.bits_upper, p);
            uint32_t* sk = a.skeys + ((size_t)t * a.B + p) * 6;
            gmm_site_keys(sh_jax, a.B_total, a.pos0 + p, sk[0], sk[1], sk[2], sk[3], sk[4], sk[5]);
        }
        return;
    }
    // ---- per-site keys split(perturbation_key, 2) (svi.py:491), then normal(site_key, leaf shape) (d3p.random.normal: word e
    // of the key's stream); site 0 = alpha_log (K), site 1 = mus_loc (K d)
    if (tid < 2) {
        uint32_t k[16], c[16];
        load_key(slot->pert_key, k);
        derive_child(k, (uint32_t)tid, 0u, D3P_TAG_SPLIT, c);
#pragma unroll
        for (int w = 0; w < 16; ++w) sh_key[tid][w] = c[w];
    } else if (tid == 64) {
        const float i1 = (float)(slot->adam_i + 1);
        slot->bc1 = 1.0f - powf(a.b1, i1);
        slot->bc2 = 1.0f - powf(a.b2, i1);
    }
    __syncthreads();
    const int n0 = a.Kc, n1 = a.P - a.Kc, nb0 = (n0 + 15) >> 4, nb1 = (n1 + 15) >> 4;
    float* noise = a.noise + (size_t)t * a.P;
    for (int blk = tid; blk < nb0 + nb1; blk += 256) {
        const int site = blk >= nb0 ? 1 : 0, b = blk - (site ? nb0 : 0);
        uint32_t key[16], o[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) key[w] = sh_key[site][w];
        keystream_block(key, (uint32_t)b, o);
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int e = 16 * b + w;
            if (e < (site ? n1 : n0)) noise[(site ? n0 : 0) + e] = bits_to_normal(o[w]);
        }
    }
}

// The pending update of one step: the sums come from the fixed-point accumulator (this device's examples are the whole batch)
// or as P + 2 floats (data-parallel apply: the all-reduced [clipped sums | loss | n] of d3p_dpvi_gmm_local_sums).
struct GmmPending {
    const long long* acc;   // D3P_GMM_ACC_R x D3P_ACC_COLS(P), or nullptr
    const float* fsums;     // P + 2, or nullptr
    const float* noise;     // P normals of that step
    const GmmSlot* slot;
    const float* in[3];     // params, adam_m, adam_v before the update
    float* out[3];          // ... after (may alias `in` only in k_gmm_flush, where every column has one reader)
    float* loss_out;        // nullable
    float* grad_out;        // nullable
    int P;
    float Bf, dp_scale, clip, obs_scale, lr, b1, b2, adam_eps;
    double inv_sg;
};

__device__ __forceinline__ float gmm_pending_count(const GmmPending& u)
{
    if (u.fsums) return u.fsums[u.P + 1];
    long long n = 0;
    for (int r = 0; r < D3P_GMM_ACC_R; ++r) n += u.acc[(size_t)r * D3P_ACC_COLS(u.P) + u.P + 1];
    return n >= (1ll << 40) ? __builtin_nanf("") : (float)n;
}

__device__ __forceinline__ float gmm_pending_loss(const GmmPending& u, float factor)
{
    float ls;
    if (u.fsums) {
        ls = u.fsums[u.P];
    } else {
        long long lo = 0, hi = 0;
        for (int r = 0; r < D3P_GMM_ACC_R; ++r) {
            lo += u.acc[(size_t)r * D3P_ACC_COLS(u.P) + u.P];
            hi += u.acc[(size_t)r * D3P_ACC_COLS(u.P) + u.P + 2];
        }
        ls = (float)loss_join(hi, lo);
    }
    return __fmul_rn(__fmul_rn(__fdiv_rn(ls, u.Bf), u.obs_scale), factor);  // svi.py:342, :306
}

// mean over the padded batch (svi.py:343-346), Gaussian mechanism (svi.py:365-375), rescale (svi.py:377), numpyro Adam
// (svi.py:379-393) for column c.  Every rounding is explicit (no fused multiply-add left to the compiler): the head kernel,
// the flush kernel and every workgroup that recomputes a Dirichlet column must agree bit for bit.
__device__ __forceinline__ float gmm_apply_column(const GmmPending& u, int c, float n, float factor, float& m, float& v, float& g)
{
    float tot;
    if (u.fsums) {
        tot = u.fsums[c];
    } else {
        long long s = 0;
        for (int r = 0; r < D3P_GMM_ACC_R; ++r) s += u.acc[(size_t)r * D3P_ACC_COLS(u.P) + c];
        tot = (float)((double)s * u.inv_sg);
    }
    const float noise_scale = __fmul_rn(u.dp_scale, __fdiv_rn(u.clip, n));
    g = __fmul_rn(__fmul_rn(__fadd_rn(__fdiv_rn(tot, u.Bf), __fmul_rn(u.noise[c], noise_scale)), u.obs_scale), factor);
    m = __fadd_rn(__fmul_rn(1.0f - u.b1, g), __fmul_rn(u.b1, u.in[1][c]));
    v = __fadd_rn(__fmul_rn(__fmul_rn(1.0f - u.b2, g), g), __fmul_rn(u.b2, u.in[2][c]));
    const float mhat = __fdiv_rn(m, u.slot->bc1), vhat = __fdiv_rn(v, u.slot->bc2);
    return __fsub_rn(u.in[0][c], __fdiv_rn(__fmul_rn(u.lr, mhat), __fadd_rn(__fsqrt_rn(vhat), u.adam_eps)));
}

// One link of the key chain of the NEXT batch of steps, made by an extra workgroup of a k_gmm_head launch (consecutive launches
// are ordered by the stream, so link i simply continues where link i - 1 left ws.chain_key): k_gmm_prep_chain's 0.9 us per step
// -- serial, in front of every batch of 64 steps -- hides behind the head kernel's latency chain.  slot == nullptr: no link.
struct GmmLink {
    uint32_t* key;        // running key of the chain (16 words)
    GmmSlot* slot;        // the slot of the step the link is for
    const int32_t* step;  // optimiser step counter of the state (constant during the run)
    int32_t step_add;
    uint32_t batch_i;
};

__device__ __forceinline__ void gmm_chain_link(const GmmLink& l)
{
    const int lane = threadIdx.x & 63, q = lane & 3, child = (lane >> 2) < 3 ? (lane >> 2) : 0;
    const uint32_t p0 = l.key[q], p1 = l.key[4 + q], p2 = l.key[8 + q], p3 = l.key[12 + q];
    uint32_t ka, kb;
    derive_child_quad_regs(p0, p1, p2, p3, (uint32_t)child, D3P_TAG_SPLIT, 0u, ka, kb);
    if (lane < 4) {  // the next state key (child 0)
        l.key[4 + q] = ka;
        l.key[8 + q] = kb;
        l.key[12 + q] = 0u;
    } else if (lane < 12) {  // gradient key (child 1), perturbation key (child 2)
        uint32_t* dst = lane < 8 ? l.slot->grad_key : l.slot->pert_key;
        dst[q] = p0;
        dst[4 + q] = ka;
        dst[8 + q] = kb;
        dst[12 + q] = 0u;
    } else if (lane == 12) {
        l.slot->adam_i = *l.step + l.step_add;
        l.slot->batch_i = l.batch_i;
    }
}

struct GmmHeadArgs {
    GmmPending prev;          // valid when apply_prev
    int apply_prev;
    const float* params;      // parameters of THIS step when there is no pending update (= prev.out[0] otherwise)
    long long* acc_zero;      // this step's accumulator
    uint32_t acc_words;
    double* pack;
    double* dir;
    double* lqs;
    const uint32_t* skeys;    // B x 6
    const uint8_t* mask;      // nullable
    uint32_t B;
    int K;
    GmmLink link;             // link.slot != nullptr: the LAST workgroup of the grid makes this link and nothing else
};

__global__ void __launch_bounds__(256) k_gmm_head(GmmHeadArgs a)
{
    __shared__ double sh_alpha[32], sh_psi1[32], sh_dir[512];
    const int tid = threadIdx.x, K = a.K;
    const uint32_t n_blocks = gridDim.x - (a.link.slot ? 1u : 0u);
    if (blockIdx.x == n_blocks) {  // (only when there is a link)
        if (tid < 64) gmm_chain_link(a.link);
        return;
    }
    const uint32_t gtid = blockIdx.x * 256u + (uint32_t)tid, gsize = n_blocks * 256u;
    float n = 0.f, factor = 0.f;
    if (a.apply_prev) {
        n = gmm_pending_count(a.prev);
        factor = (n == 0.f) ? 0.f : __fdiv_rn(a.prev.Bf, n);  // svi.py:305
    }
    if (tid < K) {
        float x, m, v, g;
        if (a.apply_prev) x = gmm_apply_column(a.prev, tid, n, factor, m, v, g);
        else x = a.params[tid];
        sh_alpha[tid] = exp((double)x);
    }
    if (a.apply_prev) {
        for (uint32_t c = gtid; c < (uint32_t)a.prev.P; c += gsize) {
            float m, v, g;
            const float x = gmm_apply_column(a.prev, (int)c, n, factor, m, v, g);
            a.prev.out[0][c] = x;
            a.prev.out[1][c] = m;
            a.prev.out[2][c] = v;
        }
        if (gtid == 0 && a.prev.loss_out) *a.prev.loss_out = gmm_pending_loss(a.prev, factor);
    }
    for (uint32_t i = gtid; i < a.acc_words; i += gsize) a.acc_zero[i] = 0;
    __syncthreads();
    if (blockIdx.x == 0 && tid >= 64 && tid < 128) {
        // pack (doubles): [alpha_k (K) | psi(alpha_k) (K) | psi(A0), A0, lgamma(A0) - sum lgamma(alpha_k) - lgamma(K)]
        const int k = tid - 64;
        double alpha = 0.0, lg = 0.0;
        if (k < K) {
            alpha = sh_alpha[k];
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
    // Dirichlet part: whole examples per workgroup (gmm_dirichlet_thread)
    if (tid < K) sh_psi1[tid] = digamma_d(sh_alpha[tid] + 1.0);
    __syncthreads();
    const uint32_t epw = 256u / (uint32_t)K, pl = (uint32_t)tid / (uint32_t)K, k = (uint32_t)tid % (uint32_t)K;
    for (uint32_t p0 = blockIdx.x * epw; p0 < a.B; p0 += n_blocks * epw) {  // (one pass: the grid covers the batch)
        const uint32_t p = p0 + pl;
        const bool p_ok = pl < epw && p < a.B && !(a.mask && a.mask[p] == 0);  // masked examples are skipped by k_gmm_px
        uint32_t kp0 = 0, kp1 = 0;
        if (p_ok) {
            kp0 = a.skeys[(size_t)p * 6];
            kp1 = a.skeys[(size_t)p * 6 + 1];
        }
        gmm_dirichlet_thread(p_ok, p, k, K, kp0, kp1, sh_alpha[k], sh_psi1[k], sh_dir, a.dir, a.lqs, nullptr);
    }
}

struct GmmFlushArgs {
    GmmPending prev;
    int32_t* step;  // nullable: the state's optimiser step counter, set to the applied step's index + 1
    const uint32_t* key_src;  // nullable: the chain's running key after the run's last link (links made in k_gmm_head launches) ...
    uint32_t* key_dst;        // ... goes to the state's key slot
};

__global__ void __launch_bounds__(256) k_gmm_flush(GmmFlushArgs a)
{
    const uint32_t gtid = blockIdx.x * 256u + threadIdx.x, gsize = gridDim.x * 256u;
    const float n = gmm_pending_count(a.prev);
    const float factor = (n == 0.f) ? 0.f : __fdiv_rn(a.prev.Bf, n);
    for (uint32_t c = gtid; c < (uint32_t)a.prev.P; c += gsize) {
        float m, v, g;
        const float x = gmm_apply_column(a.prev, (int)c, n, factor, m, v, g);
        a.prev.out[0][c] = x;
        a.prev.out[1][c] = m;
        a.prev.out[2][c] = v;
        if (a.prev.grad_out) a.prev.grad_out[c] = g;
    }
    if (gtid == 0) {
        if (a.prev.loss_out) *a.prev.loss_out = gmm_pending_loss(a.prev, factor);
        if (a.step) *a.step = a.prev.slot->adam_i + 1;
    }
    if (a.key_src && gtid < 16u) a.key_dst[gtid] = a.key_src[gtid];
}

// out[c] = this rank's [clipped sums | loss sum | n] as floats (d3p_dpvi_gmm_local_sums)
__global__ void __launch_bounds__(256) k_gmm_fold(const long long* __restrict__ acc, int P, double inv_sg, float* __restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P + 2) return;
    const size_t PA = D3P_ACC_COLS(P);
    if (c < P) {
        long long s = 0;
        for (int r = 0; r < D3P_GMM_ACC_R; ++r) s += acc[r * PA + c];
        out[c] = (float)((double)s * inv_sg);
    } else if (c == P) {
        long long lo = 0, hi = 0;
        for (int r = 0; r < D3P_GMM_ACC_R; ++r) { lo += acc[r * PA + P]; hi += acc[r * PA + P + 2]; }
        out[c] = (float)loss_join(hi, lo);
    } else {
        long long n = 0;
        for (int r = 0; r < D3P_GMM_ACC_R; ++r) n += acc[r * PA + P + 1];
        out[c] = n >= (1ll << 40) ? __builtin_nanf("") : (float)n;
    }
}

struct GmmWorkspace {
    double* pack;
    double* dir;          // B x K x D3P_GMM_DIR_ROW
    double* lqs;          // B
    long long* acc;       // 2 x D3P_GMM_ACC_R x D3P_ACC_COLS(P): steps alternate
    float* pp[3];         // second buffer of the ping-ponged optimiser state
    GmmSlot* slots[2];    // gmm_step_batch(B) slots, alternating by prepared batch (the head of a batch's first step still
    float* noise[2];      // ... reads the last slot / noise row of the batch before)
    uint32_t* idx;        // step_batch x B
    uint32_t* skeys;      // step_batch x B x 6
    uint32_t* chain_key;  // 16: key chain between prepared batches
};

static size_t gmm_carve(const d3p_gmm_model* m, uint32_t B, char* base, GmmWorkspace* ws)
{
    const size_t K = (size_t)m->K, P = K + K * (size_t)m->d, SB = gmm_step_batch(B);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up_g(bytes, 256); return base ? base + o : nullptr; };
    char* q;
    q = take((2 * K + 3) * sizeof(double)); if (ws) ws->pack = (double*)q;
    q = take((size_t)B * K * D3P_GMM_DIR_ROW * sizeof(double)); if (ws) ws->dir = (double*)q;
    q = take((size_t)B * sizeof(double)); if (ws) ws->lqs = (double*)q;
    q = take(2 * (size_t)D3P_GMM_ACC_R * D3P_ACC_COLS(P) * sizeof(long long)); if (ws) ws->acc = (long long*)q;
    for (int i = 0; i < 3; ++i) { q = take(P * sizeof(float)); if (ws) ws->pp[i] = (float*)q; }
    for (int i = 0; i < 2; ++i) { q = take(SB * sizeof(GmmSlot)); if (ws) ws->slots[i] = (GmmSlot*)q; }
    for (int i = 0; i < 2; ++i) { q = take(SB * P * sizeof(float)); if (ws) ws->noise[i] = (float*)q; }
    q = take(SB * B * sizeof(uint32_t)); if (ws) ws->idx = (uint32_t*)q;
    q = take(SB * B * 6 * sizeof(uint32_t)); if (ws) ws->skeys = (uint32_t*)q;
    q = take(16 * sizeof(uint32_t)); if (ws) ws->chain_key = (uint32_t*)q;
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

// SUM: a resident grid (as many workgroups per CU as the launch bound allows, on 256 CUs; fewer when the batch is small) strides
// over the examples
template <bool SUM>
static int gmm_launch_px(hipStream_t s, const d3p_gmm_model* model, const GmmArgs& a)
{
    const int KH = (model->K + 1) / 2 <= 8 ? 8 : 16, DS = (model->d + 63) / 64;
    // workgroups per CU = the kernel's launch bound (measured at K = 16, d = 64, B = 8192: 3 -> 57.4, 4 (128 VGPRs, 68 bytes of
    // scratch) -> 61.2, 2 -> 77.7 us per step)
    const int occ = KH * DS <= 8 ? 3 : KH * DS <= 16 ? 2 : 1;
    const uint32_t P = (uint32_t)(model->K + model->K * model->d);
    const uint32_t wgs_all = (uint32_t)cdiv((uint64_t)a.B, 4), wgs_res = 256u * (uint32_t)occ;
    const dim3 grid(SUM ? (wgs_all < wgs_res ? wgs_all : wgs_res) : wgs_all), block(256);
    const size_t lds = SUM ? (size_t)(P + 2) * sizeof(float) : 0;
    const bool full = SUM && model->K == 2 * KH && model->d == 64 * DS;
#define D3P_GMM_LAUNCH(KH_, DS_)                                                                 \
    if (full)                                                                                    \
        hipLaunchKernelGGL((k_gmm_px<KH_, DS_, SUM, true, SUM>), grid, block, lds, s, a);        \
    else if (model->K % 2 == 0)                                                                  \
        hipLaunchKernelGGL((k_gmm_px<KH_, DS_, SUM, true>), grid, block, lds, s, a);             \
    else                                                                                         \
        hipLaunchKernelGGL((k_gmm_px<KH_, DS_, SUM, false>), grid, block, lds, s, a)
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
    return align_up_g((2 * k + 3) * sizeof(double), 256) + align_up_g((size_t)B * k * D3P_GMM_DIR_ROW * sizeof(double), 256) +
           align_up_g((size_t)B * sizeof(double), 256) + align_up_g((size_t)B * k * sizeof(float), 256);
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
    const size_t k = (size_t)model->K;
    char* q = (char*)workspace_dev;
    double* pack = (double*)q; q += align_up_g((2 * k + 3) * sizeof(double), 256);
    double* dir = (double*)q; q += align_up_g((size_t)B * k * D3P_GMM_DIR_ROW * sizeof(double), 256);
    double* lqs = (double*)q; q += align_up_g((size_t)B * sizeof(double), 256);
    float* gdraw = (float*)q;
    hipLaunchKernelGGL(k_gmm_pack, dim3(1), dim3(64), 0, s, params_dev, model->K, pack);
    hipLaunchKernelGGL(k_gmm_mask_meta, dim3(1), dim3(256), 0, s, mask_dev, B, meta_dev);
    hipLaunchKernelGGL(k_gmm_dirichlet, dim3(cdiv(B, 256u / (uint32_t)model->K)), dim3(256), 0, s, (const double*)pack, jax_key_dev, B, B, 0u,
                       model->K, dir, lqs, gdraw);
    GmmArgs a;
    gmm_fill(&a, model, params_dev, X_dev, nullptr, mask_dev, B, jax_key_dev, 1.0f);
    a.pack = pack;
    a.dir = dir;
    a.lqs = lqs;
    a.gdraw = gdraw;
    a.meta = meta_dev;
    a.px_loss = px_loss_dev;
    a.px_grads = px_grads_dev;
    a.latents_out = latents_out_dev;
    return gmm_launch_px<false>(s, model, a);
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

// num_steps consecutive DPSVI.update calls (svi.py:395-434) for the mixture model, enqueued on `stream` without host
// synchronisation (launch structure: the block comment above k_gmm_prep_chain).
// stage: 0 = whole updates; 1 = local sums of ONE step only (keys without touching the state, the rank's P + 2 sums folded
// into sums_io); 2 = apply ONE step only (sums_io holds the reduced sums of the whole batch of B_total examples).
// B_total / pos0: the B examples are positions pos0 .. of a global batch of B_total (0 / 0: B is the whole batch).
// batch_key_dev != nullptr (run loop): the batch of step t is get_batch(first_batch + t, batch_key) of
// subsample_batchify_data (minibatch.py:226-237) over n_rows rows; otherwise the B rows of X_dev are the batch.
static int gmm_enqueue_steps(hipStream_t s, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                             const float* X_dev, const uint8_t* mask_dev, uint32_t B, uint32_t num_steps, float* losses_dev,
                             float* grad_out_dev, const GmmWorkspace& ws, const uint32_t* batch_key_dev = nullptr,
                             uint32_t first_batch = 0, uint32_t n_rows = 0, int stage = 0, uint32_t B_total = 0, uint32_t pos0 = 0,
                             float* sums_io = nullptr)
{
    int rc;
    const int K = model->K, P = K + K * model->d;
    const uint32_t SB = gmm_step_batch(B);
    const size_t acc_words = (size_t)D3P_GMM_ACC_R * D3P_ACC_COLS(P);
    const double sg = 1099511627776.0 / (double)hyper->clip;  // 2^40 / C: |sum of clipped gradients| <= B C < 2^63 / sg
    float* const caller[3] = {state->params, state->adam_m, state->adam_v};
    auto buf = [&](uint32_t t, int i) { return (t & 1u) ? ws.pp[i] : caller[i]; };
    auto pending = [&](uint32_t t, GmmPending* u) {  // the update of step t of this run
        memset(u, 0, sizeof(*u));
        u->acc = ws.acc + (size_t)(t & 1u) * acc_words;
        u->noise = ws.noise[(t / SB) & 1u] + (size_t)(t % SB) * P;
        u->slot = ws.slots[(t / SB) & 1u] + (t % SB);
        for (int i = 0; i < 3; ++i) { u->in[i] = buf(t, i); u->out[i] = buf(t + 1, i); }
        u->loss_out = losses_dev ? losses_dev + t : nullptr;
        u->P = P;
        u->Bf = (float)B;
        u->dp_scale = hyper->dp_scale;
        u->clip = hyper->clip;
        u->obs_scale = 1.0f / model->inv_obs;
        u->lr = hyper->lr;
        u->b1 = hyper->b1;
        u->b2 = hyper->b2;
        u->adam_eps = hyper->adam_eps;
        u->inv_sg = 1.0 / sg;
    };
    int bits_lower = 0, bits_upper = 0;
    if (batch_key_dev) {
        uint32_t v = n_rows - 1, bits = 0;
        while (v) { ++bits; v >>= 1; }  // util.py:230
        bits_lower = (int)bits >> 1;
        bits_upper = (int)bits - bits_lower;
    }
    const uint32_t* cur_key = state->rng_key + 16 * (state->key_slot & 1);
    uint32_t* final_key = state->rng_key + 16 * ((state->key_slot + (int)num_steps) & 1);
    for (uint32_t t0 = 0; t0 < num_steps; t0 += SB) {
        const uint32_t Kb = num_steps - t0 < SB ? num_steps - t0 : SB, par = (t0 / SB) & 1u;
        const bool last = t0 + Kb == num_steps;
        // whole updates: the chain of every batch but the first is made, link by link, in the head launches of the batch before
        // it (GmmLink); the staged forms (one step only) keep the chain kernel
        const bool links = stage == 0;
        const uint32_t Kb_next = (links && !last) ? (num_steps - (t0 + Kb) < SB ? num_steps - (t0 + Kb) : SB) : 0u;
        if (t0 == 0 || !links) {
            GmmChainArgs ca;
            ca.in_key = t0 == 0 ? cur_key : ws.chain_key;
            ca.out_key = last ? (stage == 1 ? nullptr : final_key) : ws.chain_key;
            ca.step = state->step;
            ca.step_add = (int32_t)t0;
            ca.batch0 = first_batch + t0;
            ca.slots = ws.slots[par];
            ca.K = (int)Kb;
            hipLaunchKernelGGL(k_gmm_prep_chain, dim3(1), dim3(64), 0, s, ca);
        }
        GmmPrepArgs pa;
        pa.slots = ws.slots[par];
        pa.batch_key = batch_key_dev;
        pa.idx = ws.idx;
        pa.skeys = ws.skeys;
        pa.noise = ws.noise[par];
        pa.B = stage == 2 ? 0u : B;
        pa.B_total = B_total ? B_total : B;
        pa.pos0 = pos0;
        pa.capacity = n_rows;
        pa.bits_lower = bits_lower;
        pa.bits_upper = bits_upper;
        pa.Kc = K;
        pa.P = P;
        pa.b1 = hyper->b1;
        pa.b2 = hyper->b2;
        hipLaunchKernelGGL(k_gmm_prep_steps, dim3(cdiv(pa.B, 256) + 1, Kb), dim3(256), 0, s, pa);
        if ((rc = check_launch("k_gmm_prep"))) return rc;
        if (stage == 2) break;
        for (uint32_t t = t0; t < t0 + Kb; ++t) {
            const uint32_t* skeys = ws.skeys + (size_t)(t - t0) * B * 6;
            GmmHeadArgs ha;
            memset(&ha, 0, sizeof(ha));
            ha.apply_prev = t > 0 ? 1 : 0;
            if (t > 0) pending(t - 1, &ha.prev);
            ha.params = buf(t, 0);
            ha.acc_zero = ws.acc + (size_t)(t & 1u) * acc_words;
            ha.acc_words = (uint32_t)acc_words;
            ha.pack = ws.pack;
            ha.dir = ws.dir;
            ha.lqs = ws.lqs;
            ha.skeys = skeys;
            ha.mask = mask_dev;
            ha.B = B;
            ha.K = K;
            if (t - t0 < Kb_next) {  // link t - t0 of the next batch (its slots: the other buffer)
                ha.link.key = ws.chain_key;
                ha.link.slot = ws.slots[par ^ 1u] + (t - t0);
                ha.link.step = state->step;
                ha.link.step_add = (int32_t)(t0 + Kb + (t - t0));
                ha.link.batch_i = first_batch + t0 + Kb + (t - t0);
            }
            hipLaunchKernelGGL(k_gmm_head, dim3(cdiv(B, 256u / (uint32_t)K) + (ha.link.slot ? 1u : 0u)), dim3(256), 0, s, ha);
            GmmArgs a;
            gmm_fill(&a, model, buf(t, 0), X_dev, batch_key_dev ? ws.idx + (size_t)(t - t0) * B : nullptr, mask_dev, B, nullptr,
                     hyper->clip);
            a.skeys = skeys;
            a.pack = ws.pack;
            a.dir = ws.dir;
            a.lqs = ws.lqs;
            a.acc = ha.acc_zero;
            a.sg = sg;
            if ((rc = gmm_launch_px<true>(s, model, a))) return rc;
        }
    }
    if (stage == 1) {
        hipLaunchKernelGGL(k_gmm_fold, dim3(cdiv(P + 2, 256)), dim3(256), 0, s, (const long long*)ws.acc, P, 1.0 / sg, sums_io);
        return check_launch("d3p_dpvi_gmm_local_sums");
    }
    GmmFlushArgs fa;
    pending(num_steps - 1, &fa.prev);
    for (int i = 0; i < 3; ++i) fa.prev.out[i] = caller[i];
    fa.prev.grad_out = grad_out_dev;
    if (stage == 2) {
        fa.prev.acc = nullptr;
        fa.prev.fsums = sums_io;
        fa.prev.Bf = (float)(B_total ? B_total : B);
    }
    fa.step = state->step;
    fa.key_src = nullptr;
    fa.key_dst = nullptr;
    if (stage == 0 && num_steps > SB) {  // the last batch's links were made in head launches: their key is in ws.chain_key
        fa.key_src = ws.chain_key;
        fa.key_dst = final_key;
    }
    hipLaunchKernelGGL(k_gmm_flush, dim3(cdiv(P, 256)), dim3(256), 0, s, fa);
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
    return gmm_enqueue_steps((hipStream_t)stream, model, hyper, state, X_dev, mask_dev, B, 1, loss_dev, grad_out_dev, ws);
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
    return gmm_enqueue_steps((hipStream_t)stream, model, hyper, state, X_dev, mask_dev, B_local, 1, nullptr, nullptr, ws, nullptr, 0, 0,
                             1, B_total, pos0, sums_dev);
}

int d3p_dpvi_gmm_apply(void* stream, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                       float* sums_dev, uint32_t B_total, uint32_t B_local, float* loss_dev, float* grad_out_dev,
                       void* workspace_dev, size_t workspace_bytes)
{
    if (int rc = gmm_check_common(model, hyper, state, B_local, workspace_dev, workspace_bytes, "d3p_dpvi_gmm_apply")) return rc;
    D3P_REQUIRE(sums_dev && B_total >= 1, "d3p_dpvi_gmm_apply: null pointer or empty batch");
    GmmWorkspace ws;
    gmm_carve(model, B_local, (char*)workspace_dev, &ws);
    return gmm_enqueue_steps((hipStream_t)stream, model, hyper, state, nullptr, nullptr, B_local, 1, loss_dev, grad_out_dev, ws, nullptr, 0,
                             0, 2, B_total, 0, sums_dev);
}

int d3p_dpvi_gmm_run(void* stream, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                     const uint32_t* batch_key_dev, uint32_t first_batch, const float* X_dev, uint32_t n_rows, uint32_t B,
                     uint32_t num_steps, float* losses_dev, void* workspace_dev, size_t workspace_bytes)
{
    if (int rc = gmm_check_common(model, hyper, state, B, workspace_dev, workspace_bytes, "d3p_dpvi_gmm_run")) return rc;
    D3P_REQUIRE(X_dev && batch_key_dev, "d3p_dpvi_gmm_run: null pointer");
    D3P_REQUIRE(B <= n_rows, "d3p_dpvi_gmm_run: batch larger than the table");
    if (num_steps == 0) return D3P_OK;
    GmmWorkspace ws;
    gmm_carve(model, B, (char*)workspace_dev, &ws);
    return gmm_enqueue_steps((hipStream_t)stream, model, hyper, state, X_dev, nullptr, B, num_steps, losses_dev, nullptr, ws, batch_key_dev,
                             first_batch, n_rows);
}

}  // extern "C"
