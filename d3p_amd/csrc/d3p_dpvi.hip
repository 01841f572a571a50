// DP-VI update step for Bayesian logistic regression + AutoDiagonalNormal on gfx950:
//   k_step_keys   : ChaCha key schedule of DPSVI.update (svi.py:413-414, :259, :491), batchifier
//                   fold_in + Feistel indices (minibatch.py:230-231), per-example threefry sample
//                   keys (svi.py:289-290)
//   k_logreg_main : fused gather -> per-example ELBO gradient -> joint L2 norm -> clip -> sum
//                   (svi.py:238-348) -- one wavefront per example row, X read once, B x P never
//                   materialised; MODE 1 materialises px_grads for the stage-level API
//   k_finalize    : deterministic partial reduction, mean, Gaussian mechanism with ChaCha20 noise,
//                   rescale (svi.py:350-377), Adam (svi.py:379-393), derived-parameter pack
#include "d3p_device.h"
#include "d3p_host.h"

namespace d3p {

// Per-step slot in the workspace (all device memory, 4-byte words).
struct StepSlot {
    uint32_t site_keys[2][16];  // split(perturbation_key, 2)          (svi.py:491)
    uint32_t batch_key[16];     // fold_in(batchifier_state, i)        (minibatch.py:115, :230)
    uint32_t jax_key[2];        // random_bits(gradient_key, 32, (2,)) (random/__init__.py:155)
    uint32_t counts[2];         // [0] raw selected, [1] valid examples of the padded batch
    int32_t adam_i;             // optimiser step index used by this step
    uint32_t batch_i;           // batch index used by this step
    uint32_t pad[2];
};

struct Workspace {
    StepSlot* slot;
    float* pack;          // [loc | s | sg | q | lc] x D
    uint32_t* idx;        // B
    uint32_t* skeys;      // B x 2
    float* partials;      // max_blocks x (P + 2)
    float* sums;          // P + 2 (multi-GPU path)
    void* poisson_ws;
    size_t poisson_bytes;
    uint32_t max_blocks;
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define D3P_MAIN_MAX_BLOCKS 2048u

static size_t carve(const d3p_logreg_model* m, const d3p_batch_source* src, char* base, Workspace* ws)
{
    const size_t D = (size_t)m->d + (m->intercept ? 1 : 0), P = 2 * D, B = src->B;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return base ? base + o : nullptr; };
    char* p;
    p = take(sizeof(StepSlot)); if (ws) ws->slot = (StepSlot*)p;
    p = take(5 * D * sizeof(float)); if (ws) ws->pack = (float*)p;
    p = take(B * sizeof(uint32_t)); if (ws) ws->idx = (uint32_t*)p;
    p = take(2 * B * sizeof(uint32_t)); if (ws) ws->skeys = (uint32_t*)p;
    p = take((size_t)D3P_MAIN_MAX_BLOCKS * (P + 2) * sizeof(float)); if (ws) ws->partials = (float*)p;
    p = take((P + 2) * sizeof(float)); if (ws) ws->sums = (float*)p;
    size_t pb = 0;
    if (src->kind == D3P_BATCH_POISSON) pb = d3p_poisson_select_workspace((uint32_t)src->n_rows);
    p = take(pb); if (ws) { ws->poisson_ws = p; ws->poisson_bytes = pb; ws->max_blocks = D3P_MAIN_MAX_BLOCKS; }
    return off;
}

// ------------------------------------------------------------------------------------------
// derived per-column parameters ("pack"): loc, s = softplus(u), sg = sigmoid(u),
// q = inv_obs * sg / s, lc = log(prior_std) - log(s)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_column(const d3p_logreg_model& m, int D, int e, float loc, float u,
                                            float* __restrict__ pack)
{
    const float s = softplus_f(u), sg = sigmoid_f(u);
    const float ps = (e < m.d) ? m.prior_w : m.prior_b;
    pack[e] = loc;
    pack[D + e] = s;
    pack[2 * D + e] = sg;
    pack[3 * D + e] = m.inv_obs * sg / s;
    pack[4 * D + e] = logf(ps) - logf(s);
}

__global__ void k_pack(d3p_logreg_model m, const float* __restrict__ params, float* __restrict__ pack)
{
    const int D = m.d + (m.intercept ? 1 : 0);
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < D) pack_column(m, D, e, params[e], params[D + e], pack);
}

// ------------------------------------------------------------------------------------------
// key schedule + sampler
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t feistel_permute_dev(const uint32_t* rc, uint32_t capacity, int bits_lower,
                                                        int bits_upper, uint32_t position)
{
    const uint32_t mask_lower = (1u << bits_lower) - 1u, mask_upper = (1u << bits_upper) - 1u;
    uint32_t x = position;
    do {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const uint32_t k0 = rc[3 * j], k1 = rc[3 * j + 1], k2 = rc[3 * j + 2];
            const uint32_t xu = x >> bits_lower, xl = x & mask_lower;
            const uint32_t yu = xl ^ ((((xu * k1) >> bits_upper) ^ k2) & mask_lower);
            const uint32_t yl = (xu * k0) & mask_upper;
            x = (yu << bits_upper) | yl;
        }
    } while (x >= capacity);
    return x;
}

// Sample key of the guide's latent draw for batch position p (see oracle d3po_px_sample_key):
// px_key = split(jax_key, B)[p]; guide_seed = split(px_key)[1]; sample_key = split(guide_seed)[1].
__device__ __forceinline__ void px_sample_key(uint32_t j0, uint32_t j1, uint32_t B, uint32_t p, uint32_t& o0,
                                              uint32_t& o1)
{
    const uint32_t px0 = tf_iota_word(j0, j1, 2ull * B, 2ull * p);
    const uint32_t px1 = tf_iota_word(j0, j1, 2ull * B, 2ull * p + 1);
    uint32_t a, b, g0, g1;
    threefry2x32(px0, px1, 0u, 2u, a, g0);  // split(key,2): counts [0,1 | 2,3]; key1 = (y1(0,2), y1(1,3))
    threefry2x32(px0, px1, 1u, 3u, a, g1);
    threefry2x32(g0, g1, 0u, 2u, a, o0);
    threefry2x32(g0, g1, 1u, 3u, b, o1);
}

struct KeysArgs {
    const uint32_t* state_keys;  // current state key (16 words)
    uint32_t* state_keys_out;    // next state key (the other ping-pong slot)
    const int32_t* adam_step;
    const uint32_t* batch_key;   // nullable
    const uint32_t* batch_index; // nullable
    StepSlot* slot;
    uint32_t* idx;               // B (FEISTEL)
    uint32_t* skeys;             // B x 2
    uint32_t B;
    uint32_t capacity;
    int bits_lower, bits_upper;
    int kind;
};

__global__ void __launch_bounds__(256) k_step_keys(KeysArgs a)
{
    __shared__ uint32_t sh_next[16], sh_grad[16], sh_pert[16], sh_site[2][16], sh_bkey[16], sh_jax[2], sh_rc[32];
    const int tid = threadIdx.x;
    const int32_t adam_i = *a.adam_step;
    const uint32_t* cur = a.state_keys;
    const uint32_t bi = a.batch_index ? *a.batch_index : 0u;
    if (tid < 3) {  // split(state_key, 3): next, gradient, perturbation (svi.py:208-211, :413-414)
        uint32_t k[16], c[16];
        load_key(cur, k);
        derive_child(k, (uint32_t)tid, 0u, D3P_TAG_SPLIT, c);
        uint32_t* dst = tid == 0 ? sh_next : (tid == 1 ? sh_grad : sh_pert);
#pragma unroll
        for (int w = 0; w < 16; ++w) dst[w] = c[w];
    } else if (tid == 3 && a.batch_key) {  // fold_in(batchifier_state, i) (minibatch.py:115, :230)
        uint32_t k[16], c[16];
        load_key(a.batch_key, k);
        derive_child(k, 0u, bi, D3P_TAG_FOLD, c);
#pragma unroll
        for (int w = 0; w < 16; ++w) sh_bkey[w] = c[w];
    }
    __syncthreads();
    if (tid == 0) {  // convert_to_jax_rng_key(gradient_key) (svi.py:259; random/__init__.py:155)
        uint32_t k[16], o[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) k[w] = sh_grad[w];
        keystream_block(k, 0u, o);
        sh_jax[0] = o[0];
        sh_jax[1] = o[1];
    } else if (tid == 1 || tid == 2) {  // per-site keys split(perturbation_key, 2) (svi.py:491)
        uint32_t k[16], c[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) k[w] = sh_pert[w];
        derive_child(k, (uint32_t)(tid - 1), 0u, D3P_TAG_SPLIT, c);
#pragma unroll
        for (int w = 0; w < 16; ++w) sh_site[tid - 1][w] = c[w];
    } else if ((tid == 3 || tid == 4) && a.kind == D3P_BATCH_FEISTEL) {  // round constants (util.py:240-246)
        uint32_t k[16], o[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) k[w] = sh_bkey[w];
        keystream_block(k, (uint32_t)(tid - 3), o);
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int g = 16 * (tid - 3) + w;
            if (g < 30) sh_rc[g] = (g % 3 == 0) ? (o[w] | 1u) : o[w];
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < 16) {
            a.state_keys_out[tid] = sh_next[tid];
            a.slot->site_keys[0][tid] = sh_site[0][tid];
            a.slot->site_keys[1][tid] = sh_site[1][tid];
            a.slot->batch_key[tid] = a.batch_key ? sh_bkey[tid] : 0u;
        }
        if (tid < 2) a.slot->jax_key[tid] = sh_jax[tid];
        if (tid == 0) {
            a.slot->adam_i = adam_i;
            a.slot->batch_i = bi;
            if (a.kind != D3P_BATCH_POISSON) {  // POISSON: written by the select kernels
                a.slot->counts[0] = a.B;
                a.slot->counts[1] = a.B;
            }
        }
    }
    const uint32_t p = blockIdx.x * blockDim.x + tid;
    if (p < a.B) {
        if (a.kind == D3P_BATCH_FEISTEL)
            a.idx[p] = feistel_permute_dev(sh_rc, a.capacity, a.bits_lower, a.bits_upper, p);
        uint32_t s0, s1;
        px_sample_key(sh_jax[0], sh_jax[1], a.B, p, s0, s1);
        a.skeys[2 * p] = s0;
        a.skeys[2 * p + 1] = s1;
    }
}

// ------------------------------------------------------------------------------------------
// fused main kernel
// ------------------------------------------------------------------------------------------
struct MainArgs {
    const float* X;
    const float* y;
    const uint32_t* idx;     // nullable: row = p
    const uint8_t* mask;     // nullable
    const uint32_t* counts;  // nullable: valid iff p < counts[1]
    const uint32_t* skeys;   // B x 2 threefry sample keys (unused with eps_ext)
    const float* eps_ext;    // nullable: B x D
    const float* pack;       // 5 x D
    float* partials;         // gridDim.x x (P + 2)            (MODE 0)
    float* px_grads;         // B x P                          (MODE 1)
    float* px_loss;          // B                              (MODE 1)
    const float* meta;       // {n, factor}                    (MODE 1)
    uint32_t B;
    int d, D, half, icpt;
    uint64_t row_lo, row_hi;
    float A_scale;   // inv_obs * lik_scale
    float c1_w, c1_b;  // inv_obs / prior^2
    float hz_w, hz_b;  // 0.5 / prior^2
    float inv_obs, lik_scale, obs_scale, clip;
};

// Lane l of the wave that owns an example holds, for k < NK and i < V, the column pair
//   c0 = 64*V*k + V*l + i   (< half)      and      c1 = c0 + half   (< D)
// which is exactly one threefry2x32 call of jax's iota layout (words c0 and c0+half of the D-word
// stream), so on-chip eps generation wastes no words; V = 4 makes both X loads 16-byte wide.
template <int V, int NK, int MODE>
__global__ void __launch_bounds__(512) k_logreg_main(MainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NC = V * NK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int D = a.D, half = a.half, P = 2 * D;
    const uint32_t total_waves = gridDim.x * W;
    const uint32_t gw = blockIdx.x * W + wave;

    int c0[NC], c1[NC];
    bool ok0[NC], ok1[NC];
    float loc0[NC], s0[NC], sg0[NC], q0[NC], lc0[NC], loc1[NC], s1[NC], sg1[NC], q1[NC], lc1[NC];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int n = k * V + i;
            c0[n] = 64 * V * k + V * lane + i;
            c1[n] = c0[n] + half;
            ok0[n] = c0[n] < half;
            ok1[n] = ok0[n] && (c1[n] < D);
            loc0[n] = ok0[n] ? a.pack[c0[n]] : 0.f;
            s0[n] = ok0[n] ? a.pack[D + c0[n]] : 0.f;
            sg0[n] = ok0[n] ? a.pack[2 * D + c0[n]] : 0.f;
            q0[n] = ok0[n] ? a.pack[3 * D + c0[n]] : 0.f;
            lc0[n] = ok0[n] ? a.pack[4 * D + c0[n]] : 0.f;
            loc1[n] = ok1[n] ? a.pack[c1[n]] : 0.f;
            s1[n] = ok1[n] ? a.pack[D + c1[n]] : 0.f;
            sg1[n] = ok1[n] ? a.pack[2 * D + c1[n]] : 0.f;
            q1[n] = ok1[n] ? a.pack[3 * D + c1[n]] : 0.f;
            lc1[n] = ok1[n] ? a.pack[4 * D + c1[n]] : 0.f;
        }

    float accg0[NC], acch0[NC], accg1[NC], acch1[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) accg0[n] = acch0[n] = accg1[n] = acch1[n] = 0.f;
    float loss_acc = 0.f, n_acc = 0.f;
    const uint32_t n_valid = a.counts ? a.counts[1] : a.B;

    for (uint32_t p = gw; p < a.B; p += total_waves) {
        const uint32_t row_g = a.idx ? a.idx[p] : p;
        bool valid = (p < n_valid) && (a.mask ? a.mask[p] != 0 : true);
        const bool mine = (uint64_t)row_g >= a.row_lo && (uint64_t)row_g < a.row_hi;
        if (MODE == 0) {
            if (!(valid && mine)) continue;
        }
        const size_t row = (size_t)((uint64_t)row_g - a.row_lo);
        const float* xrow = a.X + row * (size_t)a.d;

        // ---- gather the example's feature row (coalesced; 16 B per lane when V == 4)
        float x0[NC], x1[NC];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (V == 4) {
                const int n = k * 4;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (ok0[n]) v0 = *reinterpret_cast<const float4*>(xrow + c0[n]);
                if (ok1[n]) v1 = *reinterpret_cast<const float4*>(xrow + c1[n]);
                x0[n] = v0.x; x0[n + 1] = v0.y; x0[n + 2] = v0.z; x0[n + 3] = v0.w;
                x1[n] = v1.x; x1[n + 1] = v1.y; x1[n + 2] = v1.z; x1[n + 3] = v1.w;
            } else {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const int n = k * V + i;
                    x0[n] = ok0[n] ? (c0[n] < a.d ? xrow[c0[n]] : 1.0f) : 0.f;  // column d = intercept
                    x1[n] = ok1[n] ? (c1[n] < a.d ? xrow[c1[n]] : 1.0f) : 0.f;
                }
            }
        }
        const float yv = a.y[row];

        // ---- guide noise eps_i (svi.py:289-290): parity mode reads it, otherwise threefry on chip
        float e0[NC], e1[NC];
        if (a.eps_ext) {
            const float* er = a.eps_ext + (size_t)p * D;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if (V == 4) {
                    const int n = k * 4;
                    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                    if (ok0[n]) v0 = *reinterpret_cast<const float4*>(er + c0[n]);
                    if (ok1[n]) v1 = *reinterpret_cast<const float4*>(er + c1[n]);
                    e0[n] = v0.x; e0[n + 1] = v0.y; e0[n + 2] = v0.z; e0[n + 3] = v0.w;
                    e1[n] = v1.x; e1[n + 1] = v1.y; e1[n + 2] = v1.z; e1[n + 3] = v1.w;
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const int n = k * V + i;
                        e0[n] = ok0[n] ? er[c0[n]] : 0.f;
                        e1[n] = ok1[n] ? er[c1[n]] : 0.f;
                    }
                }
            }
        } else {
            const uint32_t k0 = a.skeys[2 * p], k1 = a.skeys[2 * p + 1];
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                uint32_t b0, b1;
                threefry2x32(k0, k1, (uint32_t)c0[n], ok1[n] ? (uint32_t)c1[n] : 0u, b0, b1);
                e0[n] = ok0[n] ? bits_to_normal(b0) : 0.f;
                e1[n] = ok1[n] ? bits_to_normal(b1) : 0.f;
            }
        }

        // ---- z = loc + s * eps, logit t = x . z
        float z0[NC], z1[NC];
        float tp = 0.f;
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            z0[n] = __fmaf_rn(s0[n], e0[n], loc0[n]);
            z1[n] = __fmaf_rn(s1[n], e1[n], loc1[n]);
            tp = __fmaf_rn(x0[n], z0[n], tp);
            tp = __fmaf_rn(x1[n], z1[n], tp);
        }
        const float t = wave_sum(tp);
        const float sp = softplus_f(t);
        const float A = a.A_scale * (sigmoid_f(t) - yv);

        // ---- per-example gradient, its squared norm and the latent part of the loss
        float g0[NC], h0[NC], g1[NC], h1[NC];
        float n2 = 0.f, lp = 0.f;
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const bool ic0 = a.icpt && (c0[n] == a.d), ic1 = a.icpt && (c1[n] == a.d);
            g0[n] = __fmaf_rn(ic0 ? a.c1_b : a.c1_w, z0[n], A * x0[n]);
            g1[n] = __fmaf_rn(ic1 ? a.c1_b : a.c1_w, z1[n], A * x1[n]);
            h0[n] = __fmaf_rn(g0[n] * e0[n], sg0[n], -q0[n]);
            h1[n] = __fmaf_rn(g1[n] * e1[n], sg1[n], -q1[n]);
            n2 = __fmaf_rn(g0[n], g0[n], n2);
            n2 = __fmaf_rn(h0[n], h0[n], n2);
            n2 = __fmaf_rn(g1[n], g1[n], n2);
            n2 = __fmaf_rn(h1[n], h1[n], n2);
            lp += __fmaf_rn((ic0 ? a.hz_b : a.hz_w) * z0[n], z0[n], __fmaf_rn(-0.5f * e0[n], e0[n], lc0[n]));
            lp += __fmaf_rn((ic1 ? a.hz_b : a.hz_w) * z1[n], z1[n], __fmaf_rn(-0.5f * e1[n], e1[n], lc1[n]));
        }
        n2 = wave_sum(n2);
        lp = wave_sum(lp);
        // L_i = inv_obs * ((logq - logp) - lik_scale * loglik)   (svi.py:278-281)
        const float L = a.inv_obs * (lp - a.lik_scale * (yv * t - sp));

        if (MODE == 0) {
            // clip factor 1/max(1, ||g||/C) (svi.py:121-122) folded into the running sum (svi.py:343-346)
            const float cf = 1.0f / fmaxf(1.0f, __fsqrt_rn(n2) / a.clip);
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                accg0[n] = __fmaf_rn(cf, g0[n], accg0[n]);
                acch0[n] = __fmaf_rn(cf, h0[n], acch0[n]);
                accg1[n] = __fmaf_rn(cf, g1[n], accg1[n]);
                acch1[n] = __fmaf_rn(cf, h1[n], acch1[n]);
            }
            loss_acc += L;
            n_acc += 1.0f;
        } else {
            const float m = (valid && mine) ? 1.0f : 0.0f;  // loss * mask => zero loss and gradient (svi.py:281)
            float* gr = a.px_grads + (size_t)p * P;
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                if (ok0[n]) { gr[c0[n]] = g0[n] * m; gr[D + c0[n]] = h0[n] * m; }
                if (ok1[n]) { gr[c1[n]] = g1[n] * m; gr[D + c1[n]] = h1[n] * m; }
            }
            if (lane == 0) a.px_loss[p] = L * m * a.obs_scale * a.meta[1];  // svi.py:306
        }
    }

    if (MODE == 0) {
        // ---- workgroup reduction through LDS, one partial row per workgroup (fixed order)
        float* mine = lds + (size_t)wave * P;
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            if (ok0[n]) { mine[c0[n]] = accg0[n]; mine[D + c0[n]] = acch0[n]; }
            if (ok1[n]) { mine[c1[n]] = accg1[n]; mine[D + c1[n]] = acch1[n]; }
        }
        float* tail = lds + (size_t)W * P;
        if (lane == 0) { tail[2 * wave] = loss_acc; tail[2 * wave + 1] = n_acc; }
        __syncthreads();
        float* out = a.partials + (size_t)blockIdx.x * (P + 2);
        for (int c = threadIdx.x; c < P; c += blockDim.x) {
            float s = 0.f;
            for (int w = 0; w < W; ++w) s += lds[(size_t)w * P + c];
            out[c] = s;
        }
        if (threadIdx.x < 2) {
            float s = 0.f;
            for (int w = 0; w < W; ++w) s += tail[2 * w + threadIdx.x];
            out[P + threadIdx.x] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// partial reduction / finalize
// ------------------------------------------------------------------------------------------
#define D3P_FIN_COLS 32
#define D3P_FIN_ROWG 8

// column sum over `nparts` rows for the 32 columns of this workgroup; result valid for rg == 0.
__device__ __forceinline__ float column_sum(const float* __restrict__ parts, uint32_t nparts, uint32_t stride,
                                            uint32_t col, bool col_ok, int c, int rg, float* lds)
{
    float s = 0.f;
    if (col_ok)
        for (uint32_t r = rg; r < nparts; r += D3P_FIN_ROWG) s += parts[(size_t)r * stride + col];
    lds[rg * D3P_FIN_COLS + c] = s;
    __syncthreads();
    float tot = 0.f;
    if (rg == 0)
        for (int g = 0; g < D3P_FIN_ROWG; ++g) tot += lds[g * D3P_FIN_COLS + c];
    __syncthreads();
    return tot;
}

__device__ __forceinline__ float block_sum_column(const float* __restrict__ parts, uint32_t nparts, uint32_t stride,
                                                  uint32_t col, float* lds)
{
    // all 256 threads cooperate; every thread returns the total (fixed order)
    float s = 0.f;
    for (uint32_t r = threadIdx.x; r < nparts; r += blockDim.x) s += parts[(size_t)r * stride + col];
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

__global__ void __launch_bounds__(256)
k_reduce_partials(const float* __restrict__ parts, uint32_t nparts, uint32_t ncols, float* __restrict__ sums)
{
    __shared__ float lds[256];
    const int c = threadIdx.x % D3P_FIN_COLS, rg = threadIdx.x / D3P_FIN_COLS;
    const uint32_t col = blockIdx.x * D3P_FIN_COLS + c;
    const float tot = column_sum(parts, nparts, ncols, col, col < ncols, c, rg, lds);
    if (rg == 0 && col < ncols) sums[col] = tot;
}

struct FinalArgs {
    const float* parts;  // nparts x (P + 2)
    uint32_t nparts;
    StepSlot* slot;
    float* params;
    float* adam_m;
    float* adam_v;
    int32_t* adam_step;
    uint32_t* batch_index;  // nullable
    float* pack;
    float* loss_out;  // nullable
    float* grad_out;  // nullable
    uint32_t B;
    d3p_logreg_model m;
    d3p_dpsvi_hyper h;
};

__device__ __forceinline__ float chacha_normal_at(const uint32_t* __restrict__ key, uint32_t e)
{
    uint32_t k[16], o[16];
    load_key(key, k);
    keystream_block(k, e >> 4, o);
    uint32_t w = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) w = ((e & 15u) == (uint32_t)t) ? o[t] : w;
    return bits_to_normal(w);
}

__device__ __forceinline__ void adam_update(float& x, float& m, float& v, float g, int i, const d3p_dpsvi_hyper& h)
{
    // jax.example_libraries.optimizers.adam as wrapped by numpyro.optim.Adam
    m = (1.0f - h.b1) * g + h.b1 * m;
    v = (1.0f - h.b2) * g * g + h.b2 * v;
    const float mhat = m / (1.0f - powf(h.b1, (float)(i + 1)));
    const float vhat = v / (1.0f - powf(h.b2, (float)(i + 1)));
    x = x - h.lr * mhat / (sqrtf(vhat) + h.adam_eps);
}

__global__ void __launch_bounds__(256) k_finalize(FinalArgs a)
{
    __shared__ float lds[256];
    const int D = a.m.d + (a.m.intercept ? 1 : 0), P = 2 * D;
    const uint32_t stride = P + 2;
    const int c = threadIdx.x % D3P_FIN_COLS, rg = threadIdx.x / D3P_FIN_COLS;
    const uint32_t col = blockIdx.x * D3P_FIN_COLS + c;
    const float n = block_sum_column(a.parts, a.nparts, stride, P + 1, lds);
    const float tot = column_sum(a.parts, a.nparts, stride, col, col < (uint32_t)P, c, rg, lds);
    const float Bf = (float)a.B;
    const float factor = (n == 0.0f) ? 0.0f : Bf / n;              // svi.py:305
    const float obs_scale = 1.0f / a.m.inv_obs;
    const int adam_i = a.slot->adam_i;
    if (rg == 0 && col < (uint32_t)P) {
        const int site = col >= (uint32_t)D, e = col - site * D;
        const float avg = tot / Bf;                                 // svi.py:343-346
        const float scale = a.h.dp_scale * (a.h.clip / n);          // svi.py:365-366 (n == 0 -> inf, as the reference)
        const float z = chacha_normal_at(a.slot->site_keys[site], (uint32_t)e);  // svi.py:485-488
        const float g = (avg + z * scale) * obs_scale * factor;     // svi.py:375
        if (a.grad_out) a.grad_out[col] = g;
        float x = a.params[col], m = a.adam_m[col], v = a.adam_v[col];
        adam_update(x, m, v, g, adam_i, a.h);
        a.params[col] = x;
        a.adam_m[col] = m;
        a.adam_v[col] = v;
        // refresh the derived columns for the next step
        if (site == 0) {
            a.pack[e] = x;
        } else {
            const float s = softplus_f(x), sg = sigmoid_f(x);
            const float ps = (e < a.m.d) ? a.m.prior_w : a.m.prior_b;
            a.pack[D + e] = s;
            a.pack[2 * D + e] = sg;
            a.pack[3 * D + e] = a.m.inv_obs * sg / s;
            a.pack[4 * D + e] = logf(ps) - logf(s);
        }
    }
    if (blockIdx.x == 0) {
        const float loss_sum = block_sum_column(a.parts, a.nparts, stride, P, lds);
        if (threadIdx.x == 0) {
            if (a.loss_out) *a.loss_out = (loss_sum / Bf) * obs_scale * factor;  // svi.py:342, :306
            *a.adam_step = adam_i + 1;
            if (a.batch_index) *a.batch_index = a.slot->batch_i + 1u;
        }
    }
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
    const float scale = 1.0f / fmaxf(1.0f, __fsqrt_rn(ss) / c);  // svi.py:121-122
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

// ------------------------------------------------------------------------------------------
// host-side launch logic
// ------------------------------------------------------------------------------------------
struct MainGeom {
    int V, NK, W;
    uint32_t blocks;
    size_t lds;
};

static int main_geometry(const d3p_logreg_model* m, uint32_t B, MainGeom* g)
{
    const int D = m->d + (m->intercept ? 1 : 0), half = (D + 1) / 2, P = 2 * D;
    const bool vec = !m->intercept && (m->d % 8 == 0);
    g->V = vec ? 4 : 1;
    const int need = (half + 64 * g->V - 1) / (64 * g->V);
    g->NK = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 0;
    if (g->NK == 0)
        return fail(D3P_E_UNSUPPORTED, "logreg kernel: latent dimension %d exceeds the supported maximum (%d)", D,
                    2 * 64 * g->V * 8);
    // waves per workgroup: 8, reduced until the LDS reduction buffer fits 64 KiB
    int W = 8;
    while (W > 1 && (size_t)W * P * sizeof(float) + 2 * W * sizeof(float) > 64 * 1024) W >>= 1;
    if ((size_t)W * P * sizeof(float) + 2 * W * sizeof(float) > 160 * 1024)
        return fail(D3P_E_UNSUPPORTED, "logreg kernel: P = %d does not fit the LDS reduction buffer", P);
    g->W = W;
    // two examples per wave, capped so the partial slab stays small
    uint64_t waves = ((uint64_t)B + 1) / 2;
    uint64_t blocks = (waves + W - 1) / W;
    if (blocks < 1) blocks = 1;
    if (blocks > D3P_MAIN_MAX_BLOCKS) blocks = D3P_MAIN_MAX_BLOCKS;
    g->blocks = (uint32_t)blocks;
    g->lds = (size_t)W * P * sizeof(float) + 2 * W * sizeof(float);
    return D3P_OK;
}

template <int MODE>
static int launch_main(hipStream_t s, const MainGeom& g, const MainArgs& a)
{
#define D3P_LAUNCH(V_, NK_)                                                                                         \
    hipLaunchKernelGGL((k_logreg_main<V_, NK_, MODE>), dim3(g.blocks), dim3(64 * g.W), MODE == 0 ? g.lds : 0, s, a); \
    return check_launch("k_logreg_main")
    if (g.V == 4) {
        switch (g.NK) {
        case 1: D3P_LAUNCH(4, 1);
        case 2: D3P_LAUNCH(4, 2);
        case 4: D3P_LAUNCH(4, 4);
        default: D3P_LAUNCH(4, 8);
        }
    } else {
        switch (g.NK) {
        case 1: D3P_LAUNCH(1, 1);
        case 2: D3P_LAUNCH(1, 2);
        case 4: D3P_LAUNCH(1, 4);
        default: D3P_LAUNCH(1, 8);
        }
    }
#undef D3P_LAUNCH
}

static void fill_model_scalars(const d3p_logreg_model* m, MainArgs* a)
{
    const int D = m->d + (m->intercept ? 1 : 0);
    a->d = m->d;
    a->D = D;
    a->half = (D + 1) / 2;
    a->icpt = m->intercept ? 1 : 0;
    a->A_scale = m->inv_obs * m->lik_scale;
    a->c1_w = m->inv_obs / (m->prior_w * m->prior_w);
    a->c1_b = m->inv_obs / (m->prior_b * m->prior_b);
    a->hz_w = 0.5f / (m->prior_w * m->prior_w);
    a->hz_b = 0.5f / (m->prior_b * m->prior_b);
    a->inv_obs = m->inv_obs;
    a->lik_scale = m->lik_scale;
    a->obs_scale = 1.0f / m->inv_obs;
}

static int validate(const d3p_logreg_model* m, const d3p_dpsvi_hyper* h, const d3p_dpsvi_state* st,
                    const d3p_batch_source* src)
{
    D3P_REQUIRE(m && h && st && src, "null argument struct");
    D3P_REQUIRE(m->d >= 1, "model.d must be >= 1");
    D3P_REQUIRE(m->prior_w > 0.f && m->prior_b > 0.f, "prior scales must be positive");
    D3P_REQUIRE(m->inv_obs > 0.f, "inv_obs must be positive");
    D3P_REQUIRE(h->clip != 0.f, "The clipping threshold must be greater than 0.");  // svi.py:119-120
    D3P_REQUIRE(std::isfinite(h->clip), "clipping_threshold must be finite!");             // svi.py:187-188
    D3P_REQUIRE(st->rng_key && st->params && st->adam_m && st->adam_v && st->step, "null state pointer");
    D3P_REQUIRE(src->B >= 1, "batch size must be >= 1");
    D3P_REQUIRE(src->kind == D3P_BATCH_EXPLICIT || src->kind == D3P_BATCH_FEISTEL || src->kind == D3P_BATCH_POISSON,
                "unknown batch source kind");
    D3P_REQUIRE(src->row_lo <= src->row_hi && src->row_hi <= src->n_rows, "bad row range");
    if (src->kind != D3P_BATCH_EXPLICIT) {
        D3P_REQUIRE(src->batch_key && src->batch_index, "sampled batch sources need batch_key and batch_index");
        D3P_REQUIRE(src->n_rows >= 1 && src->n_rows <= 0xFFFFFFFFull, "n_rows must be in [1, 2^32)");
        D3P_REQUIRE(src->B <= src->n_rows, "batch size exceeds the number of rows");
    } else {
        D3P_REQUIRE(src->n_rows == src->B, "explicit batches: n_rows must equal B");
    }
    return D3P_OK;
}

static inline int bit_length_u32(uint32_t v)
{
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

// step_keys (+ Poisson select) + fused main kernel; leaves per-workgroup partials in ws.partials
static int enqueue_local(hipStream_t s, const d3p_logreg_model* m, const d3p_dpsvi_hyper* h, const d3p_dpsvi_state* st,
                         const d3p_batch_source* src, const float* X, const float* y, const float* eps,
                         const Workspace& ws, const MainGeom& g, bool keys, bool main, int slot)
{
    if (keys) {
        KeysArgs ka;
        ka.state_keys = st->rng_key + 16 * (slot & 1);
        ka.state_keys_out = st->rng_key + 16 * ((slot & 1) ^ 1);
        ka.adam_step = st->step;
        ka.batch_key = src->kind == D3P_BATCH_EXPLICIT ? nullptr : src->batch_key;
        ka.batch_index = src->kind == D3P_BATCH_EXPLICIT ? nullptr : src->batch_index;
        ka.slot = ws.slot;
        ka.idx = ws.idx;
        ka.skeys = ws.skeys;
        ka.B = src->B;
        ka.capacity = (uint32_t)src->n_rows;
        const int bits = bit_length_u32(ka.capacity - 1);
        ka.bits_lower = bits >> 1;
        ka.bits_upper = bits - ka.bits_lower;
        ka.kind = src->kind;
        hipLaunchKernelGGL(k_step_keys, dim3(cdiv(src->B, 256)), dim3(256), 0, s, ka);
        int rc = check_launch("k_step_keys");
        if (rc) return rc;
        if (src->kind == D3P_BATCH_POISSON) {
            rc = d3p_poisson_select((void*)s, ws.slot->batch_key, src->q, (uint32_t)src->n_rows, src->B, src->suppress,
                                    ws.idx, ws.slot->counts, ws.poisson_ws, ws.poisson_bytes);
            if (rc) return rc;
        }
    }
    if (main) {
        MainArgs a;
        memset(&a, 0, sizeof(a));
        fill_model_scalars(m, &a);
        a.X = X;
        a.y = y;
        a.idx = src->kind == D3P_BATCH_EXPLICIT ? nullptr : ws.idx;
        a.mask = src->kind == D3P_BATCH_EXPLICIT ? src->mask : nullptr;
        a.counts = ws.slot->counts;
        a.skeys = ws.skeys;
        a.eps_ext = eps;
        a.pack = ws.pack;
        a.partials = ws.partials;
        a.B = src->B;
        a.row_lo = src->row_lo;
        a.row_hi = src->row_hi;
        a.clip = h->clip;
        return launch_main<0>(s, g, a);
    }
    return D3P_OK;
}

static int enqueue_finalize(hipStream_t s, const d3p_logreg_model* m, const d3p_dpsvi_hyper* h,
                            const d3p_dpsvi_state* st, const d3p_batch_source* src, const float* parts, uint32_t nparts,
                            const Workspace& ws, float* loss, float* grad_out)
{
    const int D = m->d + (m->intercept ? 1 : 0), P = 2 * D;
    FinalArgs fa;
    fa.parts = parts;
    fa.nparts = nparts;
    fa.slot = ws.slot;
    fa.params = st->params;
    fa.adam_m = st->adam_m;
    fa.adam_v = st->adam_v;
    fa.adam_step = st->step;
    fa.batch_index = src->kind == D3P_BATCH_EXPLICIT ? nullptr : src->batch_index;
    fa.pack = ws.pack;
    fa.loss_out = loss;
    fa.grad_out = grad_out;
    fa.B = src->B;
    fa.m = *m;
    fa.h = *h;
    hipLaunchKernelGGL(k_finalize, dim3(cdiv(P, D3P_FIN_COLS)), dim3(256), 0, s, fa);
    return check_launch("k_finalize");
}

static int enqueue_pack(hipStream_t s, const d3p_logreg_model* m, const d3p_dpsvi_state* st, const Workspace& ws)
{
    const int D = m->d + (m->intercept ? 1 : 0);
    hipLaunchKernelGGL(k_pack, dim3(cdiv(D, 256)), dim3(256), 0, s, *m, (const float*)st->params, ws.pack);
    return check_launch("k_pack");
}

}  // namespace d3p

using namespace d3p;

extern "C" {

size_t d3p_dpvi_logreg_workspace(const d3p_logreg_model* model, const d3p_batch_source* src)
{
    if (!model || !src) return 0;
    return carve(model, src, nullptr, nullptr);
}

#define D3P_PREP_WS()                                                                                       \
    int rc__ = validate(model, hyper, state, src);                                                          \
    if (rc__) return rc__;                                                                                  \
    D3P_REQUIRE(workspace_dev, "null workspace");                                                           \
    if (workspace_bytes < carve(model, src, nullptr, nullptr))                                              \
        return fail(D3P_E_WORKSPACE, "workspace too small (%zu < %zu)", workspace_bytes,                    \
                    carve(model, src, nullptr, nullptr));                                                   \
    Workspace ws;                                                                                           \
    carve(model, src, (char*)workspace_dev, &ws);                                                           \
    MainGeom geom;                                                                                          \
    rc__ = main_geometry(model, src->B, &geom);                                                             \
    if (rc__) return rc__;                                                                                  \
    hipStream_t s = (hipStream_t)stream

int d3p_dpvi_logreg_local_sums(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                               const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev,
                               const float* y_dev, const float* eps_dev, float* sums_dev, void* workspace_dev,
                               size_t workspace_bytes)
{
    D3P_PREP_WS();
    D3P_REQUIRE(X_dev && y_dev && sums_dev, "null data pointer");
    const int P = 2 * (model->d + (model->intercept ? 1 : 0));
    int rc = enqueue_pack(s, model, state, ws);
    if (rc) return rc;
    rc = enqueue_local(s, model, hyper, state, src, X_dev, y_dev, eps_dev, ws, geom, true, true, state->key_slot);
    if (rc) return rc;
    hipLaunchKernelGGL(k_reduce_partials, dim3(cdiv(P + 2, D3P_FIN_COLS)), dim3(256), 0, s, (const float*)ws.partials,
                       geom.blocks, (uint32_t)(P + 2), sums_dev);
    return check_launch("k_reduce_partials");
}

int d3p_dpvi_logreg_finalize(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                             const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* sums_dev,
                             float* loss_dev, float* grad_out_dev, void* workspace_dev, size_t workspace_bytes)
{
    D3P_PREP_WS();
    D3P_REQUIRE(sums_dev, "null sums pointer");
    return enqueue_finalize(s, model, hyper, state, src, sums_dev, 1u, ws, loss_dev, grad_out_dev);
}

int d3p_dpvi_logreg_run(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                        const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev,
                        const float* y_dev, uint32_t num_steps, float* losses_dev, void* workspace_dev,
                        size_t workspace_bytes)
{
    D3P_PREP_WS();
    D3P_REQUIRE(X_dev && y_dev, "null data pointer");
    D3P_REQUIRE(src->row_lo == 0 && src->row_hi == src->n_rows, "d3p_dpvi_logreg_run is the single-GPU path");
    int rc = enqueue_pack(s, model, state, ws);
    if (rc) return rc;
    for (uint32_t t = 0; t < num_steps; ++t) {
        rc = enqueue_local(s, model, hyper, state, src, X_dev, y_dev, nullptr, ws, geom, true, true,
                           state->key_slot + (int)t);
        if (rc) return rc;
        rc = enqueue_finalize(s, model, hyper, state, src, ws.partials, geom.blocks, ws, losses_dev ? losses_dev + t : nullptr,
                              nullptr);
        if (rc) return rc;
    }
    return D3P_OK;
}

int d3p_dpvi_logreg_time_main_kernel(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                                     const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev,
                                     const float* y_dev, void* workspace_dev, size_t workspace_bytes, int reps,
                                     float* avg_us)
{
    D3P_PREP_WS();
    D3P_REQUIRE(X_dev && y_dev && avg_us && reps >= 1, "bad arguments");
    int rc = enqueue_pack(s, model, state, ws);
    if (rc) return rc;
    rc = enqueue_local(s, model, hyper, state, src, X_dev, y_dev, nullptr, ws, geom, true, false, state->key_slot);
    if (rc) return rc;
    hipEvent_t e0, e1;
    D3P_HIP_TRY(hipEventCreate(&e0));
    D3P_HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) {  // warm-up
        rc = enqueue_local(s, model, hyper, state, src, X_dev, y_dev, nullptr, ws, geom, false, true, 0);
        if (rc) return rc;
    }
    // one event pair per launch: the average excludes the gaps between launches
    double total_ms = 0.0;
    for (int i = 0; i < reps; ++i) {
        D3P_HIP_TRY(hipEventRecord(e0, s));
        rc = enqueue_local(s, model, hyper, state, src, X_dev, y_dev, nullptr, ws, geom, false, true, 0);
        if (rc) return rc;
        D3P_HIP_TRY(hipEventRecord(e1, s));
        D3P_HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        D3P_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = (float)(total_ms * 1000.0 / reps);
    return D3P_OK;
}

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
    D3P_REQUIRE(model && params_dev && X_dev && y_dev && px_loss_dev && px_grads_dev && meta_dev && workspace_dev,
                "d3p_logreg_px_grads: null pointer");
    D3P_REQUIRE(eps_dev || jax_key_dev, "d3p_logreg_px_grads: either eps_dev or jax_key_dev must be given");
    D3P_REQUIRE(B >= 1, "d3p_logreg_px_grads: B must be >= 1");
    D3P_REQUIRE(model->d >= 1 && model->prior_w > 0.f && model->prior_b > 0.f && model->inv_obs > 0.f,
                "d3p_logreg_px_grads: bad model");
    if (workspace_bytes < px_ws_bytes(model, B))
        return fail(D3P_E_WORKSPACE, "d3p_logreg_px_grads: workspace too small (%zu < %zu)", workspace_bytes,
                    px_ws_bytes(model, B));
    hipStream_t s = (hipStream_t)stream;
    const int D = model->d + (model->intercept ? 1 : 0);
    float* pack = (float*)workspace_dev;
    uint32_t* skeys = (uint32_t*)((char*)workspace_dev + align_up(5 * (size_t)D * sizeof(float), 256));
    MainGeom g;
    int rc = main_geometry(model, B, &g);
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

}  // extern "C"
