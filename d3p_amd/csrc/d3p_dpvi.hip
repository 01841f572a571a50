// Fused DPSVI.update path for Bayesian logistic regression + AutoDiagonalNormal on gfx950.
//
// Per batch of K <= D3P_STEP_BATCH steps (everything here is independent of the parameters, so it
// is hoisted off the per-step critical path and amortised over K steps):
//   k_chain    : the serial ChaCha key chain of DPSVI.update: (next, gradient, perturbation) =
//                split(state_key, 3) for K consecutive steps (svi.py:208-211, :413-414)
//   k_sampler  : per step, in parallel: jax key (random/__init__.py:155), per-example threefry
//                sample keys (svi.py:289-290), batchifier fold_in + Feistel indices
//                (minibatch.py:230-231, util.py:240-301), the P Gaussian-mechanism normals
//                (svi.py:485-491) and the Adam bias corrections
// Per step (the critical path, two launches):
//   k_logreg_main : fused gather -> per-example ELBO gradient -> joint L2 norm -> clip -> sum
//                   (svi.py:238-348); d3p_logreg_kernel.h
//   k_finalize    : deterministic reduction of the per-workgroup partial rows, mean, Gaussian
//                   mechanism, rescale (svi.py:350-377), Adam (svi.py:379-393), derived columns
#include "d3p_logreg_kernel.h"

#define D3P_STEP_BATCH 32

namespace d3p {

// Per-step record produced by k_chain / k_sampler (device memory).
struct StepSlot {
    uint32_t grad_key[16];   // split(state_key, 3)[1]
    uint32_t pert_key[16];   // split(state_key, 3)[2]
    uint32_t batch_key[16];  // fold_in(batchifier_state, i)
    uint32_t jax_key[2];     // random_bits(gradient_key, 32, (2,))
    uint32_t counts[2];      // [0] raw selected, [1] valid examples of the padded batch
    int32_t adam_i;          // optimiser step index of this step
    uint32_t batch_i;        // batch index of this step
    float bc1, bc2;          // 1 - b1^(i+1), 1 - b2^(i+1)
};

// Running state of the key chain between batches.
struct Sched {
    uint32_t key[16];
    int32_t adam_i;
    uint32_t batch_i;
    uint32_t pad[2];
};

struct Workspace {
    Sched* sched;
    StepSlot* slots;  // D3P_STEP_BATCH
    float* pack;      // [loc | s | sg | q | lc] x D
    uint32_t* idx;    // D3P_STEP_BATCH x B
    uint32_t* skeys;  // D3P_STEP_BATCH x 2B
    float* noise;     // D3P_STEP_BATCH x P
    float* partials;  // max_blocks x (P + 2)
    unsigned long long* stamps;  // 2 x max_blocks
    void* poisson_ws;
    size_t poisson_bytes;
};

static size_t carve(const d3p_logreg_model* m, const d3p_batch_source* src, char* base, Workspace* ws,
                    Workspace* ws2 = nullptr)
{
    const size_t D = (size_t)m->d + (m->intercept ? 1 : 0), P = 2 * D, B = src->B, K = D3P_STEP_BATCH;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return base ? base + o : nullptr; };
    char* p;
    p = take(sizeof(Sched)); if (ws) ws->sched = (Sched*)p;
    p = take(K * sizeof(StepSlot)); if (ws) ws->slots = (StepSlot*)p;
    p = take(5 * D * sizeof(float)); if (ws) ws->pack = (float*)p;
    p = take(K * B * sizeof(uint32_t)); if (ws) ws->idx = (uint32_t*)p;
    p = take(K * 2 * B * sizeof(uint32_t)); if (ws) ws->skeys = (uint32_t*)p;
    p = take(K * P * sizeof(float)); if (ws) ws->noise = (float*)p;
    p = take((size_t)D3P_MAIN_MAX_BLOCKS * (P + 2) * sizeof(float)); if (ws) ws->partials = (float*)p;
    p = take((size_t)D3P_MAIN_MAX_BLOCKS * 2 * sizeof(unsigned long long)); if (ws) ws->stamps = (unsigned long long*)p;
    size_t pb = 0;
    if (src->kind == D3P_BATCH_POISSON) pb = d3p_poisson_select_workspace((uint32_t)src->n_rows);
    p = take(pb); if (ws) { ws->poisson_ws = p; ws->poisson_bytes = pb; }
    if (ws2) {
        *ws2 = ws ? *ws : Workspace();
    }
    p = take(K * sizeof(StepSlot)); if (ws2) ws2->slots = (StepSlot*)p;
    p = take(K * B * sizeof(uint32_t)); if (ws2) ws2->idx = (uint32_t*)p;
    p = take(K * 2 * B * sizeof(uint32_t)); if (ws2) ws2->skeys = (uint32_t*)p;
    p = take(K * P * sizeof(float)); if (ws2) ws2->noise = (float*)p;
    p = take(pb); if (ws2) ws2->poisson_ws = p;
    return off;
}

// ------------------------------------------------------------------------------------------
// key chain
// ------------------------------------------------------------------------------------------
__global__ void k_sched_init(const uint32_t* __restrict__ state_key, const int32_t* __restrict__ adam_step,
                             const uint32_t* __restrict__ batch_index, Sched* __restrict__ sched)
{
    if (threadIdx.x < 16) sched->key[threadIdx.x] = state_key[threadIdx.x];
    if (threadIdx.x == 0) {
        sched->adam_i = *adam_step;
        sched->batch_i = batch_index ? *batch_index : 0u;
    }
}

__global__ void k_sched_finish(const Sched* __restrict__ sched, uint32_t* __restrict__ state_key_out)
{
    if (threadIdx.x < 16) state_key_out[threadIdx.x] = sched->key[threadIdx.x];
}

// One wavefront; lanes 0..2 each derive one child of split(cur, 3), lane 0's child is the next key.
__global__ void __launch_bounds__(64) k_chain(Sched* __restrict__ sched, StepSlot* __restrict__ slots, int K)
{
    const int lane = threadIdx.x;
    uint32_t cur[16], child[16];
    load_key(sched->key, cur);
    const int32_t adam0 = sched->adam_i;
    const uint32_t batch0 = sched->batch_i;
    for (int t = 0; t < K; ++t) {
        derive_child(cur, (uint32_t)(lane < 3 ? lane : 0), 0u, D3P_TAG_SPLIT, child);
        if (lane == 1) {
#pragma unroll
            for (int w = 0; w < 16; ++w) slots[t].grad_key[w] = child[w];
        } else if (lane == 2) {
#pragma unroll
            for (int w = 0; w < 16; ++w) slots[t].pert_key[w] = child[w];
        } else if (lane == 3) {
            slots[t].adam_i = adam0 + t;
            slots[t].batch_i = batch0 + (uint32_t)t;
        }
#pragma unroll
        for (int w = 0; w < 16; ++w) cur[w] = __builtin_amdgcn_readfirstlane(child[w]);
    }
    if (lane < 16) {
        uint32_t v = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) v = (lane == w) ? cur[w] : v;
        sched->key[lane] = v;
    }
    if (lane == 0) {
        sched->adam_i = adam0 + K;
        sched->batch_i = batch0 + (uint32_t)K;
    }
}

// ------------------------------------------------------------------------------------------
// sampler: grid (ceil(B/256) + 1, K); blockIdx.y = step within the batch; the extra x-block of every
// step produces the Gaussian-mechanism normals and the slot scalars.
// ------------------------------------------------------------------------------------------
struct SamplerArgs {
    StepSlot* slots;
    const uint32_t* batch_key;  // batchifier state (nullable: EXPLICIT)
    uint32_t* idx;              // K x B
    uint32_t* skeys;            // K x 2B
    float* noise;               // K x P
    uint32_t B;
    uint32_t capacity;
    int bits_lower, bits_upper;
    int kind;
    int D;
    float b1, b2;
};

__global__ void __launch_bounds__(256) k_sampler(SamplerArgs a)
{
    __shared__ uint32_t sh_key[2][16], sh_jax[2], sh_rc[32];
    const int tid = threadIdx.x;
    const int t = blockIdx.y;
    StepSlot* slot = a.slots + t;
    const bool aux = blockIdx.x == gridDim.x - 1;
    if (!aux) {
        if (tid == 0) {  // convert_to_jax_rng_key(gradient_key) (svi.py:259; random/__init__.py:155)
            uint32_t k[16], o[16];
            load_key(slot->grad_key, k);
            keystream_block(k, 0u, o);
            sh_jax[0] = o[0];
            sh_jax[1] = o[1];
        } else if (tid == 64 && a.kind == D3P_BATCH_FEISTEL) {  // fold_in(batchifier_state, i) (minibatch.py:230)
            uint32_t k[16], c[16];
            load_key(a.batch_key, k);
            derive_child(k, 0u, slot->batch_i, D3P_TAG_FOLD, c);
#pragma unroll
            for (int w = 0; w < 16; ++w) sh_key[0][w] = c[w];
        }
        __syncthreads();
        if ((tid == 64 || tid == 128) && a.kind == D3P_BATCH_FEISTEL) {  // round constants (util.py:240-246)
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
        const uint32_t p = blockIdx.x * blockDim.x + tid;
        if (p < a.B) {
            if (a.kind == D3P_BATCH_FEISTEL)
                a.idx[(size_t)t * a.B + p] = feistel_permute_dev(sh_rc, a.capacity, a.bits_lower, a.bits_upper, p);
            uint32_t s0, s1;
            px_sample_key(sh_jax[0], sh_jax[1], a.B, p, s0, s1);
            a.skeys[((size_t)t * a.B + p) * 2] = s0;
            a.skeys[((size_t)t * a.B + p) * 2 + 1] = s1;
        }
        return;
    }
    // ---- aux block: per-site keys split(perturbation_key, 2) (svi.py:491), then the normals
    if (tid < 2) {
        uint32_t k[16], c[16];
        load_key(slot->pert_key, k);
        derive_child(k, (uint32_t)tid, 0u, D3P_TAG_SPLIT, c);
#pragma unroll
        for (int w = 0; w < 16; ++w) sh_key[tid][w] = c[w];
    } else if (tid == 64) {
        uint32_t k[16], o[16];
        load_key(slot->grad_key, k);
        keystream_block(k, 0u, o);
        slot->jax_key[0] = o[0];
        slot->jax_key[1] = o[1];
    } else if (tid == 128) {
        if (a.batch_key) {
            uint32_t k[16], c[16];
            load_key(a.batch_key, k);
            derive_child(k, 0u, slot->batch_i, D3P_TAG_FOLD, c);
#pragma unroll
            for (int w = 0; w < 16; ++w) slot->batch_key[w] = c[w];
        }
        if (a.kind != D3P_BATCH_POISSON) {  // POISSON: written by the select kernels
            slot->counts[0] = a.B;
            slot->counts[1] = a.B;
        }
        const float ip1 = (float)(slot->adam_i + 1);
        slot->bc1 = 1.0f - powf(a.b1, ip1);
        slot->bc2 = 1.0f - powf(a.b2, ip1);
    }
    __syncthreads();
    // noise[site * D + e] = normal(site_key[site])[e]  (svi.py:487): ChaCha block e/16, word e%16
    const int blocks_per_site = (a.D + 15) / 16;
    for (int j = tid; j < 2 * blocks_per_site; j += blockDim.x) {
        const int site = j / blocks_per_site, b = j % blocks_per_site;
        uint32_t k[16], o[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) k[w] = sh_key[site][w];
        keystream_block(k, (uint32_t)b, o);
        float* dst = a.noise + (size_t)t * 2 * a.D + (size_t)site * a.D;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int e = 16 * b + w;
            if (e < a.D) dst[e] = bits_to_normal(o[w]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// partial reduction / finalize: one workgroup of 16 waves per 64 columns; wave w sums rows
// w, w+16, ... (coalesced 256-byte reads), wave 0 adds the 16 wave sums in fixed order.
// ------------------------------------------------------------------------------------------
#define D3P_FIN_W 16

__device__ __forceinline__ float strided_rows_sum(const float* __restrict__ parts, uint32_t nparts, uint32_t stride,
                                                  uint32_t col, int wave)
{
    // 16 independent loads are issued before the first add (a plain accumulate loop serialises them)
    float s = 0.f;
    for (uint32_t r0 = wave; r0 < nparts; r0 += 16 * D3P_FIN_W) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t r = r0 + j * D3P_FIN_W;
            v[j] = r < nparts ? parts[(size_t)r * stride + col] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) s += v[j];
    }
    return s;
}

// Sum of one column over all rows by a single wave (lanes stride the rows); all lanes get the total.
__device__ __forceinline__ float wave_column_sum(const float* __restrict__ parts, uint32_t nparts, uint32_t stride,
                                                 uint32_t col, int lane)
{
    float s = 0.f;
    for (uint32_t r0 = lane; r0 < nparts; r0 += 8 * 64) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t r = r0 + j * 64;
            v[j] = r < nparts ? parts[(size_t)r * stride + col] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    return wave_sum(s);
}

__global__ void __launch_bounds__(1024)
k_reduce_partials(const float* __restrict__ parts, uint32_t nparts, uint32_t ncols, float* __restrict__ sums)
{
    __shared__ float lds[D3P_FIN_W][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col = blockIdx.x * 64 + lane;
    lds[wave][lane] = col < ncols ? strided_rows_sum(parts, nparts, ncols, col, wave) : 0.f;
    __syncthreads();
    if (wave == 0 && col < ncols) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < D3P_FIN_W; ++w) tot += lds[w][lane];
        sums[col] = tot;
    }
}

struct FinalArgs {
    const float* parts;  // nparts x (P + 2)
    uint32_t nparts;
    const StepSlot* slot;
    const float* noise;  // P standard normals of this step
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

__global__ void __launch_bounds__(1024) k_finalize(FinalArgs a)
{
    __shared__ float lds[D3P_FIN_W][64];
    const int D = a.m.d + (a.m.intercept ? 1 : 0), P = 2 * D;
    const uint32_t stride = P + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col = blockIdx.x * 64 + lane;
    const bool col_ok = col < (uint32_t)P;
    // operands of the element-wise tail: requested before the reduction so their latency overlaps it
    float x = 0.f, m = 0.f, v = 0.f, z = 0.f, bc1 = 1.f, bc2 = 1.f;
    if (wave == 0 && col_ok) {
        x = a.params[col];
        m = a.adam_m[col];
        v = a.adam_v[col];
        z = a.noise[col];
        bc1 = a.slot->bc1;
        bc2 = a.slot->bc2;
    }
    lds[wave][lane] = col_ok ? strided_rows_sum(a.parts, a.nparts, stride, col, wave) : 0.f;
    // wave 0 of every workgroup: number of valid examples; wave 0 of workgroup 0 also the loss sum
    float n = 0.f, loss_sum = 0.f;
    if (wave == 0) n = wave_column_sum(a.parts, a.nparts, stride, P + 1, lane);
    if (wave == 0 && blockIdx.x == 0) loss_sum = wave_column_sum(a.parts, a.nparts, stride, P, lane);
    __syncthreads();
    if (wave != 0) return;
    const float Bf = (float)a.B;
    const float factor = (n == 0.0f) ? 0.0f : Bf / n;  // svi.py:305
    const float obs_scale = 1.0f / a.m.inv_obs;
    if (col_ok) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < D3P_FIN_W; ++w) tot += lds[w][lane];
        const int site = col >= (uint32_t)D, e = col - site * D;
        const float avg = tot / Bf;                           // svi.py:343-346
        const float scale = a.h.dp_scale * (a.h.clip / n);    // svi.py:365-366 (n == 0 -> inf, as the reference)
        const float g = (avg + z * scale) * obs_scale * factor;  // svi.py:487-488, :375
        if (a.grad_out) a.grad_out[col] = g;
        // numpyro.optim.Adam (jax.example_libraries.optimizers.adam)
        m = (1.0f - a.h.b1) * g + a.h.b1 * m;
        v = (1.0f - a.h.b2) * g * g + a.h.b2 * v;
        const float mhat = m / bc1, vhat = v / bc2;
        x = x - a.h.lr * mhat / (sqrtf(vhat) + a.h.adam_eps);
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
    if (blockIdx.x == 0 && lane == 0) {
        if (a.loss_out) *a.loss_out = (loss_sum / Bf) * obs_scale * factor;  // svi.py:342, :306
        *a.adam_step = a.slot->adam_i + 1;
        if (a.batch_index) *a.batch_index = a.slot->batch_i + 1u;
    }
}

// ------------------------------------------------------------------------------------------
// host-side launch logic
// ------------------------------------------------------------------------------------------
static int validate(const d3p_logreg_model* m, const d3p_dpsvi_hyper* h, const d3p_dpsvi_state* st,
                    const d3p_batch_source* src)
{
    D3P_REQUIRE(m && h && st && src, "null argument struct");
    D3P_REQUIRE(m->d >= 1, "model.d must be >= 1");
    D3P_REQUIRE(m->prior_w > 0.f && m->prior_b > 0.f, "prior scales must be positive");
    D3P_REQUIRE(m->inv_obs > 0.f, "inv_obs must be positive");
    D3P_REQUIRE(h->clip != 0.f, "The clipping threshold must be greater than 0.");  // svi.py:119-120
    D3P_REQUIRE(std::isfinite(h->clip), "clipping_threshold must be finite!");       // svi.py:187-188
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

struct Ctx {
    hipStream_t s;
    const d3p_logreg_model* m;
    const d3p_dpsvi_hyper* h;
    const d3p_dpsvi_state* st;
    const d3p_batch_source* src;
    Workspace ws;
    MainGeom g;
    int D, P;
    Workspace ws2;  // second slot buffer (slots / idx / skeys / noise) for the pipelined run loop
};

static int enqueue_sched_init(const Ctx& c)
{
    const bool sampled = c.src->kind != D3P_BATCH_EXPLICIT;
    hipLaunchKernelGGL(k_sched_init, dim3(1), dim3(64), 0, c.s, (const uint32_t*)(c.st->rng_key + 16 * (c.st->key_slot & 1)),
                       (const int32_t*)c.st->step, sampled ? (const uint32_t*)c.src->batch_index : nullptr, c.ws.sched);
    hipLaunchKernelGGL(k_pack, dim3(cdiv(c.D, 256)), dim3(256), 0, c.s, *c.m, (const float*)c.st->params, c.ws.pack);
    return check_launch("k_sched_init");
}

// key chain + sampler for the next K steps
static int enqueue_batch_prep(const Ctx& c, int K)
{
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, c.s, c.ws.sched, c.ws.slots, K);
    SamplerArgs sa;
    sa.slots = c.ws.slots;
    sa.batch_key = c.src->kind == D3P_BATCH_EXPLICIT ? nullptr : c.src->batch_key;
    sa.idx = c.ws.idx;
    sa.skeys = c.ws.skeys;
    sa.noise = c.ws.noise;
    sa.B = c.src->B;
    sa.capacity = (uint32_t)c.src->n_rows;
    const int bits = bit_length_u32(sa.capacity - 1);
    sa.bits_lower = bits >> 1;
    sa.bits_upper = bits - sa.bits_lower;
    sa.kind = c.src->kind;
    sa.D = c.D;
    sa.b1 = c.h->b1;
    sa.b2 = c.h->b2;
    hipLaunchKernelGGL(k_sampler, dim3(cdiv(c.src->B, 256) + 1, K), dim3(256), 0, c.s, sa);
    int rc = check_launch("k_sampler");
    if (rc) return rc;
    if (c.src->kind == D3P_BATCH_POISSON) {
        for (int t = 0; t < K; ++t) {
            rc = d3p_poisson_select((void*)c.s, c.ws.slots[t].batch_key, c.src->q, (uint32_t)c.src->n_rows, c.src->B,
                                    c.src->suppress, c.ws.idx + (size_t)t * c.src->B, c.ws.slots[t].counts,
                                    c.ws.poisson_ws, c.ws.poisson_bytes);
            if (rc) return rc;
        }
    }
    return D3P_OK;
}

static int enqueue_main(const Ctx& c, int t, const float* X, const float* y, const float* eps, bool stamps,
                        hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    MainArgs a;
    memset(&a, 0, sizeof(a));
    fill_model_scalars(c.m, &a);
    a.X = X;
    a.y = y;
    a.idx = c.src->kind == D3P_BATCH_EXPLICIT ? nullptr : c.ws.idx + (size_t)t * c.src->B;
    a.mask = c.src->kind == D3P_BATCH_EXPLICIT ? c.src->mask : nullptr;
    a.counts = c.ws.slots[t].counts;
    a.skeys = c.ws.skeys + (size_t)t * 2 * c.src->B;
    a.eps_ext = eps;
    a.pack = c.ws.pack;
    a.partials = c.ws.partials;
    a.B = c.src->B;
    a.row_lo = c.src->row_lo;
    a.row_hi = c.src->row_hi;
    a.clip = c.h->clip;
    a.stamps = stamps ? c.ws.stamps : nullptr;
    if (const char* e = getenv("D3P_DBG")) a.dbg = atoi(e);
    return launch_main<0>(c.s, c.g, a, e0, e1);
}

static int enqueue_finalize(const Ctx& c, int t, const float* parts, uint32_t nparts, float* loss, float* grad_out)
{
    FinalArgs fa;
    fa.parts = parts;
    fa.nparts = nparts;
    fa.slot = c.ws.slots + t;
    fa.noise = c.ws.noise + (size_t)t * c.P;
    fa.params = c.st->params;
    fa.adam_m = c.st->adam_m;
    fa.adam_v = c.st->adam_v;
    fa.adam_step = c.st->step;
    fa.batch_index = c.src->kind == D3P_BATCH_EXPLICIT ? nullptr : c.src->batch_index;
    fa.pack = c.ws.pack;
    fa.loss_out = loss;
    fa.grad_out = grad_out;
    fa.B = c.src->B;
    fa.m = *c.m;
    fa.h = *c.h;
    hipLaunchKernelGGL(k_finalize, dim3(cdiv(c.P, 64)), dim3(64 * D3P_FIN_W), 0, c.s, fa);
    return check_launch("k_finalize");
}

static int enqueue_sched_finish(const Ctx& c, int steps_done)
{
    hipLaunchKernelGGL(k_sched_finish, dim3(1), dim3(64), 0, c.s, (const Sched*)c.ws.sched,
                       c.st->rng_key + 16 * ((c.st->key_slot + steps_done) & 1));
    return check_launch("k_sched_finish");
}

static int make_ctx(Ctx* c, void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                    const d3p_dpsvi_state* state, const d3p_batch_source* src, void* workspace_dev,
                    size_t workspace_bytes)
{
    int rc = validate(model, hyper, state, src);
    if (rc) return rc;
    D3P_REQUIRE(workspace_dev, "null workspace");
    const size_t need = carve(model, src, nullptr, nullptr);
    if (workspace_bytes < need) return fail(D3P_E_WORKSPACE, "workspace too small (%zu < %zu)", workspace_bytes, need);
    carve(model, src, (char*)workspace_dev, &c->ws, &c->ws2);
    rc = main_geometry(model, src->B, &c->g);
    if (rc) return rc;
    c->s = (hipStream_t)stream;
    c->m = model;
    c->h = hyper;
    c->st = state;
    c->src = src;
    c->D = model->d + (model->intercept ? 1 : 0);
    c->P = 2 * c->D;
    return D3P_OK;
}

}  // namespace d3p

using namespace d3p;

extern "C" {

size_t d3p_dpvi_logreg_workspace(const d3p_logreg_model* model, const d3p_batch_source* src)
{
    if (!model || !src) return 0;
    return carve(model, src, nullptr, nullptr);
}

int d3p_dpvi_logreg_local_sums(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                               const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev,
                               const float* y_dev, const float* eps_dev, float* sums_dev, void* workspace_dev,
                               size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(X_dev && y_dev && sums_dev, "null data pointer");
    if ((rc = enqueue_sched_init(c))) return rc;
    if ((rc = enqueue_batch_prep(c, 1))) return rc;
    if ((rc = enqueue_main(c, 0, X_dev, y_dev, eps_dev, false))) return rc;
    hipLaunchKernelGGL(k_reduce_partials, dim3(cdiv(c.P + 2, 64)), dim3(64 * D3P_FIN_W), 0, c.s,
                       (const float*)c.ws.partials, c.g.blocks, (uint32_t)(c.P + 2), sums_dev);
    return check_launch("k_reduce_partials");
}

int d3p_dpvi_logreg_finalize(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                             const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* sums_dev,
                             float* loss_dev, float* grad_out_dev, void* workspace_dev, size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(sums_dev, "null sums pointer");
    if ((rc = enqueue_finalize(c, 0, sums_dev, 1u, loss_dev, grad_out_dev))) return rc;
    return enqueue_sched_finish(c, 1);
}

int d3p_dpvi_logreg_run(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                        const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev,
                        const float* y_dev, uint32_t num_steps, float* losses_dev, void* workspace_dev,
                        size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(X_dev && y_dev, "null data pointer");
    D3P_REQUIRE(src->row_lo == 0 && src->row_hi == src->n_rows, "d3p_dpvi_logreg_run is the single-GPU path");
    if ((rc = enqueue_sched_init(c))) return rc;
    // The key chain + sampler of batch b+1 run on an auxiliary stream while the update steps of batch
    // b run on `stream`: they only depend on keys, never on parameters.  Slots are double-buffered.
    hipStream_t aux = nullptr;
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_init = nullptr;
    // Measured on MI355X (round 1): the concurrent sampler workgroups delay the dispatch of the main kernel
    // more than the overlap saves (19.2 vs 18.1 us/step), so the pipeline is opt-in.
    const bool pipelined = num_steps > D3P_STEP_BATCH && getenv("D3P_AUX_STREAM") != nullptr;
    auto cleanup = [&]() {
        for (int i = 0; i < 2; ++i) {
            if (ev_ready[i]) (void)hipEventDestroy(ev_ready[i]);
            if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
        }
        if (ev_init) (void)hipEventDestroy(ev_init);
        if (aux) (void)hipStreamDestroy(aux);
    };
#define D3P_TRY_CLEAN(expr)                                                                    \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            cleanup();                                                                         \
            return fail(D3P_E_HIP, "%s: %s", #expr, hipGetErrorString(e__));                   \
        }                                                                                      \
    } while (0)
    if (pipelined) {
        D3P_TRY_CLEAN(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            D3P_TRY_CLEAN(hipEventCreateWithFlags(&ev_ready[i], hipEventDisableTiming));
            D3P_TRY_CLEAN(hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming));
        }
        D3P_TRY_CLEAN(hipEventCreateWithFlags(&ev_init, hipEventDisableTiming));
        D3P_TRY_CLEAN(hipEventRecord(ev_init, c.s));
        D3P_TRY_CLEAN(hipStreamWaitEvent(aux, ev_init, 0));
    }
    const uint32_t n_batches = (num_steps + D3P_STEP_BATCH - 1) / D3P_STEP_BATCH;
    auto batch_len = [&](uint32_t b) {
        const uint32_t rem = num_steps - b * D3P_STEP_BATCH;
        return (int)(rem < D3P_STEP_BATCH ? rem : D3P_STEP_BATCH);
    };
    Ctx cb[2] = {c, c};  // views of the two slot buffers
    cb[1].ws = c.ws2;
    Ctx ca[2] = {cb[0], cb[1]};  // the same, enqueuing on the auxiliary stream
    if (pipelined) ca[0].s = ca[1].s = aux;
    if (num_steps > 0) {
        if ((rc = enqueue_batch_prep(ca[0], batch_len(0)))) { cleanup(); return rc; }
        if (pipelined) D3P_TRY_CLEAN(hipEventRecord(ev_ready[0], aux));
    }
    for (uint32_t b = 0; b < n_batches; ++b) {
        const int cur = pipelined ? (int)(b & 1) : 0, nxt = cur ^ 1;
        if (pipelined && b + 1 < n_batches) {
            if (b >= 1) D3P_TRY_CLEAN(hipStreamWaitEvent(aux, ev_done[nxt], 0));  // buffer `nxt` consumed
            if ((rc = enqueue_batch_prep(ca[nxt], batch_len(b + 1)))) { cleanup(); return rc; }
            D3P_TRY_CLEAN(hipEventRecord(ev_ready[nxt], aux));
        }
        if (pipelined) D3P_TRY_CLEAN(hipStreamWaitEvent(c.s, ev_ready[cur], 0));
        const int K = batch_len(b);
        for (int t = 0; t < K; ++t) {
            if ((rc = enqueue_main(cb[cur], t, X_dev, y_dev, nullptr, false))) { cleanup(); return rc; }
            if ((rc = enqueue_finalize(cb[cur], t, c.ws.partials, c.g.blocks,
                                       losses_dev ? losses_dev + (size_t)b * D3P_STEP_BATCH + t : nullptr, nullptr))) {
                cleanup();
                return rc;
            }
        }
        if (pipelined) D3P_TRY_CLEAN(hipEventRecord(ev_done[cur], c.s));
        if (!pipelined && b + 1 < n_batches)
            if ((rc = enqueue_batch_prep(cb[0], batch_len(b + 1)))) return rc;
    }
    cleanup();  // events/streams are released once their pending work completes (HIP defers destruction)
#undef D3P_TRY_CLEAN
    return enqueue_sched_finish(c, (int)num_steps);
}

int d3p_dpvi_logreg_time_main_kernel(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                                     const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev,
                                     const float* y_dev, void* workspace_dev, size_t workspace_bytes, int reps,
                                     float* avg_us, float* avg_event_us)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(X_dev && y_dev && avg_us && reps >= 1, "bad arguments");
    if ((rc = enqueue_sched_init(c))) return rc;
    if ((rc = enqueue_batch_prep(c, 1))) return rc;
    hipEvent_t e0, e1;
    D3P_HIP_TRY(hipEventCreate(&e0));
    D3P_HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i)  // warm-up
        if ((rc = enqueue_main(c, 0, X_dev, y_dev, nullptr, false))) return rc;
    const uint32_t nb = c.g.blocks;
    unsigned long long* host = (unsigned long long*)malloc((size_t)D3P_MAIN_MAX_BLOCKS * 2 * sizeof(unsigned long long));
    if (!host) return fail(D3P_E_HIP, "out of host memory");
    double dev_us = 0.0, ev_ms = 0.0;
    for (int i = 0; i < reps; ++i) {
        rc = enqueue_main(c, 0, X_dev, y_dev, nullptr, true, e0, e1);
        if (rc) { free(host); return rc; }
        hipError_t e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipMemcpyAsync(host, c.ws.stamps, (size_t)D3P_MAIN_MAX_BLOCKS * 2 * sizeof(unsigned long long),
                                                hipMemcpyDeviceToHost, c.s);
        if (e == hipSuccess) e = hipStreamSynchronize(c.s);
        if (e != hipSuccess) { free(host); return fail(D3P_E_HIP, "timing: %s", hipGetErrorString(e)); }
        unsigned long long t0 = ~0ull, t1 = 0ull;
        const char* dbg = getenv("D3P_DBG");
        const int SS = (dbg && (atoi(dbg) & 32)) ? 8 : 2;
        for (uint32_t b = 0; b < nb && SS * b + 1 < 2 * D3P_MAIN_MAX_BLOCKS; ++b) {
            if (host[SS * b] < t0) t0 = host[SS * b];
            if (host[SS * b + 1] > t1) t1 = host[SS * b + 1];
        }
        if (SS == 8 && i == reps - 1) {  // diagnostic: phase stamps relative to the first entry, in us
            for (uint32_t b = 0; b < nb && b < 256; b += 37)
                fprintf(stderr, "wg %3u: entry %.2f  packsync %.2f  eps-done %.2f  dot-done %.2f  loop-done %.2f  "
                        "red-sync %.2f  exit %.2f  clk %.0f MHz\n", b, (host[8*b]-t0)*0.01, (host[8*b+2]-t0)*0.01, (host[8*b+3]-t0)*0.01,
                        (host[8*b+4]-t0)*0.01, (host[8*b+5]-t0)*0.01, (host[8*b+6]-t0)*0.01, (host[8*b+1]-t0)*0.01, (double)host[8*b+7] / ((host[8*b+1]-host[8*b])*0.01));
        }
        dev_us += (double)(t1 - t0) * 0.01;  // wall_clock64 ticks at 100 MHz
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ev_ms += ms;
    }
    free(host);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = (float)(dev_us / reps);
    if (avg_event_us) *avg_event_us = (float)(ev_ms * 1000.0 / reps);
    return D3P_OK;
}

}  // extern "C"
