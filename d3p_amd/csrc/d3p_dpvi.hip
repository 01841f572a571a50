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
//
// Pipelined variant (default for Feistel subsampling, d3p_dpvi_logreg_run): the main kernel is bound
// by VALU issue, 60 % of it the generation of the guide noise eps, while k_finalize keeps only 16 of
// 256 CUs busy.  k_carrier therefore runs k_finalize's work for step u next to workgroups that do
// the parameter-independent work of LATER steps: key chain (u+4), derived keys (u+3), sample keys +
// Gaussian-mechanism normals (u+2), Feistel indices + eps (u+1).  The main kernel then only
// gathers X, reads eps and does the parameter-dependent arithmetic.
#include "d3p_logreg_kernel.h"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the entry points are resolved at run time from the RCCL torch has loaded

#define D3P_STEP_BATCH 32
#define D3P_RING 8        // ring of step slots used by the pipelined run loop (> look-ahead)
#define D3P_LOOKAHEAD 4

namespace d3p {

struct Workspace {
    Sched* sched;
    StepSlot* slots;  // D3P_STEP_BATCH
    float* pack;      // [loc | s | sg | q | lc] x D
    uint32_t* idx;    // D3P_STEP_BATCH x B
    uint32_t* skeys;  // D3P_STEP_BATCH x 2B
    uint32_t* plist;  // D3P_STEP_BATCH x B: dense owned-position lists
    float* noise;     // D3P_STEP_BATCH x P
    float* eps;       // B x D: guide noise of the NEXT step, staged by k_carrier
    long long* acc;   // 3 x D3P_ACC_R x (P + 2) fixed-point accumulators of the one-launch step
    float* scratch_state;  // 3P + 4 floats: stand-in state for the timing entry point
    float* pp_state;       // 3P floats: second buffer of the ping-ponged optimiser state (one-launch-per-step path)
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
    p = take(K * B * sizeof(uint32_t)); if (ws) ws->plist = (uint32_t*)p;
    p = take(K * P * sizeof(float)); if (ws) ws->noise = (float*)p;
    p = take(B * D * sizeof(float)); if (ws) ws->eps = (float*)p;
    p = take(3 * (size_t)D3P_ACC_R * (P + 2) * sizeof(long long)); if (ws) ws->acc = (long long*)p;
    p = take((3 * P + 4) * sizeof(float)); if (ws) ws->scratch_state = (float*)p;
    p = take(3 * P * sizeof(float)); if (ws) ws->pp_state = (float*)p;
    p = take((size_t)D3P_MAIN_MAX_BLOCKS * (P + 2) * sizeof(float)); if (ws) ws->partials = (float*)p;
    p = take((size_t)D3P_MAIN_MAX_BLOCKS * 2 * sizeof(unsigned long long)); if (ws) ws->stamps = (unsigned long long*)p;
    size_t pb = 0;
    if (src->kind == D3P_BATCH_POISSON) pb = K * d3p_poisson_select_workspace((uint32_t)src->n_rows);
    p = take(pb); if (ws) { ws->poisson_ws = p; ws->poisson_bytes = pb; }
    if (ws2) {
        *ws2 = ws ? *ws : Workspace();
    }
    p = take(K * sizeof(StepSlot)); if (ws2) ws2->slots = (StepSlot*)p;
    p = take(K * B * sizeof(uint32_t)); if (ws2) ws2->idx = (uint32_t*)p;
    p = take(K * 2 * B * sizeof(uint32_t)); if (ws2) ws2->skeys = (uint32_t*)p;
    p = take(K * B * sizeof(uint32_t)); if (ws2) ws2->plist = (uint32_t*)p;
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
    uint64_t row_lo, row_hi;  // rows held by this rank: sample keys are only needed for those
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
            bool owned = true;
            if (a.kind == D3P_BATCH_FEISTEL) {
                const uint32_t r = feistel_permute_dev(sh_rc, a.capacity, a.bits_lower, a.bits_upper, p);
                a.idx[(size_t)t * a.B + p] = r;
                owned = (uint64_t)r >= a.row_lo && (uint64_t)r < a.row_hi;
            }
            if (owned) {  // six dependent threefry calls: skipped for positions another rank processes
                uint32_t s0, s1;
                px_sample_key(sh_jax[0], sh_jax[1], a.B, p, s0, s1);
                a.skeys[((size_t)t * a.B + p) * 2] = s0;
                a.skeys[((size_t)t * a.B + p) * 2 + 1] = s1;
            }
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
// dense list of the batch positions a rank processes: valid (p < counts[1]) and row in [row_lo, row_hi).
// One workgroup per step; positions stay in ascending order (deterministic).  Needed for Poisson batches
// (padding) and for row-sharded multi-GPU runs, where only ~B / world of the global positions are owned.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_owned_list(StepSlot* __restrict__ slots, const uint32_t* __restrict__ idx, uint32_t B, uint64_t row_lo, uint64_t row_hi,
             uint32_t* __restrict__ plist)
{
    __shared__ uint32_t part[1024];
    const int t = blockIdx.y;
    const uint32_t* ix = idx + (size_t)t * B;
    uint32_t* out = plist + (size_t)t * B;
    const uint32_t n_valid = slots[t].counts[1];
    const uint32_t chunk = (B + 1023u) / 1024u;
    const uint32_t p0 = threadIdx.x * chunk;
    uint32_t c = 0;
    for (uint32_t i = 0; i < chunk; ++i) {
        const uint32_t p = p0 + i;
        if (p < B && p < n_valid) {
            const uint64_t r = ix[p];
            c += (r >= row_lo && r < row_hi) ? 1u : 0u;
        }
    }
    part[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = (threadIdx.x >= (unsigned)off) ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t w = part[threadIdx.x] - c;
    for (uint32_t i = 0; i < chunk; ++i) {
        const uint32_t p = p0 + i;
        if (p < B && p < n_valid) {
            const uint64_t r = ix[p];
            if (r >= row_lo && r < row_hi) out[w++] = p;
        }
    }
    if (threadIdx.x == 1023) slots[t].n_owned = part[1023];
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
    // optional piggy-backed key-chain step of the NEXT batch (one extra workgroup, one wavefront)
    Sched* chain_sched;
    StepSlot* chain_slot;
    int chain_t, chain_last;
};

__device__ __forceinline__ void finalize_role(const FinalArgs& a, uint32_t blk, float (*lds)[64])
{
    const int D = a.m.d + (a.m.intercept ? 1 : 0), P = 2 * D;
    const uint32_t stride = P + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col = blk * 64 + lane;
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
    if (wave == 0 && blk == 0) loss_sum = wave_column_sum(a.parts, a.nparts, stride, P, lane);
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
            float s, sg;
            guide_scale(a.m.guide_transform, x, s, sg);
            const float ps = (e < a.m.d) ? a.m.prior_w : a.m.prior_b;
            a.pack[D + e] = s;
            a.pack[2 * D + e] = sg;
            a.pack[3 * D + e] = a.m.inv_obs * sg / s;
            a.pack[4 * D + e] = logf(ps) - logf(s);
        }
    }
    if (blk == 0 && lane == 0) {
        if (a.loss_out) *a.loss_out = (loss_sum / Bf) * obs_scale * factor;  // svi.py:342, :306
        *a.adam_step = a.slot->adam_i + 1;
        if (a.batch_index) *a.batch_index = a.slot->batch_i + 1u;
    }
}

__global__ void __launch_bounds__(1024) k_finalize(FinalArgs a, uint32_t n_fin)
{
    __shared__ float lds[D3P_FIN_W][64];
    if (blockIdx.x >= n_fin) {  // the piggy-backed chain step (independent of everything else in this launch)
        if (threadIdx.x < 64) chain_step(a.chain_sched, a.chain_slot, a.chain_t, a.chain_last);
        return;
    }
    finalize_role(a, blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------
// carrier kernel: finalize of step u + parameter-independent stages of later steps
// ------------------------------------------------------------------------------------------
struct CarrierArgs {
    FinalArgs fin;
    int n_fin;           // finalize workgroups (0: no finalize in this launch)
    Sched* sched;
    StepSlot* slots;     // ring of D3P_RING
    const uint32_t* bkey;
    uint32_t* skeys;     // ring x 2B
    float* noise;        // ring x P
    uint32_t* idx;       // B (next step)
    float* eps;          // B x D (next step)
    int tA, tB, tC, tD;  // step (relative to the start of the run) each stage works on, -1 = idle
    uint32_t B;
    int D;
    uint32_t capacity;
    int bits_lower, bits_upper;
    float b1, b2;
    int n_idx_wg, n_eps_wg, n_sk_wg;
};

// stage A: (next, gradient, perturbation) = split(chain_key, 3)   (svi.py:208-211, :413-414)
__device__ __forceinline__ void stage_chain(Sched* sched, StepSlot* slot, int t)
{
    const int lane = threadIdx.x & 63;
    uint32_t cur[16], child[16];
    load_key(sched->key, cur);
    const int32_t adam0 = sched->adam_i;
    const uint32_t batch0 = sched->batch_i;
    derive_child(cur, (uint32_t)(lane < 3 ? lane : 0), 0u, D3P_TAG_SPLIT, child);
    uint32_t* dst = lane == 0 ? slot->next_key : (lane == 1 ? slot->grad_key : slot->pert_key);
    if (lane < 3) {
#pragma unroll
        for (int w = 0; w < 16; ++w) dst[w] = child[w];
    }
    if (lane == 0) {
#pragma unroll
        for (int w = 0; w < 16; ++w) sched->key[w] = child[w];
    } else if (lane == 3) {
        slot->adam_i = adam0 + t;
        slot->batch_i = batch0 + (uint32_t)t;
    }
}

// stage B: jax key, per-site keys, batch key -- one ChaCha block each, four lanes of one wave
__device__ __forceinline__ void stage_keys(StepSlot* slot, const uint32_t* __restrict__ bkey)
{
    const int lane = threadIdx.x & 63;
    if (lane == 0) {  // convert_to_jax_rng_key(gradient_key) (svi.py:259; random/__init__.py:155)
        uint32_t k[16], o[16];
        load_key(slot->grad_key, k);
        keystream_block(k, 0u, o);
        slot->jax_key[0] = o[0];
        slot->jax_key[1] = o[1];
    } else if (lane == 1 || lane == 2) {  // split(perturbation_key, 2) (svi.py:491)
        uint32_t k[16], c[16];
        load_key(slot->pert_key, k);
        derive_child(k, (uint32_t)(lane - 1), 0u, D3P_TAG_SPLIT, c);
#pragma unroll
        for (int w = 0; w < 16; ++w) slot->site_keys[lane - 1][w] = c[w];
    } else if (lane == 3 && bkey) {  // fold_in(batchifier_state, i) (minibatch.py:230)
        uint32_t k[16], c[16];
        load_key(bkey, k);
        derive_child(k, 0u, slot->batch_i, D3P_TAG_FOLD, c);
#pragma unroll
        for (int w = 0; w < 16; ++w) slot->batch_key[w] = c[w];
    }
}

// stage C: Feistel round constants, Gaussian-mechanism normals, Adam bias terms, sample keys
__device__ __forceinline__ void stage_derive(const CarrierArgs& a, StepSlot* slot, uint32_t* skeys, float* noise)
{
    const int tid = threadIdx.x;
    if (tid < 2) {  // round constants (util.py:240-246)
        uint32_t k[16], o[16];
        load_key(slot->batch_key, k);
        keystream_block(k, (uint32_t)tid, o);
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int g = 16 * tid + w;
            if (g < 30) slot->rc[g] = (g % 3 == 0) ? (o[w] | 1u) : o[w];
        }
    } else if (tid == 63) {
        slot->counts[0] = a.B;
        slot->counts[1] = a.B;
        const float ip1 = (float)(slot->adam_i + 1);
        slot->bc1 = 1.0f - powf(a.b1, ip1);
        slot->bc2 = 1.0f - powf(a.b2, ip1);
    }
    const int blocks_per_site = (a.D + 15) / 16;
    for (int j = tid - 64; j >= 0 && j < 2 * blocks_per_site; j += (int)blockDim.x) {  // svi.py:487
        const int site = j / blocks_per_site, b = j % blocks_per_site;
        uint32_t k[16], o[16];
        load_key(slot->site_keys[site], k);
        keystream_block(k, (uint32_t)b, o);
        float* dst = noise + (size_t)site * a.D;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int e = 16 * b + w;
            if (e < a.D) dst[e] = bits_to_normal(o[w]);
        }
    }
}

// stage C (sample keys): 256 positions per workgroup on 4 of its waves -- six dependent threefry calls per
// position, so few waves per SIMD finish soonest   (svi.py:289-290 + numpyro key plumbing)
__device__ __forceinline__ void stage_sample_keys(const StepSlot* slot, uint32_t* __restrict__ skeys, uint32_t B,
                                                  uint32_t wg)
{
    if (threadIdx.x >= 256) return;
    const uint32_t p = wg * 256 + threadIdx.x;
    if (p >= B) return;
    uint32_t s0, s1;
    px_sample_key(slot->jax_key[0], slot->jax_key[1], B, p, s0, s1);
    skeys[2 * p] = s0;
    skeys[2 * p + 1] = s1;
}

// stage D (eps): one wavefront per example, jax.random.normal(sample_key, (D,)) in JAX's word layout
__device__ __forceinline__ void stage_eps(const uint32_t* __restrict__ skeys, float* __restrict__ eps, uint32_t B, int D,
                                          uint32_t wg, uint32_t n_wg)
{
    const int lane = threadIdx.x & 63;
    const uint32_t waves = blockDim.x >> 6;
    for (uint32_t p = wg * waves + (threadIdx.x >> 6); p < B; p += n_wg * waves) {
    const uint32_t k0 = skeys[2 * p], k1 = skeys[2 * p + 1];
    const int half = (D + 1) >> 1;
    float* row = eps + (size_t)p * D;
    if ((D & 7) == 0 && (half & 255) == 0) {  // 16-byte stores: lane owns pairs 4l..4l+3 (+256k); all lanes active
        for (int j = 4 * lane; j < half; j += 256) {
            float v0[4], v1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t b0, b1;
                threefry2x32(k0, k1, (uint32_t)(j + i), (uint32_t)(j + i + half), b0, b1);
                v0[i] = bits_to_normal_wu(b0);
                v1[i] = bits_to_normal_wu(b1);
            }
            *reinterpret_cast<float4*>(row + j) = make_float4(v0[0], v0[1], v0[2], v0[3]);
            *reinterpret_cast<float4*>(row + j + half) = make_float4(v1[0], v1[1], v1[2], v1[3]);
        }
    } else {
        for (int j0 = 0; j0 < half; j0 += 64) {  // all lanes iterate together (wave-uniform erf_inv tail)
            const int j = j0 + lane;
            const bool ok = j < half, ok1 = ok && (j + half < D);
            uint32_t b0, b1;
            threefry2x32(k0, k1, (uint32_t)j, ok1 ? (uint32_t)(j + half) : 0u, b0, b1);
            const float v0 = bits_to_normal_wu(b0), v1 = bits_to_normal_wu(b1);
            if (ok) row[j] = v0;
            if (ok1) row[j + half] = v1;
        }
    }
    }
}

__global__ void __launch_bounds__(1024) k_carrier(CarrierArgs a)
{
    __shared__ float lds[D3P_FIN_W][64];
    __shared__ uint32_t sh_rc[32];
    uint32_t blk = blockIdx.x;
    if (blk < (uint32_t)a.n_fin) {
        finalize_role(a.fin, blk, lds);
        return;
    }
    blk -= a.n_fin;
    if (blk == 0) {  // stages A and B on two different waves (they work on different steps)
        const int wave = threadIdx.x >> 6;
        if (wave == 0 && a.tA >= 0) stage_chain(a.sched, a.slots + (a.tA % D3P_RING), a.tA);
        if (wave == 1 && a.tB >= 0) stage_keys(a.slots + (a.tB % D3P_RING), a.bkey);
        return;
    }
    if (blk == 1) {
        if (a.tC >= 0) {
            const int r = a.tC % D3P_RING;
            stage_derive(a, a.slots + r, a.skeys + (size_t)r * 2 * a.B, a.noise + (size_t)r * 2 * a.D);
        }
        return;
    }
    blk -= 2;
    if (blk < (uint32_t)a.n_sk_wg) {
        const int r = a.tC % D3P_RING;
        stage_sample_keys(a.slots + r, a.skeys + (size_t)r * 2 * a.B, a.B, blk);
        return;
    }
    blk -= a.n_sk_wg;
    if (a.tD < 0) return;
    const int rD = a.tD % D3P_RING;
    if (blk < (uint32_t)a.n_idx_wg) {  // Feistel indices of the next step (util.py:273-300)
        if (threadIdx.x < 32) sh_rc[threadIdx.x] = a.slots[rD].rc[threadIdx.x];
        __syncthreads();
        const uint32_t p = blk * blockDim.x + threadIdx.x;
        if (p < a.B) a.idx[p] = feistel_permute_dev(sh_rc, a.capacity, a.bits_lower, a.bits_upper, p);
        return;
    }
    blk -= a.n_idx_wg;
    stage_eps(a.skeys + (size_t)rD * 2 * a.B, a.eps, a.B, a.D, blk, (uint32_t)a.n_eps_wg);
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

static bool need_owned_list(const d3p_batch_source* src)
{
    if (src->kind == D3P_BATCH_EXPLICIT) return false;
    return src->kind == D3P_BATCH_POISSON || src->row_lo != 0 || src->row_hi != src->n_rows;
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

static int enqueue_chain(const Ctx& c, int K)
{
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, c.s, c.ws.sched, c.ws.slots, K);
    return check_launch("k_chain");
}

static int enqueue_sampler(const Ctx& c, int K);

// key chain + sampler for the next K steps
static int enqueue_batch_prep(const Ctx& c, int K)
{
    int rc = enqueue_chain(c, K);
    if (rc) return rc;
    return enqueue_sampler(c, K);
}

static int enqueue_sampler(const Ctx& c, int K)
{
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
    sa.row_lo = c.src->kind == D3P_BATCH_EXPLICIT ? 0 : c.src->row_lo;
    sa.row_hi = c.src->kind == D3P_BATCH_EXPLICIT ? ~0ull : c.src->row_hi;
    hipLaunchKernelGGL(k_sampler, dim3(cdiv(c.src->B, 256) + 1, K), dim3(256), 0, c.s, sa);
    int rc = check_launch("k_sampler");
    if (rc) return rc;
    if (c.src->kind == D3P_BATCH_POISSON) {  // all K draws in one set of launches (blockIdx.y = step)
        rc = d3p_poisson_select_batch((void*)c.s, 0, c.ws.slots[0].batch_key, sizeof(StepSlot) / sizeof(uint32_t), c.src->q,
                                      (uint32_t)c.src->n_rows, c.src->B, c.src->suppress, c.ws.idx, c.src->B,
                                      c.ws.slots[0].counts, sizeof(StepSlot) / sizeof(uint32_t), (uint32_t)K,
                                      c.ws.poisson_ws, c.ws.poisson_bytes);
        if (rc) return rc;
    }
    if (need_owned_list(c.src)) {
        hipLaunchKernelGGL(k_owned_list, dim3(1, K), dim3(1024), 0, c.s, c.ws.slots, (const uint32_t*)c.ws.idx, c.src->B,
                           (uint64_t)c.src->row_lo, (uint64_t)c.src->row_hi, c.ws.plist);
        if ((rc = check_launch("k_owned_list"))) return rc;
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
    if (need_owned_list(c.src)) {
        a.plist = c.ws.plist + (size_t)t * c.src->B;
        a.n_list = &c.ws.slots[t].n_owned;
    }
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

static int enqueue_finalize(const Ctx& c, int t, const float* parts, uint32_t nparts, float* loss, float* grad_out,
                            StepSlot* chain_slot = nullptr, int chain_t = 0, int chain_last = 0)
{
    FinalArgs fa;
    fa.chain_sched = c.ws.sched;
    fa.chain_slot = chain_slot;
    fa.chain_t = chain_t;
    fa.chain_last = chain_last;
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
    const uint32_t n_fin = cdiv(c.P, 64);
    hipLaunchKernelGGL(k_finalize, dim3(n_fin + (chain_slot ? 1u : 0u)), dim3(64 * D3P_FIN_W), 0, c.s, fa, n_fin);
    return check_launch("k_finalize");
}

static int enqueue_sched_finish(const Ctx& c, int steps_done)
{
    hipLaunchKernelGGL(k_sched_finish, dim3(1), dim3(64), 0, c.s, (const Sched*)c.ws.sched,
                       c.st->rng_key + 16 * ((c.st->key_slot + steps_done) & 1));
    return check_launch("k_sched_finish");
}

__global__ void k_copy_key(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst)
{
    if (threadIdx.x < 16) dst[threadIdx.x] = src[threadIdx.x];
}

static bool use_carrier(const Ctx& c)
{
    // Measured on MI355X (round 1): 21.7 us/step against 17.8 us/step for the fused-RNG path -- every extra
    // role pays its own launch ramp / cold instruction cache, and the VALU work of eps is the same wherever it
    // runs.  Kept as an opt-in experiment.
    return c.src->kind == D3P_BATCH_FEISTEL && getenv("D3P_CARRIER") != nullptr;
}

// carrier launch number j of a run of n steps (j = -D3P_LOOKAHEAD .. n-1): finalize(j) if j >= 0 plus the
// look-ahead stages that still have a step to work on
static int enqueue_carrier(const Ctx& c, int j, int n, float* loss_out)
{
    CarrierArgs ca;
    memset(&ca, 0, sizeof(ca));
    auto in_run = [&](int t) { return (t >= 0 && t < n) ? t : -1; };
    ca.tA = in_run(j + 4);
    ca.tB = in_run(j + 3);
    ca.tC = in_run(j + 2);
    ca.tD = in_run(j + 1);
    const bool fin = j >= 0;
    if (!fin && ca.tA < 0 && ca.tB < 0 && ca.tC < 0 && ca.tD < 0) return D3P_OK;
    if (fin) {
        const int r = j % D3P_RING;
        FinalArgs& fa = ca.fin;
        fa.parts = c.ws.partials;
        fa.nparts = c.g.blocks;
        fa.slot = c.ws.slots + r;
        fa.noise = c.ws.noise + (size_t)r * c.P;
        fa.params = c.st->params;
        fa.adam_m = c.st->adam_m;
        fa.adam_v = c.st->adam_v;
        fa.adam_step = c.st->step;
        fa.batch_index = c.src->batch_index;
        fa.pack = c.ws.pack;
        fa.loss_out = loss_out;
        fa.grad_out = nullptr;
        fa.B = c.src->B;
        fa.m = *c.m;
        fa.h = *c.h;
        ca.n_fin = (int)cdiv(c.P, 64);
    }
    ca.sched = c.ws.sched;
    ca.slots = c.ws.slots;
    ca.bkey = c.src->batch_key;
    ca.skeys = c.ws.skeys;
    ca.noise = c.ws.noise;
    ca.idx = c.ws.idx;
    ca.eps = c.ws.eps;
    ca.B = c.src->B;
    ca.D = c.D;
    ca.capacity = (uint32_t)c.src->n_rows;
    const int bits = bit_length_u32(ca.capacity - 1);
    ca.bits_lower = bits >> 1;
    ca.bits_upper = bits - ca.bits_lower;
    ca.b1 = c.h->b1;
    ca.b2 = c.h->b2;
    ca.n_idx_wg = ca.tD >= 0 ? (int)cdiv(c.src->B, 1024) : 0;
    ca.n_sk_wg = ca.tC >= 0 ? (int)cdiv(c.src->B, 256) : 0;
    // one 1024-thread workgroup per CU: keep the whole launch within 256 workgroups (a second scheduling
    // round would double its duration); the eps role grid-strides over the examples
    ca.n_eps_wg = ca.tD >= 0 ? (int)cdiv(c.src->B, 16) : 0;
    const int others = ca.n_fin + 2 + ca.n_sk_wg + ca.n_idx_wg;
    if (ca.n_eps_wg > 0 && others + ca.n_eps_wg > 256) ca.n_eps_wg = (256 - others) > 16 ? (256 - others) : 16;
    if (const char* e = getenv("D3P_CARRIER_SKIP")) {  // developer ablation (results invalid): bit0 eps, 1 sample keys, 2 idx, 3 C, 4 A/B
        const int m = atoi(e);
        if (m & 1) ca.n_eps_wg = 0;
        if (m & 2) ca.n_sk_wg = 0;
        if (m & 4) ca.n_idx_wg = 0;
        if (m & 8) ca.tC = -1;
        if (m & 16) ca.tA = ca.tB = -1;
        if (m & 32) ca.n_fin = 0;
    }
    hipLaunchKernelGGL(k_carrier, dim3(ca.n_fin + 2 + ca.n_sk_wg + ca.n_idx_wg + ca.n_eps_wg), dim3(1024), 0, c.s, ca);
    return check_launch("k_carrier");
}

static int enqueue_main_staged(const Ctx& c, int j, const float* X, const float* y, bool stamps,
                               hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    MainArgs a;
    memset(&a, 0, sizeof(a));
    fill_model_scalars(c.m, &a);
    a.X = X;
    a.y = y;
    a.idx = c.ws.idx;
    a.counts = c.ws.slots[j % D3P_RING].counts;
    a.skeys = nullptr;
    a.eps_ext = c.ws.eps;
    a.pack = c.ws.pack;
    a.partials = c.ws.partials;
    a.B = c.src->B;
    a.row_lo = c.src->row_lo;
    a.row_hi = c.src->row_hi;
    a.clip = c.h->clip;
    a.stamps = stamps ? c.ws.stamps : nullptr;
    return launch_main<0>(c.s, c.g, a, e0, e1);
}

static int run_pipelined(const Ctx& c, const float* X, const float* y, int n, float* losses)
{
    int rc;
    for (int j = -D3P_LOOKAHEAD; j < n; ++j) {
        if (j >= 0 && (rc = enqueue_main_staged(c, j, X, y, false))) return rc;
        if ((rc = enqueue_carrier(c, j, n, (j >= 0 && losses) ? losses + j : nullptr))) return rc;
    }
    if (n > 0) {  // the state key after n steps is the chain key recorded by step n-1
        hipLaunchKernelGGL(k_copy_key, dim3(1), dim3(64), 0, c.s, (const uint32_t*)c.ws.slots[(n - 1) % D3P_RING].next_key,
                           c.st->rng_key + 16 * ((c.st->key_slot + n) & 1));
        return check_launch("k_copy_key");
    }
    return D3P_OK;
}

// ------------------------------------------------------------------------------------------
// one launch per step (MODE 2)
// ------------------------------------------------------------------------------------------
static bool use_fused_step(const Ctx& c)
{
    return getenv("D3P_NO_FUSED_STEP") == nullptr && getenv("D3P_CARRIER") == nullptr;
}

static void fill_fuse_common(const Ctx& c, StepFuse* f, int g)
{
    const size_t PA = (size_t)c.P + 2;
    f->acc_prev = c.ws.acc + (size_t)((g + 2) % 3) * D3P_ACC_R * PA;
    f->acc_cur = c.ws.acc + (size_t)(g % 3) * D3P_ACC_R * PA;
    f->acc_next = c.ws.acc + (size_t)((g + 1) % 3) * D3P_ACC_R * PA;
    f->R = D3P_ACC_R;
    f->adam_step = c.st->step;
    f->batch_index = c.src->kind == D3P_BATCH_EXPLICIT ? nullptr : c.src->batch_index;
    f->dp_scale = c.h->dp_scale;
    f->lr = c.h->lr;
    f->b1 = c.h->b1;
    f->b2 = c.h->b2;
    f->adam_eps = c.h->adam_eps;
    f->prior_w = c.m->prior_w;
    f->prior_b = c.m->prior_b;
    // gradient columns: every workgroup partial is bounded by 16 * C per step, sums by B * C -> 2^40 / C keeps
    // B up to 2^22 inside int64 with a resolution of C * 2^-40
    f->sg = 1099511627776.0 / (double)fabsf(c.h->clip);
    f->inv_sg = 1.0 / f->sg;
    // loss column: bound per example by inv_obs * (50 D + 1e3 * lik_scale); keep 2^61 of headroom
    const double bound = (double)c.m->inv_obs * (50.0 * c.D + 1.0e3 * (double)c.m->lik_scale) * (double)c.src->B + 1.0;
    int e = 0;
    (void)frexp(bound, &e);
    f->sl = ldexp(1.0, 61 - e);
    f->inv_sl = 1.0 / f->sl;
}

// step `g` of the run (slot `t` of buffer `cur`); prev = slot of step g-1 (nullable for g == 0)
static int enqueue_fused_step(const Ctx& c, int g, int t, const StepSlot* prev_slot, const float* prev_noise, const float* X,
                              const float* y, float* prev_loss, StepSlot* chain_slot, int chain_t, int chain_last,
                              bool flush_only, bool stamps = false, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    MainArgs a;
    memset(&a, 0, sizeof(a));
    fill_model_scalars(c.m, &a);
    a.X = X;
    a.y = y;
    a.idx = c.src->kind == D3P_BATCH_EXPLICIT ? nullptr : c.ws.idx + (size_t)t * c.src->B;
    a.mask = c.src->kind == D3P_BATCH_EXPLICIT ? c.src->mask : nullptr;
    a.counts = c.ws.slots[t].counts;
    if (need_owned_list(c.src) && !flush_only) {
        a.plist = c.ws.plist + (size_t)t * c.src->B;
        a.n_list = &c.ws.slots[t].n_owned;
    }
    a.skeys = c.ws.skeys + (size_t)t * 2 * c.src->B;
    a.pack = c.ws.pack;
    a.partials = c.ws.partials;
    a.B = c.src->B;
    a.row_lo = c.src->row_lo;
    a.row_hi = c.src->row_hi;
    a.clip = c.h->clip;
    a.stamps = stamps ? c.ws.stamps : nullptr;
    if (const char* e = getenv("D3P_DBG")) a.dbg = atoi(e);
    fill_fuse_common(c, &a.fuse, g);
    {
        // Launch g applies the update of step g - 1: it reads state buffer (g - 1) & 1 and publishes to buffer g & 1
        // (buffer 0 = the caller's arrays, buffer 1 = workspace).  The flush launch (one workgroup) always
        // publishes to the caller's arrays; the first launch has nothing to apply and only reads them.
        const size_t P = (size_t)c.P;
        float* const bufs[2][3] = {{c.st->params, c.st->adam_m, c.st->adam_v},
                                   {c.ws.pp_state, c.ws.pp_state + P, c.ws.pp_state + 2 * P}};
        const int in = g > 0 ? ((g - 1) & 1) : 0, out = flush_only ? 0 : (g & 1);
        a.fuse.params_in = bufs[in][0];
        a.fuse.m_in = bufs[in][1];
        a.fuse.v_in = bufs[in][2];
        a.fuse.params_out = bufs[out][0];
        a.fuse.m_out = bufs[out][1];
        a.fuse.v_out = bufs[out][2];
    }
    a.fuse.apply_prev = prev_slot != nullptr;
    a.fuse.prev_noise = prev_noise;
    a.fuse.prev_meta = prev_slot ? reinterpret_cast<const StepMeta*>(&prev_slot->adam_i) : nullptr;
    a.fuse.prev_loss_out = prev_loss;
    a.fuse.flush_only = flush_only ? 1 : 0;
    a.fuse.chain_sched = c.ws.sched;
    a.fuse.chain_slot = chain_slot;
    a.fuse.chain_t = chain_t;
    a.fuse.chain_last = chain_last;
    MainGeom g2 = c.g;
    if (flush_only) g2.blocks = 1;
    if (chain_slot) g2.blocks += 1;
    return launch_main<2>(c.s, g2, a, e0, e1);
}

// ---- RCCL, resolved lazily with dlopen so that libd3p_hip.so has no link-time dependency on it (single-GPU users
// never touch it) and shares the copy torch.distributed already loaded when there is one.
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
};

static const RcclApi* rccl_api()
{
    static RcclApi api;
    static int state = 0;  // 0 = not tried, 1 = ok, -1 = unavailable
    if (state == 0) {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
            api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
        }
        state = (h && api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GetErrorString) ? 1 : -1;
    }
    return state == 1 ? &api : nullptr;
}

static int enqueue_sampler(const Ctx& c, int K);
static int enqueue_chain(const Ctx& c, int K);

// comm != nullptr: data-parallel run -- after every step launch the rank's fixed-point accumulator (R x (P + 2) int64) is
// sum-all-reduced in place on the same stream (the ONE collective of the step); the next launch applies the global sums.
static int run_fused_steps(const Ctx& c, const float* X, const float* y, uint32_t num_steps, float* losses, ncclComm_t comm = nullptr)
{
    int rc;
    const size_t acc_bytes = 3 * (size_t)D3P_ACC_R * (c.P + 2) * sizeof(long long);
    D3P_HIP_TRY(hipMemsetAsync(c.ws.acc, 0, acc_bytes, c.s));
    const uint32_t n_batches = (num_steps + D3P_STEP_BATCH - 1) / D3P_STEP_BATCH;
    auto batch_len = [&](uint32_t b) {
        const uint32_t rem = num_steps - b * D3P_STEP_BATCH;
        return (int)(rem < D3P_STEP_BATCH ? rem : D3P_STEP_BATCH);
    };
    Ctx cb[2] = {c, c};
    cb[1].ws = c.ws2;
    cb[1].ws.partials = c.ws.partials;
    cb[1].ws.acc = c.ws.acc;
    cb[1].ws.stamps = c.ws.stamps;
    if (num_steps == 0) return D3P_OK;
    if ((rc = enqueue_chain(cb[0], batch_len(0)))) return rc;
    if ((rc = enqueue_sampler(cb[0], batch_len(0)))) return rc;
    const StepSlot* prev_slot = nullptr;
    const float* prev_noise = nullptr;
    int g = 0;
    for (uint32_t b = 0; b < n_batches; ++b) {
        const int cur = (int)(b & 1), nxt = cur ^ 1;
        const int K = batch_len(b), K_next = (b + 1 < n_batches) ? batch_len(b + 1) : 0;
        for (int t = 0; t < K; ++t, ++g) {
            static const bool no_piggy = getenv("D3P_NO_CHAIN_PIGGYBACK") != nullptr;
            StepSlot* cslot = (!no_piggy && t < K_next) ? cb[nxt].ws.slots + t : nullptr;
            if ((rc = enqueue_fused_step(cb[cur], g, t, prev_slot, prev_noise, X, y, (losses && g > 0) ? losses + g - 1 : nullptr,
                                         cslot, t, t == K_next - 1, false)))
                return rc;
            if (comm) {
                const size_t words = (size_t)D3P_ACC_R * (c.P + 2);
                long long* acc = c.ws.acc + (size_t)(g % 3) * words;
                const ncclResult_t r = rccl_api()->AllReduce(acc, acc, words, ncclInt64, ncclSum, comm, c.s);
                if (r != ncclSuccess) return fail(D3P_E_HIP, "ncclAllReduce: %s", rccl_api()->GetErrorString(r));
            }
            prev_slot = cb[cur].ws.slots + t;
            prev_noise = cb[cur].ws.noise + (size_t)t * c.P;
        }
        if (b + 1 < n_batches) {
            static const bool no_piggy2 = getenv("D3P_NO_CHAIN_PIGGYBACK") != nullptr;
            if (no_piggy2 && (rc = enqueue_chain(cb[nxt], K_next))) return rc;
            if ((rc = enqueue_sampler(cb[nxt], K_next))) return rc;
        }
    }
    // apply the update of the last step
    const int last_buf = (int)((n_batches - 1) & 1);
    if ((rc = enqueue_fused_step(cb[last_buf], g, 0, prev_slot, prev_noise, X, y, losses ? losses + g - 1 : nullptr, nullptr, 0, 0,
                                 true)))
        return rc;
    return D3P_OK;
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
    if (need_owned_list(src)) {
        // a rank processes ~B * (rows held / rows total) positions (Poisson: <= B valid ones): size the grid for
        // that, never beyond one workgroup per CU (waves loop over further items)
        const double frac = (double)(src->row_hi - src->row_lo) / (double)(src->n_rows ? src->n_rows : 1);
        const uint64_t expected = (uint64_t)((double)src->B * frac + 0.999);
        uint32_t blocks = cdiv(expected > 0 ? expected : 1, c->g.W);
        if (blocks > 256u) blocks = 256u;
        if (blocks < 1u) blocks = 1u;
        c->g.blocks = blocks;
    }
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
    D3P_REQUIRE(X_dev && sums_dev, "null data pointer");
    if (int rcm = validate_model(model, y_dev, "d3p_dpvi_logreg")) return rcm;
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

/* ---- stepwise API for the data-parallel loop: begin -> [prepare(K) -> K x (step_sums -> all-reduce ->
 *      step_finalize)]* -> end.  Per step only the fused kernel, the partial reduction and finalize are launched. */
int d3p_dpvi_logreg_begin(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                          const d3p_dpsvi_state* state, const d3p_batch_source* src, void* workspace_dev,
                          size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    return enqueue_sched_init(c);
}

int d3p_dpvi_logreg_prepare(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                            const d3p_dpsvi_state* state, const d3p_batch_source* src, uint32_t num_steps,
                            void* workspace_dev, size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(num_steps >= 1 && num_steps <= D3P_STEP_BATCH, "d3p_dpvi_logreg_prepare: 1 <= num_steps <= 32");
    return enqueue_batch_prep(c, (int)num_steps);
}

int d3p_dpvi_logreg_step_sums(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                              const d3p_dpsvi_state* state, const d3p_batch_source* src, uint32_t t,
                              const float* X_dev, const float* y_dev, const float* eps_dev, float* sums_dev,
                              void* workspace_dev, size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(X_dev && sums_dev, "null data pointer");
    if (int rcm = validate_model(model, y_dev, "d3p_dpvi_logreg")) return rcm;
    D3P_REQUIRE(t < D3P_STEP_BATCH, "d3p_dpvi_logreg_step_sums: t must be < 32");
    if ((rc = enqueue_main(c, (int)t, X_dev, y_dev, eps_dev, false))) return rc;
    hipLaunchKernelGGL(k_reduce_partials, dim3(cdiv(c.P + 2, 64)), dim3(64 * D3P_FIN_W), 0, c.s,
                       (const float*)c.ws.partials, c.g.blocks, (uint32_t)(c.P + 2), sums_dev);
    return check_launch("k_reduce_partials");
}

int d3p_dpvi_logreg_step_finalize(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                                  const d3p_dpsvi_state* state, const d3p_batch_source* src, uint32_t t,
                                  const float* sums_dev, float* loss_dev, float* grad_out_dev, void* workspace_dev,
                                  size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(sums_dev, "null sums pointer");
    D3P_REQUIRE(t < D3P_STEP_BATCH, "d3p_dpvi_logreg_step_finalize: t must be < 32");
    return enqueue_finalize(c, (int)t, sums_dev, 1u, loss_dev, grad_out_dev);
}

int d3p_dpvi_logreg_end(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                        const d3p_dpsvi_state* state, const d3p_batch_source* src, uint32_t steps_done,
                        void* workspace_dev, size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    return enqueue_sched_finish(c, (int)steps_done);
}

/* ---- one-launch-per-step form of the data-parallel loop.  `buf` selects one of the two slot buffers
 *      (alternate it per prepared batch so that the previous step's slot survives a batch boundary). */
int d3p_dpvi_logreg_prepare_buf(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                                const d3p_dpsvi_state* state, const d3p_batch_source* src, uint32_t num_steps, int buf,
                                void* workspace_dev, size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(num_steps >= 1 && num_steps <= D3P_STEP_BATCH, "d3p_dpvi_logreg_prepare_buf: 1 <= num_steps <= 32");
    D3P_REQUIRE(buf == 0 || buf == 1, "d3p_dpvi_logreg_prepare_buf: buf must be 0 or 1");
    Ctx cb = c;
    if (buf) {
        cb.ws = c.ws2;
        cb.ws.partials = c.ws.partials;
        cb.ws.acc = c.ws.acc;
        cb.ws.stamps = c.ws.stamps;
    }
    return enqueue_batch_prep(cb, (int)num_steps);
}

int d3p_dpvi_logreg_acc_layout(const d3p_logreg_model* model, const d3p_batch_source* src, size_t* offset_bytes,
                               size_t* words_per_buffer)
{
    D3P_REQUIRE(model && src && offset_bytes && words_per_buffer, "d3p_dpvi_logreg_acc_layout: null pointer");
    Workspace ws;
    carve(model, src, (char*)nullptr + 256, &ws);  // offsets relative to a fake base
    *offset_bytes = (size_t)((char*)ws.acc - ((char*)nullptr + 256));
    *words_per_buffer = (size_t)D3P_ACC_R * (2 * ((size_t)model->d + (model->intercept ? 1 : 0)) + 2);
    return D3P_OK;
}

int d3p_dpvi_logreg_acc_reset(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                              const d3p_dpsvi_state* state, const d3p_batch_source* src, void* workspace_dev,
                              size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_HIP_TRY(hipMemsetAsync(c.ws.acc, 0, 3 * (size_t)D3P_ACC_R * (c.P + 2) * sizeof(long long), c.s));
    return D3P_OK;
}

/* Fused step `g` (0-based since acc_reset) using slot `t` of buffer `buf`: applies the pending update of step
 * g-1 (slot prev_t of prev_buf, sums in accumulator (g-1) % 3 -- all-reduced by the caller on N GPUs), then
 * accumulates this rank's clipped sums into accumulator g % 3.  flush_only: only apply the pending update. */
int d3p_dpvi_logreg_fused_step(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                               const d3p_dpsvi_state* state, const d3p_batch_source* src, uint32_t g, uint32_t t, int buf,
                               int have_prev, uint32_t prev_t, int prev_buf, const float* X_dev, const float* y_dev,
                               float* prev_loss_dev, int flush_only, void* workspace_dev, size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(flush_only || X_dev, "null data pointer");
    if (!flush_only)
        if (int rcm = validate_model(model, y_dev, "d3p_dpvi_logreg_fused_step")) return rcm;
    D3P_REQUIRE(t < D3P_STEP_BATCH && prev_t < D3P_STEP_BATCH, "slot index must be < 32");
    Ctx cb[2] = {c, c};
    cb[1].ws = c.ws2;
    cb[1].ws.partials = c.ws.partials;
    cb[1].ws.acc = c.ws.acc;
    cb[1].ws.stamps = c.ws.stamps;
    const Ctx& cur = cb[buf & 1];
    const StepSlot* prev_slot = have_prev ? cb[prev_buf & 1].ws.slots + prev_t : nullptr;
    const float* prev_noise = have_prev ? cb[prev_buf & 1].ws.noise + (size_t)prev_t * c.P : nullptr;
    return enqueue_fused_step(cur, (int)g, (int)t, prev_slot, prev_noise, X_dev, y_dev, prev_loss_dev, nullptr, 0, 0,
                              flush_only != 0);
}

int d3p_comm_unique_id(uint8_t* id_out, size_t id_bytes)
{
    D3P_REQUIRE(id_out && id_bytes >= sizeof(ncclUniqueId), "d3p_comm_unique_id: buffer of at least 128 bytes required");
    const RcclApi* api = rccl_api();
    if (!api) return fail(D3P_E_UNSUPPORTED, "d3p_comm_unique_id: librccl.so could not be loaded");
    ncclUniqueId id;
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(D3P_E_HIP, "ncclGetUniqueId: %s", api->GetErrorString(r));
    memcpy(id_out, &id, sizeof(id));
    return D3P_OK;
}

int d3p_comm_init(const uint8_t* id, size_t id_bytes, int32_t nranks, int32_t rank, void** comm_out)
{
    D3P_REQUIRE(id && comm_out && id_bytes >= sizeof(ncclUniqueId) && nranks >= 1 && rank >= 0 && rank < nranks,
                "d3p_comm_init: bad arguments");
    const RcclApi* api = rccl_api();
    if (!api) return fail(D3P_E_UNSUPPORTED, "d3p_comm_init: librccl.so could not be loaded");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = api->CommInitRank(&comm, nranks, uid, rank);
    if (r != ncclSuccess) return fail(D3P_E_HIP, "ncclCommInitRank: %s", api->GetErrorString(r));
    *comm_out = (void*)comm;
    return D3P_OK;
}

int d3p_comm_destroy(void* comm)
{
    if (!comm) return D3P_OK;
    const RcclApi* api = rccl_api();
    if (!api) return fail(D3P_E_UNSUPPORTED, "d3p_comm_destroy: librccl.so could not be loaded");
    const ncclResult_t r = api->CommDestroy((ncclComm_t)comm);
    if (r != ncclSuccess) return fail(D3P_E_HIP, "ncclCommDestroy: %s", api->GetErrorString(r));
    return D3P_OK;
}

int d3p_dpvi_logreg_run_dist(void* stream, void* comm, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                             const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev, const float* y_dev,
                             uint32_t num_steps, float* losses_dev, void* workspace_dev, size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(X_dev || src->row_lo == src->row_hi, "null data pointer");
    if (int rcm = validate_model(model, y_dev ? (const void*)y_dev : (src->row_lo == src->row_hi ? (const void*)model : nullptr),
                                 "d3p_dpvi_logreg_run_dist"))
        return rcm;
    D3P_REQUIRE(src->kind != D3P_BATCH_EXPLICIT, "d3p_dpvi_logreg_run_dist: needs an on-device sampler (Feistel or Poisson)");
    if (comm && !rccl_api()) return fail(D3P_E_UNSUPPORTED, "d3p_dpvi_logreg_run_dist: librccl.so could not be loaded");
    if ((rc = enqueue_sched_init(c))) return rc;
    if ((rc = run_fused_steps(c, X_dev, y_dev, num_steps, losses_dev, (ncclComm_t)comm))) return rc;
    return enqueue_sched_finish(c, (int)num_steps);
}

int d3p_dpvi_logreg_run(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                        const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev,
                        const float* y_dev, uint32_t num_steps, float* losses_dev, void* workspace_dev,
                        size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(X_dev, "null data pointer");
    if (int rcm = validate_model(model, y_dev, "d3p_dpvi_logreg_run")) return rcm;
    D3P_REQUIRE(src->row_lo == 0 && src->row_hi == src->n_rows, "d3p_dpvi_logreg_run is the single-GPU path");
    if ((rc = enqueue_sched_init(c))) return rc;
    if (use_carrier(c)) return run_pipelined(c, X_dev, y_dev, (int)num_steps, losses_dev);
    if (use_fused_step(c)) {
        if ((rc = run_fused_steps(c, X_dev, y_dev, num_steps, losses_dev))) return rc;
        return enqueue_sched_finish(c, (int)num_steps);
    }
    // The serial ChaCha key chain of batch b+1 (one wavefront, ~1.8 us per step) runs on an auxiliary stream
    // while the update steps of batch b run on `stream`; the two streams meet once per batch.  (Putting the
    // sampler there too made things slower: its workgroups delay the dispatch of the main kernel.)
    // Measured: even this costs more than it hides (19.3-20.1 vs 17.6 us/step), so it is opt-in; by default the
    // chain step of the next batch rides in every k_finalize launch as one extra workgroup instead.
    const bool pipelined = num_steps > D3P_STEP_BATCH && getenv("D3P_AUX_STREAM") != nullptr;
    const bool piggyback = !pipelined && getenv("D3P_NO_CHAIN_PIGGYBACK") == nullptr;
    hipStream_t aux = nullptr;
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_init = nullptr;
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
    cb[1].ws.partials = c.ws.partials;
    Ctx ca[2] = {cb[0], cb[1]};  // the same, enqueuing on the auxiliary stream
    if (pipelined) ca[0].s = ca[1].s = aux;
    if (num_steps > 0) {
        if ((rc = enqueue_chain(ca[0], batch_len(0)))) { cleanup(); return rc; }
        if (pipelined) {
            D3P_TRY_CLEAN(hipEventRecord(ev_ready[0], aux));
            D3P_TRY_CLEAN(hipStreamWaitEvent(c.s, ev_ready[0], 0));
        }
        if ((rc = enqueue_sampler(cb[0], batch_len(0)))) { cleanup(); return rc; }
    }
    for (uint32_t b = 0; b < n_batches; ++b) {
        const bool dbuf = pipelined || piggyback;
        const int cur = dbuf ? (int)(b & 1) : 0, nxt = dbuf ? (cur ^ 1) : 0;
        const int K_next = (b + 1 < n_batches) ? batch_len(b + 1) : 0;
        if (pipelined && b + 1 < n_batches) {
            if (b >= 1) D3P_TRY_CLEAN(hipStreamWaitEvent(aux, ev_done[nxt], 0));  // buffer `nxt` consumed
            if ((rc = enqueue_chain(ca[nxt], batch_len(b + 1)))) { cleanup(); return rc; }
            D3P_TRY_CLEAN(hipEventRecord(ev_ready[nxt], aux));
        }
        const int K = batch_len(b);
        for (int t = 0; t < K; ++t) {
            if ((rc = enqueue_main(cb[cur], t, X_dev, y_dev, nullptr, false))) { cleanup(); return rc; }
            StepSlot* cslot = (piggyback && t < K_next) ? cb[nxt].ws.slots + t : nullptr;
            if ((rc = enqueue_finalize(cb[cur], t, c.ws.partials, c.g.blocks,
                                       losses_dev ? losses_dev + (size_t)b * D3P_STEP_BATCH + t : nullptr, nullptr,
                                       cslot, t, t == K_next - 1))) {
                cleanup();
                return rc;
            }
        }
        if (b + 1 < n_batches) {
            if (pipelined) {
                D3P_TRY_CLEAN(hipEventRecord(ev_done[cur], c.s));
                D3P_TRY_CLEAN(hipStreamWaitEvent(c.s, ev_ready[nxt], 0));
            } else if (!piggyback && (rc = enqueue_chain(cb[0], batch_len(b + 1)))) {
                return rc;
            }
            if ((rc = enqueue_sampler(cb[nxt], batch_len(b + 1)))) { cleanup(); return rc; }
        }
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
    D3P_REQUIRE(X_dev && avg_us && reps >= 1, "bad arguments");
    if (int rcm = validate_model(model, y_dev, "d3p_dpvi_logreg_time_main_kernel")) return rcm;
    if ((rc = enqueue_sched_init(c))) return rc;
    const bool staged = use_carrier(c);
    if (staged) {
        for (int j = -D3P_LOOKAHEAD; j < 0; ++j)
            if ((rc = enqueue_carrier(c, j, 1, nullptr))) return rc;
    } else if ((rc = enqueue_batch_prep(c, 1))) {
        return rc;
    }
    const bool fused = !staged && use_fused_step(c);
    Ctx ct = c;  // the fused step applies an update in its prologue: let it work on a scratch copy of the state
    d3p_dpsvi_state st_scratch = *c.st;
    if (fused) {
        const size_t P = (size_t)c.P;
        D3P_HIP_TRY(hipMemcpyAsync(c.ws.scratch_state, c.st->params, P * sizeof(float), hipMemcpyDeviceToDevice, c.s));
        D3P_HIP_TRY(hipMemcpyAsync(c.ws.scratch_state + P, c.st->adam_m, P * sizeof(float), hipMemcpyDeviceToDevice, c.s));
        D3P_HIP_TRY(hipMemcpyAsync(c.ws.scratch_state + 2 * P, c.st->adam_v, P * sizeof(float), hipMemcpyDeviceToDevice, c.s));
        D3P_HIP_TRY(hipMemsetAsync(c.ws.acc, 0, 3 * (size_t)D3P_ACC_R * (P + 2) * sizeof(long long), c.s));
        st_scratch.params = c.ws.scratch_state;
        st_scratch.adam_m = c.ws.scratch_state + P;
        st_scratch.adam_v = c.ws.scratch_state + 2 * P;
        st_scratch.step = reinterpret_cast<int32_t*>(c.ws.scratch_state + 3 * P);
        ct.st = &st_scratch;
    }
    d3p_batch_source src_scratch = *c.src;
    if (fused) {
        src_scratch.batch_index = reinterpret_cast<uint32_t*>(c.ws.scratch_state + 3 * c.P + 1);
        ct.src = &src_scratch;
    }
    auto launch = [&](bool stamps, hipEvent_t a0, hipEvent_t a1) {
        if (fused)  // same launch as step 1 of a run: previous step = slot 0, key-chain workgroup piggy-backed
            return enqueue_fused_step(ct, 1, 0, c.ws.slots, c.ws.noise, X_dev, y_dev, nullptr, c.ws2.slots, 0, 0, false, stamps,
                                      a0, a1);
        return staged ? enqueue_main_staged(c, 0, X_dev, y_dev, stamps, a0, a1)
                      : enqueue_main(c, 0, X_dev, y_dev, nullptr, stamps, a0, a1);
    };
    if (fused)  // prime the accumulator the timed launches read as "previous step" with real sums
        if ((rc = enqueue_fused_step(ct, 0, 0, nullptr, nullptr, X_dev, y_dev, nullptr, nullptr, 0, 0, false))) return rc;
    hipEvent_t e0, e1;
    D3P_HIP_TRY(hipEventCreate(&e0));
    D3P_HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i)  // warm-up
        if ((rc = launch(false, nullptr, nullptr))) return rc;
    const uint32_t nb = c.g.blocks;
    unsigned long long* host = (unsigned long long*)malloc((size_t)D3P_MAIN_MAX_BLOCKS * 2 * sizeof(unsigned long long));
    if (!host) return fail(D3P_E_HIP, "out of host memory");
    double dev_us = 0.0, ev_ms = 0.0;
    for (int i = 0; i < reps; ++i) {
        rc = launch(true, e0, e1);
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
