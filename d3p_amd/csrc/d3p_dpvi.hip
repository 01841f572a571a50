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
// One launch per step (default): the MODE-2 step kernel accumulates the clipped sums with 64-bit fixed-point atomics
// and the NEXT launch's prologue applies mean / noise / Adam (d3p_logreg_kernel.h), so k_finalize disappears from the
// per-step path; the key-chain step of the next batch rides along as one extra workgroup.
#include "d3p_logreg_kernel.h"
#include "d3p_logreg_persist.h"
#include "d3p_logreg_chain.h"
#include "d3p_logreg_wide.h"
#include "d3p_ipc_arena.h"

#include <dlfcn.h>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>
#include <rccl/rccl.h>  // types only: the entry points are resolved at run time from the RCCL torch has loaded

// steps prepared (key chain, sampler, noise) and chained into one launch at a time: measured 32 / 64 / 128 ->
// 87.4 / 89.4 / 91.5 k steps/s on the headline config (fewer launch boundaries and sampler launches per step)
#define D3P_STEP_BATCH 128

namespace d3p {

struct Workspace {
    Sched* sched;
    StepSlot* slots;  // D3P_STEP_BATCH
    float* pack;      // [loc | s | sg | q | lc] x D
    uint32_t* idx;    // D3P_STEP_BATCH x B
    uint32_t* skeys;  // D3P_STEP_BATCH x 2B
    uint32_t* plist;  // D3P_STEP_BATCH x B: dense owned-position lists
    float* noise;     // D3P_STEP_BATCH x P
    long long* acc;   // 3 x D3P_ACC_R x (P + 2) fixed-point accumulators of the one-launch step
    float* scratch_state;  // 3P + 4 floats: stand-in state for the timing entry point
    float* pp_state;       // 3P floats: second buffer of the ping-ponged optimiser state (one-launch-per-step path)
    uint32_t* chain_bar;   // chained form: (D3P_STEP_BATCH + 1) x D3P_BAR_WORDS arrival counters + 16 words (abort flag)
    uint32_t* xflags;      // in-launch exchange of a data-parallel run: D3P_STEP_BATCH x D3P_XCHG_WGS flags, 128 bytes apart
    long long* xsum;       // ... and the world's sums of step g in row g % 3 (3 x cols)
    unsigned long long* ll_state;  // data-parallel updater form: the optimiser state as tagged words -- parameters 2 x cols, m cols, v cols
    unsigned long long* own_mask;  // row-sharded Feistel batches: per step ceil(B / 64) ballot words "this rank holds the row of position p"
    uint32_t* pshard;      // sharded Poisson selection: [D3P_STEP_BATCH] the shard's selected counts | [D3P_STEP_BATCH] selected in the shards above
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
    p = take(3 * (size_t)D3P_ACC_R * D3P_ACC_COLS(P) * sizeof(long long)); if (ws) ws->acc = (long long*)p;
    p = take((3 * P + 4) * sizeof(float)); if (ws) ws->scratch_state = (float*)p;
    p = take(3 * P * sizeof(float)); if (ws) ws->pp_state = (float*)p;
    p = take(((size_t)(D3P_STEP_BATCH + 1) * D3P_BAR_WORDS + 16) * sizeof(uint32_t)); if (ws) ws->chain_bar = (uint32_t*)p;
    p = take((size_t)D3P_STEP_BATCH * D3P_XCHG_WGS * 32 * sizeof(uint32_t)); if (ws) ws->xflags = (uint32_t*)p;
    p = take(3 * (size_t)D3P_ACC_COLS(P) * sizeof(long long)); if (ws) ws->xsum = (long long*)p;
    p = take(4 * (size_t)D3P_ACC_COLS(P) * sizeof(unsigned long long)); if (ws) ws->ll_state = (unsigned long long*)p;
    p = take(2 * (size_t)D3P_STEP_BATCH * sizeof(uint32_t)); if (ws) ws->pshard = (uint32_t*)p;
    p = take(K * ((B + 63) / 64) * sizeof(unsigned long long)); if (ws) ws->own_mask = (unsigned long long*)p;
    p = take((size_t)D3P_MAIN_MAX_BLOCKS * (P + 2) * sizeof(float)); if (ws) ws->partials = (float*)p;
    p = take((size_t)D3P_MAIN_MAX_BLOCKS * 4 * sizeof(unsigned long long)); if (ws) ws->stamps = (unsigned long long*)p;  // >= 2 x 256 x 16 phase stamps
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
// state_key != nullptr: the schedule starts from the state (what k_sched_init would have stored) -- one launch less on the
// path of a single update() call.
__global__ void __launch_bounds__(64) k_chain(Sched* __restrict__ sched, StepSlot* __restrict__ slots, int K,
                                              const uint32_t* __restrict__ state_key = nullptr,
                                              const int32_t* __restrict__ adam_step = nullptr,
                                              const uint32_t* __restrict__ batch_index = nullptr)
{
    const int lane = threadIdx.x;
    uint32_t cur[16], child[16];
    load_key(state_key ? state_key : sched->key, cur);
    const int32_t adam0 = state_key ? *adam_step : sched->adam_i;
    const uint32_t batch0 = state_key ? (batch_index ? *batch_index : 0u) : sched->batch_i;
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
// Start and end of a device-resident run in ONE launch each (a run of a few steps is dominated by launch latencies: a
// 20-step run spent ~180 of its 350 us outside the step kernel).
// k_run_init (1 workgroup): wave 0 walks the serial key chain of the first batch -- split(key, 3) per step with the 4-lane
//   ChaCha block, three children in three quads (0.7 us per step instead of 1.9 for the one-lane form of k_chain) -- starting
//   from the state's key and counters; the other waves zero the fixed-point accumulators and the run's status words.
//   Replaces k_sched_init + k_chain + two memsets (and k_pack, which only the two-kernel path reads).
// k_flush (1 workgroup): applies the update that is still pending after the last step (the arithmetic of the step kernels'
//   prologue), leaves the state in the caller's arrays, stores the final key of the schedule into the state's other key slot
//   and copies the run's status words to pinned host memory.  Replaces the flush launch of the step kernel + k_sched_finish
//   + the device-to-host copy of d3p_dpvi_logreg_run_status.
// ------------------------------------------------------------------------------------------
struct RunInitCopy {  // d3p_dpvi_logreg_run_from: the optimiser state is copied from `src` into the run's arrays here
    const float* src[3];
    float* dst[3];
    int32_t* step_dst;
    uint32_t* batch_index_dst;  // the run's batch-index word (by-value start)
    uint32_t batch0;
    int n, by_value;
    uint32_t* bar;       // arrival counters of the first chained launch, zeroed here (nullable)
    uint32_t bar_words;
    // data-parallel updater form (d3p_logreg_chain.h): the optimiser state the run starts from as tagged words {fp32 | ll_tag} --
    // the parameters into row ll_tag & 1 of ll[0] (the other row is given the tag ll_tag - 1, which no step of this run waits
    // for: whatever an earlier run left there can never be taken for a parameter), m and v into ll[1], ll[2]
    unsigned long long* ll[3];  // nullable
    const float* ll_src[3];
    uint32_t ll_tag, ll_cols;   // cols = D3P_ACC_COLS(P): the row stride of ll[0]
};

__global__ void __launch_bounds__(256) k_run_init(const uint32_t* __restrict__ state_key, const int32_t* __restrict__ adam_step,
                                                  const uint32_t* __restrict__ batch_index, Sched* __restrict__ sched,
                                                  StepSlot* __restrict__ slots, int K, long long* __restrict__ acc, uint32_t acc_words,
                                                  uint32_t* __restrict__ status, RunInitCopy cp)
{
    const int tid = threadIdx.x;
    if (tid >= 64) {  // waves 1..3: zero the accumulators (3 x R x cols int64) and the status words; copy the state if asked to
        for (uint32_t i = tid - 64; i < acc_words; i += 192) acc[i] = 0;
        if (tid < 64 + 16) status[tid - 64] = 0u;
        if (cp.bar)
            for (uint32_t i = tid - 64; i < cp.bar_words; i += 192) cp.bar[i] = 0u;
        if (cp.dst[0]) {
            for (int j = 0; j < 3; ++j)
                for (int i = tid - 64; i < cp.n; i += 192) cp.dst[j][i] = cp.src[j][i];
            if (tid == 64) {
                *cp.step_dst = *adam_step;
                if (cp.by_value) *cp.batch_index_dst = cp.batch0;
            }
        }
        if (cp.ll[0]) {
            const unsigned long long t = (unsigned long long)cp.ll_tag << 32, told = (unsigned long long)(cp.ll_tag - 1u) << 32;
            for (uint32_t i = tid - 64; i < cp.ll_cols; i += 192) {
                const bool par = i < (uint32_t)cp.n;
                cp.ll[0][(size_t)(cp.ll_tag & 1u) * cp.ll_cols + i] = t | (par ? __float_as_uint(cp.ll_src[0][i]) : 0u);
                cp.ll[0][(size_t)((cp.ll_tag & 1u) ^ 1u) * cp.ll_cols + i] = told;
                cp.ll[1][i] = t | (par ? __float_as_uint(cp.ll_src[1][i]) : 0u);
                cp.ll[2][i] = t | (par ? __float_as_uint(cp.ll_src[2][i]) : 0u);
            }
        }
        return;
    }
    const int lane = tid, q = lane & 3, child = (lane >> 2) < 3 ? (lane >> 2) : 0;
    const int32_t adam0 = *adam_step;
    const uint32_t batch0 = cp.by_value ? cp.batch0 : (batch_index ? *batch_index : 0u);
    const uint32_t p0 = state_key[q];
    uint32_t p1 = state_key[4 + q], p2 = state_key[8 + q], p3 = state_key[12 + q];
    for (int t = 0; t < K; ++t) {
        uint32_t a, b;
        derive_child_quad_regs(p0, p1, p2, p3, (uint32_t)child, D3P_TAG_SPLIT, 0u, a, b);
        if (lane >= 4 && lane < 12) {  // gradient key (child 1), perturbation key (child 2)
            uint32_t* dst = lane < 8 ? slots[t].grad_key : slots[t].pert_key;
            dst[q] = p0;
            dst[4 + q] = a;
            dst[8 + q] = b;
            dst[12 + q] = 0u;
        } else if (lane == 12) {
            slots[t].adam_i = adam0 + t;
            slots[t].batch_i = batch0 + (uint32_t)t;
        }
        // the next state key is child 0 (lanes 0..3): every quad continues from it
        p1 = __shfl(a, q);
        p2 = __shfl(b, q);
        p3 = 0u;
    }
    if (lane < 4) {
        sched->key[q] = p0;
        sched->key[4 + q] = p1;
        sched->key[8 + q] = p2;
        sched->key[12 + q] = p3;
    }
    if (lane == 0) {
        sched->adam_i = adam0 + K;
        sched->batch_i = batch0 + (uint32_t)K;
    }
}


struct FlushArgs {
    const long long* acc_prev;  // nrep x cols: the local replicas, or the one row of world sums of a data-parallel chained run
    int nrep;
    const float* noise;         // P normals of the last step
    const StepSlot* slot;       // its slot
    const float* state_in[3];
    float* state_out[3];        // the caller's arrays
    float* loss_out;            // nullable
    int32_t* adam_step;
    uint32_t* batch_index;      // nullable
    const Sched* sched;
    uint32_t* key_out;          // the state's key slot after the run
    const uint32_t* status;
    unsigned long long* host_status;  // nullable: pinned host record {abort, nonfinite, tag}
    unsigned long long host_tag;
    int dbg_print;
    int P, B;
    float dp_scale, clip, obs_scale, lr, b1, b2, adam_eps;
    double inv_sg;
    // data-parallel updater form: nothing is pending (the last step's updater applied its update and reported its loss); the state
    // is unpacked from the tagged words {fp32 | ll_tag} (parameters: row ll_tag & 1) into the caller's arrays
    const unsigned long long* ll[3];  // nullable
    uint32_t ll_tag;
};

__global__ void __launch_bounds__(1024) k_flush(FlushArgs a)
{
    const int tid = threadIdx.x, PA = D3P_ACC_COLS(a.P);
    if (tid < 16) a.key_out[tid] = a.sched->key[tid];
    const uint32_t aborted = a.status[0];
    if (tid == 0 && aborted && a.dbg_print) {  // D3P_DBG=64: where the waits of the stopped launch stood
        printf("[d3p] run stopped, code %#x; earliest step a wait ran out at, by kind:", aborted);
        for (int k = 1; k < 8; ++k)
            if (a.status[8 + k]) printf(" kind %d: step %u;", k, 0x1000u - a.status[8 + k]);
        printf("\n");
    }
    // the run's status for the host: ONE 16-byte store {abort code | non-finite flag, tag} into THIS workspace's pinned slot
    auto status_to_host = [&](uint32_t code) {
        if (tid == 0 && a.host_status) {
            d3p_u32x4 rec;
            rec.x = code;
            rec.y = a.status[1];
            rec.z = (uint32_t)a.host_tag;
            rec.w = (uint32_t)(a.host_tag >> 32);
            *reinterpret_cast<d3p_u32x4*>(a.host_status) = rec;
        }
    };
    status_to_host(aborted);
    if (aborted) {  // the pending sums are incomplete: leave the state where the run stopped
        if (tid == 0 && a.loss_out) *a.loss_out = __builtin_nanf("");
        return;
    }
    if (a.ll[0]) {
        bool bad = false;
        for (int col = tid; col < a.P; col += blockDim.x) {
            const unsigned long long wx = a.ll[0][(size_t)(a.ll_tag & 1u) * PA + col], wm = a.ll[1][col], wv = a.ll[2][col];
            bad |= (uint32_t)(wx >> 32) != a.ll_tag || (uint32_t)(wm >> 32) != a.ll_tag || (uint32_t)(wv >> 32) != a.ll_tag;
            a.state_out[0][col] = __uint_as_float((uint32_t)wx);
            a.state_out[1][col] = __uint_as_float((uint32_t)wm);
            a.state_out[2][col] = __uint_as_float((uint32_t)wv);
        }
        // (cannot happen once the launches are complete: every word of the final epoch was stored by the last updater.  Reported as
        // a stopped run rather than handed on silently.  `bad` is per thread -- its own columns -- and only thread 0 reports and writes
        // the counters: the workgroup votes first.)
        bad = __syncthreads_or(bad) != 0;
        if (bad) status_to_host(abort_code(D3P_ABORT_RELEASE, 0xfff, 4u));
        // The run's counters: ONE writer, in the run's last kernel, from the schedule (k_run_init / the key-chain links leave the
        // counts after the run's last prepared step there).  The step launches do not store them: a word that a different workgroup
        // plain-stores every step keeps the value of whichever XCD's L2 is written back last (svi.py:379-393, :432-434: the
        // returned state's optimiser step advances once per update).
        if (tid == 0 && !bad) {
            *a.adam_step = a.sched->adam_i;
            if (a.batch_index) *a.batch_index = a.sched->batch_i;
        }
        return;
    }
    long long nll = 0;
    for (int r = 0; r < a.nrep; ++r) nll += a.acc_prev[(size_t)r * PA + a.P + 1];
    const float n = nll >= (1ll << 40) ? __builtin_nanf("") : (float)nll;
    const float Bf = (float)a.B;
    const float factor = (n == 0.0f) ? 0.0f : Bf / n;
    const float inv_B = 1.0f / Bf, inv_bc1 = 1.0f / a.slot->bc1, inv_bc2 = 1.0f / a.slot->bc2;
    const float noise_scale = a.dp_scale * (a.clip / n), out_scale = a.obs_scale * factor;
    // (the parameters the step ran with, asked by the reporter of an EMPTY batch's loss below: state_in IS state_out after an odd number
    //  of launches, so every thread notes what it reads before it stores -- thread 0 reading the columns afterwards saw updated ones)
    int x_bad = 0;
    for (int col = tid; col < a.P; col += blockDim.x) {
        long long sll = 0;
        for (int r = 0; r < a.nrep; ++r) sll += a.acc_prev[(size_t)r * PA + col];
        const float tot = (float)((double)sll * a.inv_sg);
        const float g = __fmaf_rn(a.noise[col], noise_scale, tot * inv_B) * out_scale;
        const float x0 = a.state_in[0][col];
        x_bad |= !(fabsf(x0) <= 3.402823466e38f);
        const float mm = (1.0f - a.b1) * g + a.b1 * a.state_in[1][col];
        const float vv = (1.0f - a.b2) * g * g + a.b2 * a.state_in[2][col];
        const float xx = x0 - a.lr * (mm * inv_bc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv * inv_bc2) + a.adam_eps);
        a.state_out[0][col] = xx;
        a.state_out[1][col] = mm;
        a.state_out[2][col] = vv;
    }
    x_bad = __syncthreads_or(x_bad);
    if (tid == 0) {
        long long lll = 0, lhh = 0;
        for (int r = 0; r < a.nrep; ++r) {
            lll += a.acc_prev[(size_t)r * PA + a.P];
            lhh += a.acc_prev[(size_t)r * PA + a.P + 2];
        }
        if (a.loss_out) {
            float lv = ((float)loss_join(lhh, lll) / Bf) * a.obs_scale * factor;
            if (n == 0.0f) lv = x_bad ? __builtin_nanf("") : 0.0f;   // empty_batch_loss (d3p_device.h) on the values noted above
            *a.loss_out = lv;
        }
        *a.adam_step = a.slot->adam_i + 1;
        if (a.batch_index) *a.batch_index = a.slot->batch_i + 1u;
    }
}

// Pinned host records of the runs' status words (written by k_flush as ONE 16-byte store): spare d3p_dpvi_logreg_run_status its
// copy.  One slot per WORKSPACE (keyed by the address of its status words) -- two in-process ranks on two streams
// (XchgComm.local_group) each report into their own slot -- and every claim stamps the slot's expected tag with a fresh sequence
// number, so a record is only ever read as the result of the LAST flush enqueued for that workspace: not a record of an earlier run
// (a run form that ends without k_flush invalidates the slot and the reader copies the device words), not one of a workspace that
// lived at the same address before.
struct StatusSlots {
    static constexpr int N = 256;
    struct Entry { int slot; unsigned long long seq; bool valid; };
    std::mutex mu;
    unsigned long long* table = nullptr;   // N x 2 words {abort | nonfinite << 32, tag}
    bool tried = false;
    unsigned long long seq = 0;
    int next = 0;
    uintptr_t owner[N] = {};
    std::unordered_map<uintptr_t, Entry> by_ws;
    unsigned long long* base()
    {
        if (!tried) {
            tried = true;
            void* q = nullptr;
            if (hipHostMalloc(&q, (size_t)N * 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); q = nullptr; }
            if (q) memset(q, 0, (size_t)N * 16);
            table = (unsigned long long*)q;
        }
        return table;
    }
};
static StatusSlots& status_slots() { static StatusSlots s; return s; }

// the slot the run's k_flush will report into (nullptr: none -- the reader copies) and the tag it must carry
static unsigned long long* status_slot_claim(const uint32_t* status_words, unsigned long long* tag_out)
{
    StatusSlots& S = status_slots();
    std::lock_guard<std::mutex> lk(S.mu);
    unsigned long long* t = S.base();
    *tag_out = 0ull;
    if (!t) return nullptr;
    const uintptr_t key = (uintptr_t)status_words;
    auto it = S.by_ws.find(key);
    if (it == S.by_ws.end()) {
        const int slot = S.next;
        S.next = (S.next + 1) % StatusSlots::N;
        if (S.owner[slot]) S.by_ws.erase(S.owner[slot]);   // (a pending flush of the evicted workspace carries ITS tag: never taken for ours)
        S.owner[slot] = key;
        it = S.by_ws.emplace(key, StatusSlots::Entry{slot, 0ull, false}).first;
    }
    it->second.seq = ++S.seq;
    it->second.valid = true;
    *tag_out = it->second.seq;
    return t + 2 * (size_t)it->second.slot;
}

// a run form that ends without k_flush: whatever the slot holds is not this run's
static void status_slot_invalidate(const uint32_t* status_words)
{
    StatusSlots& S = status_slots();
    std::lock_guard<std::mutex> lk(S.mu);
    auto it = S.by_ws.find((uintptr_t)status_words);
    if (it != S.by_ws.end()) it->second.valid = false;
}

// after the stream is idle: the record of the workspace's last run, if its flush reported one
static bool status_slot_read(const uint32_t* status_words, uint32_t* aborted, uint32_t* nonfinite)
{
    StatusSlots& S = status_slots();
    std::lock_guard<std::mutex> lk(S.mu);
    auto it = S.by_ws.find((uintptr_t)status_words);
    if (it == S.by_ws.end() || !it->second.valid || !S.table) return false;
    const volatile unsigned long long* rec = S.table + 2 * (size_t)it->second.slot;
    const unsigned long long w0 = rec[0], w1 = rec[1];
    if (w1 != it->second.seq) return false;
    *aborted = (uint32_t)w0;
    *nonfinite = (uint32_t)(w0 >> 32);
    return true;
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
    unsigned long long* own_mask;  // nullable: per step ceil(B / 64) words, bit l of word w = "position 64 w + l is this rank's" (k_owned_pack)
    int ppt;                  // batch positions per thread (1, or 4 for the large padded batches of a row-sharded rank: the per-step prefix --
                              // two dependent ChaCha blocks -- is then made by a quarter of the blocks)
};

// The sampler work of x-block bx (of gx) for step t of the batch.
// (every ChaCha block of the sampler runs on a quad of lanes -- chacha20_block_quad, ~310 instructions instead of ~970 for the
// one-lane form: the sampler is a chain of three dependent derivations per block, i.e. latency, and sits on the start-up path
// of every run)
__device__ __forceinline__ void sampler_block(const SamplerArgs& a, int bx, int gx, int t)
{
    __shared__ uint32_t sh_key[2][16], sh_jax[2], sh_rc[32];
    const int tid = threadIdx.x, quad = tid >> 2, q = tid & 3;
    StepSlot* slot = a.slots + t;
    const bool aux = bx == gx - 1;
    const uint32_t slot_batch_i = slot->batch_i;
    auto store_child = [&](uint32_t* dst, const uint32_t* parent, uint32_t ka, uint32_t kb) {  // child key = constants | block words 0..7 | 0
        dst[q] = parent[q];
        dst[4 + q] = ka;
        dst[8 + q] = kb;
        dst[12 + q] = 0u;
    };
    if (!aux) {
        if (quad == 0) {  // convert_to_jax_rng_key(gradient_key) (svi.py:259; random/__init__.py:155): words 0, 1 of block 0
            uint32_t ka, kb, kc, kd;
            keystream_block_quad(slot->grad_key, 0u, ka, kb, kc, kd);
            if (q < 2) sh_jax[q] = ka;
        } else if (quad == 16 && a.kind == D3P_BATCH_FEISTEL) {  // fold_in(batchifier_state, i) (minibatch.py:230)
            uint32_t ka, kb;
            derive_child_quad(a.batch_key, 0u, D3P_TAG_FOLD, slot_batch_i, ka, kb);
            store_child(sh_key[0], a.batch_key, ka, kb);
        }
        __syncthreads();
        if ((quad == 16 || quad == 32) && a.kind == D3P_BATCH_FEISTEL) {  // round constants: keystream blocks 0, 1 (util.py:240-246)
            const uint32_t b = quad == 16 ? 0u : 1u;
            uint32_t w[4];
            keystream_block_quad(sh_key[0], b, w[0], w[1], w[2], w[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g = 16 * (int)b + 4 * i + q;
                if (g < 30) sh_rc[g] = (g % 3 == 0) ? (w[i] | 1u) : w[i];
            }
        }
        __syncthreads();
        for (int j = 0; j < a.ppt; ++j) {
        const uint32_t p = ((uint32_t)bx * (uint32_t)a.ppt + (uint32_t)j) * blockDim.x + tid;
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
            if (a.own_mask) {  // (the wave's positions 64 w .. 64 w + 63: lanes beyond B are not active and read as 0)
                const unsigned long long bal = __ballot(owned);
                if ((tid & 63) == 0) a.own_mask[(size_t)t * ((a.B + 63) / 64) + (p >> 6)] = bal;
            }
        }
        }
        return;
    }
    // ---- aux block: per-site keys split(perturbation_key, 2) (svi.py:491), then the normals
    if (quad < 2) {
        uint32_t ka, kb;
        derive_child_quad(slot->pert_key, (uint32_t)quad, D3P_TAG_SPLIT, 0u, ka, kb);
        store_child(sh_key[quad], slot->pert_key, ka, kb);
    } else if (quad == 16) {
        uint32_t ka, kb, kc, kd;
        keystream_block_quad(slot->grad_key, 0u, ka, kb, kc, kd);
        if (q < 2) slot->jax_key[q] = ka;
    } else if (quad == 32) {
        if (a.batch_key) {
            uint32_t ka, kb;
            derive_child_quad(a.batch_key, 0u, D3P_TAG_FOLD, slot_batch_i, ka, kb);
            store_child(slot->batch_key, a.batch_key, ka, kb);
        }
        if (q == 0) {
            if (a.kind != D3P_BATCH_POISSON) {  // POISSON: written by the select kernels
                slot->counts[0] = a.B;
                slot->counts[1] = a.B;
            }
            const float ip1 = (float)(slot->adam_i + 1);
            slot->bc1 = 1.0f - powf(a.b1, ip1);
            slot->bc2 = 1.0f - powf(a.b2, ip1);
        }
    }
    __syncthreads();
    // noise[site * D + e] = normal(site_key[site])[e]  (svi.py:487): ChaCha block e/16, word e%16; one quad per block
    const int blocks_per_site = (a.D + 15) / 16;
    for (int j = quad; j < 2 * blocks_per_site; j += (int)blockDim.x / 4) {
        const int site = j / blocks_per_site, b = j % blocks_per_site;
        uint32_t w[4];
        keystream_block_quad(sh_key[site], (uint32_t)b, w[0], w[1], w[2], w[3]);
        float* dst = a.noise + (size_t)t * 2 * a.D + (size_t)site * a.D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 16 * b + 4 * i + q;
            if (e < a.D) dst[e] = bits_to_normal(w[i]);
        }
    }
}

__global__ void __launch_bounds__(256) k_sampler(SamplerArgs a) { sampler_block(a, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y); }

// ------------------------------------------------------------------------------------------
// dense list of the batch positions a rank processes: valid (p < counts[1]) and row in [row_lo, row_hi).
// One workgroup per step; positions stay in ascending order (deterministic).  Needed for Poisson batches
// (padding) and for row-sharded multi-GPU runs, where only ~B / world of the global positions are owned.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_owned_list(StepSlot* __restrict__ slots, const uint32_t* __restrict__ idx, uint32_t B, uint64_t row_lo, uint64_t row_hi,
             uint32_t* __restrict__ plist)
{
    // Ordered compaction of the valid, owned positions of one step (blockIdx.y).  Positions are taken in rounds of 1024
    // consecutive ones (thread t <-> position 1024 r + t: coalesced index reads), 32 rounds per pass; a thread keeps its 32
    // flags in a bit mask, wave ballots give the counts per (round, wave), one wave scans them, and the second half
    // writes every owned position to base + count of owned positions before it.
    __shared__ uint32_t cnt[32 * 16 + 1];
    __shared__ uint32_t base_s;
    const int t = blockIdx.y;
    const uint32_t* ix = idx + (size_t)t * B;
    uint32_t* out = plist + (size_t)t * B;
    const uint32_t n_valid = slots[t].counts[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0u;
    __syncthreads();
    for (uint32_t p0 = 0; p0 < B; p0 += 32u * 1024u) {
        uint32_t mask = 0u;
#pragma unroll 4
        for (int r = 0; r < 32; ++r) {
            const uint32_t p = p0 + 1024u * (uint32_t)r + threadIdx.x;
            bool own = false;
            if (p < B && p < n_valid) {
                const uint64_t row = ix[p];
                own = row >= row_lo && row < row_hi;
            }
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(own);
            if (lane == 0) cnt[16 * r + wave] = (uint32_t)__popcll(bal);
            mask |= (own ? 1u : 0u) << r;
        }
        __syncthreads();
        if (wave == 0) {  // exclusive scan of the 512 counts: 8 consecutive entries per lane, then a wave prefix
            uint32_t v[8], s = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = cnt[8 * lane + j]; s += v[j]; }
            uint32_t inc = s;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            uint32_t run = base_s + inc - s;
#pragma unroll
            for (int j = 0; j < 8; ++j) { cnt[8 * lane + j] = run; run += v[j]; }
            if (lane == 63) cnt[512] = run;  // owned positions so far, including this pass
        }
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < 32; ++r) {
            const bool own = (mask >> r) & 1u;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(own);
            if (own) {
                const uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                out[cnt[16 * r + wave] + before] = p0 + 1024u * (uint32_t)r + threadIdx.x;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s = cnt[512];
        __syncthreads();
    }
    if (threadIdx.x == 0) slots[t].n_owned = base_s;
}

// The same list from the sampler's ballot words (row-sharded Feistel batches: every position of the padded batch is valid): thread w
// takes word w -- an exclusive scan of the popcounts gives its offset, the set bits its positions, ascending.  One workgroup per
// step, B / 64 threads' worth of work: 3 us where k_owned_list's 32 rounds over the B indices took 17 (a short run's start-up).
__global__ void __launch_bounds__(1024)
k_owned_pack(StepSlot* __restrict__ slots, const unsigned long long* __restrict__ own_mask, uint32_t B, uint32_t* __restrict__ plist)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t base_s;
    const int t = blockIdx.y;
    const uint32_t nwords = (B + 63u) / 64u;
    const unsigned long long* mk = own_mask + (size_t)t * nwords;
    uint32_t* out = plist + (size_t)t * B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0u;
    __syncthreads();
    for (uint32_t w0 = 0; w0 < nwords; w0 += 1024u) {
        const uint32_t w = w0 + threadIdx.x;
        const unsigned long long m = w < nwords ? mk[w] : 0ull;
        const uint32_t c = (uint32_t)__popcll(m);
        uint32_t inc = c;   // inclusive prefix over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint32_t before = base_s + inc - c;
        for (int v = 0; v < wave; ++v) before += wave_tot[v];
        unsigned long long r = m;
        while (r) {
            const int b = __builtin_ctzll(r);
            out[before++] = 64u * w + (uint32_t)b;
            r &= r - 1ull;
        }
        __syncthreads();
        if (threadIdx.x == 1023) base_s = before;   // (the last thread's end = everything so far)
        __syncthreads();
    }
    if (threadIdx.x == 0) slots[t].n_owned = base_s;
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
// host-side launch logic
// ------------------------------------------------------------------------------------------
static int validate(const d3p_logreg_model* m, const d3p_dpsvi_hyper* h, const d3p_dpsvi_state* st,
                    const d3p_batch_source* src)
{
    D3P_REQUIRE(m && h && st && src, "null argument struct");
    D3P_REQUIRE(m->d >= 1, "model.d must be >= 1");
    D3P_REQUIRE(m->prior_w > 0.f && m->prior_b > 0.f, "prior scales must be positive");
    D3P_REQUIRE(m->inv_obs > 0.f, "inv_obs must be positive");
    // svi.py:119-120 rejects C == 0; a negative C would turn 1 / max(1, norm / C) into "no clipping at all" there and
    // flip every clipped gradient here, so C <= 0 is refused altogether (as the mixture-model and VAE entry points do)
    D3P_REQUIRE(h->clip > 0.f || std::isnan(h->clip), "The clipping threshold must be greater than 0.");
    D3P_REQUIRE(std::isfinite(h->clip), "clipping_threshold must be finite!");       // svi.py:187-188
    D3P_REQUIRE(std::isfinite(h->dp_scale) && h->dp_scale >= 0.f, "dp_scale must be finite and >= 0");
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

// the run's sticky status words (StepFuse::status): behind the arrival counters of the chained launch
static inline uint32_t* run_status_words(const Workspace& ws) { return ws.chain_bar + (size_t)(D3P_STEP_BATCH + 1) * D3P_BAR_WORDS; }
// D3P_DBG: developer ablation / phase-stamp switches of the step kernel (0 in production); read once per process
static int dev_dbg_flags()
{
    static const int v = [] { const char* e = getenv("D3P_DBG"); return e ? atoi(e) : 0; }();
    return v;
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
    uint64_t items_expected;  // batch positions this rank expects to process per step (B, or its share of a row-sharded batch)
    const d3p_dpsvi_state* from = nullptr;  // d3p_dpvi_logreg_run_from: the run starts from THIS state (read only), st receives the result
    bool batch0_by_value = false;           // ... and from batch index batch0 instead of *src->batch_index
    uint32_t batch0 = 0;
    Workspace ws2;  // second slot buffer (slots / idx / skeys / noise) for the pipelined run loop
};

// zeroed status words for a run form that does not end with k_flush: the reader must not take an earlier run's pinned record
static hipError_t reset_status_words(const Ctx& c)
{
    status_slot_invalidate(run_status_words(c.ws));
    return hipMemsetAsync(run_status_words(c.ws), 0, 16 * sizeof(uint32_t), c.s);
}

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
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, c.s, c.ws.sched, c.ws.slots, K, (const uint32_t*)nullptr,
                       (const int32_t*)nullptr, (const uint32_t*)nullptr);
    return check_launch("k_chain");
}

// enqueue_sched_init + enqueue_chain in two launches instead of three (the schedule is initialised inside k_chain)
static int enqueue_sched_init_chain(const Ctx& c, int K)
{
    const bool sampled = c.src->kind != D3P_BATCH_EXPLICIT;
    hipLaunchKernelGGL(k_pack, dim3(cdiv(c.D, 256)), dim3(256), 0, c.s, *c.m, (const float*)c.st->params, c.ws.pack);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, c.s, c.ws.sched, c.ws.slots, K,
                       (const uint32_t*)(c.st->rng_key + 16 * (c.st->key_slot & 1)), (const int32_t*)c.st->step,
                       sampled ? (const uint32_t*)c.src->batch_index : nullptr);
    return check_launch("k_chain");
}

struct Xchg;
static int enqueue_sampler(const Ctx& c, int K, Xchg* xchg = nullptr);
static int enqueue_xchg_poisson_counts(hipStream_t s, Xchg* x, const uint32_t* local, int K, uint32_t cutoff, int suppress, uint32_t* counts,
                                       size_t counts_stride_words, uint32_t* above_out, uint32_t* n_owned, size_t n_owned_stride_words,
                                       uint32_t* status);

// key chain + sampler for the next K steps
static int enqueue_batch_prep(const Ctx& c, int K)
{
    int rc = enqueue_chain(c, K);
    if (rc) return rc;
    return enqueue_sampler(c, K);
}

static void fill_sampler_args(const Ctx& c, SamplerArgs* out)
{
    SamplerArgs& sa = *out;
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
    sa.own_mask = (c.src->kind == D3P_BATCH_FEISTEL && need_owned_list(c.src)) ? c.ws.own_mask : nullptr;
    // (a rank of a row-sharded job derives sample keys for its own ~ B / world positions only: four positions per thread cost little
    // and the step's prefix is made by a quarter of the blocks -- 22 -> ~ 12 us for 20 steps at B = 32768)
    sa.ppt = (need_owned_list(c.src) && c.src->kind == D3P_BATCH_FEISTEL && c.src->B >= 16384u) ? 4 : 1;
}

static int enqueue_sampler(const Ctx& c, int K, Xchg* xchg)
{
    SamplerArgs sa;
    fill_sampler_args(c, &sa);
    hipLaunchKernelGGL(k_sampler, dim3(cdiv(c.src->B, 256u * (uint32_t)sa.ppt) + 1, K), dim3(256), 0, c.s, sa);
    int rc = check_launch("k_sampler");
    if (rc) return rc;
    static const bool full_mask = getenv("D3P_POISSON_FULL_MASK") != nullptr;   // developer switch: every rank makes the whole mask (round 3)
    if (c.src->kind == D3P_BATCH_POISSON && xchg && !full_mask) {
        // Data-parallel run with the one-shot exchange: the rank makes the Bernoulli mask of ITS rows only (1 / world of the ChaCha20
        // blocks); the shards' selected counts of the K steps travel in one tagged exchange (d3p_xchg_poisson_counts);
        // the rank then writes its valid selected rows at their global batch positions and its dense list of owned positions.
        // Bit for bit the rows and positions the whole mask gives (minibatch.py:29-39, :119-124).
        const size_t stride = sizeof(StepSlot) / sizeof(uint32_t);
        uint32_t* local = c.ws.pshard, *above = c.ws.pshard + D3P_STEP_BATCH;
        if ((rc = d3p_poisson_shard_flags((void*)c.s, 0, c.ws.slots[0].batch_key, stride, c.src->q, (uint32_t)c.src->n_rows, (uint32_t)c.src->row_lo,
                                          (uint32_t)c.src->row_hi, (uint32_t)K, local, c.ws.poisson_ws, c.ws.poisson_bytes)))
            return rc;
        if ((rc = enqueue_xchg_poisson_counts(c.s, xchg, local, K, c.src->B, c.src->suppress, c.ws.slots[0].counts, stride, above,
                                              &c.ws.slots[0].n_owned, stride, run_status_words(c.ws))))
            return rc;
        return d3p_poisson_shard_write((void*)c.s, (uint32_t)c.src->n_rows, (uint32_t)c.src->row_lo, (uint32_t)c.src->row_hi, c.src->B,
                                       c.ws.slots[0].counts, stride, above, c.ws.idx, c.ws.plist, c.src->B, (uint32_t)K, c.ws.poisson_ws,
                                       c.ws.poisson_bytes);
    }
    if (c.src->kind == D3P_BATCH_POISSON) {  // all K draws in one set of launches (blockIdx.y = step)
        rc = d3p_poisson_select_batch((void*)c.s, 0, c.ws.slots[0].batch_key, sizeof(StepSlot) / sizeof(uint32_t), c.src->q,
                                      (uint32_t)c.src->n_rows, c.src->B, c.src->suppress, c.ws.idx, c.src->B,
                                      c.ws.slots[0].counts, sizeof(StepSlot) / sizeof(uint32_t), (uint32_t)K,
                                      c.ws.poisson_ws, c.ws.poisson_bytes);
        if (rc) return rc;
    }
    if (need_owned_list(c.src) && c.src->kind == D3P_BATCH_FEISTEL) {  // from the sampler's ballot words (every padded position is valid)
        hipLaunchKernelGGL(k_owned_pack, dim3(1, K), dim3(1024), 0, c.s, c.ws.slots, (const unsigned long long*)c.ws.own_mask, c.src->B, c.ws.plist);
        if ((rc = check_launch("k_owned_pack"))) return rc;
    } else if (need_owned_list(c.src)) {
        hipLaunchKernelGGL(k_owned_list, dim3(1, K), dim3(1024), 0, c.s, c.ws.slots, (const uint32_t*)c.ws.idx, c.src->B,
                           (uint64_t)c.src->row_lo, (uint64_t)c.src->row_hi, c.ws.plist);
        if ((rc = check_launch("k_owned_list"))) return rc;
    }
    return D3P_OK;
}

// The 16-wave kernels need more dynamic LDS than the default limit: the attribute is per function AND per device, so it is set once
// for every (kernel, current device) pair -- a process that drives a second GPU sets it there too -- and a failure is reported
// instead of surfacing as an unexplained launch error.
static int ensure_dynamic_lds(const void* fn, size_t bytes, const char* what)
{
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& d : done)
        if (d.first == fn && d.second == dev) return D3P_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(D3P_E_HIP, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize, %zu) on device %d: %s", what, bytes, dev, hipGetErrorString(e));
    }
    done.emplace_back(fn, dev);
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
    a.dbg = dev_dbg_flags();
    if (c.g.wide) {  // wide rows: column-chunked kernel, same partial-row output
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(k_logreg_wide<false>), 160 * 1024, "k_logreg_wide")) return rc_;
        if (e0)
            hipExtLaunchKernelGGL(k_logreg_wide<false>, dim3(c.g.blocks), dim3(64 * D3P_WIDE_W), c.g.lds, c.s, e0, e1, 0, a);
        else
            hipLaunchKernelGGL(k_logreg_wide<false>, dim3(c.g.blocks), dim3(64 * D3P_WIDE_W), c.g.lds, c.s, a);
        return check_launch("k_logreg_wide");
    }
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

// ------------------------------------------------------------------------------------------
// kernel timing of the run loops (d3p_dpvi_logreg_kernel_timing_*): event pairs around the step-kernel launches
// ------------------------------------------------------------------------------------------
struct KernelTiming {
    bool on = false;
    std::vector<hipEvent_t> ev;    // start, stop, start, stop, ... of the launches timed so far
    std::vector<hipEvent_t> pool;  // created when timing is switched on: event creation stays off the enqueue path of the timed runs
    uint32_t steps = 0;
};
static KernelTiming g_kt;

// a fresh (start, stop) pair for a launch that covers `steps` DP-VI steps; nullptrs when timing is off
static void timing_pair(int steps, hipEvent_t* e0, hipEvent_t* e1)
{
    *e0 = *e1 = nullptr;
    if (!g_kt.on) return;
    hipEvent_t a = nullptr, b = nullptr;
    if (g_kt.pool.size() >= 2) {
        a = g_kt.pool.back(); g_kt.pool.pop_back();
        b = g_kt.pool.back(); g_kt.pool.pop_back();
    } else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        if (a) (void)hipEventDestroy(a);
        (void)hipGetLastError();
        return;
    }
    g_kt.ev.push_back(a);
    g_kt.ev.push_back(b);
    g_kt.steps += (uint32_t)steps;
    *e0 = a;
    *e1 = b;
}

// ------------------------------------------------------------------------------------------
// one launch per step (MODE 2)
// ------------------------------------------------------------------------------------------
static bool use_fused_step(const Ctx& c)
{
    static const bool off = getenv("D3P_NO_FUSED_STEP") != nullptr;  // two-kernel steps (main + finalize), kept for comparison
    return !off && !c.g.wide;  // wide rows run as two-kernel steps with the column-chunked main kernel
}

static void fill_fuse_common(const Ctx& c, StepFuse* f, int g)
{
    const size_t PA = (size_t)D3P_ACC_COLS(c.P);
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
    f->status = run_status_words(c.ws);
    // gradient columns: every workgroup partial is bounded by 16 * C per step, sums by B * C -> 2^40 / C keeps
    // B up to 2^22 inside int64 with a resolution of C * 2^-40
    f->sg = 1099511627776.0 / (double)fabsf(c.h->clip);
    f->inv_sg = 1.0 / f->sg;
    // (the loss takes two columns, coarse and fine, at fixed scales: D3P_ACC_COLS in d3p_logreg_kernel.h)
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
    a.dbg = dev_dbg_flags();
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

// Chained form (MODE 3): the K steps of the prepared batch in ONE launch of K x (nw + 1) workgroups (see ChainFuse in
// d3p_logreg_kernel.h).  On by default for the single-GPU run loop; D3P_NO_CHAINED_STEPS=1 falls back to one launch per step.
// d3p_dpvi_logreg_set_run_form: 0 = automatic, 1 = one launch per step.  Per THREAD (a caller that sets it around one run -- DPSVI.run_steps'
// fallback -- cannot flip the form of a run another thread is enqueueing), and a run reads it ONCE, when it starts (run_fused_steps).
static thread_local int g_run_form = 0;
static bool use_chained_steps(const Ctx& c)
{
    static const bool off = getenv("D3P_NO_CHAINED_STEPS") != nullptr;
    return !off && g_run_form != 1 && c.src->kind != D3P_BATCH_EXPLICIT;
}

// Persistent form (MODE 4, d3p_logreg_persist.h): the same K steps in ONE launch of RESIDENT workgroups that loop over the
// steps and overlap the noise generation of step t + 1 with the arrival / update latency of step t.  Used when the batch
// geometry is the specialised one (d = 512, no intercept, softplus guide, one example per wave) and every workgroup of the
// launch is resident at once (one 1024-thread workgroup per CU); selected with D3P_PERSISTENT_STEPS=1.
static bool use_persistent_steps(const Ctx& c)
{
    static const bool off = getenv("D3P_PERSISTENT_STEPS") == nullptr;  // opt-in while it only ties the chained form
    if (off || !c.g.full || c.g.V != 4 || c.g.NK != 1 || c.g.W != D3P_PERSIST_W || c.D != D3P_PERSIST_D) return false;
    if (c.m->guide_transform != D3P_GUIDE_SOFTPLUS || c.m->family != D3P_FAMILY_LOGREG) return false;
    if ((uint64_t)c.g.blocks * D3P_PERSIST_W < (uint64_t)c.src->B) return false;  // one example per wave and step
    if (c.g.blocks != D3P_PERSIST_NW) return false;  // workgroup b owns gradient columns 4 b .. 4 b + 3
    if (need_owned_list(c.src)) return false;        // every wave owns one fixed batch position
    static int resident = -1;  // workgroups of this kernel the device holds at once (queried once per process)
    if (resident < 0) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_logreg_persist<false>, 64 * D3P_PERSIST_W, persist_lds_bytes()) != hipSuccess)
            resident = 0;
        else
            resident = cus * per_cu;
        (void)hipGetLastError();
    }
    return (int)c.g.blocks <= resident;
}

// D3P_DBG=32: reads the phase stamps the stamped chained kernels left for the last two steps of a launch and prints the
// anatomy of the last step (fourth launch of the process = steady state) to stderr.
static int print_chain_anatomy(const Ctx& c, uint32_t blocks = 0, bool upd = false)
{
    static unsigned long long host[2 * 256 * 16];
    D3P_HIP_TRY(hipMemcpyAsync(host, c.ws.stamps, sizeof(host), hipMemcpyDeviceToHost, c.s));
    D3P_HIP_TRY(hipStreamSynchronize(c.s));
    static int printed = 0;
    if (upd && (dev_dbg_flags() & 0x700) == 0x700 && printed == 3) {  // updater form: publication / parameter-seen times per wave
        ++printed;
        if (!blocks) blocks = c.g.blocks;
        const uint32_t nb = blocks < 256u ? blocks : 256u;
        unsigned long long base = ~0ull;
        for (uint32_t b = 0; b < nb; ++b) {
            const unsigned long long v = host[((size_t)0 * 256 + b) * 16 + 8];
            if (v != 0ull && v < base) base = v;
        }
        if (base == ~0ull) { fprintf(stderr, "  (no updater found in the stamped step)\n"); return D3P_OK; }
        fprintf(stderr, "  updaters of the step before the last: publication by their waves 0, 1 (us after the first publication):");
        for (uint32_t b = 0; b < nb; ++b)
            if (host[((size_t)0 * 256 + b) * 16 + 8] != 0ull)
                fprintf(stderr, " wg %u: %.2f %.2f;", b, (double)(long long)(host[((size_t)b) * 16 + 8] - base) * 0.01,
                        (double)(long long)(host[((size_t)b) * 16 + 9] - base) * 0.01);
        fprintf(stderr, "\n  last step, when the even waves had their parameter words (us after the first publication):\n");
        for (uint32_t b = 0; b < nb; b += 9) {
            fprintf(stderr, "  wg %3u:", b);
            for (int w = 0; w < 8; ++w) fprintf(stderr, " %5.2f", (double)(long long)(host[((size_t)256 + b) * 16 + w] - base) * 0.01);
            fprintf(stderr, "\n");
        }
        return D3P_OK;
    }
    if ((dev_dbg_flags() & 256) && printed == 3) {  // per-WAVE "examples done" times of the last step (16-wave form), us after the workgroup's first wave
        ++printed;
        if (!blocks) blocks = c.g.blocks;
        for (uint32_t b = 0; b < (blocks < 128u ? blocks : 128u); b += 9) {
            unsigned long long mn = ~0ull;
            for (int w = 0; w < 16; ++w) mn = host[((size_t)256 + b) * 16 + w] < mn ? host[((size_t)256 + b) * 16 + w] : mn;
            fprintf(stderr, "  wg %3u waves:", b);
            for (int w = 0; w < 16; ++w) fprintf(stderr, " %5.2f", (double)(host[((size_t)256 + b) * 16 + w] - mn) * 0.01);
            fprintf(stderr, "\n");
        }
        return D3P_OK;
    }
    if (printed++ == 3) {  // fourth launch of the process: steady state
        if (!blocks) blocks = c.g.blocks;
        const uint32_t nb = blocks < 256u ? blocks : 256u;
        auto at = [&](int rec, uint32_t b, int k) { return (double)host[((size_t)rec * 256 + b) * 16 + k] * 0.01; };
        // what a step waits for: the LAST arrival of the previous step
        double last_arr = 0.0, last_acked = 0.0;
        for (uint32_t b = 0; b < nb; ++b) {
            last_arr = at(0, b, 12) > last_arr ? at(0, b, 12) : last_arr;
            last_acked = at(0, b, 10) > last_acked ? at(0, b, 10) : last_acked;
        }
        const char* names[13] = {"entry", "staging barrier (lean kernel)", "prologue done", "-", "-", "examples done", "reduction barrier", "release seen",
                                 "param-independent work done", "atomics issued", "atomics acknowledged", "arrival barrier",
                                 "arrival returned"};
        fprintf(stderr, "chained step anatomy (us, relative to the last arrival of the previous step; mean / min / max over %u workgroups)\n", nb);
        for (int k : {0, 8, 7, 2, 1, 5, 6, 9, 10, 11, 12}) {
            double sum = 0.0, mn = 1e30, mx = -1e30;
            uint32_t who = 0;
            for (uint32_t b = 0; b < nb; ++b) {
                const double v = at(1, b, k) - last_arr;
                sum += v; mn = v < mn ? v : mn;
                if (v > mx) { mx = v; who = b; }
            }
            fprintf(stderr, "  %-30s %7.2f %7.2f %7.2f  (latest: workgroup %u)\n", names[k], sum / nb, mn, mx, who);
        }
        if (getenv("D3P_ANATOMY_ROWS")) {  // one line per workgroup: release seen, prologue done, examples done, reduction barrier, arrival
            for (uint32_t b = 0; b < nb; ++b)
                fprintf(stderr, "  wg %3u: %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f\n", b, at(1, b, 7) - last_arr, at(1, b, 2) - last_arr,
                        at(1, b, 1) - last_arr, at(1, b, 5) - last_arr, at(1, b, 6) - last_arr, at(1, b, 12) - last_arr);
        }
        fprintf(stderr, "  previous step: last acknowledgement %.2f us before its last arrival\n", last_arr - last_acked);
        if (upd) {  // the updater of the previous step: the workgroup whose tail stamps are set
            for (uint32_t b = 0; b < nb; ++b)
                if (host[((size_t)0 * 256 + b) * 16 + 15] != 0ull)
                    fprintf(stderr, "  updater of the previous step (workgroup %u), us after its arrival returned: replicas + state loaded %.2f, rows sent %.2f, "
                                    "world's rows collected %.2f, parameters published %.2f\n", b, at(0, b, 3) - at(0, b, 12), at(0, b, 4) - at(0, b, 12),
                            at(0, b, 14) - at(0, b, 12), at(0, b, 15) - at(0, b, 12));
        }
        const int order[9][2] = {{7, 2}, {2, 5}, {5, 6}, {6, 9}, {9, 10}, {10, 11}, {11, 12}, {8, 7}, {0, 8}};
        for (auto& o : order) {
            double sum = 0.0, mx = -1e30;
            for (uint32_t b = 0; b < nb; ++b) {
                const double v = at(1, b, o[1]) - at(1, b, o[0]);
                sum += v; mx = v > mx ? v : mx;
            }
            fprintf(stderr, "  phase %-28s -> %-28s mean %6.2f max %6.2f\n", names[o[0]], names[o[1]], sum / nb, mx);
        }
    }
    return D3P_OK;
}

struct Xchg;
static void xchg_fill_dev(Xchg* x, XchgDev* d, int K);  // (defined with the exchange, below) takes K epochs of the exchange

// can this run's chained launch be the kernel of d3p_logreg_chain.h?
// w16: the 16-wave form (single-rank runs; any batch size: waves take further items in pairs); else the 8-wave form, which
// prepares at most two items per wave ahead (data-parallel runs, and D3P_CHAIN_W8=1 for A/B measurements)
static bool chain_w16_enabled()
{
    static const bool off = getenv("D3P_CHAIN_W8") != nullptr;
    return !off;
}
static bool lean_chain_ok(const Ctx& c, bool w16)
{
    static const bool off = getenv("D3P_NO_LEAN_CHAIN") != nullptr || getenv("D3P_NO_PIPELINED_STEPS") != nullptr ||
                            getenv("D3P_MAIN_W") != nullptr;
    return !off && c.g.full && c.g.V == 4 && c.g.NK == 1 && c.g.W == 16 && c.m->d == D3P_CHAIN_D &&
           c.m->family == D3P_FAMILY_LOGREG && (w16 || c.items_expected <= 18ull * c.g.blocks);
}
// Data-parallel chained runs: the 16-wave UPDATER form of d3p_logreg_chain.h (round 4; any item count) unless D3P_XCHG_W8=1 asks
// for round 2's 8-wave form with its two exchange workgroups per step (kept for A/B measurements).
static bool xchg_updater_form(const Ctx& c)
{
    static const bool w8 = getenv("D3P_XCHG_W8") != nullptr;
    return !w8 && chain_w16_enabled() && lean_chain_ok(c, true);
}
// workgroups per step of the 16-wave form: two items per wave; up to 12 % more items than 2 x 16 x 128 (a Poisson batch padded
// to its 0.99 quantile) still run on 128 workgroups -- two steps side by side on 256 CUs matter more than a few waves taking a
// third item; larger batches take the whole chip per step.  D3P_CHAIN_NW overrides (sweeps).
static uint32_t chain16_blocks(uint64_t items)
{
    static const int env_nw = [] { const char* e = getenv("D3P_CHAIN_NW"); return e ? atoi(e) : 0; }();
    if (env_nw >= 1 && env_nw <= 480) return (uint32_t)env_nw;  // (the slot row of a step holds 512 arrival slots)
    static const uint32_t cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 2)
            n = 256;
        (void)hipGetLastError();
        return (uint32_t)n;
    }();
    const uint64_t per_wg = 2ull * 16ull;
    uint64_t nw = (items + per_wg - 1) / per_wg;
    if (nw < 1) nw = 1;
    const uint64_t half = cus / 2;
    if (nw > half && items <= 36ull * half) nw = half;  // (up to 36 instead of 32 items per workgroup)
    if (nw > cus) nw = cus;
    return (uint32_t)nw;
}

static int enqueue_chained_batch(const Ctx& c, int g0, int K, const StepSlot* prev_slot0, const float* prev_noise0, const float* X,
                                 const float* y, float* losses, StepSlot* chain_slots, int K_next, Xchg* xchg = nullptr)
{
    MainArgs a;
    memset(&a, 0, sizeof(a));
    fill_model_scalars(c.m, &a);
    a.X = X;
    a.y = y;
    a.pack = c.ws.pack;
    a.partials = c.ws.partials;
    a.B = c.src->B;
    a.row_lo = c.src->row_lo;
    a.row_hi = c.src->row_hi;
    a.clip = c.h->clip;
    fill_fuse_common(c, &a.fuse, g0);
    a.fuse.chain_sched = c.ws.sched;
    ChainFuse& cf = a.chain;
    cf.nw = (int)c.g.blocks;
    cf.g0 = g0;
    cf.K = K;
    cf.slots = c.ws.slots;
    cf.idx_base = c.ws.idx;
    cf.skeys_base = c.ws.skeys;
    cf.plist_base = need_owned_list(c.src) ? c.ws.plist : nullptr;
    cf.noise_base = c.ws.noise;
    cf.prev_slot0 = prev_slot0;
    cf.prev_noise0 = prev_noise0;
    cf.acc_base = c.ws.acc;
    const size_t P = (size_t)c.P;
    cf.state[0][0] = c.st->params; cf.state[0][1] = c.st->adam_m; cf.state[0][2] = c.st->adam_v;
    cf.state[1][0] = c.ws.pp_state; cf.state[1][1] = c.ws.pp_state + P; cf.state[1][2] = c.ws.pp_state + 2 * P;
    cf.losses = losses;
    cf.bar = c.ws.chain_bar;
    cf.abort_flag = run_status_words(c.ws);
    cf.chain_slots = chain_slots;
    cf.K_next = chain_slots ? K_next : 0;
    // arrival counters of this launch (the abort flag behind them is sticky for the whole run); the first launch of a run finds
    // them zeroed by k_run_init
    if (g0 > 0) D3P_HIP_TRY(hipMemsetAsync(c.ws.chain_bar, 0, (size_t)(D3P_STEP_BATCH + 1) * D3P_BAR_WORDS * sizeof(uint32_t), c.s));
    if (use_persistent_steps(c)) {
        static const bool stamps = getenv("D3P_PERSIST_STAMPS") != nullptr;  // developer diagnostic, read once
        if (stamps) {
            a.stamps = c.ws.stamps;
            hipLaunchKernelGGL(k_logreg_persist<true>, dim3(c.g.blocks), dim3(64 * D3P_PERSIST_W), persist_lds_bytes(), c.s, a);
            int rc = check_launch("k_logreg_persist");
            if (rc) return rc;
            static unsigned long long host[4 * 32 * 16];  // (the kernel stamps the first 32 steps of a launch)
            D3P_HIP_TRY(hipMemcpyAsync(host, c.ws.stamps, sizeof(host), hipMemcpyDeviceToHost, c.s));
            D3P_HIP_TRY(hipStreamSynchronize(c.s));
            static int printed = 0;
            if (printed++ == 2) {  // third launch of the process: steady state
                for (int wg = 0; wg < 4; ++wg)
                    for (int t = 8; t < 12 && t < K; ++t) {
                        const unsigned long long* h = host + ((size_t)wg * 32 + t) * 16;
                        const unsigned long long z = host[((size_t)0 * 32 + t) * 16 + 2];  // wg 0, wave 0, after barrier S
                        fprintf(stderr, "wg %3d t %2d | w0: start %.2f C %.2f R %.2f arr1 %.2f A %.2f rows-read %.2f publ %.2f seen2 %.2f | w5: start %.2f C %.2f R %.2f A %.2f\n",
                                wg * 85, t, ((double)h[2] - z) * 0.01, ((double)h[3] - z) * 0.01, ((double)h[4] - z) * 0.01, ((double)h[5] - z) * 0.01,
                                ((double)h[7] - z) * 0.01, ((double)h[6] - z) * 0.01, ((double)h[1] - z) * 0.01, ((double)h[12] - z) * 0.01,
                                ((double)h[8] - z) * 0.01, ((double)h[9] - z) * 0.01, ((double)h[10] - z) * 0.01, ((double)h[11] - z) * 0.01);
                    }
            }
            return D3P_OK;
        }
        hipEvent_t e0, e1;
        timing_pair(K, &e0, &e1);
        if (e0)
            hipExtLaunchKernelGGL(k_logreg_persist<false>, dim3(c.g.blocks), dim3(64 * D3P_PERSIST_W), persist_lds_bytes(), c.s, e0, e1,
                                  0, a);
        else
            hipLaunchKernelGGL(k_logreg_persist<false>, dim3(c.g.blocks), dim3(64 * D3P_PERSIST_W), persist_lds_bytes(), c.s, a);
        return check_launch("k_logreg_persist");
    }
    {
        // The headline shape (d = 512, no intercept, logistic regression; at most ~two examples per wave of an 8-wave
        // workgroup) runs the kernel written for it (d3p_logreg_chain.h); D3P_NO_LEAN_CHAIN=1 keeps the generic template.
        const bool icpt = c.g.tail;  // 512 features + intercept (D = 513): the ICPT instantiations
        // (data-parallel runs: the 16-wave updater form; D3P_XCHG_W8=1: the 8-wave form with two exchange workgroups per step)
        const bool w16 = (!xchg || xchg_updater_form(c)) && chain_w16_enabled();
        if (lean_chain_ok(c, w16)) {
            const uint32_t nw = w16 ? chain16_blocks(c.items_expected) : c.g.blocks;
            ChainArgs ca;
            memset(&ca, 0, sizeof(ca));
            ca.X = X;
            ca.y = y;
            ca.idx_base = c.ws.idx;
            ca.skeys_base = c.ws.skeys;
            ca.plist_base = need_owned_list(c.src) ? c.ws.plist : nullptr;
            ca.noise_base = c.ws.noise;
            ca.slots = c.ws.slots;
            ca.prev_slot0 = prev_slot0;
            ca.prev_noise0 = prev_noise0;
            ca.acc_base = c.ws.acc;
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) ca.state[i][j] = cf.state[i][j];
            ca.losses = losses;
            ca.bar = cf.bar;
            ca.status = cf.abort_flag;
            ca.chain_sched = c.ws.sched;
            ca.chain_slots = cf.chain_slots;
            ca.adam_step = a.fuse.adam_step;
            ca.batch_index = a.fuse.batch_index;
            ca.row_lo = c.src->row_lo;
            ca.sg = a.fuse.sg; ca.inv_sg = a.fuse.inv_sg;
            ca.B = c.src->B;
            ca.nw = (int)nw;
            ca.g0 = g0;
            ca.K = K;
            ca.K_next = cf.K_next;
            ca.A_scale = a.A_scale; ca.c1 = a.c1_w; ca.hz = a.hz_w; ca.inv_obs = a.inv_obs; ca.lik_scale = a.lik_scale;
            ca.obs_scale = a.obs_scale; ca.clip = a.clip; ca.dp_scale = a.fuse.dp_scale; ca.lr = a.fuse.lr; ca.b1 = a.fuse.b1;
            ca.b2 = a.fuse.b2; ca.adam_eps = a.fuse.adam_eps; ca.log_prior = logf(c.m->prior_w);
            ca.c1_b = a.c1_b; ca.hz_b = a.hz_b; ca.log_prior_b = logf(c.m->prior_b);
            ca.gexp = a.gexp;
            ca.dbg = dev_dbg_flags();
            if (xchg) {  // data-parallel run: the step's exchange rides in the launch
                xchg_fill_dev(xchg, &ca.x, K);
                ca.x.xflag = c.ws.xflags;
                ca.x.xsum = c.ws.xsum;
                if (w16) {  // updater form: the state travels as tagged words, no flags
                    static const bool self_trip = getenv("D3P_XCHG_SELF_TRIP") != nullptr;   // developer switch, read once
                    ca.x.self_trip = self_trip ? 1 : 0;
                    const size_t PAc = (size_t)D3P_ACC_COLS(c.P);
                    ca.ll_state[0] = c.ws.ll_state;
                    ca.ll_state[1] = c.ws.ll_state + 2 * PAc;
                    ca.ll_state[2] = c.ws.ll_state + 3 * PAc;
                } else {
                    D3P_HIP_TRY(hipMemsetAsync(c.ws.xflags, 0, (size_t)D3P_STEP_BATCH * D3P_XCHG_WGS * 32 * sizeof(uint32_t), c.s));
                }
            }
            const int W = w16 ? 16 : D3P_CHAIN_W;
            const dim3 grid((uint32_t)K * (nw + (w16 ? 0u : 1u) + ((xchg && !w16) ? (uint32_t)D3P_XCHG_WGS : 0u))), block(64 * W);
            const bool plist = ca.plist_base != nullptr;
            const size_t lds = chain_lds_bytes(icpt, W);
            const bool stamped = (ca.dbg & 32) && K >= 2 && (!xchg || (w16 && !icpt));  // D3P_DBG=32: the stamped instantiation + the phase anatomy on stderr
            if (stamped) ca.stamps = c.ws.stamps;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (!stamped) timing_pair(K, &e0, &e1);
#define D3P_CHAIN_LAUNCH(PL_, ST_, IC_, XC_)                                                                       \
    do {                                                                                                           \
        if (e0) hipExtLaunchKernelGGL((k_logreg_chain<PL_, ST_, IC_, XC_>), grid, block, lds, c.s, e0, e1, 0, ca); \
        else hipLaunchKernelGGL((k_logreg_chain<PL_, ST_, IC_, XC_>), grid, block, lds, c.s, ca);                  \
    } while (0)
#define D3P_CHAIN16_LAUNCH(PL_, ST_, IC_)                                                                                     \
    do {                                                                                                                      \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_logreg_chain<PL_, ST_, IC_, false, 16>), 96 * 1024,   \
                                         "k_logreg_chain (16-wave form)"))                                                    \
            return rc_;                                                                                                       \
        if (e0) hipExtLaunchKernelGGL((k_logreg_chain<PL_, ST_, IC_, false, 16>), grid, block, lds, c.s, e0, e1, 0, ca);        \
        else hipLaunchKernelGGL((k_logreg_chain<PL_, ST_, IC_, false, 16>), grid, block, lds, c.s, ca);                         \
    } while (0)
#define D3P_CHAIN16_XCHG_LAUNCH(PL_, ST_, IC_)                                                                                   \
    do {                                                                                                                      \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_logreg_chain<PL_, ST_, IC_, true, 16>), 96 * 1024,    \
                                         "k_logreg_chain (data-parallel 16-wave form)"))                                      \
            return rc_;                                                                                                       \
        if (e0) hipExtLaunchKernelGGL((k_logreg_chain<PL_, ST_, IC_, true, 16>), grid, block, lds, c.s, e0, e1, 0, ca);         \
        else hipLaunchKernelGGL((k_logreg_chain<PL_, ST_, IC_, true, 16>), grid, block, lds, c.s, ca);                          \
    } while (0)
            if (w16 && xchg) {
                if (stamped) {  // (the anatomy is taken at the production shape: owned lists, no intercept)
                    if (plist) D3P_CHAIN16_XCHG_LAUNCH(true, true, false); else D3P_CHAIN16_XCHG_LAUNCH(false, true, false);
                } else {
                    if (icpt) { if (plist) D3P_CHAIN16_XCHG_LAUNCH(true, false, true); else D3P_CHAIN16_XCHG_LAUNCH(false, false, true); }
                    else { if (plist) D3P_CHAIN16_XCHG_LAUNCH(true, false, false); else D3P_CHAIN16_XCHG_LAUNCH(false, false, false); }
                }
            } else if (w16) {
                if (stamped) {
                    if (icpt) { if (plist) D3P_CHAIN16_LAUNCH(true, true, true); else D3P_CHAIN16_LAUNCH(false, true, true); }
                    else { if (plist) D3P_CHAIN16_LAUNCH(true, true, false); else D3P_CHAIN16_LAUNCH(false, true, false); }
                } else {
                    if (icpt) { if (plist) D3P_CHAIN16_LAUNCH(true, false, true); else D3P_CHAIN16_LAUNCH(false, false, true); }
                    else { if (plist) D3P_CHAIN16_LAUNCH(true, false, false); else D3P_CHAIN16_LAUNCH(false, false, false); }
                }
            } else if (xchg) {  // (no stamped form of the data-parallel kernel)
                if (icpt) { if (plist) D3P_CHAIN_LAUNCH(true, false, true, true); else D3P_CHAIN_LAUNCH(false, false, true, true); }
                else { if (plist) D3P_CHAIN_LAUNCH(true, false, false, true); else D3P_CHAIN_LAUNCH(false, false, false, true); }
            } else if (stamped) {
                if (icpt) { if (plist) D3P_CHAIN_LAUNCH(true, true, true, false); else D3P_CHAIN_LAUNCH(false, true, true, false); }
                else { if (plist) D3P_CHAIN_LAUNCH(true, true, false, false); else D3P_CHAIN_LAUNCH(false, true, false, false); }
            } else {
                if (icpt) { if (plist) D3P_CHAIN_LAUNCH(true, false, true, false); else D3P_CHAIN_LAUNCH(false, false, true, false); }
                else { if (plist) D3P_CHAIN_LAUNCH(true, false, false, false); else D3P_CHAIN_LAUNCH(false, false, false, false); }
            }
#undef D3P_CHAIN_LAUNCH
#undef D3P_CHAIN16_LAUNCH
#undef D3P_CHAIN16_XCHG_LAUNCH
            int rc = check_launch("k_logreg_chain");
            if (rc || !stamped) return rc;
            return print_chain_anatomy(c, nw, xchg != nullptr);
        }
    }
    MainGeom g2 = c.g;
    {
        // Pipelined geometry (D3P_NO_PIPELINED_STEPS=1 keeps one 16-wave workgroup per CU): the same number of workgroups
        // with 8 waves each, so two of them are resident per CU and the noise generation of step t + 1 overlaps with the
        // exchange of step t (see `pregen` in k_logreg_main); every wave then goes through twice as many examples.
        static const bool off = getenv("D3P_NO_PIPELINED_STEPS") != nullptr || getenv("D3P_MAIN_W") != nullptr;
        // Only while a wave then has (about) the two examples it can prepare ahead: beyond that the waiting workgroup of
        // the next step just halves the occupancy of the working one (B = 32768: 44 -> 55 us per step).  The 12 % slack
        // admits Poisson batches padded to a quantile above 16 x workgroups (a few waves then take a third example and
        // generate its noise in the loop): 63.2 -> 68.6 k steps/s at q = 4096 / 1e6.
        if (!off && c.g.full && c.g.V == 4 && c.g.NK == 1 && c.g.W == 16 && (uint64_t)c.src->B <= 18ull * c.g.blocks) {
            g2.W = 8;
            g2.lds = main_lds_bytes(c.D, g2.W);
            cf.pregen = 1;
        }
    }
    g2.blocks = (uint32_t)K * (c.g.blocks + 1u);
    a.dbg = dev_dbg_flags();
    if ((a.dbg & 32) && K >= 2) {  // developer diagnostic (D3P_DBG=32): phase stamps of the last two steps, printed to stderr
        a.stamps = c.ws.stamps;
        int rc = launch_main<3>(c.s, g2, a);
        if (rc) return rc;
        return print_chain_anatomy(c);
    }
    hipEvent_t e0, e1;
    timing_pair(K, &e0, &e1);
    return launch_main<3>(c.s, g2, a, e0, e1);
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

int rccl_allreduce_f32(void* comm, float* buf, size_t count, hipStream_t s)
{
    const RcclApi* api = rccl_api();
    if (!api) return fail(D3P_E_UNSUPPORTED, "librccl.so could not be loaded");
    const ncclResult_t r = api->AllReduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)comm, s);
    if (r != ncclSuccess) return fail(D3P_E_HIP, "ncclAllReduce: %s", api->GetErrorString(r));
    return D3P_OK;
}

// ---- One-shot full-mesh exchange of the step message: xGMI is point to point (7 links per GPU), so for
// a message of 8 KB the latency-optimal all-reduce is not a ring (2 (n - 1) dependent hops) but ONE hop: every rank folds its
// 4 accumulator replicas, writes the folded vector straight into an inbox slot on each of its peers (and its own), raises a
// flag there, waits for the n flags of its own inbox and adds the n vectors locally.  int64 sums are exact and commutative, so
// every rank obtains bitwise the same totals -- the property the RCCL path has too (it stays as the checker and fallback).
// Inboxes are allocated uncached (fine-grained) and shared between the processes with hipIpc handles; all cross-device
// accesses are system-scope.  A row carries its own arrival signal (xchg_ll_store in d3p_logreg_chain.h: every 8-byte word
// holds 32 data bits and the epoch's tag), so a sender neither waits for write acknowledgements nor sends a flag behind the
// data.  Slots are double-buffered by the parity of the step count: a rank can only start exchange e + 1 after every peer
// has finished READING in exchange e (it needs their row of e + 1, sent after their exchange e), so slot parity e is free
// again when exchange e + 2 writes it.  Waits are bounded and raise status[0].
struct Xchg {
    int world, rank;
    uint32_t words;                    // int64 words per message (one folded accumulator row)
    unsigned long long epoch;          // exchanges done (host-side count; the device sees it as an argument)
    char* inbox;                       // this rank's inbox: ll[2][world][words] of 16 bytes
    size_t inbox_bytes;
    char* peer[D3P_XCHG_MAX_WORLD];    // the peers' inboxes mapped into this process (peer[rank] == inbox)
    bool opened[D3P_XCHG_MAX_WORLD];
    // behind the rows, in the same allocation (so the peers' mappings cover it): the COUNT box of the sharded Poisson selection,
    // ll[2][world][D3P_STEP_BATCH] words of 16 bytes, with its own epoch (one exchange per prepared batch of steps)
    size_t cbox_off;
    unsigned long long cepoch;
    bool in_arena;                     // the inbox is a range of the process's hipIpc arena (d3p_ipc_arena.h)
};

static inline size_t xchg_inbox_bytes(int world, uint32_t words) { return align_up((size_t)2 * world * words * 16, 256); }
static inline size_t xchg_cbox_bytes(int world) { return align_up((size_t)2 * world * D3P_STEP_BATCH * 16, 256); }

// The shards' selected counts of the K steps of a prepared batch, all-gathered through the count box (same tagged-word protocol
// as the rows: no flag, no acknowledgement).  Thread t <-> step t: total selected, valid after truncate / suppress
// (minibatch.py:119-124) -> counts[0], counts[1]; selected in the shards with HIGHER rows (= higher ranks: the table is sharded
// contiguously in rank order) -> above_out[t]; the rank's owned valid positions -> n_owned.
struct XchgCountArgs {
    const uint32_t* local;
    int K;
    uint32_t cutoff;
    int suppress;
    uint32_t* counts;
    size_t counts_stride;
    uint32_t* above_out;
    uint32_t* n_owned;
    size_t n_owned_stride;
    int world, rank;
    unsigned long long epoch;
    char* peer[D3P_XCHG_MAX_WORLD];
    size_t cbox_off;
    uint32_t* status;
};

__global__ void __launch_bounds__(D3P_STEP_BATCH) k_xchg_counts(XchgCountArgs a)
{
    const int t = threadIdx.x;
    if (t >= a.K) return;
    const unsigned parity = (unsigned)(a.epoch & 1ull);
    const uint32_t tag = (uint32_t)a.epoch;
    const uint32_t mine = a.local[t];
    for (int p = 0; p < a.world; ++p)
        if (p != a.rank) xchg_ll_store(a.peer[p] + a.cbox_off, ((size_t)parity * a.world + a.rank) * D3P_STEP_BATCH + t, (long long)mine, tag);
    uint32_t total = mine, above = 0u;
    bool ok = true;
    for (int p = 0; p < a.world && ok; ++p) {
        if (p == a.rank) continue;
        unsigned long long w0 = 0ull, w1 = 0ull;
        ok = false;
        for (uint32_t spins = 0; spins < D3P_WAIT_ROUNDS_PEERS; ++spins) {
            xchg_ll_fetch(a.peer[a.rank] + a.cbox_off, ((size_t)parity * a.world + p) * D3P_STEP_BATCH + t, &w0, &w1);
            if (xchg_ll_valid(w0, w1, tag)) { ok = true; break; }
            if ((spins & 63u) == 63u && a.status && __hip_atomic_load(a.status, __ATOMIC_RELAXED, D3P_AGENT) != 0u) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (!ok) {
            if (a.status) chain_raise(a.status, abort_code(D3P_ABORT_XCHG_KERNEL, t, (uint32_t)p));
            break;
        }
        const uint32_t v = (uint32_t)xchg_ll_value(w0, w1);
        total += v;
        if (p > a.rank) above += v;
    }
    if (!ok) { total = 0u; above = 0u; }   // (the run is stopped: nothing of this batch is used)
    const uint32_t valid = a.suppress ? (total <= a.cutoff ? total : 0u) : (total < a.cutoff ? total : a.cutoff);
    a.counts[(size_t)t * a.counts_stride] = total;
    a.counts[(size_t)t * a.counts_stride + 1] = valid;
    a.above_out[t] = above;
    const uint32_t room = valid > above ? valid - above : 0u;
    a.n_owned[(size_t)t * a.n_owned_stride] = room < mine ? room : mine;
}

struct XchgArgs {
    long long* acc;  // R x words: this rank's replicas of the step's accumulator; on return row 0 holds the global totals, rows 1.. zeros
    int R;
    uint32_t words;
    int world, rank;
    unsigned long long epoch;  // 1-based count of this exchange
    char* peer[D3P_XCHG_MAX_WORLD];
    uint32_t* status;  // nullable: [0] raised when a wait runs out
};

__global__ void __launch_bounds__(1024) k_xchg(XchgArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned parity = (unsigned)(a.epoch & 1ull);
    const uint32_t tag = (uint32_t)a.epoch;
    // fold the replicas, deliver the folded row to every inbox (slot [parity][my rank]) and clear the replicas: row 0 is where
    // the world's sums are added up below
    for (uint32_t c = tid; c < a.words; c += blockDim.x) {
        long long v = 0;
        for (int r = 0; r < a.R; ++r) {
            v += a.acc[(size_t)r * a.words + c];
            a.acc[(size_t)r * a.words + c] = 0;
        }
        for (int p = 0; p < a.world; ++p) xchg_ll_store(a.peer[p], ((size_t)parity * a.world + a.rank) * a.words + c, v, tag);
    }
    __threadfence();
    __syncthreads();
    // wave w takes the row of rank w in this rank's own inbox, four columns per lane in flight, asking again until every word
    // carries this epoch's tag
    if (wave < a.world) {
        const size_t row = ((size_t)parity * a.world + wave) * a.words;
        unsigned long long* tot = reinterpret_cast<unsigned long long*>(a.acc);
        for (uint32_t c0 = 0; c0 < a.words; c0 += 256) {
            unsigned long long w0[4], w1[4];
            bool ok = false;
            for (uint32_t spins = 0; spins < 2u * D3P_WAIT_ROUNDS_PEERS; ++spins) {  // (a round here is ~ 0.1 us: 13 s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t c = c0 + 64u * i + lane;
                    xchg_ll_fetch(a.peer[a.rank], row + (c < a.words ? c : a.words - 1u), &w0[i], &w1[i]);
                }
                bool all = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) all = all && xchg_ll_valid(w0[i], w1[i], tag);
                if (all) { ok = true; break; }
                if ((spins & 63u) == 63u && a.status && __hip_atomic_load(a.status, __ATOMIC_RELAXED, D3P_AGENT) != 0u) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (!ok) {  // aborted: the run is over (status[0]); the accumulator is left cleared
                if (a.status) chain_raise(a.status, abort_code(D3P_ABORT_XCHG_KERNEL, 0, (uint32_t)wave));
                break;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t c = c0 + 64u * i + lane;
                if (c < a.words) atomicAdd(tot + c, (unsigned long long)xchg_ll_value(w0[i], w1[i]));
            }
        }
    }
}

// Test / rehearsal helper: the OTHER world - 1 ranks of an exchange played by one workgroup on this GPU.  For each of the next
// `n` exchanges it waits until this rank's row of that epoch has arrived in the inbox of its next peer (i.e. the rank has sent)
// and then delivers all-zero rows of every other rank, tagged with the epoch, to this rank's inbox -- a faithful protocol partner
// (slot parity, tags, one row per peer and epoch) that contributes nothing to the sums.  With it ONE GPU runs the 8-rank code
// paths of the data-parallel kernels (seven rows collected per round, sends to seven inboxes): tests/test_dist.py.
struct XchgSimArgs {
    char* peer_next;   // inbox of rank (rank + 1) % world: where this rank's rows are watched
    char* mine;        // this rank's inbox
    int world, rank;
    uint32_t words;
    unsigned long long epoch0;
    uint32_t n;
};

__global__ void __launch_bounds__(1024) k_xchg_simulate_peers(XchgSimArgs a)
{
    for (uint32_t i = 0; i < a.n; ++i) {
        const unsigned long long epoch = a.epoch0 + i + 1ull;
        const unsigned parity = (unsigned)(epoch & 1ull);
        const uint32_t tag = (uint32_t)epoch;
        for (uint32_t c = threadIdx.x; c < a.words; c += blockDim.x) {
            bool ok = false;
            for (uint32_t spins = 0; spins < D3P_WAIT_ROUNDS_PEERS; ++spins) {
                unsigned long long w0, w1;
                xchg_ll_fetch(a.peer_next, ((size_t)parity * a.world + a.rank) * a.words + c, &w0, &w1);
                if (((uint32_t)(w0 >> 32) & 0x7fffffffu) == (tag & 0x7fffffffu) && ((uint32_t)(w1 >> 32) & 0x7fffffffu) == (tag & 0x7fffffffu)) { ok = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ok) return;   // (the run never sent: it was stopped, or never started)
            for (int p = 0; p < a.world; ++p)
                if (p != a.rank) xchg_ll_store(a.mine, ((size_t)parity * a.world + p) * a.words + c, 0ll, tag);
        }
    }
}

static void xchg_fill_dev(Xchg* x, XchgDev* d, int K)
{
    d->world = x->world;
    d->rank = x->rank;
    for (int p = 0; p < x->world; ++p) d->peer[p] = x->peer[p];
    d->epoch0 = x->epoch;
    x->epoch += (unsigned long long)K;
}

static int enqueue_xchg_poisson_counts(hipStream_t s, Xchg* x, const uint32_t* local, int K, uint32_t cutoff, int suppress, uint32_t* counts,
                                       size_t counts_stride_words, uint32_t* above_out, uint32_t* n_owned, size_t n_owned_stride_words,
                                       uint32_t* status)
{
    XchgCountArgs a;
    memset(&a, 0, sizeof(a));
    a.local = local;
    a.K = K;
    a.cutoff = cutoff;
    a.suppress = suppress;
    a.counts = counts;
    a.counts_stride = counts_stride_words;
    a.above_out = above_out;
    a.n_owned = n_owned;
    a.n_owned_stride = n_owned_stride_words;
    a.world = x->world;
    a.rank = x->rank;
    a.epoch = ++x->cepoch;
    for (int p = 0; p < x->world; ++p) a.peer[p] = x->peer[p];
    a.cbox_off = x->cbox_off;
    a.status = status;
    hipLaunchKernelGGL(k_xchg_counts, dim3(1), dim3(D3P_STEP_BATCH), 0, s, a);
    return check_launch("k_xchg_counts");
}

static int enqueue_xchg(hipStream_t s, Xchg* x, long long* acc, int R, uint32_t* status)
{
    XchgArgs a;
    memset(&a, 0, sizeof(a));
    a.acc = acc;
    a.R = R;
    a.words = x->words;
    a.world = x->world;
    a.rank = x->rank;
    a.epoch = ++x->epoch;
    for (int p = 0; p < x->world; ++p) a.peer[p] = x->peer[p];
    a.status = status;
    hipLaunchKernelGGL(k_xchg, dim3(1), dim3(1024), 0, s, a);
    return check_launch("k_xchg");
}

static int enqueue_chain(const Ctx& c, int K);

// comm != nullptr: data-parallel run -- after every step launch the rank's fixed-point accumulator (R x (P + 2) int64) is
// sum-all-reduced in place on the same stream (the ONE collective of the step); the next launch applies the global sums.
// The whole run: k_run_init (schedule from the state, key chain of the first batch, zeroed accumulators and status) ->
// per batch the sampler and the step launches -> k_flush (last pending update, final key, status to the host record).
static int run_fused_steps(const Ctx& c, const float* X, const float* y, uint32_t num_steps, float* losses, ncclComm_t comm = nullptr,
                           Xchg* xchg = nullptr)
{
    int rc;
    const uint32_t acc_words = 3u * D3P_ACC_R * (uint32_t)D3P_ACC_COLS(c.P);
    const bool sampled = c.src->kind != D3P_BATCH_EXPLICIT;
    uint32_t* key_out = c.st->rng_key + 16 * ((c.st->key_slot + (int)num_steps) & 1);
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
    if (num_steps == 0) {  // nothing to run: the key and the status words are still defined afterwards
        D3P_HIP_TRY(reset_status_words(c));
        const d3p_dpsvi_state* st0 = c.from ? c.from : c.st;
        D3P_HIP_TRY(hipMemcpyAsync(key_out, st0->rng_key + 16 * (st0->key_slot & 1), 16 * sizeof(uint32_t), hipMemcpyDeviceToDevice, c.s));
        if (c.from) {
            D3P_HIP_TRY(hipMemcpyAsync(c.st->params, c.from->params, c.P * sizeof(float), hipMemcpyDeviceToDevice, c.s));
            D3P_HIP_TRY(hipMemcpyAsync(c.st->adam_m, c.from->adam_m, c.P * sizeof(float), hipMemcpyDeviceToDevice, c.s));
            D3P_HIP_TRY(hipMemcpyAsync(c.st->adam_v, c.from->adam_v, c.P * sizeof(float), hipMemcpyDeviceToDevice, c.s));
            D3P_HIP_TRY(hipMemcpyAsync(c.st->step, c.from->step, sizeof(int32_t), hipMemcpyDeviceToDevice, c.s));
        }
        return D3P_OK;
    }
    RunInitCopy cp;
    memset(&cp, 0, sizeof(cp));
    const d3p_dpsvi_state* st0 = c.from ? c.from : c.st;  // where the run starts from
    if (c.from) {
        cp.src[0] = c.from->params; cp.src[1] = c.from->adam_m; cp.src[2] = c.from->adam_v;
        cp.dst[0] = c.st->params; cp.dst[1] = c.st->adam_m; cp.dst[2] = c.st->adam_v;
        cp.step_dst = c.st->step;
        cp.n = c.P;
        cp.by_value = c.batch0_by_value ? 1 : 0;
        cp.batch0 = c.batch0;
        cp.batch_index_dst = c.src->batch_index;
    }
    // (data-parallel with the one-shot exchange: chained too when the shape has the dedicated kernel -- the exchange then rides
    // in the launch; D3P_XCHG_PER_STEP=1 keeps one step launch + one exchange launch per step)
    static const bool xchg_per_step = getenv("D3P_XCHG_PER_STEP") != nullptr;
    // updater form of the data-parallel chained launch: the optimiser state travels as tagged words, nothing is pending between
    // launches or at the end of the run
    const bool chain_ok = use_chained_steps(c);   // (the run's form is decided here, once)
    const bool upd = xchg && !comm && chain_ok && !xchg_per_step && xchg_updater_form(c);
    if (upd) {
        const size_t PAc = (size_t)D3P_ACC_COLS(c.P);
        cp.ll[0] = c.ws.ll_state; cp.ll[1] = c.ws.ll_state + 2 * PAc; cp.ll[2] = c.ws.ll_state + 3 * PAc;
        cp.ll_src[0] = st0->params; cp.ll_src[1] = st0->adam_m; cp.ll_src[2] = st0->adam_v;
        cp.ll_tag = (uint32_t)(xchg->epoch + 1ull);   // the first step of the run is exchange epoch + 1
        cp.ll_cols = (uint32_t)PAc;
        cp.n = c.P;
    }
    const bool init_zeroes_bar = !comm && chain_ok;  // the first chained launch's arrival counters: no memset launch
    if (init_zeroes_bar) {
        cp.bar = c.ws.chain_bar;
        cp.bar_words = (uint32_t)((batch_len(0) + 1) * D3P_BAR_WORDS);
    }
    // (Measured and not kept: the sampler's workgroups inside this launch, each waiting for the key chain to pass its step -- the
    // launch then takes 36 us for 20 steps, exactly the 18 + 17 us of the two launches: the sampler's ~15 us are LATENCY of the
    // last step's blocks (serial ChaCha derivations, Feistel walk, six dependent threefry calls), not throughput.)
    hipLaunchKernelGGL(k_run_init, dim3(1), dim3(256), 0, c.s, (const uint32_t*)(st0->rng_key + 16 * (st0->key_slot & 1)),
                       (const int32_t*)st0->step, sampled ? (const uint32_t*)c.src->batch_index : nullptr, c.ws.sched, cb[0].ws.slots,
                       batch_len(0), c.ws.acc, acc_words, run_status_words(c.ws), cp);
    if ((rc = check_launch("k_run_init"))) return rc;
    if ((rc = enqueue_sampler(cb[0], batch_len(0), xchg))) return rc;
    static const bool no_piggy = getenv("D3P_NO_CHAIN_PIGGYBACK") != nullptr;  // developer switch, read once
    const StepSlot* prev_slot = nullptr;
    const float* prev_noise = nullptr;
    int g = 0;
    const bool chained = !comm && chain_ok && (!xchg || upd || (lean_chain_ok(c, false) && !xchg_per_step));
    const bool persist = chained && use_persistent_steps(c);
    for (uint32_t b = 0; b < n_batches; ++b) {
        const int cur = (int)(b & 1), nxt = cur ^ 1;
        const int K = batch_len(b), K_next = (b + 1 < n_batches) ? batch_len(b + 1) : 0;
        if (chained) {
            if ((rc = enqueue_chained_batch(cb[cur], g, K, prev_slot, prev_noise, X, y, losses, no_piggy ? nullptr : cb[nxt].ws.slots,
                                            K_next, xchg)))
                return rc;
            g += K;
            if (!persist && !upd) {  // the persistent / updater forms apply every update inside the launch: nothing is pending afterwards
                prev_slot = cb[cur].ws.slots + (K - 1);
                prev_noise = cb[cur].ws.noise + (size_t)(K - 1) * c.P;
            }
        }
        for (int t = 0; !chained && t < K; ++t, ++g) {
            StepSlot* cslot = (!no_piggy && t < K_next) ? cb[nxt].ws.slots + t : nullptr;
            // (data-parallel loop: only every 16th launch is bracketed, so that event creation does not sit on the host's
            // enqueue path of every step)
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (!(comm || xchg) || (g & 15) == 0) timing_pair(1, &e0, &e1);
            if ((rc = enqueue_fused_step(cb[cur], g, t, prev_slot, prev_noise, X, y, (losses && g > 0) ? losses + g - 1 : nullptr,
                                         cslot, t, t == K_next - 1, false, false, e0, e1)))
                return rc;
            if (comm) {
                const size_t words = (size_t)D3P_ACC_R * D3P_ACC_COLS(c.P);
                long long* acc = c.ws.acc + (size_t)(g % 3) * words;
                const ncclResult_t r = rccl_api()->AllReduce(acc, acc, words, ncclInt64, ncclSum, comm, c.s);
                if (r != ncclSuccess) return fail(D3P_E_HIP, "ncclAllReduce: %s", rccl_api()->GetErrorString(r));
            } else if (xchg) {  // one-shot full-mesh exchange of the folded accumulator (8 KB per peer)
                long long* acc = c.ws.acc + (size_t)(g % 3) * D3P_ACC_R * D3P_ACC_COLS(c.P);
                if ((rc = enqueue_xchg(c.s, xchg, acc, D3P_ACC_R, run_status_words(c.ws)))) return rc;
            }
            prev_slot = cb[cur].ws.slots + t;
            prev_noise = cb[cur].ws.noise + (size_t)t * c.P;
        }
        if (b + 1 < n_batches) {
            if (no_piggy && (rc = enqueue_chain(cb[nxt], K_next))) return rc;
            if ((rc = enqueue_sampler(cb[nxt], K_next, xchg))) return rc;
        }
    }
    if (persist) {  // state, step counters and losses are already final: only the key is left
        status_slot_invalidate(run_status_words(c.ws));
        hipLaunchKernelGGL(k_sched_finish, dim3(1), dim3(64), 0, c.s, (const Sched*)c.ws.sched, key_out);
        return check_launch("k_sched_finish");
    }
    // apply the update of the last step, store the final key, report the status
    FlushArgs fa;
    memset(&fa, 0, sizeof(fa));
    const size_t words = (size_t)D3P_ACC_R * D3P_ACC_COLS(c.P);
    fa.acc_prev = c.ws.acc + (size_t)((g + 2) % 3) * words;
    fa.nrep = D3P_ACC_R;
    if (chained && xchg) {  // the exchange workgroups left the world's sums in one row per step
        fa.acc_prev = c.ws.xsum + (size_t)((g + 2) % 3) * D3P_ACC_COLS(c.P);
        fa.nrep = 1;
    }
    if (upd) {  // nothing pending: unpack the tagged state of the epoch after the run's last
        const size_t PAc = (size_t)D3P_ACC_COLS(c.P);
        fa.ll[0] = c.ws.ll_state; fa.ll[1] = c.ws.ll_state + 2 * PAc; fa.ll[2] = c.ws.ll_state + 3 * PAc;
        fa.ll_tag = (uint32_t)(xchg->epoch + 1ull);
    }
    fa.noise = prev_noise;
    fa.slot = prev_slot;
    {
        const size_t P = (size_t)c.P;
        float* const bufs[2][3] = {{c.st->params, c.st->adam_m, c.st->adam_v}, {c.ws.pp_state, c.ws.pp_state + P, c.ws.pp_state + 2 * P}};
        const int in = g > 0 ? ((g - 1) & 1) : 0;  // launch g - 1 published to buffer (g - 1) & 1
        for (int j = 0; j < 3; ++j) { fa.state_in[j] = bufs[in][j]; fa.state_out[j] = bufs[0][j]; }
    }
    fa.loss_out = losses ? losses + g - 1 : nullptr;
    fa.adam_step = c.st->step;
    fa.batch_index = sampled ? c.src->batch_index : nullptr;
    fa.sched = c.ws.sched;
    fa.key_out = key_out;
    fa.status = run_status_words(c.ws);
    fa.dbg_print = (dev_dbg_flags() & 64) ? 1 : 0;
    fa.host_status = status_slot_claim(run_status_words(c.ws), &fa.host_tag);
    fa.P = c.P;
    fa.B = (int)c.src->B;
    fa.dp_scale = c.h->dp_scale; fa.clip = c.h->clip; fa.obs_scale = 1.0f / c.m->inv_obs;
    fa.lr = c.h->lr; fa.b1 = c.h->b1; fa.b2 = c.h->b2; fa.adam_eps = c.h->adam_eps;
    fa.inv_sg = 1.0 / (1099511627776.0 / (double)fabsf(c.h->clip));
    hipLaunchKernelGGL(k_flush, dim3(1), dim3(1024), 0, c.s, fa);
    return check_launch("k_flush");
}

// launch geometry of the step kernels for this model and batch source (no device memory involved)
static int fill_geometry(Ctx* c, const d3p_logreg_model* model, const d3p_batch_source* src)
{
    int rc = main_geometry(model, src->B, &c->g);
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
        c->items_expected = expected;
    } else {
        c->items_expected = src->B;
    }
    c->m = model;
    c->src = src;
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
    rc = fill_geometry(c, model, src);
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
    D3P_REQUIRE(X_dev && sums_dev, "null data pointer");
    if (int rcm = validate_model(model, y_dev, "d3p_dpvi_logreg")) return rcm;
    if ((rc = enqueue_sched_init_chain(c, 1))) return rc;
    if ((rc = enqueue_sampler(c, 1))) return rc;
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
    D3P_REQUIRE(num_steps >= 1 && num_steps <= D3P_STEP_BATCH, "d3p_dpvi_logreg_prepare: 1 <= num_steps <= 128");
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
    D3P_REQUIRE(t < D3P_STEP_BATCH, "d3p_dpvi_logreg_step_sums: t must be < 128");
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
    D3P_REQUIRE(t < D3P_STEP_BATCH, "d3p_dpvi_logreg_step_finalize: t must be < 128");
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
    D3P_REQUIRE(num_steps >= 1 && num_steps <= D3P_STEP_BATCH, "d3p_dpvi_logreg_prepare_buf: 1 <= num_steps <= 128");
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
    *words_per_buffer = (size_t)D3P_ACC_R * D3P_ACC_COLS(2 * ((size_t)model->d + (model->intercept ? 1 : 0)));
    return D3P_OK;
}

int d3p_dpvi_logreg_fused_step_supported(const d3p_logreg_model* model, const d3p_batch_source* src)
{
    if (!model || !src) return 0;
    Ctx c;
    c.m = model;
    c.src = src;
    if (fill_geometry(&c, model, src) != D3P_OK) return 0;
    return c.g.wide ? 0 : 1;
}

int d3p_dpvi_logreg_acc_reset(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                              const d3p_dpsvi_state* state, const d3p_batch_source* src, void* workspace_dev,
                              size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_HIP_TRY(hipMemsetAsync(c.ws.acc, 0, 3 * (size_t)D3P_ACC_R * D3P_ACC_COLS(c.P) * sizeof(long long), c.s));
    D3P_HIP_TRY(reset_status_words(c));
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
    D3P_REQUIRE(t < D3P_STEP_BATCH && prev_t < D3P_STEP_BATCH, "slot index must be < 128");
    // rows too wide for the register-tiled kernel have no one-launch step (the column-chunked kernel only leaves partial rows):
    // launching it with that geometry would compute nonsense -- the caller takes the two-kernel steps (step_sums / step_finalize)
    if (c.g.wide)
        return fail(D3P_E_UNSUPPORTED, "d3p_dpvi_logreg_fused_step: rows of %d columns run as two-kernel steps (d3p_dpvi_logreg_step_sums / "
                    "_step_finalize); ask d3p_dpvi_logreg_fused_step_supported first", c.D);
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

int d3p_xchg_create(int32_t world, int32_t rank, uint32_t words, void** xchg_out, uint8_t* handle_out, size_t handle_bytes)
{
    D3P_REQUIRE(xchg_out && handle_out && handle_bytes >= D3P_IPC_HANDLE_BYTES, "d3p_xchg_create: null pointer or handle buffer < 80 bytes");
    D3P_REQUIRE(world >= 1 && world <= D3P_XCHG_MAX_WORLD && rank >= 0 && rank < world && words >= 1, "d3p_xchg_create: bad arguments");
    Xchg* x = new Xchg();
    x->world = world;
    x->rank = rank;
    x->words = words;
    x->epoch = 0;
    x->cbox_off = xchg_inbox_bytes(world, words);
    x->cepoch = 0;
    x->inbox_bytes = x->cbox_off + xchg_cbox_bytes(world);
    // the inbox: a zeroed range of the process's hipIpc arena (d3p_ipc_arena.h), or an allocation of its own when it does not fit
    IpcRange r;
    if (int rc = ipc_range_create(x->inbox_bytes, &r, handle_out, "d3p_xchg_create")) { delete x; return rc; }
    x->inbox = r.ptr;
    x->inbox_bytes = r.bytes;
    x->in_arena = r.in_arena;
    for (int i = 0; i < D3P_XCHG_MAX_WORLD; ++i) { x->peer[i] = nullptr; x->opened[i] = false; }
    x->peer[rank] = x->inbox;
    *xchg_out = x;
    return D3P_OK;
}

int d3p_xchg_connect(void* xchg, const uint8_t* handles, size_t handle_stride)
{
    D3P_REQUIRE(xchg && handles && handle_stride >= D3P_IPC_HANDLE_BYTES, "d3p_xchg_connect: bad arguments (handles are 80 bytes)");
    Xchg* x = (Xchg*)xchg;
    for (int p = 0; p < x->world; ++p) {
        if (p == x->rank) continue;
        if (int rc = ipc_peer_open(handles + (size_t)p * handle_stride, &x->peer[p], &x->opened[p], "d3p_xchg_connect", p)) return rc;
    }
    return D3P_OK;
}

int d3p_xchg_connect_local(void* xchg, void* const* peers, int32_t world)
{
    D3P_REQUIRE(xchg && peers, "d3p_xchg_connect_local: null pointer");
    Xchg* x = (Xchg*)xchg;
    D3P_REQUIRE(world == x->world, "d3p_xchg_connect_local: group size differs from the one the exchange was created for");
    for (int p = 0; p < world; ++p) {
        const Xchg* q = (const Xchg*)peers[p];
        D3P_REQUIRE(q && q->rank == p && q->words == x->words && q->world == world, "d3p_xchg_connect_local: peers must be the group's exchanges in rank order");
        x->peer[p] = q->inbox;
    }
    return D3P_OK;
}

// Unmap the peers' inboxes: the first half of a teardown (every rank disconnects -> barrier -> every rank destroys; a rank that frees
// its inbox while a peer still has it mapped breaks the exporter's NEXT hipIpcGetMemHandle with dmabuf IPC).  Idempotent.
int d3p_xchg_disconnect(void* xchg)
{
    if (!xchg) return D3P_OK;
    Xchg* x = (Xchg*)xchg;
    for (int p = 0; p < x->world; ++p)
        if (p != x->rank && x->peer[p]) {
            ipc_peer_close(x->peer[p], x->opened[p]);   // (a range of a peer's arena stays mapped: nothing to undo)
            x->opened[p] = false;
            x->peer[p] = nullptr;
        }
    return D3P_OK;
}

int d3p_xchg_destroy(void* xchg)
{
    if (!xchg) return D3P_OK;
    Xchg* x = (Xchg*)xchg;
    for (int p = 0; p < x->world; ++p)
        if (p != x->rank) ipc_peer_close(x->peer[p], x->opened[p]);
    IpcRange r;
    r.ptr = x->inbox; r.bytes = x->inbox_bytes; r.in_arena = x->in_arena;
    ipc_range_destroy(&r);
    delete x;
    return D3P_OK;
}

int d3p_xchg_poisson_counts(void* stream, void* xchg, const uint32_t* shard_counts_dev, uint32_t num_steps, uint32_t cutoff, int suppress,
                            uint32_t* counts_dev, size_t counts_stride_words, uint32_t* above_dev, uint32_t* n_owned_dev,
                            size_t n_owned_stride_words)
{
    D3P_REQUIRE(xchg && shard_counts_dev && counts_dev && above_dev && n_owned_dev, "d3p_xchg_poisson_counts: null pointer");
    D3P_REQUIRE(num_steps >= 1 && num_steps <= D3P_STEP_BATCH, "d3p_xchg_poisson_counts: 1 <= num_steps <= 128 (one prepared batch)");
    D3P_REQUIRE(counts_stride_words >= 2 && n_owned_stride_words >= 1, "d3p_xchg_poisson_counts: strides too small");
    return enqueue_xchg_poisson_counts((hipStream_t)stream, (Xchg*)xchg, shard_counts_dev, (int)num_steps, cutoff, suppress, counts_dev,
                                       counts_stride_words, above_dev, n_owned_dev, n_owned_stride_words, nullptr);
}

int d3p_xchg_simulate_peers(void* stream, void* xchg, uint32_t num_exchanges)
{
    D3P_REQUIRE(xchg, "d3p_xchg_simulate_peers: null exchange");
    Xchg* x = (Xchg*)xchg;
    D3P_REQUIRE(x->world >= 2, "d3p_xchg_simulate_peers: needs an exchange of at least two ranks");
    const int nxt = (x->rank + 1) % x->world;
    D3P_REQUIRE(x->peer[nxt] && x->peer[nxt] != x->inbox, "d3p_xchg_simulate_peers: the peers' inboxes must be mapped (d3p_xchg_connect / _connect_local)");
    if (num_exchanges == 0) return D3P_OK;
    XchgSimArgs a;
    memset(&a, 0, sizeof(a));
    a.peer_next = x->peer[nxt];
    a.mine = x->inbox;
    a.world = x->world;
    a.rank = x->rank;
    a.words = x->words;
    a.epoch0 = x->epoch;
    a.n = num_exchanges;
    hipLaunchKernelGGL(k_xchg_simulate_peers, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    return check_launch("k_xchg_simulate_peers");
}

int d3p_xchg_allreduce(void* stream, void* xchg, long long* acc_dev, int32_t replicas)
{
    D3P_REQUIRE(xchg && acc_dev && replicas >= 1, "d3p_xchg_allreduce: bad arguments");
    return enqueue_xchg((hipStream_t)stream, (Xchg*)xchg, acc_dev, replicas, nullptr);
}

int d3p_dpvi_logreg_run_xchg(void* stream, void* xchg, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                             const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev, const float* y_dev,
                             uint32_t num_steps, float* losses_dev, void* workspace_dev, size_t workspace_bytes)
{
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, src, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(xchg, "d3p_dpvi_logreg_run_xchg: null exchange");
    D3P_REQUIRE(X_dev || src->row_lo == src->row_hi, "null data pointer");
    if (int rcm = validate_model(model, y_dev ? (const void*)y_dev : (src->row_lo == src->row_hi ? (const void*)model : nullptr),
                                 "d3p_dpvi_logreg_run_xchg"))
        return rcm;
    D3P_REQUIRE(src->kind != D3P_BATCH_EXPLICIT, "d3p_dpvi_logreg_run_xchg: needs an on-device sampler (Feistel or Poisson)");
    D3P_REQUIRE(((Xchg*)xchg)->words == (uint32_t)D3P_ACC_COLS(c.P), "d3p_dpvi_logreg_run_xchg: the exchange was created for another message size");
    return run_fused_steps(c, X_dev, y_dev, num_steps, losses_dev, nullptr, (Xchg*)xchg);
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
    return run_fused_steps(c, X_dev, y_dev, num_steps, losses_dev, (ncclComm_t)comm);
}

// d3p_dpvi_logreg_run as a function of an immutable state (DPSVI.update / the fori_loop body return a NEW state,
// svi.py:395-434): the run starts from `from` (key slot from->key_slot, optimiser state, step counter: read only) and from
// batch index first_batch (by value), and leaves its result in `state` (key in slot num_steps & 1 of state->rng_key; its
// key_slot is taken as 0) -- the copies and the batch-index word that the caller would otherwise prepare with four small
// launches happen inside the run's first kernel.  Fused-step configurations only (the default).
int d3p_dpvi_logreg_run_from(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                             const d3p_dpsvi_state* state, const d3p_dpsvi_state* from, const d3p_batch_source* src,
                             uint32_t first_batch, const float* X_dev, const float* y_dev, uint32_t num_steps, float* losses_dev,
                             void* workspace_dev, size_t workspace_bytes)
{
    D3P_REQUIRE(from && from->rng_key && from->params && from->adam_m && from->adam_v && from->step, "d3p_dpvi_logreg_run_from: null source state");
    D3P_REQUIRE(src && workspace_dev, "d3p_dpvi_logreg_run_from: null pointer");
    d3p_batch_source s2 = *src;
    if (s2.kind != D3P_BATCH_EXPLICIT && !s2.batch_index) s2.batch_index = reinterpret_cast<uint32_t*>(workspace_dev);  // placeholder for validate(); set below
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, &s2, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(X_dev, "null data pointer");
    if (int rcm = validate_model(model, y_dev, "d3p_dpvi_logreg_run_from")) return rcm;
    D3P_REQUIRE(src->row_lo == 0 && src->row_hi == src->n_rows, "d3p_dpvi_logreg_run_from is the single-GPU path");
    D3P_REQUIRE(state->key_slot == 0, "d3p_dpvi_logreg_run_from: state->key_slot must be 0");
    // the run's own batch-index word: a spare word of the workspace (k_run_init stores first_batch there, k_flush advances it)
    s2.batch_index = reinterpret_cast<uint32_t*>(c.ws.scratch_state + 3 * c.P + 2);
    if (!use_fused_step(c)) {  // two-kernel steps (wide rows, D3P_NO_FUSED_STEP): copy here, then the in-place run
        hipStream_t hs = (hipStream_t)stream;
        const size_t pb = (size_t)c.P * sizeof(float);
        D3P_HIP_TRY(hipMemcpyAsync(state->params, from->params, pb, hipMemcpyDeviceToDevice, hs));
        D3P_HIP_TRY(hipMemcpyAsync(state->adam_m, from->adam_m, pb, hipMemcpyDeviceToDevice, hs));
        D3P_HIP_TRY(hipMemcpyAsync(state->adam_v, from->adam_v, pb, hipMemcpyDeviceToDevice, hs));
        D3P_HIP_TRY(hipMemcpyAsync(state->step, from->step, sizeof(int32_t), hipMemcpyDeviceToDevice, hs));
        D3P_HIP_TRY(hipMemcpyAsync(state->rng_key, from->rng_key + 16 * (from->key_slot & 1), 16 * sizeof(uint32_t), hipMemcpyDeviceToDevice, hs));
        D3P_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)s2.batch_index, (int)first_batch, 1, hs));
        return d3p_dpvi_logreg_run(stream, model, hyper, state, &s2, X_dev, y_dev, num_steps, losses_dev, workspace_dev, workspace_bytes);
    }
    c.src = &s2;
    c.from = from;
    c.batch0_by_value = true;
    c.batch0 = first_batch;
    return run_fused_steps(c, X_dev, y_dev, num_steps, losses_dev);
}

// The data-parallel runs (d3p_dpvi_logreg_run_dist / _run_xchg) likewise as functions of an immutable state: `comm` (RCCL) or
// `xchg` (one-shot exchange) or neither; the rank's shard is src->row_lo .. row_hi.
int d3p_dpvi_logreg_run_dist_from(void* stream, void* comm, void* xchg, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                                  const d3p_dpsvi_state* state, const d3p_dpsvi_state* from, const d3p_batch_source* src,
                                  uint32_t first_batch, const float* X_dev, const float* y_dev, uint32_t num_steps, float* losses_dev,
                                  void* workspace_dev, size_t workspace_bytes)
{
    D3P_REQUIRE(from && from->rng_key && from->params && from->adam_m && from->adam_v && from->step, "d3p_dpvi_logreg_run_dist_from: null source state");
    D3P_REQUIRE(src && workspace_dev, "d3p_dpvi_logreg_run_dist_from: null pointer");
    D3P_REQUIRE(!(comm && xchg), "d3p_dpvi_logreg_run_dist_from: give one of comm / xchg");
    d3p_batch_source s2 = *src;
    if (s2.kind != D3P_BATCH_EXPLICIT && !s2.batch_index) s2.batch_index = reinterpret_cast<uint32_t*>(workspace_dev);  // placeholder for validate(); set below
    Ctx c;
    int rc = make_ctx(&c, stream, model, hyper, state, &s2, workspace_dev, workspace_bytes);
    if (rc) return rc;
    D3P_REQUIRE(X_dev || src->row_lo == src->row_hi, "null data pointer");
    if (int rcm = validate_model(model, y_dev ? (const void*)y_dev : (src->row_lo == src->row_hi ? (const void*)model : nullptr),
                                 "d3p_dpvi_logreg_run_dist_from"))
        return rcm;
    D3P_REQUIRE(src->kind != D3P_BATCH_EXPLICIT, "d3p_dpvi_logreg_run_dist_from: needs an on-device sampler (Feistel or Poisson)");
    D3P_REQUIRE(state->key_slot == 0, "d3p_dpvi_logreg_run_dist_from: state->key_slot must be 0");
    if (comm && !rccl_api()) return fail(D3P_E_UNSUPPORTED, "d3p_dpvi_logreg_run_dist_from: librccl.so could not be loaded");
    if (xchg)
        D3P_REQUIRE(((Xchg*)xchg)->words == (uint32_t)D3P_ACC_COLS(c.P), "d3p_dpvi_logreg_run_dist_from: the exchange was created for another message size");
    if (!use_fused_step(c)) return fail(D3P_E_UNSUPPORTED, "d3p_dpvi_logreg_run_dist_from: the data-parallel run needs the fused step");
    s2.batch_index = reinterpret_cast<uint32_t*>(c.ws.scratch_state + 3 * c.P + 2);
    c.src = &s2;
    c.from = from;
    c.batch0_by_value = true;
    c.batch0 = first_batch;
    return run_fused_steps(c, X_dev, y_dev, num_steps, losses_dev, (ncclComm_t)comm, (Xchg*)xchg);
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
    if (use_fused_step(c)) return run_fused_steps(c, X_dev, y_dev, num_steps, losses_dev);
    if ((rc = enqueue_sched_init(c))) return rc;
    D3P_HIP_TRY(reset_status_words(c));  // (no waits, no fixed-point sums: stays 0)
    // Two-kernel steps (D3P_NO_FUSED_STEP): main + finalize per step; the key-chain step of the next batch rides in every
    // k_finalize launch as one extra workgroup.  (Running the chain or the sampler on an auxiliary stream was measured
    // slower -- 19.3-20.1 vs 17.6 us/step -- and is not kept.)
    const uint32_t n_batches = (num_steps + D3P_STEP_BATCH - 1) / D3P_STEP_BATCH;
    auto batch_len = [&](uint32_t b) {
        const uint32_t rem = num_steps - b * D3P_STEP_BATCH;
        return (int)(rem < D3P_STEP_BATCH ? rem : D3P_STEP_BATCH);
    };
    Ctx cb[2] = {c, c};  // views of the two slot buffers
    cb[1].ws = c.ws2;
    cb[1].ws.partials = c.ws.partials;
    if (num_steps > 0 && (rc = enqueue_batch_prep(cb[0], batch_len(0)))) return rc;
    for (uint32_t b = 0; b < n_batches; ++b) {
        const int cur = (int)(b & 1), nxt = cur ^ 1;
        const int K = batch_len(b), K_next = (b + 1 < n_batches) ? batch_len(b + 1) : 0;
        for (int t = 0; t < K; ++t) {
            if ((rc = enqueue_main(cb[cur], t, X_dev, y_dev, nullptr, false))) return rc;
            StepSlot* cslot = t < K_next ? cb[nxt].ws.slots + t : nullptr;
            if ((rc = enqueue_finalize(cb[cur], t, c.ws.partials, c.g.blocks,
                                       losses_dev ? losses_dev + (size_t)b * D3P_STEP_BATCH + t : nullptr, nullptr, cslot, t,
                                       t == K_next - 1)))
                return rc;
        }
        if (b + 1 < n_batches && (rc = enqueue_sampler(cb[nxt], K_next))) return rc;
    }
    return enqueue_sched_finish(c, (int)num_steps);
}

int d3p_dpvi_logreg_set_run_form(int form)
{
    D3P_REQUIRE(form == 0 || form == 1, "d3p_dpvi_logreg_set_run_form: form must be 0 (automatic) or 1 (one launch per step)");
    g_run_form = form;
    return D3P_OK;
}

int d3p_dpvi_logreg_kernel_timing_enable(int enable)
{
    g_kt.on = enable != 0;
    if (g_kt.on) {  // a pool of events for the launches to come (64 pairs; more are created as needed)
        while (g_kt.pool.size() < 128) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); break; }
            g_kt.pool.push_back(e);
        }
    }
    return D3P_OK;
}

int d3p_dpvi_logreg_kernel_timing_read(double* total_us_out, uint32_t* launches_out, uint32_t* steps_out)
{
    D3P_REQUIRE(total_us_out && launches_out && steps_out, "d3p_dpvi_logreg_kernel_timing_read: null pointer");
    double us = 0.0;
    uint32_t n = 0;
    hipError_t err = hipSuccess;
    for (size_t i = 0; i + 1 < g_kt.ev.size(); i += 2) {
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventSynchronize(g_kt.ev[i + 1]);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, g_kt.ev[i], g_kt.ev[i + 1]);
        if (err == hipSuccess) { us += 1000.0 * ms; ++n; }
        (void)hipEventDestroy(g_kt.ev[i]);
        (void)hipEventDestroy(g_kt.ev[i + 1]);
    }
    *total_us_out = us;
    *launches_out = n;
    *steps_out = g_kt.steps;
    g_kt.ev.clear();
    g_kt.steps = 0;
    if (err != hipSuccess) return fail(D3P_E_HIP, "kernel timing: %s", hipGetErrorString(err));
    return D3P_OK;
}

int d3p_dpvi_logreg_chain_status(void* stream, const d3p_logreg_model* model, const d3p_batch_source* src, void* workspace_dev,
                                 size_t workspace_bytes, int32_t* aborted_out)
{
    D3P_REQUIRE(model && src && workspace_dev && aborted_out, "d3p_dpvi_logreg_chain_status: null pointer");
    if (workspace_bytes < d3p_dpvi_logreg_workspace(model, src)) return fail(D3P_E_WORKSPACE, "workspace too small");
    Workspace ws;
    carve(model, src, (char*)workspace_dev, &ws);
    uint32_t flag = 0;
    D3P_HIP_TRY(hipMemcpyAsync(&flag, ws.chain_bar + (size_t)(D3P_STEP_BATCH + 1) * D3P_BAR_WORDS, sizeof(flag), hipMemcpyDeviceToHost,
                               (hipStream_t)stream));
    D3P_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *aborted_out = (int32_t)flag;
    return D3P_OK;
}

int d3p_dpvi_logreg_chain_grid(const d3p_logreg_model* model, const d3p_batch_source* src, int data_parallel,
                               uint32_t* workgroups_per_step_out, int32_t* waves_out)
{
    D3P_REQUIRE(model && src && workgroups_per_step_out && waves_out, "d3p_dpvi_logreg_chain_grid: null pointer");
    Ctx c = Ctx();
    if (int rc = fill_geometry(&c, model, src)) return rc;
    *workgroups_per_step_out = 0u;
    *waves_out = 0;
    if (!use_chained_steps(c)) return D3P_OK;
    const bool w16 = (!data_parallel || xchg_updater_form(c)) && chain_w16_enabled();
    if (!lean_chain_ok(c, w16)) return D3P_OK;
    *workgroups_per_step_out = w16 ? chain16_blocks(c.items_expected) : c.g.blocks;
    *waves_out = w16 ? 16 : 8;
    return D3P_OK;
}

int d3p_dpvi_logreg_run_status(void* stream, const d3p_logreg_model* model, const d3p_batch_source* src, void* workspace_dev,
                               size_t workspace_bytes, int32_t* aborted_out, int32_t* nonfinite_out)
{
    D3P_REQUIRE(model && src && workspace_dev && aborted_out && nonfinite_out, "d3p_dpvi_logreg_run_status: null pointer");
    if (workspace_bytes < d3p_dpvi_logreg_workspace(model, src)) return fail(D3P_E_WORKSPACE, "workspace too small");
    Workspace ws;
    carve(model, src, (char*)workspace_dev, &ws);
    D3P_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    // the run's last launch (k_flush) left the words in a pinned host record, tagged with this workspace's status address
    uint32_t words[2] = {0u, 0u};
    if (status_slot_read(run_status_words(ws), &words[0], &words[1])) {
        *aborted_out = (int32_t)words[0];
        *nonfinite_out = (int32_t)words[1];
        return D3P_OK;
    }
    D3P_HIP_TRY(hipMemcpy(words, run_status_words(ws), sizeof(words), hipMemcpyDeviceToHost));
    *aborted_out = (int32_t)words[0];
    *nonfinite_out = (int32_t)words[1];
    return D3P_OK;
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
    if ((rc = enqueue_batch_prep(c, 1))) return rc;
    const bool fused = use_fused_step(c);
    Ctx ct = c;  // the fused step applies an update in its prologue: let it work on a scratch copy of the state
    d3p_dpsvi_state st_scratch = *c.st;
    if (fused) {
        const size_t P = (size_t)c.P;
        D3P_HIP_TRY(hipMemcpyAsync(c.ws.scratch_state, c.st->params, P * sizeof(float), hipMemcpyDeviceToDevice, c.s));
        D3P_HIP_TRY(hipMemcpyAsync(c.ws.scratch_state + P, c.st->adam_m, P * sizeof(float), hipMemcpyDeviceToDevice, c.s));
        D3P_HIP_TRY(hipMemcpyAsync(c.ws.scratch_state + 2 * P, c.st->adam_v, P * sizeof(float), hipMemcpyDeviceToDevice, c.s));
        D3P_HIP_TRY(hipMemsetAsync(c.ws.acc, 0, 3 * (size_t)D3P_ACC_R * D3P_ACC_COLS(P) * sizeof(long long), c.s));
        D3P_HIP_TRY(reset_status_words(c));
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
        return enqueue_main(c, 0, X_dev, y_dev, nullptr, stamps, a0, a1);
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
        const int SS = (dev_dbg_flags() & 32) ? 8 : 2;
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
