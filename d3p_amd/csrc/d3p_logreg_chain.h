// The headline kernel: the chained DP-VI launch for Bayesian logistic regression with d = 512 features, no intercept
// (D = 512 latents, P = 1024 parameters) -- BASELINE configs[1] / the north_star workload -- written for nothing else.
//
// k_logreg_main<.., MODE 3, ..> (d3p_logreg_kernel.h) is one template for every shape, likelihood family, noise source and
// launch form; its chained instantiation carries ~20 derived pointers and two argument structs through the whole kernel:
// 106 SGPRs with ~390 v_readlane / v_writelane spill moves and 56 kernel-argument reloads (s_load + s_waitcnt lgkmcnt(0))
// scattered over 37 KB of code, many of them between the release of the previous step and this workgroup's arrival -- the
// serial chain that bounds the step (phase stamps, profiles/r02_chain_anatomy.txt: 3.4 us for the gradients of two examples
// and 1.2 us for an 8-row LDS reduction, each ~5 x their instruction count).  This kernel does the same step with the same
// protocol (fixed-point accumulator replicas, group arrival counters + flags, bounded waits, status words) and the same
// arithmetic, but keeps only what the step needs live:
//
//   phase 0  parameter-independent, BEFORE the release of step t - 1: index -> row -> features of the wave's two examples,
//            their threefry sample keys, their guide noise (svi.py:289-290) into the wave's own LDS row, the Gaussian-mechanism
//            normals and bias corrections of the pending update;
//   phase 1  wave 0 polls the 8 group flags of step t - 1 (+ the abort flag) with one load per round; workgroup barrier;
//   phase 2  update prologue, redundantly per workgroup: thread e owns latent e -- columns e (auto_loc) and D + e (auto_scale) --
//            sums the 4 replicas, mean, noise, rescale (svi.py:343-375), Adam (svi.py:379-393), derived columns into LDS;
//            workgroup 0 publishes the state;
//   phase 3  per example: z, logit (DPP wave sum), gradient, joint norm, clip factor, accumulate (svi.py:238-346);
//   phase 4  LDS reduction of the 8 waves, 2 fixed-point int64 atomics per thread, arrival.
//
// Geometry, two forms (template parameter W_):
//   W = 8  (round 2; still the data-parallel XCHG form): 8-wave workgroups, two resident per CU (<= 128 VGPRs), every wave takes
//          at most two items: k1 = 8 bid + wave, k2 = k1 + 8 nw; the host picks nw >= ceil(items / 16); one more workgroup per
//          step runs the key chain (and two more the exchange).
//   W = 16 (round 3, the single-rank default): 16-wave workgroups, ONE per CU, nw = ceil(items / 32) -- 128 workgroups per step
//          at B = 4096, so two consecutive steps are resident side by side on the two halves of the chip (the pipelining moves
//          from "two workgroups per CU" to "two steps per chip"): half the accumulator atomics (131 k instead of 262 k per
//          step) and half the redundant prologue reads, the cost that scales with the workgroup count; a CU never holds a
//          noise-generating workgroup beside one on the step's critical path; a thread owns ONE parameter column (prologue and
//          reduction); the key-chain step rides at the tail of workgroup 0 (no extra workgroup: 2 x 128 = the chip's 256 CUs).
//          Waves with more than two items (B = 32768: eight) take the further ones in pairs, noise generated in the loop.
// PLIST: the items are the entries of the step's dense owned-position list (Poisson padding, row-sharded ranks).
//
// Data-parallel runs (XCHG, W = 16; round 4) -- the UPDATER form.  What a data-parallel step adds to the serial chain between two
// steps is the sum over the ranks.  Round 3 ran it as: arrival flags -> poll (workgroups 0 / 1) -> fold the replicas -> tagged
// stores to every inbox -> poll the own inbox -> sum -> store the sums -> wait for the acknowledgement -> barrier -> flag -> poll
// (next step) -> load the sums -> update, redundantly in every workgroup: 3.9 us on top of the single-rank step.  Now:
//   * the workgroup whose arrival completes the step (it knows from the value its arrival atomic returns: no flag, no poll) does
//     the exchange in its tail, one accumulator column per thread, no barrier and no LDS: fold (the replica loads, the state
//     words and the step's normals in ONE round trip) -> tagged stores to every inbox -> tagged loads from the own inbox until
//     the world's rows are there -> noise, rescale, Adam (svi.py:343-393) ONCE -> the new parameter as ONE tagged 8-byte word
//     {fp32 | epoch} (ll_state), m and v likewise for the next updater -- nothing is waited for after the stores;
//   * the next step's workgroups poll the parameter word of their own column: the poll IS the load, and the 127 other workgroups
//     no longer repeat the update (no replica / m / v loads, no Adam arithmetic between the release and the staging barrier).
// No update is pending across launch boundaries or at the end of a run (k_run_init writes the tagged state, k_flush unpacks it).
#pragma once
#include "d3p_logreg_kernel.h"

namespace d3p {

#define D3P_CHAIN_D 512
#define D3P_CHAIN_W 8

// In-launch exchange of a data-parallel run (world > 0): one more workgroup per step waits for the step's arrivals, folds the
// rank's accumulator replicas, writes the folded row into every rank's inbox (d3p_xchg_*: system-scope stores over xGMI),
// waits for the world's rows, leaves their sum in replica 0 of the step's accumulator (replicas 1.. zeroed) and raises the
// step's exchange flag -- which is what the NEXT step's workgroups then wait for instead of the arrival flags.  The next
// step's parameter-independent work (row gathers, noise) overlaps the exchange; there is no launch boundary per step.
#define D3P_XCHG_MAX_WORLD 16
#define D3P_XCHG_WGS 2  // exchange workgroups per step, each with its share of the columns: a lane keeps its share of a rank's
                        // row in flight while it waits (four registers per value), and ONE workgroup's share of 17 values per
                        // lane took the kernel from 77 to 143 VGPRs -- past the 128 that two resident workgroups per CU allow
struct XchgDev {
    int world, rank;                   // world == 0: no exchange
    char* peer[D3P_XCHG_MAX_WORLD];    // inboxes: ll[2][world][words] of 16 bytes (see xchg_ll_store)
    unsigned long long epoch0;         // exchanges done before this launch; step t of the launch is exchange epoch0 + t + 1
    uint32_t* xflag;                   // K x D3P_XCHG_WGS flags, 32 words apart, zeroed before the launch
    long long* xsum;                   // 3 x cols: the world's sums of step g in row g % 3 -- ONE row, which is all the next
                                       // step's prologues read in a data-parallel run (not the 4 local replicas)
    int self_trip;                     // updater form, developer switch (D3P_XCHG_SELF_TRIP=1): the rank's OWN row takes the trip through
                                       // its inbox too, like a peer's -- on one GPU a stand-in for the link trip of a real peer's row
};

// One int64 of an exchange message travels as two 8-byte words {32 data bits | 32-bit tag}, tag = the low word of the
// exchange's epoch.  An aligned 8-byte store is performed as a whole, so a receiver that finds the epoch's tag in both words
// has the value: no flag behind the data, hence no wait for the write acknowledgement between the two and no second trip over
// the link (the low-latency protocol of RCCL's small messages).  The slot last held the tag of epoch - 2, or 0 at the start
// (epochs count from 1), never the one waited for.
__device__ __forceinline__ void xchg_ll_store(char* inbox, size_t value_index, long long v, uint32_t tag)
{
    unsigned long long* w = reinterpret_cast<unsigned long long*>(inbox) + 2 * value_index;
    const unsigned long long t = (unsigned long long)tag << 32;
    __hip_atomic_store(w, t | (unsigned long long)(uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 1, t | ((unsigned long long)v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void xchg_ll_fetch(const char* inbox, size_t value_index, unsigned long long* w0, unsigned long long* w1)
{
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(inbox) + 2 * value_index;
    *w0 = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *w1 = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ bool xchg_ll_valid(unsigned long long w0, unsigned long long w1, uint32_t tag)
{
    return (uint32_t)(w0 >> 32) == tag && (uint32_t)(w1 >> 32) == tag;
}

__device__ __forceinline__ long long xchg_ll_value(unsigned long long w0, unsigned long long w1)
{
    return (long long)((w1 << 32) | (w0 & 0xffffffffull));
}

// The per-element arithmetic of an example runs on PAIRS of adjacent elements: a lane's 4 + 4 elements of every operand come out
// of 16-byte loads (table row, LDS columns) or are produced in element order (noise), so elements (0, 1) and (2, 3) already sit in
// adjacent registers and v_pk_fma_f32 / v_pk_mul_f32 process two of them per issue slot without a single move.  Component-wise
// IEEE: bit for bit the scalar formulation, element by element; only the order of the per-lane partial sums changes.
typedef float d3p_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d3p_v2f pk_fma(d3p_v2f a_, d3p_v2f b_, d3p_v2f c_) { return __builtin_elementwise_fma(a_, b_, c_); }
__device__ __forceinline__ d3p_v2f pk_splat(float x) { return d3p_v2f{x, x}; }
struct Quad { d3p_v2f lo, hi; };   // four adjacent elements
__device__ __forceinline__ Quad quad_of(const float4& f) { return Quad{d3p_v2f{f.x, f.y}, d3p_v2f{f.z, f.w}}; }
__device__ __forceinline__ Quad quad_lds(const float* p_) { return quad_of(*reinterpret_cast<const float4*>(p_)); }
// sigmoid and log(1 + exp(-|t|)) on the hardware's exp2 / log2 / rcp: 1 + exp(-|t|) is in (1, 2], a normal number, so the raw
// v_log_f32 needs none of the denormal handling __logf carries (ten instructions on the step's critical path)
__device__ __forceinline__ float chain_sigmoid(float t) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * t)); }
__device__ __forceinline__ float chain_softplus(float t)
{
    return fmaxf(t, 0.0f) + 0.693147180559945309f * __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(t)));
}

struct ChainArgs {
    const float* X;
    const float* y;
    const uint32_t* idx_base;    // K x B
    const uint32_t* skeys_base;  // K x 2B
    const uint32_t* plist_base;  // K x B (PLIST)
    const float* noise_base;     // K x P
    StepSlot* slots;             // K slots of this batch
    const StepSlot* prev_slot0;  // slot of step g0 - 1 (nullptr: nothing to apply before step 0)
    const float* prev_noise0;
    long long* acc_base;         // 3 x R x D3P_ACC_COLS(P)
    float* state[2][3];          // ping-ponged {params, m, v}
    float* losses;               // nullable
    uint32_t* bar;               // K x D3P_BAR_WORDS arrival counters + chain progress word
    uint32_t* status;            // [0] abort, [1] non-finite
    Sched* chain_sched;
    StepSlot* chain_slots;       // slots of the NEXT batch (key chain), K_next of them
    int32_t* adam_step;
    uint32_t* batch_index;
    unsigned long long* stamps;  // STAMPS
    uint64_t row_lo;
    double sg, inv_sg;
    uint32_t B;
    int nw, g0, K, K_next;
    float A_scale, c1, hz, inv_obs, lik_scale, obs_scale, clip, dp_scale, lr, b1, b2, adam_eps, log_prior;
    float c1_b, hz_b, log_prior_b;  // ICPT: the intercept's prior
    int gexp;
    XchgDev x;
    // data-parallel 16-wave form (UPD): the optimiser state as self-validating 8-byte words {fp32 bits | tag}, tag = low word of
    // the exchange epoch of the step the value is FOR.  [0]: parameters, two rows of D3P_ACC_COLS(P) words (row = tag & 1: the
    // updater of step e writes row (e + 1) & 1 while row e & 1 may still be read); [1], [2]: Adam's m and v, one row each,
    // rewritten in place (only the updaters read them, one after the other)
    unsigned long long* ll_state[3];
    int dbg;  // developer switches (D3P_DBG): 2 = raised wave priority on the critical path, 4 = no gradient atomics (STAMPS only)
};

// ICPT: the same 512 features plus the intercept (examples/logistic_regression.py:49-66): D = 513 latents.  jax's iota
// layout for an odd D pads the counter array, so the pairs of one threefry call are (c, c + 257), c < 256, and (256, pad):
// lanes own latents 4 l .. 4 l + 3 and 257 + 4 l .. 260 + 4 l (the very last one, 512, is the intercept: x = 1, its own
// prior), and latent 256 -- the "tail" -- is a feature column every lane carries redundantly (one more threefry call per
// example, added once after the wave sums).  In LDS the second half is stored from index 260 on, so that its 16-byte
// reads stay aligned: latent j lives at j (j <= 256) or j + 3 (j >= 257), arrays are 520 long.
#define D3P_CHAIN_DL(ICPT) ((ICPT) ? 520 : 512)
static inline size_t chain_lds_bytes(bool icpt, int W = D3P_CHAIN_W)
{
    return (size_t)(5 * D3P_CHAIN_DL(icpt) + W * 2 * D3P_CHAIN_DL(icpt) + 2 * W + 4 + 32) * sizeof(float);
}

// XCHG: the data-parallel form (a.x.world > 0) -- its own instantiation, so that the single-rank kernel does not carry the
// exchange workgroup's registers (61 SGPRs / 77 VGPRs alone, 82 / 80 with it: 2-3 % of the single-rank step)
// W_: waves per workgroup (8 or 16, see the header); RU: accumulator replicas in use (<= D3P_ACC_R; the others stay zero)
template <bool PLIST, bool STAMPS, bool ICPT = false, bool XCHG = false, int W_ = D3P_CHAIN_W, int RU = D3P_ACC_R>
__global__ void __launch_bounds__(64 * W_) k_logreg_chain(ChainArgs a)
{
    static_assert(W_ == 8 || W_ == 16, "8- or 16-wave workgroups");
    static_assert(RU >= 1 && RU <= D3P_ACC_R, "replicas in use");
    constexpr bool W16 = W_ == 16;
    constexpr bool UPD = XCHG && W16;   // data-parallel updater form (see the header)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int DF = D3P_CHAIN_D;                 // feature columns of a table row
    constexpr int D = DF + (ICPT ? 1 : 0), P = 2 * D, PA = D3P_ACC_COLS(P), W = W_, R = D3P_ACC_R;
    constexpr int HALF = ICPT ? 257 : 256;          // second index of a threefry pair = first + HALF
    constexpr int DL = D3P_CHAIN_DL(ICPT);          // LDS array length per quantity
    constexpr int C1 = ICPT ? 260 : 256;            // LDS index of the lane-0 element of the second half
    constexpr int TL = 256;                         // ICPT: latent / LDS index of the tail column
    auto lix = [](int j) { return (!ICPT || j <= 256) ? j : j + 3; };  // latent -> LDS index
    float* pk = lds;                    // [loc | s | sg | q] x DL, then the waves' shares of sum_j lc_j (W floats)
    float* red = lds + 5 * DL;          // W rows of 2 DL floats: the waves' noise, later their partial sums
    float* tail = red + W * 2 * DL;     // 2 W: loss / count per wave
    uint32_t* okw = reinterpret_cast<uint32_t*>(tail + 2 * W);             // verdict of the polling wave
    unsigned long long* stamp = reinterpret_cast<unsigned long long*>(okw + 4);
#define D3P_CSTAMP(k) if (STAMPS && threadIdx.x == 0 && !((a.dbg & 256) && (k) != 5 && (k) >= 5) && (a.dbg & 0x700) != 0x700) stamp[k] = wall_clock64();
    D3P_CSTAMP(0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (W = 16: no key-chain workgroup, and the exchange of a data-parallel run rides in the tails of workgroups 0 and 1)
    const uint32_t per = (uint32_t)a.nw + (W16 ? 0u : 1u) + ((XCHG && !W16) ? (uint32_t)D3P_XCHG_WGS : 0u);
    const int step_t = (int)(blockIdx.x / per);
    const uint32_t bid = blockIdx.x % per;

    if (XCHG && !W16 && bid > (uint32_t)a.nw) {  // ---- exchange workgroup `xj` of step `step_t`: columns c_lo .. c_lo + CH - 1
        constexpr int CH = (PA + D3P_XCHG_WGS - 1) / D3P_XCHG_WGS;
        const int xj = (int)(bid - (uint32_t)a.nw - 1u), c_lo = xj * CH, cn = (PA - c_lo < CH ? PA - c_lo : CH);
        const size_t words = (size_t)R * PA;
        long long* acc = a.acc_base + (size_t)((a.g0 + step_t) % 3) * words;
        if (tid == 64) okw[1] = 0u;
        if (wave == 0) {  // every compute workgroup of the step has added its sums
            const uint32_t ng = (uint32_t)a.nw < D3P_BAR_GROUPS ? (uint32_t)a.nw : D3P_BAR_GROUPS;
            const bool ok = chain_wait_groups(a.bar + (size_t)step_t * D3P_BAR_WORDS + D3P_BAR_LINE * (1 + D3P_BAR_GROUPS), ng, a.status,
                                              abort_code(D3P_ABORT_XCHG_ARRIVALS, step_t, (uint32_t)xj), D3P_WAIT_ROUNDS_PEERS);
            if (lane == 0) okw[0] = ok ? 0u : 1u;
        }
        __syncthreads();
        if (okw[0] != 0u) return;
        const unsigned long long epoch = a.x.epoch0 + (unsigned long long)step_t + 1ull;
        const unsigned parity = (unsigned)(epoch & 1ull);
        const uint32_t tag = (uint32_t)epoch;
        unsigned long long* tot = reinterpret_cast<unsigned long long*>(red);  // CH int64: the world's sums, built in LDS
        for (int c = tid; c < CH; c += 64 * W) tot[c] = 0ull;
        {  // fold the replicas (all loads of a thread in flight together), deliver the row to every inbox
            constexpr int NC = (CH + 64 * W - 1) / (64 * W);
            long long v[NC][R];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int c = tid + i * 64 * W;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    v[i][r] = c < cn ? __hip_atomic_load(acc + (size_t)r * PA + c_lo + c, __ATOMIC_RELAXED, D3P_AGENT) : 0ll;
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int c = tid + i * 64 * W;
                long long s = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) s += v[i][r];
                if (c < cn)
                    for (int p = 0; p < a.x.world; ++p)
                        xchg_ll_store(a.x.peer[p], ((size_t)parity * a.x.world + a.x.rank) * PA + c_lo + c, s, tag);
            }
        }
        __syncthreads();  // (tot zeroed)
        // the world's rows: wave w takes ranks w, w + W; a lane keeps all of its columns of one row in flight and asks again
        // until every word carries this epoch's tag
        for (int p = wave; p < a.x.world; p += W) {
            constexpr int NL = (CH + 63) / 64;
            const size_t row = ((size_t)parity * a.x.world + p) * PA + c_lo;
            unsigned long long w0[NL], w1[NL];
            bool ok = false;
            for (uint32_t spins = 0; spins < 2u * D3P_WAIT_ROUNDS_PEERS; ++spins) {  // (a round here is ~ 0.1 us)
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const int c = lane + 64 * i;
                    xchg_ll_fetch(a.x.peer[a.x.rank], row + (c < cn ? c : cn - 1), &w0[i], &w1[i]);
                }
                bool all = true;
#pragma unroll
                for (int i = 0; i < NL; ++i) all = all && xchg_ll_valid(w0[i], w1[i], tag);
                if (all) { ok = true; break; }
                if ((spins & 63u) == 63u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, D3P_AGENT) != 0u) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) {
                okw[1] = 1u;
                chain_raise(a.status, abort_code(D3P_ABORT_XCHG_ROW, step_t, (uint32_t)p));
            } else {
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const int c = lane + 64 * i;
                    if (c < cn) atomicAdd(tot + c, (unsigned long long)xchg_ll_value(w0[i], w1[i]));
                }
            }
        }
        __syncthreads();
        if (okw[1] != 0u) return;  // aborted: the run stops here (status[0])
        long long* xrow = a.x.xsum + (size_t)((a.g0 + step_t) % 3) * PA + c_lo;
        for (int c = tid; c < cn; c += 64 * W) __hip_atomic_store(xrow + c, (long long)tot[c], __ATOMIC_RELAXED, D3P_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(a.x.xflag + ((size_t)step_t * D3P_XCHG_WGS + xj) * D3P_BAR_LINE, 1u, __ATOMIC_RELAXED, D3P_AGENT);
        return;
    }

    if (!W16 && bid == (uint32_t)a.nw) {  // key-chain workgroup: split(state_key, 3) of step `step_t` of the NEXT batch
        if (step_t < a.K_next && tid < 64) {
            uint32_t* progress = a.bar + (size_t)a.K * D3P_BAR_WORDS;
            bool go = true;
            if (step_t > 0)
                go = chain_wait(progress, (uint32_t)step_t, a.status, abort_code(D3P_ABORT_KEY_CHAIN, step_t),
                                XCHG ? D3P_WAIT_ROUNDS_PEERS : D3P_WAIT_ROUNDS);
            if (go) chain_step<true>(a.chain_sched, a.chain_slots + step_t, step_t, step_t == a.K_next - 1);
            __builtin_amdgcn_s_waitcnt(0);
            if (tid == 0) __hip_atomic_store(progress, (uint32_t)step_t + 1u, __ATOMIC_RELAXED, D3P_AGENT);
        }
        return;
    }

    // ------------------------------------------------------------------ phase 0: parameter-independent
    const StepSlot* slot = a.slots + step_t;
    const uint32_t n_items = PLIST ? slot->n_owned : a.B;
    const uint32_t k1 = bid * W + (uint32_t)wave, k2 = k1 + (uint32_t)a.nw * W;
    const bool live1 = k1 < n_items, live2 = k2 < n_items;
    const uint32_t* idx = a.idx_base + (size_t)step_t * a.B;
    const uint32_t* skeys = a.skeys_base + (size_t)step_t * 2 * a.B;
    float4 xa0 = make_float4(0.f, 0.f, 0.f, 0.f), xa1 = xa0, xb0 = xa0, xb1 = xa0;
    float ya = 0.f, yb = 0.f, xta = 0.f, xtb = 0.f;  // xt*: ICPT, the tail feature
    uint32_t ka0 = 0, ka1 = 0, kb0 = 0, kb1 = 0;
    // features of one table row for this lane: first half 16-byte aligned; ICPT: the second half starts one float off
    // (scalar loads), its last element is the intercept's constant 1
    auto load_x = [&](const float* xr, float4& x0, float4& x1, float& xt) {
        x0 = *reinterpret_cast<const float4*>(xr + 4 * lane);
        if (ICPT) {
            const int f = HALF + 4 * lane;
            x1 = make_float4(xr[f], xr[f + 1], xr[f + 2], f + 3 < DF ? xr[f + 3] : 1.0f);
            xt = xr[TL];
        } else {
            x1 = *reinterpret_cast<const float4*>(xr + HALF + 4 * lane);
        }
    };
    // The table rows are requested near the END of the noise generation, not in front of it: a workgroup enters
    // when a workgroup of step t - 2 leaves, i.e. around the moment step t - 1 is released, and 128 entering workgroups asking
    // for 8.4 MB at once put that burst on the memory system exactly while the other step's workgroups poll for their release
    // and fetch the sums -- the part of the chain that is pure memory latency.  The rows are not needed before phase 3.
    size_t row_a = 0, row_b = 0;
    if (live1) {
        const uint32_t p = PLIST ? (a.plist_base + (size_t)step_t * a.B)[k1] : k1;
        row_a = (size_t)((uint64_t)idx[p] - a.row_lo);
        ka0 = skeys[2 * p];
        ka1 = skeys[2 * p + 1];
    }
    if (live2) {
        const uint32_t p = PLIST ? (a.plist_base + (size_t)step_t * a.B)[k2] : k2;
        row_b = (size_t)((uint64_t)idx[p] - a.row_lo);
        kb0 = skeys[2 * p];
        kb1 = skeys[2 * p + 1];
    }
    auto fetch_rows = [&]() {
        if (live1) { load_x(a.X + row_a * DF, xa0, xa1, xta); ya = a.y[row_a]; }
        if (live2) { load_x(a.X + row_b * DF, xb0, xb1, xtb); yb = a.y[row_b]; }
    };
    // the pending update of step g - 1: which state buffers, which slot / noise row (all known before the release)
    const int g = a.g0 + step_t;
    const StepSlot* ps = step_t > 0 ? a.slots + (step_t - 1) : a.prev_slot0;
    const bool apply_prev = !UPD && ps != nullptr;   // (UPD: the step's updater has already applied it)
    const float* prev_noise = step_t > 0 ? a.noise_base + (size_t)(step_t - 1) * P : a.prev_noise0;
    // W = 16: thread tid owns ONE parameter column -- auto_loc of latent tid (tid < 512) or auto_scale of latent tid - 512
    // (ICPT: the intercept's two columns D - 1 and 2 D - 1 are second columns of threads 64 and 128)
    const int mycol = W16 ? (tid < DF ? tid : D + (tid - DF)) : tid;
    // ICPT: the intercept's two columns are second columns of two threads of LOC waves (waves 1 and 2: their own column is the cheap
    // kind -- no softplus / sigmoid / logarithms --, so the wave that also takes the intercept's scale column ends 25 instructions
    // after the scale waves instead of 70: the staging barrier waits for the slowest wave)
    const int xcol = (W16 && ICPT) ? (tid == 64 ? D - 1 : tid == 128 ? 2 * D - 1 : -1) : -1;
    float zL = 0.f, zS = 0.f, zX = 0.f, bc1 = 1.f, bc2 = 1.f;
    // W = 16: everything of the pending update that does not depend on the sums is computed HERE, before the release: the valid
    // example count of the pending step is a function of the keys (the sampler left it in the step's slot; the count column of
    // the accumulator is only looked at for the non-finite marker), so svi.py:305's factor, the noise scale of svi.py:365 and
    // Adam's bias corrections -- five divisions -- leave the step's critical path.
    float pre_n = 0.f, pre_factor = 0.f, pre_noise_scale = 0.f, pre_out_scale = 0.f, pre_inv_B = 0.f, pre_inv_bc1 = 1.f, pre_inv_bc2 = 1.f;
    if (apply_prev) {  // Gaussian-mechanism normals of this thread's column(s) and the bias corrections
        zL = prev_noise[mycol];
        if (!W16) zS = prev_noise[D + tid];
        if (xcol >= 0) zX = prev_noise[xcol];
        bc1 = ps->bc1;
        bc2 = ps->bc2;
        if (W16) {
            pre_n = (float)ps->counts[1];
            const float Bf = (float)a.B;
            pre_factor = (pre_n == 0.0f) ? 0.0f : Bf / pre_n;  // svi.py:305
            pre_inv_B = 1.0f / Bf;
            pre_inv_bc1 = 1.0f / bc1;
            pre_inv_bc2 = 1.0f / bc2;
            pre_noise_scale = a.dp_scale * (a.clip / pre_n);   // svi.py:365-375
            pre_out_scale = a.obs_scale * pre_factor;
        }
    }
    // guide noise of both examples, kept in registers (round 3; round 2 parked it in the wave's row of the reduction buffer: 4 KB
    // of LDS reads per wave on the step's critical path, where all 16 waves of a CU read at once): lane owns columns
    // 4 lane .. + 3 and D/2 + the same, i.e. the pairs (c, c + D/2) of jax's iota layout come out of ONE threefry2x32 call
    float* er = red + (size_t)wave * 2 * DL;   // the wave's row of the reduction buffer (partial sums, phase 4)
    struct Eps { Quad v0, v1; float vt, e2; };  // v0 / v1: the lane's 4 + 4 elements; vt: ICPT, the tail latent's noise; e2: the lane's share of -0.5 |eps|^2
    // (e2: the log q term of the loss is parameter-independent, so it is summed here)
    auto gen = [&](uint32_t k0, uint32_t k1_, auto&& halfway) {   // halfway(): called in front of the last of the four pairs
        const uint32_t s0 = __builtin_amdgcn_readfirstlane(k0), s1 = __builtin_amdgcn_readfirstlane(k1_);  // wave-uniform keys
        Eps o;
        float e2 = 0.f, w0[4], w1[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (n == 3) halfway();
            uint32_t b0, b1;
            threefry2x32(s0, s1, (uint32_t)(4 * lane + n), (uint32_t)(4 * lane + n + HALF), b0, b1);
            w0[n] = bits_to_normal_wu(b0);
            w1[n] = bits_to_normal_wu(b1);
            e2 = __fmaf_rn(w0[n], w0[n], e2);
            e2 = __fmaf_rn(w1[n], w1[n], e2);
        }
        o.v0 = Quad{d3p_v2f{w0[0], w0[1]}, d3p_v2f{w0[2], w0[3]}};
        o.v1 = Quad{d3p_v2f{w1[0], w1[1]}, d3p_v2f{w1[2], w1[3]}};
        o.vt = 0.f;
        if (ICPT) {  // the tail latent: word 256 pairs with the padding of the odd counter array (jax pads with a zero)
            uint32_t b0, b1;
            threefry2x32(s0, s1, (uint32_t)TL, 0u, b0, b1);
            o.vt = bits_to_normal_wu(b0);
        }
        o.e2 = -0.5f * e2;
        return o;
    };
    Eps epa = {}, epb = {};
    // (where in phase 0 the rows are requested, same box, us per step: in front of the noise 7.00 - 7.07, between the two examples'
    // noise 6.98, three quarters through 6.90, HERE -- in front of the last of the eight pairs -- 6.87 - 6.91, behind all of it
    // 6.95 - 6.99: later, and the burst meets the other step's accumulator atomics instead)
    auto nothing = [] {};
    if (live1) epa = gen(ka0, ka1, nothing);
    if (live2) epb = gen(kb0, kb1, [&] { fetch_rows(); });
    else fetch_rows();
    // W = 16, a step with up to THREE items per wave (an owned-position list a little longer than 2 x 16 x nw: a Poisson batch
    // padded to its 0.99 quantile, a row-sharded rank's share of the batch -- Binomial(B, 1 / world) around 4096 +- 60, i.e. every
    // other step of a data-parallel run): the third item is prepared HERE too -- its table row and its noise parked in the wave's
    // own row of the reduction buffer, which is free until phase 4 -- instead of in phase 3, where index -> row -> features and
    // 2 us of noise generation sat on the step's critical path (anatomy, round 4: the workgroups with third items arrived 3.9 us
    // after the others).  And the third items are dealt out ACROSS the workgroups (item 2 nw W + wave nw + bid: one per workgroup
    // before any workgroup gets a second), not to the 16 waves of workgroup 0 first.
    const uint32_t stride = (uint32_t)a.nw * W;
    const bool third_pre = W16 && n_items > 2u * stride && n_items <= 3u * stride;
    const uint32_t k3 = 2u * stride + (uint32_t)wave * (uint32_t)a.nw + bid;
    const bool live3 = third_pre && k3 < n_items;
    float y3 = 0.f, xt3 = 0.f, e2_3 = 0.f, vt3 = 0.f;
    if (live3) {
        const uint32_t p3 = PLIST ? (a.plist_base + (size_t)step_t * a.B)[k3] : k3;
        const size_t row3 = (size_t)((uint64_t)idx[p3] - a.row_lo);
        const uint32_t q0 = skeys[2 * p3], q1 = skeys[2 * p3 + 1];
        float4 x30, x31;
        load_x(a.X + row3 * DF, x30, x31, xt3);
        y3 = a.y[row3];
        const Eps e3 = gen(q0, q1, nothing);
        *reinterpret_cast<float4*>(er + 4 * lane) = make_float4(e3.v0.lo.x, e3.v0.lo.y, e3.v0.hi.x, e3.v0.hi.y);
        *reinterpret_cast<float4*>(er + C1 + 4 * lane) = make_float4(e3.v1.lo.x, e3.v1.lo.y, e3.v1.hi.x, e3.v1.hi.y);
        *reinterpret_cast<float4*>(er + DL + 4 * lane) = x30;
        *reinterpret_cast<float4*>(er + DL + C1 + 4 * lane) = x31;
        e2_3 = e3.e2;
        vt3 = e3.vt;
    }
    // UPD: what an updater's tail needs of the step's slot -- svi.py:305, :365-375 and Adam's bias corrections, functions of the keys
    // (the sampler left them in the slot) -- is read and computed HERE by every workgroup: in the tail these scalar loads (first touch:
    // a memory round trip) sat between the arrival and the publication (round-4 anatomy: 1 us)
    float u_factor0 = 0.f, u_inv_B = 0.f, u_inv_bc1 = 1.f, u_inv_bc2 = 1.f, u_noise_scale = 0.f, u_out_scale0 = 0.f;
    if (UPD) {
        const StepSlot* ms = a.slots + step_t;
        const float n_valid = (float)ms->counts[1], Bf = (float)a.B;
        u_factor0 = (n_valid == 0.0f) ? 0.0f : Bf / n_valid;
        u_inv_B = 1.0f / Bf;
        u_inv_bc1 = 1.0f / ms->bc1;
        u_inv_bc2 = 1.0f / ms->bc2;
        u_noise_scale = a.dp_scale * (a.clip / n_valid);
        u_out_scale0 = a.obs_scale * u_factor0;
        // ... and the step's Gaussian-mechanism normals (written by the sampler launch, cold by now) are pulled into this XCD's L2:
        // an updater's load of its normal is then no slower than its other loads
        const float zwarm = (a.noise_base + (size_t)step_t * P)[tid < P ? tid : 0];
        asm volatile("" :: "v"(zwarm));
    }
    if (UPD) {
        // the kernel-argument lines an updater's tail reads (the exchange block, the state words): first touched HERE, so that the tail's
        // scalar loads hit the scalar cache instead of costing a memory round trip between the arrival and the publication
        asm volatile("" :: "s"(a.x.world), "s"(a.x.rank), "s"(a.x.self_trip), "s"(a.x.peer[0]), "s"(a.x.peer[D3P_XCHG_MAX_WORLD / 2]),
                     "s"(a.x.peer[D3P_XCHG_MAX_WORLD - 1]), "s"(a.ll_state[1]), "s"(a.ll_state[2]), "s"(a.inv_sg), "s"(a.b1), "s"(a.adam_eps),
                     "s"(a.losses), "s"(a.adam_step), "s"(a.batch_index));
    }
    D3P_CSTAMP(8)
    // From here on the workgroup is on the critical path of the step (D3P_DBG=2: raised wave priority against the co-resident
    // workgroup of the next step, which is generating its noise on the same SIMDs).
    if (a.dbg & 2) __builtin_amdgcn_s_setprio(3);

    // ------------------------------------------------------------------ phase 1: release of step t - 1
    const size_t words = (size_t)R * PA;
    long long* acc_prev = a.acc_base + (size_t)((g + 2) % 3) * words;
    long long* acc_cur = a.acc_base + (size_t)(g % 3) * words;
    long long* acc_next = a.acc_base + (size_t)((g + 1) % 3) * words;
    if (!UPD && wave == 0) {   // (UPD: no release to wait for -- every thread polls its own parameter word, phase 2)
        const uint32_t ng = (uint32_t)a.nw < D3P_BAR_GROUPS ? (uint32_t)a.nw : D3P_BAR_GROUPS;
        bool ok;
        if (XCHG)  // data-parallel: the previous step's exchange flag (global sums in place) instead of its arrival flags
            ok = chain_wait_groups(step_t > 0 ? a.x.xflag + (size_t)(step_t - 1) * D3P_XCHG_WGS * D3P_BAR_LINE : nullptr, D3P_XCHG_WGS, a.status,
                                   abort_code(D3P_ABORT_RELEASE, step_t, 1u), D3P_WAIT_ROUNDS_PEERS);
        else
            ok = chain_wait_groups(
                step_t > 0 ? a.bar + (size_t)(step_t - 1) * D3P_BAR_WORDS + D3P_BAR_LINE * (1 + D3P_BAR_GROUPS) : nullptr, ng, a.status,
                abort_code(D3P_ABORT_RELEASE, step_t));
        if (lane == 0) okw[0] = ok ? 0u : 1u;
    }
    if (!UPD) {
        __syncthreads();
        if (okw[0] != 0u) return;  // the run was aborted: no update, no publication, no arrival
        D3P_CSTAMP(7)
    }

    // ------------------------------------------------------------------ phase 2: update prologue (thread e <-> latent e)
    const int in = g > 0 ? ((g - 1) & 1) : 0, out = g & 1;
    uint32_t upd_bad = 0u;  // UPD: this wave's poll ran out, or the run was stopped
    float lc_mine = 0.f;  // this thread's share of sum_j lc_j = sum_j [log prior scale - log s_j] (example-independent loss term)
    {
        float n = 0.f, factor = 0.f;
        // nobody reads the next accumulator any more (the previous step's prologues are over; at the first step of a run
        // nobody has read it yet): zero it.  W = 16: behind the reduction barrier (below) -- the prologue's loads are counted
        // behind these stores (one in-order counter for loads and stores), so the workgroups that have any (0 - 4) waited for a
        // write acknowledgement in the step's latency chain
        if (!W16)
            for (int i = (int)bid * (64 * W) + tid; i < R * PA; i += a.nw * 64 * W)
                __hip_atomic_store(acc_next + i, 0ll, __ATOMIC_RELAXED, D3P_AGENT);
        // one latent: pending update of its two columns (e: auto_loc, D + e: auto_scale) and its derived LDS entries
        // the sums of the previous step: the local replicas, or (data-parallel) the ONE row the exchange workgroup left
        const long long* sums = XCHG ? a.x.xsum + (size_t)((g + 2) % 3) * PA : acc_prev;
        constexpr int nrep = XCHG ? 1 : RU;
        // W = 16: one parameter column c (c < D: auto_loc of latent c, else auto_scale of latent c - D): its pending update
        // and its derived LDS entries.  Same arithmetic per column as the two-column form below.
        long long nll_main = 0;  // example count of the pending step (summed in the thread's main column call, all lanes active)
        // (loads and arithmetic are separate steps so that a thread with two columns -- ICPT: the intercept's -- has the loads of
        // both in flight together: one memory round trip, not two, between the release and the staging barrier)
        struct ColData { long long s8[nrep]; float x, m, v; };
        auto col_load = [&](int c) {
            ColData d;
            if (apply_prev) {
#pragma unroll
                for (int r = 0; r < nrep; ++r) d.s8[r] = __hip_atomic_load(sums + (size_t)r * PA + c, __ATOMIC_RELAXED, D3P_AGENT);
                d.x = __hip_atomic_load(a.state[in][0] + c, __ATOMIC_RELAXED, D3P_AGENT);
                d.m = __hip_atomic_load(a.state[in][1] + c, __ATOMIC_RELAXED, D3P_AGENT);
                d.v = __hip_atomic_load(a.state[in][2] + c, __ATOMIC_RELAXED, D3P_AGENT);
            } else {
#pragma unroll
                for (int r = 0; r < nrep; ++r) d.s8[r] = 0;
                d.x = a.state[in][0][c];
                d.m = d.v = 0.f;
            }
            return d;
        };
        auto col_apply = [&](int c, const ColData& d, float z) {
            const bool is_scale = c >= D;
            const int e = is_scale ? c - D : c;
            float x = d.x;
            if (apply_prev) {
                float m = d.m, v = d.v;
                long long sm = 0;
#pragma unroll
                for (int r = 0; r < nrep; ++r) sm += d.s8[r];
                // (a workgroup that saw a non-finite partial added 2^44 to the count column: NaN from here on, like float sums)
                const float poison = nll_main >= (1ll << 40) ? __builtin_nanf("") : 0.0f;
                n = pre_n + poison;
                factor = pre_factor + poison;
                const float inv_B = pre_inv_B, inv_bc1 = pre_inv_bc1, inv_bc2 = pre_inv_bc2;
                const float noise_scale = pre_noise_scale, out_scale = pre_out_scale + poison;
                // the int64 -> double conversion as hi 2^32 + lo in ONE fused multiply-add (v_cvt_f64_i32, v_cvt_f64_u32, v_fma_f64):
                // the correctly rounded value for EVERY int64 -- the column sum over all workgroups, replicas and ranks is only
                // bounded by B 2^40 (2^55 at B = 32768), and the bit-pattern trick of fixed_point_rn (exact below 2^51 only, used
                // here in round 3) distorted any column whose clipped gradients add up beyond 2048 C: a constant feature, the
                // intercept -- a wrong gradient AND a sensitivity above the C / n the noise is calibrated for
                const double smd = i64_to_f64(sm);
                const float tot = (float)(smd * a.inv_sg);
                const float gr = __fmaf_rn(z, noise_scale, tot * inv_B) * out_scale;
                m = (1.0f - a.b1) * gr + a.b1 * m;
                v = (1.0f - a.b2) * gr * gr + a.b2 * v;
                x = x - a.lr * (m * inv_bc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v * inv_bc2) + a.adam_eps);
                if (bid == 0) {  // one workgroup publishes the state
                    __hip_atomic_store(a.state[out][0] + c, x, __ATOMIC_RELAXED, D3P_AGENT);
                    __hip_atomic_store(a.state[out][1] + c, m, __ATOMIC_RELAXED, D3P_AGENT);
                    __hip_atomic_store(a.state[out][2] + c, v, __ATOMIC_RELAXED, D3P_AGENT);
                }
            }
            const int li = lix(e);
            if (!is_scale) {
                pk[li] = x;
            } else {
                float sp, sgm;
                guide_scale(a.gexp, x, sp, sgm);
                pk[DL + li] = sp;
                pk[2 * DL + li] = sgm;
                pk[3 * DL + li] = a.inv_obs * sgm * __builtin_amdgcn_rcpf(sp);
                lc_mine += ((ICPT && e == D - 1) ? a.log_prior_b : a.log_prior) - __logf(sp);
            }
        };
        auto latent = [&](int e, float zl, float zs) {
            float xL, xS;
            if (apply_prev) {
                long long aL[nrep], aS[nrep], n8[nrep];
#pragma unroll
                for (int r = 0; r < nrep; ++r) {
                    aL[r] = __hip_atomic_load(sums + (size_t)r * PA + e, __ATOMIC_RELAXED, D3P_AGENT);
                    aS[r] = __hip_atomic_load(sums + (size_t)r * PA + D + e, __ATOMIC_RELAXED, D3P_AGENT);
                }
                xL = __hip_atomic_load(a.state[in][0] + e, __ATOMIC_RELAXED, D3P_AGENT);
                xS = __hip_atomic_load(a.state[in][0] + D + e, __ATOMIC_RELAXED, D3P_AGENT);
                float mL = __hip_atomic_load(a.state[in][1] + e, __ATOMIC_RELAXED, D3P_AGENT);
                float mS = __hip_atomic_load(a.state[in][1] + D + e, __ATOMIC_RELAXED, D3P_AGENT);
                float vL = __hip_atomic_load(a.state[in][2] + e, __ATOMIC_RELAXED, D3P_AGENT);
                float vS = __hip_atomic_load(a.state[in][2] + D + e, __ATOMIC_RELAXED, D3P_AGENT);
#pragma unroll
                for (int r = 0; r < nrep; ++r) n8[r] = __hip_atomic_load(sums + (size_t)r * PA + P + 1, __ATOMIC_RELAXED, D3P_AGENT);
                long long nll = 0, sL = 0, sS = 0;
#pragma unroll
                for (int r = 0; r < nrep; ++r) { nll += n8[r]; sL += aL[r]; sS += aS[r]; }
                // (a workgroup that saw a non-finite partial added 2^44 to the count column: NaN from here on, like float sums)
                n = nll >= (1ll << 40) ? __builtin_nanf("") : (float)nll;
                const float Bf = (float)a.B;
                factor = (n == 0.0f) ? 0.0f : Bf / n;  // svi.py:305
                const float inv_B = 1.0f / Bf, inv_bc1 = 1.0f / bc1, inv_bc2 = 1.0f / bc2;
                const float noise_scale = a.dp_scale * (a.clip / n), out_scale = a.obs_scale * factor;  // svi.py:365-375
                auto adam = [&](long long sm, float z, float& x, float& m, float& v) {
                    const float tot = (float)((double)sm * a.inv_sg);
                    const float gr = __fmaf_rn(z, noise_scale, tot * inv_B) * out_scale;
                    m = (1.0f - a.b1) * gr + a.b1 * m;
                    v = (1.0f - a.b2) * gr * gr + a.b2 * v;
                    x = x - a.lr * (m * inv_bc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v * inv_bc2) + a.adam_eps);
                };
                adam(sL, zl, xL, mL, vL);
                adam(sS, zs, xS, mS, vS);
                if (bid == 0) {  // one workgroup publishes the state
                    __hip_atomic_store(a.state[out][0] + e, xL, __ATOMIC_RELAXED, D3P_AGENT);
                    __hip_atomic_store(a.state[out][0] + D + e, xS, __ATOMIC_RELAXED, D3P_AGENT);
                    __hip_atomic_store(a.state[out][1] + e, mL, __ATOMIC_RELAXED, D3P_AGENT);
                    __hip_atomic_store(a.state[out][1] + D + e, mS, __ATOMIC_RELAXED, D3P_AGENT);
                    __hip_atomic_store(a.state[out][2] + e, vL, __ATOMIC_RELAXED, D3P_AGENT);
                    __hip_atomic_store(a.state[out][2] + D + e, vS, __ATOMIC_RELAXED, D3P_AGENT);
                }
            } else {
                xL = a.state[in][0][e];
                xS = a.state[in][0][D + e];
            }
            float sp, sgm;
            guide_scale(a.gexp, xS, sp, sgm);
            const int li = lix(e);
            pk[li] = xL;
            pk[DL + li] = sp;
            pk[2 * DL + li] = sgm;
            pk[3 * DL + li] = a.inv_obs * sgm * __builtin_amdgcn_rcpf(sp);
            lc_mine += ((ICPT && e == D - 1) ? a.log_prior_b : a.log_prior) - __logf(sp);
        };
        if (UPD) {
            // The parameters of this step: ONE tagged word per column, written by the updater of the previous step (or by
            // k_run_init / the last updater of the previous launch).  The poll is the load.  The waves leave the loop one by one
            // (a wave whose 64 words carry the tag goes on to its derived columns).  (Measured and dropped: wave 0 alone polls, the
            // others wait at a barrier and load afterwards -- fewer polls in flight, one more round trip: 8.78 vs 8.62 us per step.)
            const uint32_t etag = (uint32_t)(a.x.epoch0 + (unsigned long long)step_t + 1ull);
            const unsigned long long* xrow = a.ll_state[0] + (size_t)(etag & 1u) * PA;
            unsigned long long w1 = 0ull, w2 = 0ull;
            auto poll = [&]() {
                for (uint32_t spins = 0; spins < D3P_WAIT_ROUNDS_PEERS / 2u; ++spins) {   // (a round is a memory round trip, ~ 0.7 us: 25 s)
                    w1 = __hip_atomic_load(xrow + mycol, __ATOMIC_RELAXED, D3P_AGENT);
                    if (xcol >= 0) w2 = __hip_atomic_load(xrow + xcol, __ATOMIC_RELAXED, D3P_AGENT);
                    const bool valid = (uint32_t)(w1 >> 32) == etag && (xcol < 0 || (uint32_t)(w2 >> 32) == etag);
                    if (__ballot(!valid) == 0ull) return true;
                    if ((spins & 63u) == 63u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, D3P_AGENT) != 0u) return false;
                }
                return false;
            };
            const bool ok = poll();
            if (!ok) {
                upd_bad = 1u;
                if (lane == 0) chain_raise(a.status, abort_code(D3P_ABORT_RELEASE, step_t, 2u));
            }
            D3P_CSTAMP(7)
            // (D3P_DBG = 32 + 0x700: when each even WAVE had its parameter words -- slots 0 .. 7; an updater's working waves leave the
            // time of their publication in slots 8 ..)
            if (STAMPS && (a.dbg & 0x700) == 0x700 && lane == 0 && !(wave & 1)) { stamp[wave >> 1] = wall_clock64(); stamp[8 + (wave >> 1)] = 0ull; }
            ColData d1 = {};
            d1.x = __uint_as_float((uint32_t)w1);
            col_apply(mycol, d1, 0.f);
            if (xcol >= 0) {   // ICPT: the intercept's auto_loc (thread 64) / auto_scale (thread 128)
                ColData d2 = {};
                d2.x = __uint_as_float((uint32_t)w2);
                col_apply(xcol, d2, 0.f);
            }
        } else if (W16) {
            const ColData d1 = col_load(mycol);
            ColData d2 = {};
            if (xcol >= 0) d2 = col_load(xcol);
            if (apply_prev) {
                // the example count: ONE load instruction per wave (lane r < RU reads replica r's count column), summed over the lanes
                const long long nr = __hip_atomic_load(sums + (size_t)(lane < nrep ? lane : 0) * PA + P + 1, __ATOMIC_RELAXED, D3P_AGENT);
#pragma unroll
                for (int r = 0; r < nrep; ++r)
                    nll_main += ((long long)__builtin_amdgcn_readlane((int)(nr >> 32), r) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)nr, r);
            }
            col_apply(mycol, d1, zL);
            if (xcol >= 0) col_apply(xcol, d2, zX);   // ICPT: the intercept's auto_loc (thread 64) / auto_scale (thread 128)
        } else {
            latent(tid, zL, zS);
            if (ICPT && tid == 1) latent(D - 1, apply_prev ? prev_noise[D - 1] : 0.f, apply_prev ? prev_noise[2 * D - 1] : 0.f);  // the intercept
        }
        // (measured and dropped: this report in workgroup 0's TAIL instead -- it makes workgroup 0 the last at the staging barrier of
        // every step, 1.1 us after the average, but the tail is not free either: the workgroup that inherits the CU enters late;
        // 6.73 -> 6.88 us per step, same box)
        if (apply_prev && bid == 0 && tid == 0) {
            long long lll = 0, lhh = 0;
#pragma unroll
            for (int r = 0; r < nrep; ++r) {
                lll += __hip_atomic_load(sums + (size_t)r * PA + P, __ATOMIC_RELAXED, D3P_AGENT);
                lhh += __hip_atomic_load(sums + (size_t)r * PA + P + 2, __ATOMIC_RELAXED, D3P_AGENT);
            }
            if (a.losses && g > 0) {
                float lv = ((float)loss_join(lhh, lll) / (float)a.B) * a.obs_scale * factor;
                if (n == 0.0f) lv = empty_batch_loss(P, [&](int c) { return __hip_atomic_load(a.state[in][0] + c, __ATOMIC_RELAXED, D3P_AGENT); });
                a.losses[g - 1] = lv;
            }
            // (the run's counters -- the optimiser's step, the batch index -- are written ONCE, by k_flush, from the schedule: a word
            // that a different workgroup plain-stores every step is left with the value of whichever XCD's L2 is written back last)
        }
    }
    {
        const float lcw = wave_sum(lc_mine);   // (all lanes active again)
        if (lane == 0) {
            pk[4 * DL + wave] = lcw;
            if (UPD) reinterpret_cast<uint32_t*>(pk)[4 * DL + W + wave] = upd_bad;   // the wave's verdict rides through the staging barrier
        }
    }
    D3P_CSTAMP(2)
    __syncthreads();
    D3P_CSTAMP(1)
    if (UPD) {  // a stopped run: no arrival, nothing published (one LDS word per thread: thread t looks at the verdict of wave t % 16)
        const uint32_t vd = reinterpret_cast<const uint32_t*>(pk)[4 * DL + W + (tid & (W - 1))];
        if (__ballot(vd != 0u) != 0ull) return;
    }
    float lcsum = 0.f;   // fixed order over the waves' slots: identical in every thread
#pragma unroll
    for (int w = 0; w < W; ++w) lcsum += pk[4 * DL + w];

    // ------------------------------------------------------------------ phase 3: the wave's (up to) two examples
    const d3p_v2f zero2 = pk_splat(0.f);
    Quad accg0 = {zero2, zero2}, acch0 = accg0, accg1 = accg0, acch1 = accg0;   // clipped sums of the lane's 4 + 4 columns (loc-, scale-gradient)
    float acc_gt = 0.f, acc_ht = 0.f;  // ICPT: the tail latent's two sums (identical in every lane, lane 0 stores them)
    float loss_lane = 0.f, loss_uni = 0.f, n_acc = 0.f;  // loss: per-lane shares / wave-uniform terms (see `examples`)
    const int c0 = 4 * lane, c1 = C1 + 4 * lane;
    const bool icpt_lane = ICPT && lane == 63;  // its last second-half element is the intercept (own prior scale)
    // NE examples in lockstep: their dependency chains (LDS reads -> dot product -> DPP wave sum -> sigmoid -> gradient ->
    // two more wave sums -> clip factor) are independent, so the instructions of one fill the latency gaps of the other; the
    // derived parameter columns are read from LDS once for all of them.
    // The latent part of the loss of an example, sum_j [ hz z_j^2 - eps_j^2 / 2 + lc_j ] (log p(z) - log q(z) up to constants),
    // is split into the parameter-independent -|eps|^2 / 2 (from phase 0, `e2v`), the example-independent sum of the lc
    // column (one pass per call) and hz |z|^2, which alone stays in the per-element loop.
    auto examples = [&](auto ne_tag, const float4* x0v, const float4* x1v, const float* xtv, const float* yv, const Eps* ev) {
        constexpr int NE = decltype(ne_tag)::value;   // examples in lockstep: independent instruction streams the SIMD interleaves
        const Quad l0 = quad_lds(pk + c0), l1 = quad_lds(pk + c1), s0 = quad_lds(pk + DL + c0), s1 = quad_lds(pk + DL + c1);
        Quad x0[NE], x1[NE], z0[NE], z1[NE];
        float et[NE], zt[NE];  // ICPT: noise and latent value of the tail column
        d3p_v2f tp[NE], tq[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            x0[j] = quad_of(x0v[j]);
            x1[j] = quad_of(x1v[j]);
            et[j] = ICPT ? ev[j].vt : 0.f;
            zt[j] = ICPT ? __fmaf_rn(pk[DL + TL], et[j], pk[TL]) : 0.f;
            z0[j].lo = pk_fma(s0.lo, ev[j].v0.lo, l0.lo);
            z0[j].hi = pk_fma(s0.hi, ev[j].v0.hi, l0.hi);
            z1[j].lo = pk_fma(s1.lo, ev[j].v1.lo, l1.lo);
            z1[j].hi = pk_fma(s1.hi, ev[j].v1.hi, l1.hi);
            tp[j] = pk_fma(x0[j].hi, z0[j].hi, x0[j].lo * z0[j].lo);
            tq[j] = pk_fma(x1[j].hi, z1[j].hi, x1[j].lo * z1[j].lo);
        }
        if (STAMPS && (a.dbg & 0x700) == 0x500 && lane == 0) { asm volatile("" :: "v"(tp[0].x), "v"(tq[0].x)); stamp[wave & 15] = wall_clock64(); }
        const Quad sg0 = quad_lds(pk + 2 * DL + c0), sg1 = quad_lds(pk + 2 * DL + c1);   // (requested before the wave sums need them)
        const Quad q0 = quad_lds(pk + 3 * DL + c0), q1 = quad_lds(pk + 3 * DL + c1);
        float t[NE], A[NE], loglik[NE];
        if (NE == 2) {   // both examples' sums in one pass (wave_sum2)
            const d3p_v2f ta = tp[0] + tq[0], tb = tp[NE - 1] + tq[NE - 1];
            wave_sum2(ta.x + ta.y, tb.x + tb.y, t[0], t[NE - 1]);           // logits x . z
        } else {
            const d3p_v2f tpq = tp[0] + tq[0];
            t[0] = wave_sum(tpq.x + tpq.y);
        }
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (ICPT) t[j] = __fmaf_rn(xtv[j], zt[j], t[j]);                 // + the tail feature, once
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            A[j] = a.A_scale * (chain_sigmoid(t[j]) - yv[j]);                // d(-lik_scale inv_obs loglik)/dt
            loglik[j] = yv[j] * t[j] - chain_softplus(t[j]);
        }
        if (STAMPS && (a.dbg & 0x700) == 0x300 && lane == 0) { asm volatile("" :: "v"(A[0])); stamp[wave & 15] = wall_clock64(); }
        Quad g0[NE], g1[NE], h0[NE], h1[NE];
        float n2s[NE], gt[NE], ht[NE];
        const d3p_v2f c1p = pk_splat(a.c1);
        const d3p_v2f c1q = ICPT ? d3p_v2f{a.c1, icpt_lane ? a.c1_b : a.c1} : c1p;   // the intercept's own prior (last element of lane 63)
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const d3p_v2f Ap = pk_splat(A[j]);
            g0[j].lo = pk_fma(c1p, z0[j].lo, Ap * x0[j].lo);
            g0[j].hi = pk_fma(c1p, z0[j].hi, Ap * x0[j].hi);
            g1[j].lo = pk_fma(c1p, z1[j].lo, Ap * x1[j].lo);
            g1[j].hi = pk_fma(c1q, z1[j].hi, Ap * x1[j].hi);
            h0[j].lo = pk_fma(g0[j].lo * ev[j].v0.lo, sg0.lo, -q0.lo);
            h0[j].hi = pk_fma(g0[j].hi * ev[j].v0.hi, sg0.hi, -q0.hi);
            h1[j].lo = pk_fma(g1[j].lo * ev[j].v1.lo, sg1.lo, -q1.lo);
            h1[j].hi = pk_fma(g1[j].hi * ev[j].v1.hi, sg1.hi, -q1.hi);
            // squared norm of the example's gradient: four chains of two
            const d3p_v2f na = pk_fma(g0[j].hi, g0[j].hi, g0[j].lo * g0[j].lo), nb = pk_fma(h0[j].hi, h0[j].hi, h0[j].lo * h0[j].lo);
            const d3p_v2f nc = pk_fma(g1[j].hi, g1[j].hi, g1[j].lo * g1[j].lo), nd = pk_fma(h1[j].hi, h1[j].hi, h1[j].lo * h1[j].lo);
            const d3p_v2f nn = (na + nb) + (nc + nd);
            n2s[j] = nn.x + nn.y;
            // The latent part of the loss, sum_j [hz z_j^2 - eps_j^2 / 2 + lc_j]: its per-lane share is only ever summed -- over
            // the lanes, the examples, the waves and the workgroups -- so it is NOT wave-summed per example (round 2 did): it goes
            // into a per-lane accumulator that the wave sums once in phase 4; the wave-uniform terms (lc sum, tail latent,
            // likelihood) into a scalar one.
            const d3p_v2f zz = pk_fma(z0[j].hi, z0[j].hi, z0[j].lo * z0[j].lo) + pk_fma(z1[j].hi, z1[j].hi, z1[j].lo * z1[j].lo);
            float lpl = __fmaf_rn(a.hz, zz.x + zz.y, ev[j].e2);
            if (ICPT) lpl = __fmaf_rn((icpt_lane ? a.hz_b - a.hz : 0.0f) * z1[j].hi.y, z1[j].hi.y, lpl);  // (the intercept's own prior)
            loss_lane += lpl;
        }
        if (NE == 2) wave_sum2(n2s[0], n2s[NE - 1], n2s[0], n2s[NE - 1]);
        else n2s[0] = wave_sum(n2s[0]);
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            gt[j] = ht[j] = 0.f;
            float lpu = lcsum;
            if (ICPT) {  // the tail latent is a feature column (prior of the weights); its terms enter every sum once
                gt[j] = __fmaf_rn(a.c1, zt[j], A[j] * xtv[j]);
                ht[j] = __fmaf_rn(gt[j] * et[j], pk[2 * DL + TL], -pk[3 * DL + TL]);
                n2s[j] = __fmaf_rn(gt[j], gt[j], __fmaf_rn(ht[j], ht[j], n2s[j]));
                lpu += __fmaf_rn(a.hz * zt[j], zt[j], -0.5f * et[j] * et[j]);
            }
            // clip factor 1 / max(1, ||g|| / C) (svi.py:121-122) folded into the running sum (svi.py:343-346)
            const d3p_v2f cf = pk_splat(fminf(1.0f, a.clip * __builtin_amdgcn_rsqf(n2s[j])));
            accg0.lo = pk_fma(cf, g0[j].lo, accg0.lo); accg0.hi = pk_fma(cf, g0[j].hi, accg0.hi);
            acch0.lo = pk_fma(cf, h0[j].lo, acch0.lo); acch0.hi = pk_fma(cf, h0[j].hi, acch0.hi);
            accg1.lo = pk_fma(cf, g1[j].lo, accg1.lo); accg1.hi = pk_fma(cf, g1[j].hi, accg1.hi);
            acch1.lo = pk_fma(cf, h1[j].lo, acch1.lo); acch1.hi = pk_fma(cf, h1[j].hi, acch1.hi);
            if (ICPT) {
                acc_gt = __fmaf_rn(cf.x, gt[j], acc_gt);
                acc_ht = __fmaf_rn(cf.x, ht[j], acc_ht);
            }
            loss_uni += lpu - a.lik_scale * loglik[j];  // svi.py:278-281 (times inv_obs in phase 4)
            n_acc += 1.0f;
        }
    };
    if (live2) {  // (live2 implies live1) the common case: both in lockstep
        const float4 xs0[2] = {xa0, xb0}, xs1[2] = {xa1, xb1};
        const float xts[2] = {xta, xtb}, ys[2] = {ya, yb};
        const Eps es[2] = {epa, epb};
        examples(std::integral_constant<int, 2>{}, xs0, xs1, xts, ys, es);
    } else if (live1) {
        examples(std::integral_constant<int, 1>{}, &xa0, &xa1, &xta, &ya, &epa);
    }
    if (live3) {  // the third item, prepared in phase 0 (a lane reads back what it wrote: no barrier)
        Eps e3;
        e3.v0 = quad_lds(er + c0);
        e3.v1 = quad_lds(er + c1);
        e3.vt = vt3;
        e3.e2 = e2_3;
        const float4 x30 = *reinterpret_cast<const float4*>(er + DL + c0), x31 = *reinterpret_cast<const float4*>(er + DL + c1);
        examples(std::integral_constant<int, 1>{}, &x30, &x31, &xt3, &y3, &e3);
    }
    // further items of this wave (the grid was sized for fewer items than the step has -- more than three per wave): loaded,
    // their noise generated and consumed one at a time
    // W = 16: in pairs (B = 32768 on 256 workgroups: eight items per wave, i.e. three more pairs).
    for (uint32_t k = third_pre ? n_items : k2 + stride; k < n_items; k += (W16 ? 2u : 1u) * stride) {
        const uint32_t kb = k + stride;
        const bool two = W16 && kb < n_items;  // (wave-uniform)
        const uint32_t p = PLIST ? (a.plist_base + (size_t)step_t * a.B)[k] : k;
        const uint32_t pb = two ? (PLIST ? (a.plist_base + (size_t)step_t * a.B)[kb] : kb) : p;
        const size_t row = (size_t)((uint64_t)idx[p] - a.row_lo), rowb = (size_t)((uint64_t)idx[pb] - a.row_lo);
        float4 xs0[2], xs1[2];
        float xts[2] = {0.f, 0.f}, ys[2];
        Eps es[2];
        load_x(a.X + row * DF, xs0[0], xs1[0], xts[0]);
        ys[0] = a.y[row];
        const uint32_t q0 = skeys[2 * p], q1 = skeys[2 * p + 1];
        if (two) {
            load_x(a.X + rowb * DF, xs0[1], xs1[1], xts[1]);
            ys[1] = a.y[rowb];
            const uint32_t r0 = skeys[2 * pb], r1 = skeys[2 * pb + 1];
            es[0] = gen(q0, q1, nothing);
            es[1] = gen(r0, r1, nothing);
            examples(std::integral_constant<int, 2>{}, xs0, xs1, xts, ys, es);
        } else {
            es[0] = gen(q0, q1, nothing);
            examples(std::integral_constant<int, 1>{}, xs0, xs1, xts, ys, es);
        }
    }
    D3P_CSTAMP(5)
    if (STAMPS && (a.dbg & 0x700) == 0x100 && lane == 0) stamp[wave & 15] = wall_clock64();  // diagnostic: when each WAVE finished its examples
    // (D3P_DBG = 32 + 256: examples done; + 512: after the logit / sigmoid instead; + 1024: after the z / dot-product loop instead)

    // ------------------------------------------------------------------ phase 4: workgroup reduction, atomics, arrival
    // (the wave's row held its noise; both examples are consumed, the row now takes its partial sums: [loc-gradient | scale-
    // gradient] x DL, indexed like the LDS columns)
    *reinterpret_cast<float4*>(er + c0) = make_float4(accg0.lo.x, accg0.lo.y, accg0.hi.x, accg0.hi.y);
    *reinterpret_cast<float4*>(er + c1) = make_float4(accg1.lo.x, accg1.lo.y, accg1.hi.x, accg1.hi.y);
    *reinterpret_cast<float4*>(er + DL + c0) = make_float4(acch0.lo.x, acch0.lo.y, acch0.hi.x, acch0.hi.y);
    *reinterpret_cast<float4*>(er + DL + c1) = make_float4(acch1.lo.x, acch1.lo.y, acch1.hi.x, acch1.hi.y);
    if (ICPT && lane == 0) { er[TL] = acc_gt; er[DL + TL] = acc_ht; }
    {
        const float lw = a.inv_obs * (wave_sum(loss_lane) + loss_uni);
        if (lane == 0) { tail[2 * wave] = lw; tail[2 * wave + 1] = n_acc; }
    }
    __syncthreads();
    D3P_CSTAMP(6)
    if (W16)   // (see the prologue; same box: 6.84 - 6.88 against 6.89 - 6.92 us per step)
        for (int i = (int)bid * (64 * W) + tid; i < R * PA; i += a.nw * 64 * W)
            __hip_atomic_store(acc_next + i, 0ll, __ATOMIC_RELAXED, D3P_AGENT);
    {
        // fixed-point integer atomics: the exact, order-independent sum of the workgroups' fp32 partials
        long long* outp = acc_cur + (size_t)(bid % RU) * PA;
        bool bad = false;
        auto column_sum = [&](int c) {   // W = 16: one parameter column per thread
            const bool is_scale = c >= D;
            const int li = lix(is_scale ? c - D : c) + (is_scale ? DL : 0);
            float sc = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) sc += red[(size_t)w * 2 * DL + li];
            bool ok;
            const long long fx = fixed_point_rn(sc, a.sg, ok);
            bad |= !ok;
            if (!(STAMPS && (a.dbg & 4))) atomicAdd(reinterpret_cast<unsigned long long*>(outp + c), (unsigned long long)fx);
        };
        auto column_pair = [&](int e) {  // latent e: columns e (auto_loc) and D + e (auto_scale)
            const int li = lix(e);
            float sL = 0.f, sS = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                sL += red[(size_t)w * 2 * DL + li];
                sS += red[(size_t)w * 2 * DL + DL + li];
            }
            const double dL = (double)sL * a.sg, dS = (double)sS * a.sg;
            bad |= !(fabs(dL) < 4503599627370496.0) | !(fabs(dS) < 4503599627370496.0);
            if (!(STAMPS && (a.dbg & 4))) {  // (diagnostic instantiation only: D3P_DBG=4 switches the gradient atomics off)
                atomicAdd(reinterpret_cast<unsigned long long*>(outp + e), (unsigned long long)__double2ll_rn(dL));
                atomicAdd(reinterpret_cast<unsigned long long*>(outp + D + e), (unsigned long long)__double2ll_rn(dS));
            }
        };
        if (W16) {
            column_sum(mycol);
            if (ICPT && tid == 64) column_sum(D - 1);        // the intercept's two columns
            if (ICPT && tid == 128) column_sum(2 * D - 1);
        } else {
            column_pair(tid);
            if (ICPT && tid == 64) column_pair(D - 1);  // the intercept
        }
        if (tid < 3) {  // thread 0: loss, fine part; 1: example count; 2: loss, coarse part
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) s += tail[2 * w + (tid & 1)];
            long long hi, lo;
            const bool ok = loss_split(s, hi, lo);
            if (tid != 1) bad |= !ok;
            const long long v = tid == 0 ? lo : tid == 1 ? (long long)s : hi;
            atomicAdd(reinterpret_cast<unsigned long long*>(outp + P + tid), (unsigned long long)v);
        }
        if (bad) {  // (rare) poison the count column: the next prologue yields NaN like the reference's float sums
            atomicAdd(reinterpret_cast<unsigned long long*>(outp + P + 1), 1ull << 44);
            __hip_atomic_store(a.status + 1, 1u, __ATOMIC_RELAXED, D3P_AGENT);
        }
    }
    D3P_CSTAMP(9)
    // arrive: this workgroup's atomics (and, for workgroup 0, the published state; the zeroed accumulator) are complete
    // at the memory side before the counters move
    __builtin_amdgcn_s_waitcnt(0);
    D3P_CSTAMP(10)
    __syncthreads();
    D3P_CSTAMP(11)
    if (tid == 0) {
        uint32_t* bar = a.bar + (size_t)step_t * D3P_BAR_WORDS;
        // Arrival group = XCD.  The dispatcher deals workgroups to the 8 XCDs by blockIdx.x % 8; a step's workgroups are the nw
        // consecutive indices from step_t * nw, so for W = 16 (per == nw) the group is taken from blockIdx.x itself and its size is the
        // count of those indices with that residue -- for EVERY nw, not only multiples of 8 (bid % 8 names an XCD only then: at
        // nw = 9, 17, 43 a group's counter line and its updater moved to another XCD every step).  nw < 8: one workgroup per group.
        const uint32_t nw = (uint32_t)a.nw;
        uint32_t grp = bid % D3P_BAR_GROUPS, gsize = (nw + D3P_BAR_GROUPS - 1u - grp) / D3P_BAR_GROUPS;
        if (W16 && nw >= D3P_BAR_GROUPS && !(a.dbg & 0x1000)) {   // (D3P_DBG bit 0x1000: round 4's bid % 8 groups, for A/B runs)
            const uint32_t first = (uint32_t)step_t * nw;   // (< 2^31: K <= 128 steps of <= 256 workgroups)
            grp = blockIdx.x % D3P_BAR_GROUPS;
            gsize = (first + nw + D3P_BAR_GROUPS - 1u - grp) / D3P_BAR_GROUPS - (first + D3P_BAR_GROUPS - 1u - grp) / D3P_BAR_GROUPS;
        }
        const uint32_t prev = __hip_atomic_fetch_add(bar + D3P_BAR_LINE * (1 + grp), 1u, __ATOMIC_RELAXED, D3P_AGENT);
        if (UPD) {
            // (measured and dropped: ONE counter per step, the last 8 workgroups to arrive as updaters -- one atomic round trip less
            // for them, 128 same-address atomics per step: 8.78 vs 8.64 us per step)
            // The workgroup whose arrival completes its GROUP is one of the step's (up to 8) updaters: it adds to the step's top
            // counter on line 0 and, unless that completes the step, waits until the other groups have (a handful of waiters on a
            // word that takes 8 atomics per step).  Its role for the tail: okw[2] = 1 + group, 0 = none / the run was stopped.
            uint32_t role = 0u;
            if (prev + 1u == gsize) {
                const uint32_t ng = nw < D3P_BAR_GROUPS ? nw : D3P_BAR_GROUPS;
                role = 1u + grp;
                if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, D3P_AGENT) + 1u != ng) {
                    for (uint32_t spins = 0;; ++spins) {
                        if (__hip_atomic_load(bar, __ATOMIC_RELAXED, D3P_AGENT) >= ng) break;
                        if (spins > D3P_WAIT_ROUNDS_PEERS || ((spins & 63u) == 63u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, D3P_AGENT) != 0u)) {
                            chain_raise(a.status, abort_code(D3P_ABORT_XCHG_ARRIVALS, step_t, grp));
                            role = 0u;
                            break;
                        }
                    }
                }
            }
            okw[2] = role;
        } else if (prev + 1u == gsize) {  // this group's flag; the waiters poll the flags of all groups
            __hip_atomic_store(bar + D3P_BAR_LINE * (1 + D3P_BAR_GROUPS + grp), 1u, __ATOMIC_RELAXED, D3P_AGENT);
        }
        if (STAMPS && a.stamps) {
            if (!(a.dbg & 256)) {
                stamp[12] = wall_clock64();
                stamp[13] = (unsigned long long)prev;
            }
            if (UPD && !(a.dbg & 256)) stamp[3] = stamp[4] = stamp[14] = stamp[15] = 0ull;   // (an updater's tail sets them)
            const int rec = step_t - (a.K - 2);
            if (!UPD && rec >= 0 && bid < 256u)
                for (int k = 0; k < 16; ++k) a.stamps[((size_t)rec * 256 + bid) * 16 + k] = stamp[k];
        }
    }
    // Data-parallel, W = 16: the step's UPDATERS -- the workgroups whose arrivals completed the 8 arrival groups -- sum the step
    // over the ranks and apply the update, each for its share of the accumulator columns (chunks of 64 columns: updater j takes
    // chunks j, j + 8, ...: two waves of it work, one column per lane), with no barrier and no LDS between the threads:
    //   one round trip: the column's R replicas, its state words {x, m, v | epoch}, its Gaussian-mechanism normal and the rank's
    //   own count column;
    //   fold -> the column as two tagged 8-byte words into slot [parity][rank] of every PEER's inbox (system scope: over xGMI);
    //   tagged loads from the own inbox until the column of every peer carries this epoch's tag -> the world's sum (int64: exact,
    //   the same on every rank);
    //   mean, noise ONCE (svi.py:365-375), rescale, Adam (svi.py:379-393) -> {x | epoch + 1} for the next step's
    //   workgroups (phase 2 above), {m, v | epoch + 1} for its updaters.  Nothing is waited for after the stores.
    // (Several updaters, not the one workgroup whose arrival completes the step: a tail that reads 60 KB -- 480 cache lines --
    // through ONE CU is bound by the misses a CU keeps in flight: three memory round trips, its 16 waves publishing 2 us apart
    // (round-4 anatomy); 8 updaters on 8 CUs have 60 lines each.)
    if (UPD) {
        __syncthreads();
        const uint32_t role = okw[2];
        const uint32_t ngr = (uint32_t)a.nw < D3P_BAR_GROUPS ? (uint32_t)a.nw : D3P_BAR_GROUPS;
        constexpr int NCH = (PA + 63) / 64;
        if (role != 0u && (role - 1u) + ngr * (uint32_t)wave < (uint32_t)NCH) {
            const unsigned long long epoch = a.x.epoch0 + (unsigned long long)step_t + 1ull;
            const unsigned parity = (unsigned)(epoch & 1ull);
            const uint32_t tag = (uint32_t)epoch, ntag = tag + 1u, itag = tag & 0x7fffffffu;
            const StepSlot* ms = a.slots + step_t;
            const float* znoise = a.noise_base + (size_t)step_t * P;
            const unsigned long long* xin = a.ll_state[0] + (size_t)(tag & 1u) * PA;
            unsigned long long* xout = a.ll_state[0] + (size_t)(ntag & 1u) * PA;
            const float Bf = (float)a.B, factor0 = u_factor0, inv_B = u_inv_B, inv_bc1 = u_inv_bc1, inv_bc2 = u_inv_bc2;
            const float noise_scale = u_noise_scale, out_scale0 = u_out_scale0;   // (from phase 0)
            const bool self_trip = a.x.self_trip != 0;
            // The non-finite marker (a workgroup that saw a non-finite partial added 2^44 to the count column: NaN from here on,
            // like the reference's float sums) travels as bit 31 of the tag of every word the rank sends, so a thread needs no
            // count column of the peers; the rank's own count column is ONE load instruction per wave (lane r reads replica r).
            const long long nr = __hip_atomic_load(acc_cur + (size_t)(lane < R ? lane : 0) * PA + P + 1, __ATOMIC_RELAXED, D3P_AGENT);
            auto load_fold = [&](int c) {
                long long s8[R], sf = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) s8[r] = __hip_atomic_load(acc_cur + (size_t)r * PA + c, __ATOMIC_RELAXED, D3P_AGENT);
#pragma unroll
                for (int r = 0; r < R; ++r) sf += s8[r];
                return sf;
            };
            for (int ch = (int)(role - 1u) + (int)ngr * wave; ch < NCH; ch += (int)ngr * W) {   // (wave-uniform; one pass unless nw < 8)
                const int craw = 64 * ch + lane;
                const bool mine = craw < PA;
                const int c = mine ? craw : PA - 1;
                const bool is_par = c < P;
                const int cp = is_par ? c : 0;
                // everything this column needs, in one round trip
                unsigned long long wx = __hip_atomic_load(xin + cp, __ATOMIC_RELAXED, D3P_AGENT);
                unsigned long long wm = __hip_atomic_load(a.ll_state[1] + cp, __ATOMIC_RELAXED, D3P_AGENT);
                unsigned long long wv = __hip_atomic_load(a.ll_state[2] + cp, __ATOMIC_RELAXED, D3P_AGENT);
                const float z = znoise[cp];
                const long long sf = load_fold(c);
                long long nloc = 0;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    nloc += ((long long)__builtin_amdgcn_readlane((int)(nr >> 32), r) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)nr, r);
                const bool poison_loc = nloc >= (1ll << 40);
                if (STAMPS && lane == 0 && wave == 0 && !(a.dbg & 256)) { asm volatile("" :: "v"(sf)); stamp[3] = wall_clock64(); }
                const uint32_t stag = itag | (poison_loc ? 0x80000000u : 0u);
                if (mine)
                    for (int p = 0; p < a.x.world; ++p)
                        if (p != a.x.rank || self_trip)
                            xchg_ll_store(a.x.peer[p], ((size_t)parity * a.x.world + a.x.rank) * PA + c, sf, stag);
                if (STAMPS && lane == 0 && wave == 0 && !(a.dbg & 256)) stamp[4] = wall_clock64();
                bool alive = true;   // (wave-uniform: every exit of the wait below is taken by the whole wave)
                // The world's sum of column cc (own fold `own`), and whether any rank sent the non-finite marker.  The peers' rows are
                // asked for TOGETHER (NP rows per round: 7 = the peers of an 8-rank job in one round), again and again until every word
                // carries this epoch's tag.  NP is a compile-time constant per call: the tail runs on one or two waves per CU, i.e. at one
                // instruction per ~5 cycles, and the first form of this wait -- eight peers unrolled with run-time predicates, ~400
                // instructions around zero to seven loads -- cost 1 us between the arrival and the publication (round-4 anatomy).
                const char* inbox = a.x.peer[a.x.rank];
                const int npeers = self_trip ? a.x.world : a.x.world - 1;
                auto collect_n = [&](auto np_tag, int pos, int cc, long long& tot, bool& poisoned) {   // peers pos .. pos + NP - 1 of the list
                    constexpr int NP = decltype(np_tag)::value;   // (the list: ranks rank + 1, rank + 2, ... mod world; with self_trip it starts at rank)
                    size_t off[NP];
                    int pr = a.x.rank + (self_trip ? 0 : 1) + pos;
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        int q = pr + j;
                        q = q >= a.x.world ? q - a.x.world : q;
                        off[j] = ((size_t)parity * a.x.world + q) * PA + cc;
                    }
                    unsigned long long u0[NP], u1[NP];
                    bool ok = false;
                    for (uint32_t spins = 0; spins < D3P_WAIT_ROUNDS_PEERS / 2u; ++spins) {
#pragma unroll
                        for (int j = 0; j < NP; ++j) xchg_ll_fetch(inbox, off[j], &u0[j], &u1[j]);
                        bool valid = true;
#pragma unroll
                        for (int j = 0; j < NP; ++j)
                            valid = valid && ((uint32_t)(u0[j] >> 32) & 0x7fffffffu) == itag && ((uint32_t)(u1[j] >> 32) & 0x7fffffffu) == itag;
                        if (__ballot(!valid) == 0ull) { ok = true; break; }
                        if ((spins & 63u) == 63u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, D3P_AGENT) != 0u) break;
                    }
                    if (!ok) {
                        alive = false;
                        chain_raise(a.status, abort_code(D3P_ABORT_XCHG_ROW, step_t, (uint32_t)pos));
                        return;
                    }
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        tot += xchg_ll_value(u0[j], u1[j]);
                        poisoned = poisoned || ((u0[j] | u1[j]) >> 63) != 0ull;
                    }
                };
                auto collect = [&](int cc, long long own, long long& tot, bool& poisoned) {
                    tot = self_trip ? 0ll : own;
                    poisoned = poison_loc;
                    int pos = 0;
                    while (alive && npeers - pos >= 7) { collect_n(std::integral_constant<int, 7>{}, pos, cc, tot, poisoned); pos += 7; }
                    while (alive && npeers - pos >= 3) { collect_n(std::integral_constant<int, 3>{}, pos, cc, tot, poisoned); pos += 3; }
                    while (alive && npeers - pos >= 1) { collect_n(std::integral_constant<int, 1>{}, pos, cc, tot, poisoned); pos += 1; }
                };
                long long tot;
                bool poisoned;
                collect(c, sf, tot, poisoned);
                if (!alive) break;
                if (STAMPS && lane == 0 && wave == 0 && !(a.dbg & 256)) { asm volatile("" :: "v"(tot)); stamp[14] = wall_clock64(); }
                if (mine && is_par) {   // noise, rescale, Adam, publication
                    // (the state words were written one step -- microseconds -- ago by the previous step's updater, or by k_run_init:
                    // the tag test is a formality, but it is what makes the hand-over correct by construction)
                    bool have = true;
                    for (uint32_t spins = 0; (uint32_t)(wx >> 32) != tag || (uint32_t)(wm >> 32) != tag || (uint32_t)(wv >> 32) != tag; ++spins) {
                        if (spins > D3P_WAIT_ROUNDS) { chain_raise(a.status, abort_code(D3P_ABORT_RELEASE, step_t, 3u)); have = false; break; }
                        wx = __hip_atomic_load(xin + c, __ATOMIC_RELAXED, D3P_AGENT);
                        wm = __hip_atomic_load(a.ll_state[1] + c, __ATOMIC_RELAXED, D3P_AGENT);
                        wv = __hip_atomic_load(a.ll_state[2] + c, __ATOMIC_RELAXED, D3P_AGENT);
                    }
                    if (have) {
                        const float out_scale = out_scale0 + (poisoned ? __builtin_nanf("") : 0.0f);
                        float x = __uint_as_float((uint32_t)wx), m = __uint_as_float((uint32_t)wm), v = __uint_as_float((uint32_t)wv);
                        const float totf = (float)(i64_to_f64(tot) * a.inv_sg);
                        const float gr = __fmaf_rn(z, noise_scale, totf * inv_B) * out_scale;
                        m = (1.0f - a.b1) * gr + a.b1 * m;
                        v = (1.0f - a.b2) * gr * gr + a.b2 * v;
                        x = x - a.lr * (m * inv_bc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v * inv_bc2) + a.adam_eps);
                        if (!(fabsf(x) <= 3.402823466e38f)) {
                            // (cold) the FIRST step that runs with a non-finite parameter, + 1, in status word 2 -- performed before the
                            // parameter is published: what the reporter of an EMPTY batch's loss asks (empty_batch_loss, d3p_device.h)
                            uint32_t expect = 0u;
                            (void)__hip_atomic_compare_exchange_strong(a.status + 2, &expect, (uint32_t)g + 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, D3P_AGENT);
                            __builtin_amdgcn_s_waitcnt(0);
                        }
                        const unsigned long long nt = (unsigned long long)ntag << 32;
                        __hip_atomic_store(xout + c, nt | __float_as_uint(x), __ATOMIC_RELAXED, D3P_AGENT);
                        __hip_atomic_store(a.ll_state[1] + c, nt | __float_as_uint(m), __ATOMIC_RELAXED, D3P_AGENT);
                        __hip_atomic_store(a.ll_state[2] + c, nt | __float_as_uint(v), __ATOMIC_RELAXED, D3P_AGENT);
                    }
                }
                if (STAMPS && lane == 0 && wave == 0 && !(a.dbg & 256)) stamp[15] = wall_clock64();
                if (STAMPS && (a.dbg & 0x700) == 0x700 && lane == 0 && wave < 8) stamp[8 + wave] = wall_clock64();
                if (mine && c == P) {   // the step's loss (fine part P: `tot`; coarse part P + 2) and the run's counters
                    const long long hi_own = load_fold(P + 2);
                    long long hi;
                    bool p2;
                    collect(P + 2, hi_own, hi, p2);
                    if (alive) {
                        const float factor = factor0 + (poisoned ? __builtin_nanf("") : 0.0f);
                        // (one address per step: no two workgroups ever store to the same word.  The run's counters are k_flush's.)
                        if (a.losses) {
                            float lv = ((float)loss_join(hi, tot) / Bf) * a.obs_scale * factor;
                            // (n = 0: NaN once any parameter this step ran with was not finite.  The row the step's workgroups polled may
                            // already belong to step g + 2 -- nothing waits for this thread -- so the updaters leave a mark instead: above)
                            if (factor0 == 0.0f && !poisoned) {
                                const uint32_t nf = __hip_atomic_load(a.status + 2, __ATOMIC_RELAXED, D3P_AGENT);
                                if (nf != 0u && nf <= (uint32_t)g + 1u) lv = __builtin_nanf("");
                            }
                            a.losses[g] = lv;
                        }
                    }
                }
            }
        }
        if (STAMPS && tid == 0 && a.stamps) {
            const int rec = step_t - (a.K - 2);
            if (rec >= 0 && bid < 256u)
                for (int k = 0; k < 16; ++k) a.stamps[((size_t)rec * 256 + bid) * 16 + k] = stamp[k];
        }
    }
    // W = 16: the key-chain link of step `step_t` of the NEXT batch, behind the arrival of workgroup 0 (off the step's critical
    // path; the previous link was made by workgroup 0 of the previous step, behind ITS arrival)
    if (W16 && bid == 0 && step_t < a.K_next && tid < 64) {
        // (the 32 bytes in front of the words hold the progress word of the 8-wave form's key-chain workgroup: unused here)
        unsigned long long* ll = reinterpret_cast<unsigned long long*>(a.bar + (size_t)a.K * D3P_BAR_WORDS) + 4;
        chain_step_ll(a.chain_sched, ll, a.chain_slots + step_t, step_t, step_t == a.K_next - 1, a.status, abort_code(D3P_ABORT_KEY_CHAIN, step_t));
    }
#undef D3P_CSTAMP
}

}  // namespace d3p
