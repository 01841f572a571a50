// Shared device code of the logistic-regression DP-VI kernels: derived-parameter pack, per-example
// threefry sample keys, the fused main kernel template (MODE 0: clip + accumulate, MODE 1:
// materialise px_grads) and its launch geometry.
#pragma once
#include <type_traits>
#include "d3p_device.h"
#include "d3p_host.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

namespace d3p {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// LDS of k_logreg_main: derived columns (5 x D) | W x P reduction rows | 2W loss/count tail | 4 status words
// (+ 16 phase stamps of 8 bytes for the D3P_DBG=32 diagnostic of the chained launch)
static inline size_t main_lds_bytes(int D, int W) { return (size_t)(((5 * D + 3) & ~3) + W * 2 * D + 2 * W + 4 + 32) * sizeof(float); }

#define D3P_MAIN_MAX_BLOCKS 2048u

// ------------------------------------------------------------------------------------------
// derived per-column parameters ("pack"): loc, s = softplus(u), sg = sigmoid(u),
// q = inv_obs * sg / s, lc = log(prior_std) - log(s)
// ------------------------------------------------------------------------------------------
// scale s(u) of the guide and ds/du: softplus (AutoDiagonalNormal) or exp (hand-written example guides)
__device__ __forceinline__ void guide_scale(int gexp, float u, float& s, float& ds)
{
    if (gexp) {
        s = ds = expf(u);   // (not __expf: exp2(u log2 e) carries |u| ulp of relative error, 2e-6 at u = 27 -- this is a scale, once per column)
    } else {
        s = softplus_f(u);
        // softplus_f's log(1 + e) is exact to ~1e-7 ABSOLUTE: enough for a log-likelihood term, not for a SCALE, which enters the step
        // through its logarithm and its reciprocal: at u = -6 (a posterior standard deviation of 2.5e-3) that is 3e-5 relative, at
        // u = -15 20 %, and below u ~ -16.6 the sum 1 + exp(u) rounds to 1 and the scale is 0 (0 / 0, log 0).  Below u = -4 the scale is
        // log1p's series e (1 - e/2 + e^2/3 - e^3/4 + e^4/5) (e < 0.0184: the next term is 4e-10 relative) -- what jax.nn.softplus
        // (logaddexp) and the oracle's log1pf return.  (Five multiply-adds per column, only for such columns.)
        if (u < -4.0f) {
            const float e = u < -15.0f ? expf(u) : __expf(u);
            s = e * __fmaf_rn(e, __fmaf_rn(e, __fmaf_rn(e, __fmaf_rn(e, 0.2f, -0.25f), 0.33333334f), -0.5f), 1.0f);
        }
        ds = sigmoid_f(u);
    }
}

__device__ __forceinline__ void pack_column(const d3p_logreg_model& m, int D, int e, float loc, float u,
                                            float* __restrict__ pack)
{
    float s, sg;
    guide_scale(m.guide_transform, u, s, sg);
    const float ps = (e < m.d) ? m.prior_w : m.prior_b;
    pack[e] = loc;
    pack[D + e] = s;
    pack[2 * D + e] = sg;
    pack[3 * D + e] = m.inv_obs * sg / s;
    pack[4 * D + e] = logf(ps) - logf(s);
}

static __global__ void k_pack(d3p_logreg_model m, const float* __restrict__ params, float* __restrict__ pack)
{
    const int D = m.d + (m.intercept ? 1 : 0);
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < D) pack_column(m, D, e, params[e], params[D + e], pack);
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



// (row layout of a fixed-point accumulator replica, loss_split / loss_join: d3p_device.h)
// replicas of the fixed-point accumulator (workgroup b adds to replica b % R).  Measured, us per step: 16-wave form 8 -> 14.75,
// 4 -> 14.43, 2 -> 15.9; pipelined form 8 -> 11.2, 4 -> 10.6, 2 -> 12.3; k_logreg_chain (round 2) 8 -> 8.04, 4 -> 8.12, 2 -> 8.67
#ifndef D3P_ACC_R
#define D3P_ACC_R 4
#endif

// Per-step record produced by the key chain / sampler (device memory).
struct StepSlot {
    uint32_t next_key[16];   // split(state_key, 3)[0]: the state key after this step (ring path)
    uint32_t grad_key[16];   // split(state_key, 3)[1]
    uint32_t pert_key[16];   // split(state_key, 3)[2]
    uint32_t batch_key[16];  // fold_in(batchifier_state, i)
    uint32_t site_keys[2][16];  // split(perturbation_key, 2) (ring path)
    uint32_t rc[32];         // Feistel round constants (ring path)
    uint32_t jax_key[2];     // random_bits(gradient_key, 32, (2,))
    uint32_t counts[2];      // [0] raw selected, [1] valid examples of the padded batch
    int32_t adam_i;          // optimiser step index of this step      } these four words are read as
    uint32_t batch_i;        // batch index of this step               } one block by the one-launch
    float bc1, bc2;          // 1 - b1^(i+1), 1 - b2^(i+1)             } step kernel (StepMeta)
    uint32_t n_owned;        // entries of this step's owned-position list (k_owned_list)
    uint32_t pad[3];
};

struct StepMeta {  // view of StepSlot::{adam_i, batch_i, bc1, bc2}
    int32_t adam_i;
    uint32_t batch_i;
    float bc1, bc2;
};

// Running state of the key chain between batches.
struct Sched {
    uint32_t key[16];
    int32_t adam_i;
    uint32_t batch_i;
    uint32_t pad[2];
};

// Arrival counters of one step: 17 lines of 128 bytes -- line 1 + g = arrivals of group g (blockIdx % 8: the workgroups of
// one XCD), line 9 + g = flag of group g, set by the workgroup whose arrival completes that group; a waiter polls the 8 flags
// with ONE load instruction (lane g <-> flag g).  Counters are only ever added to, flags only ever polled, every word has a
// line of its own: measured on MI355X (tools/probes/barrier_probe.hip), counters sharing lines or waiters polling a word that
// takes atomics cost several us more.  (D3P_DBG=512: the earlier two-level form -- the last arriver of a group adds to a top
// counter on line 0, the last of those sets all 8 flags, a waiter polls its own group's flag: 10.49 vs 10.41 us per step.)
#define D3P_BAR_LINE 32
#define D3P_BAR_GROUPS 8u  // arrival groups per step (blockIdx % 8: the workgroups of one XCD; 16 and 32 groups measured the same: 10.39 / 10.43-10.49 vs 10.41 us per step)
#define D3P_BAR_WORDS ((1 + 2 * D3P_BAR_GROUPS) * D3P_BAR_LINE)
#define D3P_AGENT __HIP_MEMORY_SCOPE_AGENT
typedef unsigned int d3p_u32x4 __attribute__((ext_vector_type(4)));

template <bool CH, typename T>
__device__ __forceinline__ T ld_x(const T* p)  // cross-workgroup load: agent-scope in the chained form
{
    if (CH) return __hip_atomic_load(p, __ATOMIC_RELAXED, D3P_AGENT);
    return *p;
}

template <bool CH, typename T>
__device__ __forceinline__ void st_x(T* p, T v)
{
    if (CH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, D3P_AGENT);
    else *p = v;
}

// The abort flag of a run holds 0 or the code of the FIRST bounded wait that ran out: kind | step << 8 | detail << 20
// (d3p_dpvi_logreg_run_status hands it out as `aborted`).
#define D3P_ABORT_WAIT 1u          // a wait not told apart below
#define D3P_ABORT_XCHG_ARRIVALS 2u // exchange workgroup: the step's compute workgroups did not all arrive
#define D3P_ABORT_XCHG_ROW 3u      // exchange workgroup: the row of rank `detail` did not come
#define D3P_ABORT_RELEASE 4u       // compute workgroup: the previous step was not released (arrival flags / exchange flags)
#define D3P_ABORT_KEY_CHAIN 5u     // key-chain workgroup: the previous link did not come
#define D3P_ABORT_XCHG_KERNEL 6u   // k_xchg: the row of rank `detail` did not come
__device__ __forceinline__ uint32_t abort_code(uint32_t kind, int step, uint32_t detail = 0u)
{
    return kind | ((uint32_t)step & 0xfffu) << 8 | (detail & 0xfffu) << 20;
}
__device__ __forceinline__ void chain_raise(uint32_t* abort_flag, uint32_t code)
{
    uint32_t expected = 0u;  // (the first code stays)
    (void)__hip_atomic_compare_exchange_strong(abort_flag, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, D3P_AGENT);
    // words 8 .. 15 of the status block: per kind of wait, 0x1000 - the EARLIEST step of the launch at which one ran out (all
    // waits of a launch start when it does and run out together, so the first code alone does not say where the run stood)
    const uint32_t kind = code & 7u, step = (code >> 8) & 0xfffu;
    (void)__hip_atomic_fetch_max(abort_flag + 8 + kind, 0x1000u - step, __ATOMIC_RELAXED, D3P_AGENT);
}

// bounded wait until *p >= target; false (and the abort flag raised) when the bound is hit or another waiter gave up
// Bounds of the waits, in polling rounds (0.2 - 0.4 us each).  Inside one GPU nothing takes longer than a few steps: 2^21
// rounds (0.4 - 0.9 s) mean the launch is stuck.  A data-parallel run also waits for its PEERS, transitively in every wait of
// the launch, and those may start late (another process, its first launch, a slower host): 2^26 rounds (13 - 27 s).
#define D3P_WAIT_ROUNDS (1u << 21)
#define D3P_WAIT_ROUNDS_PEERS (1u << 26)

__device__ __forceinline__ bool chain_wait(const uint32_t* p, uint32_t target, uint32_t* abort_flag, uint32_t code = D3P_ABORT_WAIT,
                                           uint32_t rounds = D3P_WAIT_ROUNDS)
{
    for (uint32_t spins = 0;; ++spins) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, D3P_AGENT) >= target) return true;
        // (the abort flag is looked at every 64th spin only: reading it on every spin doubles the polling traffic)
        if (spins > rounds || ((spins & 63u) == 63u && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, D3P_AGENT) != 0u)) {
            chain_raise(abort_flag, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// bounded wait, made by a whole wavefront, until the `ngroups` group flags (one per 128-byte line) are all set: lane g polls
// the flag of group g and lane `ngroups` the run's abort flag -- one load instruction per round.  false when the bound is hit
// or the abort flag is (or becomes) set: the caller must then leave without applying or publishing anything.
// flags == nullptr: nothing to wait for (first step of a launch), only the abort flag is looked at.
__device__ __forceinline__ bool chain_wait_groups(const uint32_t* flags, uint32_t ngroups, uint32_t* abort_flag,
                                                  uint32_t code = D3P_ABORT_WAIT, uint32_t rounds = D3P_WAIT_ROUNDS)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t spins = 0;; ++spins) {
        uint32_t v = 1u;
        if (lane < ngroups && flags) v = __hip_atomic_load(flags + D3P_BAR_LINE * lane, __ATOMIC_RELAXED, D3P_AGENT);
        else if (lane == ngroups) v = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, D3P_AGENT);
        const unsigned long long zero = __ballot(v == 0u);
        const bool aborted = ((zero >> ngroups) & 1ull) == 0ull;  // lane `ngroups` read a non-zero abort flag
        if (aborted) return false;
        if ((zero & ((1ull << ngroups) - 1ull)) == 0ull) return true;
        if (spins > rounds) {
            if (lane == 0) chain_raise(abort_flag, code);
            return false;
        }
        // (no s_sleep between two polls: a poll waits for its own answer -- a memory round trip -- anyway; without the 128 idle cycles
        // the release is seen a little earlier: 6.73 -> 6.68 us per step, same box, five pairs of builds)
    }
}

// One step of the serial key chain for step `t` of the next batch: (next, gradient, perturbation) =
// split(chain_key, 3) (svi.py:208-211, :413-414).  `last` advances the batch counters of the schedule.
template <bool CH = false>
__device__ __forceinline__ void chain_step(Sched* sched, StepSlot* slot, int t, int last)
{
    // lanes 0-3 / 4-7 / 8-11 derive children 0 / 1 / 2 with the 4-lane ChaCha block (the call needs whole quads:
    // it is made by the first 64 threads of a workgroup, lanes >= 12 compute a discarded fourth copy).
    // CH: consecutive chain steps run in different workgroups of ONE launch, so the schedule goes through agent-scope
    // accesses (the slot is only read by later launches).
    const int lane = threadIdx.x & 63, q = lane & 3, child = (lane >> 2) < 3 ? (lane >> 2) : 0;
    const int32_t adam0 = ld_x<CH>(&sched->adam_i);
    const uint32_t batch0 = ld_x<CH>(&sched->batch_i);
    const uint32_t p0 = ld_x<CH>(&sched->key[q]), p1 = ld_x<CH>(&sched->key[4 + q]);
    const uint32_t p2 = ld_x<CH>(&sched->key[8 + q]), p3 = ld_x<CH>(&sched->key[12 + q]);
    uint32_t a, b;
    derive_child_quad_regs(p0, p1, p2, p3, (uint32_t)child, D3P_TAG_SPLIT, 0u, a, b);
    if (lane < 4) {          // next state key: only the key words and the (zero) counter/nonce change
        st_x<CH>(&sched->key[4 + q], a);
        st_x<CH>(&sched->key[8 + q], b);
        st_x<CH>(&sched->key[12 + q], 0u);
        if (lane == 0 && last) {
            st_x<CH>(&sched->adam_i, adam0 + t + 1);
            st_x<CH>(&sched->batch_i, batch0 + (uint32_t)(t + 1));
        }
    } else if (lane < 12) {  // gradient key (child 1), perturbation key (child 2)
        uint32_t* dst = lane < 8 ? slot->grad_key : slot->pert_key;
        dst[q] = p0;  // constants row
        dst[4 + q] = a;
        dst[8 + q] = b;
        dst[12 + q] = 0u;
    } else if (lane == 12) {
        slot->adam_i = adam0 + t;
        slot->batch_i = batch0 + (uint32_t)t;
    }
}

// chain_step for the links that ride in the tail of workgroup 0 of the 16-wave chained launch, one per step (d3p_logreg_chain.h).
// There the tail's length is the next-but-one step's problem (the workgroup that inherits the CU enters late), and the plain form
// spends two memory round trips before it derives anything (progress word of the previous link, then the schedule) and waits for
// its stores before it raises the progress word: measured 0.2 us per step of a long run.  Here the running key travels between
// consecutive links as eight self-validating 8-byte words {key word | link index} (`ll`, zeroed with the launch's arrival
// counters): ONE round trip that is its own readiness test, and the stores are not waited for.  Link 0 starts from the schedule
// (left by an earlier launch), the last link leaves the schedule for the next one; the counters of the schedule are constant
// during the launch (only the last link moves them).
__device__ __forceinline__ void chain_step_ll(Sched* sched, unsigned long long* ll, StepSlot* slot, int t, int last, uint32_t* abort_flag,
                                              uint32_t abort_code_)
{
    const int lane = threadIdx.x & 63, q = lane & 3, child = (lane >> 2) < 3 ? (lane >> 2) : 0;
    const int32_t adam0 = ld_x<true>(&sched->adam_i);
    const uint32_t batch0 = ld_x<true>(&sched->batch_i);
    const uint32_t p0 = ld_x<true>(&sched->key[q]);   // the constants row never changes
    uint32_t p1, p2, p3;
    if (t == 0) {
        p1 = ld_x<true>(&sched->key[4 + q]);
        p2 = ld_x<true>(&sched->key[8 + q]);
        p3 = ld_x<true>(&sched->key[12 + q]);
    } else {  // the key after link t - 1: words 4 + q, 8 + q tagged with t (a derived key's words 12 .. 15 are zero)
        unsigned long long w1 = 0ull, w2 = 0ull;
        bool ok = false;
        for (uint32_t spins = 0; spins <= D3P_WAIT_ROUNDS; ++spins) {
            w1 = __hip_atomic_load(ll + q, __ATOMIC_RELAXED, D3P_AGENT);
            w2 = __hip_atomic_load(ll + 4 + q, __ATOMIC_RELAXED, D3P_AGENT);
            const bool mine = (uint32_t)(w1 >> 32) == (uint32_t)t && (uint32_t)(w2 >> 32) == (uint32_t)t;
            if (__ballot(!mine) == 0ull) { ok = true; break; }
            if ((spins & 63u) == 63u && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, D3P_AGENT) != 0u) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) {
            if (lane == 0) chain_raise(abort_flag, abort_code_);
            return;
        }
        p1 = (uint32_t)w1;
        p2 = (uint32_t)w2;
        p3 = 0u;
    }
    uint32_t a, b;
    derive_child_quad_regs(p0, p1, p2, p3, (uint32_t)child, D3P_TAG_SPLIT, 0u, a, b);
    if (lane < 4) {
        if (!last) {
            const unsigned long long tag = (unsigned long long)(uint32_t)(t + 1) << 32;
            __hip_atomic_store(ll + q, tag | a, __ATOMIC_RELAXED, D3P_AGENT);
            __hip_atomic_store(ll + 4 + q, tag | b, __ATOMIC_RELAXED, D3P_AGENT);
        } else {  // the schedule of the next launch
            st_x<true>(&sched->key[4 + q], a);
            st_x<true>(&sched->key[8 + q], b);
            st_x<true>(&sched->key[12 + q], 0u);
            if (lane == 0) {
                st_x<true>(&sched->adam_i, adam0 + t + 1);
                st_x<true>(&sched->batch_i, batch0 + (uint32_t)(t + 1));
            }
        }
    } else if (lane < 12) {  // gradient key (child 1), perturbation key (child 2)
        uint32_t* dst = lane < 8 ? slot->grad_key : slot->pert_key;
        dst[q] = p0;
        dst[4 + q] = a;
        dst[8 + q] = b;
        dst[12 + q] = 0u;
    } else if (lane == 12) {
        slot->adam_i = adam0 + t;
        slot->batch_i = batch0 + (uint32_t)t;
    }
}

// Arguments of the one-launch-per-step mode (MODE 2) of k_logreg_main: the cross-workgroup sum of the
// clipped gradients goes through 64-bit FIXED-POINT integer atomics into R replicas (integer addition is
// associative, so the result is the exact sum of the fp32 workgroup partials and bitwise reproducible,
// unlike float atomics); the NEXT launch's prologue turns the pending sums into the parameter update.
struct StepFuse {
    long long* acc_prev;  // R x D3P_ACC_COLS(P): sums of the previous step (read in the prologue if apply_prev)
    long long* acc_cur;   // R x D3P_ACC_COLS(P): this step's sums (zeroed by the previous launch)
    long long* acc_next;  // R x D3P_ACC_COLS(P): zeroed here for the next launch
    int R;
    int apply_prev;
    int flush_only;       // apply the pending sums and return (after the last step of a run)
    // optimiser state, ping-ponged between two buffers: every workgroup reads the `_in` arrays and workgroup 0
    // publishes the updated values to the `_out` arrays, so a workgroup that starts late never sees a
    // half-published state (in == out only in the single-workgroup flush launch)
    const float* params_in;
    const float* m_in;
    const float* v_in;
    float* params_out;
    float* m_out;
    float* v_out;
    int32_t* adam_step;
    uint32_t* batch_index;       // nullable
    const float* prev_noise;     // P standard normals of the previous step
    const StepMeta* prev_meta;   // bias corrections / counters of the previous step
    float* prev_loss_out;        // nullable: loss of the previous step
    float dp_scale, lr, b1, b2, adam_eps;
    float prior_w, prior_b;
    double sg, inv_sg;           // fixed-point scale of the gradient columns (2^40 / C) and its inverse
    // run status (sticky, cleared once per run, read back by d3p_dpvi_logreg_run_status): [0] a bounded wait of the chained
    // launch ran out (the run stops advancing: no workgroup applies or publishes anything any more), [1] a workgroup
    // partial was not finite or left the fixed-point range (the next update turns the parameters and the loss into NaN, as
    // the reference's float arithmetic would)
    uint32_t* status;
    // piggy-backed key-chain step of the next batch (extra workgroup), nullable
    Sched* chain_sched;
    StepSlot* chain_slot;
    int chain_t, chain_last;
};

// Arguments of the chained form (MODE 3): ONE launch covers the K steps of a prepared batch.  The grid has
// K x (nw + 1) workgroups; workgroup i belongs to step i / (nw + 1).  Workgroups are dispatched in linear-id order, so
// every workgroup of step t is resident before any workgroup of step t + 1 is placed: a step-(t+1) workgroup can start
// its parameter-independent work (row loads, eps generation) the moment a CU frees up, and only its update prologue
// waits -- on a two-level arrival counter -- until all workgroups of step t have added their sums.  That overlaps the
// eps generation of step t + 1 with the tail of step t and removes the kernel-launch boundary between steps.
// Everything exchanged between workgroups inside the launch (accumulators, the ping-ponged optimiser state, counters,
// the key-chain schedule) is accessed with agent-scope relaxed atomics only (coherent at the memory side: no L2-wide
// flush/invalidate); waits are bounded and raise `abort_flag` instead of hanging.
struct ChainFuse {
    int nw;                      // compute workgroups per step (the (nw + 1)-th is the key-chain workgroup)
    int g0;                      // global index of step 0 of this launch (accumulator rotation, state ping-pong)
    int K;                       // steps in this launch
    StepSlot* slots;             // K slots of this batch
    const uint32_t* idx_base;    // K x B (nullable: rows are positions)
    const uint32_t* skeys_base;  // K x 2B
    const uint32_t* plist_base;  // K x B dense owned-position lists (nullable: every batch position is processed)
    const float* noise_base;     // K x P
    const StepSlot* prev_slot0;  // slot of step g0 - 1 (nullptr: nothing to apply before step 0)
    const float* prev_noise0;
    long long* acc_base;         // 3 x R x D3P_ACC_COLS(P)
    float* state[2][3];          // ping-ponged {params, m, v}
    float* losses;               // nullable; losses[g] of the run
    uint32_t* bar;               // K x D3P_BAR_WORDS arrival counters (zeroed before the launch) + chain progress word
    uint32_t* abort_flag;
    StepSlot* chain_slots;       // slots of the NEXT batch (key chain), K_next of them
    int K_next;
    int pregen;                  // pipelined geometry: noise of a wave's first two examples is generated before the release wait
};

struct MainArgs {
    const float* X;
    const float* y;
    const uint32_t* idx;     // nullable: row = p
    const uint8_t* mask;     // nullable
    const uint32_t* counts;  // nullable: valid iff p < counts[1]
    const uint32_t* plist;   // nullable: dense list of the batch positions this rank processes (valid and owned)
    const uint32_t* n_list;  // number of entries of plist
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
    int dbg;  // developer ablation switches (0 in production)
    unsigned long long* stamps;  // nullable: per-workgroup {start, end} wall_clock64 (timing entry point)
    int family, gexp;            // likelihood family (non-FULL kernels only), guide transform
    float nh_inv_var, ll_const;  // Gaussian family: -0.5 / sigma^2,  D * (log sigma + log(2 pi) / 2)
    StepFuse fuse;               // MODE 2 / 3
    ChainFuse chain;             // MODE 3 only
};

// Lane l of the wave that owns an example holds, for k < NK and i < V, the column pair
//   c0 = 64*V*k + V*l + i   (< half)      and      c1 = c0 + half   (< D)
// which is exactly one threefry2x32 call of jax's iota layout (words c0 and c0+half of the D-word
// stream), so on-chip eps generation wastes no words; V = 4 makes both X loads 16-byte wide.
// Per-example inputs whose loads are issued one example ahead of their use.
template <int NC>
struct ExLoad {
    float x0[NC], x1[NC];
    float e0[NC], e1[NC];  // guide noise when it is read from memory (issued with the row loads)
    float y;
    float xt, et;     // TAIL: feature / stored noise of the tail column (see k_logreg_main)
    uint32_t k0, k1;  // threefry sample key
    bool live;        // valid && held by this rank
};

// FULL: every lane's column pairs exist (D == 2 * 64 * V * NK, no intercept) -> no guards at all.
// EPS: where the guide noise comes from: 0 = generated on chip (threefry + erf_inv), 1 = read from
// a.eps_ext (parity mode), -1 = decided at run time.
// NK == 1 (d <= 512): 1024-thread workgroups (<= 128 VGPRs); wider rows keep more columns per lane in registers,
// so NK == 2 is built for 512-thread workgroups (<= 256 VGPRs, at most 8 waves) and NK >= 4 for 256-thread workgroups
// (one wave per SIMD: the whole 512-entry register file, AGPRs included, instead of scratch).
// TAIL (with FULL): a full tile of features PLUS the intercept column, D = 2 * 64 * V * NK + 1 -- the shape of
// examples/logistic_regression.py.  half = 64 * V * NK + 1 is odd: lane pairs (c, c + half), c < half - 1, cover columns
// 0 .. half - 2 and half .. D - 1 (the last one is the intercept); the second-half features are not 16-byte aligned and
// come through scalar loads.  The left-over pair (half - 1, -) is the "tail column": every lane computes it redundantly
// (one more threefry call per example) and lane 0 owns its accumulators.
// STAMPS: diagnostic instantiation of the chained form with phase stamps (D3P_DBG=32); production kernels carry none.
// (the scalar-load form needs fewer registers per tile slot: V = 1, NK = 2 fits 16 waves -- 114 VGPRs --, V = 1, NK = 4 eight -- 174)
#define D3P_MAIN_MAX_THREADS(V, NK) ((NK) == 1 ? 1024 : (NK) == 2 ? ((V) == 1 ? 1024 : 512) : ((NK) == 4 && (V) == 1) ? 512 : 256)
#define D3P_MAIN_MAX_WAVES(V, NK) (D3P_MAIN_MAX_THREADS(V, NK) / 64)
template <int V, int NK, int MODE, bool FULL, int EPS, bool TAIL = false, bool STAMPS = false>
__global__ void __launch_bounds__(D3P_MAIN_MAX_THREADS(V, NK)) k_logreg_main(MainArgs a_in)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool FUSE = MODE == 2 || MODE == 3;   // update applied in the prologue, fixed-point accumulators
    constexpr bool CHAIN = MODE == 3;               // K steps in one launch (see ChainFuse)
    MainArgs a = a_in;
    uint32_t bid = blockIdx.x, nblk = gridDim.x;    // workgroup index / count within the step
    int step_t = 0;
    if (CHAIN) {
        const ChainFuse& cf = a_in.chain;
        const uint32_t per = (uint32_t)cf.nw + 1u;
        step_t = (int)(blockIdx.x / per);
        bid = blockIdx.x % per;
        nblk = per;
        const int g = cf.g0 + step_t;
        const size_t Pc = 2 * (size_t)a_in.D, words = (size_t)D3P_ACC_R * D3P_ACC_COLS(Pc);
        a.idx = cf.idx_base ? cf.idx_base + (size_t)step_t * a_in.B : nullptr;
        a.counts = cf.slots[step_t].counts;
        a.skeys = cf.skeys_base + (size_t)step_t * 2 * a_in.B;
        if (cf.plist_base) {  // Poisson padding / row-sharded ranks: the step's valid, owned positions
            a.plist = cf.plist_base + (size_t)step_t * a_in.B;
            a.n_list = &cf.slots[step_t].n_owned;
        }
        StepFuse& f = a.fuse;
        f.acc_prev = cf.acc_base + (size_t)((g + 2) % 3) * words;
        f.acc_cur = cf.acc_base + (size_t)(g % 3) * words;
        f.acc_next = cf.acc_base + (size_t)((g + 1) % 3) * words;
        const int in = g > 0 ? ((g - 1) & 1) : 0, out = g & 1;
        f.params_in = cf.state[in][0]; f.m_in = cf.state[in][1]; f.v_in = cf.state[in][2];
        f.params_out = cf.state[out][0]; f.m_out = cf.state[out][1]; f.v_out = cf.state[out][2];
        const StepSlot* ps = step_t > 0 ? cf.slots + (step_t - 1) : cf.prev_slot0;
        f.apply_prev = ps != nullptr;
        f.prev_meta = ps ? reinterpret_cast<const StepMeta*>(&ps->adam_i) : nullptr;
        f.prev_noise = step_t > 0 ? cf.noise_base + (size_t)(step_t - 1) * Pc : cf.prev_noise0;
        f.prev_loss_out = (cf.losses && g > 0) ? cf.losses + (g - 1) : nullptr;
        f.flush_only = 0;
        f.status = cf.abort_flag;
        f.chain_slot = step_t < cf.K_next ? cf.chain_slots + step_t : nullptr;
        f.chain_t = step_t;
        f.chain_last = step_t == cf.K_next - 1;
    }
    const bool eps_from_mem = (EPS == 1) || (EPS < 0 && a.eps_ext != nullptr);
    const int SS = (a.dbg & 32) ? 8 : 2;  // diagnostic build: 8 phase stamps per workgroup
    const long long clk0 = (a.dbg & 32) ? clock64() : 0;
    if (!CHAIN && a.stamps && threadIdx.x == 0) a.stamps[SS * bid] = wall_clock64();
    // D3P_DBG=32 phase stamps (100 MHz wall clock).  Chained launch: thread 0 keeps 16 stamps in LDS -- no global stores
    // between the phases, whose acknowledgements the in-order vmcnt would charge to the next phase -- and copies them out
    // when the workgroup is done; the last two steps of the launch are recorded.  Other modes: straight to memory.
#define D3P_STAMP(k)                                                                                                     \
    if (CHAIN) {                                                                                                         \
        if (STAMPS && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(lds)[stamp_off + (k)] = wall_clock64();    \
    } else if ((k) < 8 && (a.dbg & 32) && a.stamps && threadIdx.x == 0) {                                                \
        a.stamps[8 * bid + (k)] = wall_clock64();                                                                        \
    }
    constexpr int NC = V * NK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int D = a.D, half = a.half, P = 2 * D;
    const uint32_t total_waves = (nblk - ((CHAIN || (MODE == 2 && a.fuse.chain_slot)) ? 1u : 0u)) * W;
    const uint32_t gw = bid * W + wave;

    // ---- stage the derived parameter columns in LDS once per workgroup (5 x D floats)
    float* pk = lds;                       // [loc | s | sg | q | lc]
    float* red = lds + ((5 * D + 3) & ~3); // W x P reduction buffer (MODE 0) + 2W tail
    const int stamp_off = (((5 * D + 3) & ~3) + W * P + 2 * W + 4) / 2;  // in 8-byte units, behind the 4 status words
    D3P_STAMP(0)
    // Update prologue of MODE 2 / 3 (finish_prologue below).  Measured: issuing its loads here and keeping them in registers
    // across the eps generation costs 12 extra VGPRs -> 128 VGPRs + scratch and a slower kernel (14.8 vs 12.5 us), so the
    // prologue runs in one piece.
    long long pend_n8[D3P_ACC_R];
    float pend_bc1 = 1.f, pend_bc2 = 1.f;
    // per-polling-wave verdicts of the chained launch's wait (see finish_prologue): 4 words behind the loss / count tail
    auto okw = [&]() { return reinterpret_cast<uint32_t*>(red + (size_t)W * P + 2 * W); };
    if (FUSE) {
        const StepFuse& f = a.fuse;
        const int PA = D3P_ACC_COLS(P);
        if (bid == nblk - 1 && (CHAIN || f.chain_slot)) {  // piggy-backed key-chain workgroup
            if (f.chain_slot && threadIdx.x < 64) {
                bool go = true;
                if (CHAIN && step_t > 0)  // the schedule is serial: wait for the chain step of the previous step
                    go = chain_wait(a.chain.bar + (size_t)a.chain.K * D3P_BAR_WORDS, (uint32_t)step_t, a.chain.abort_flag);
                if (go) chain_step<CHAIN>(f.chain_sched, f.chain_slot, f.chain_t, f.chain_last);
                if (CHAIN) {
                    __builtin_amdgcn_s_waitcnt(0);
                    if (threadIdx.x == 0)
                        __hip_atomic_store(a.chain.bar + (size_t)a.chain.K * D3P_BAR_WORDS, (uint32_t)step_t + 1u, __ATOMIC_RELAXED,
                                           D3P_AGENT);
                }
            }
            return;
        }
        // zero the accumulator of the NEXT step (last read one step ago, never touched in this one).  Chained form: the
        // workgroups of step t - 1 may still be reading it in their prologue, so it is zeroed after the wait below.
        // (grid-stride over the COMPUTE workgroups: a small grid -- few examples per rank, wide rows -- has fewer threads than
        // accumulator words, and words left uncleared would keep the sums of step g - 3, the count column included)
        if (!CHAIN) {
            const int ncomp = (int)nblk - (f.chain_slot ? 1 : 0);
            for (int i = bid * blockDim.x + threadIdx.x; i < D3P_ACC_R * PA; i += ncomp * (int)blockDim.x) f.acc_next[i] = 0;
        }
    } else {
        for (int i = threadIdx.x; i < 5 * D; i += blockDim.x) pk[i] = a.pack[i];
    }
    auto pack_column = [&](int col, float x) {
        const int site = col >= D, e = col - site * D;
        if (site == 0) {
            pk[e] = x;
        } else {
            float sp, sgm;
            guide_scale(a.gexp, x, sp, sgm);
            const float ps = (e < a.d) ? a.fuse.prior_w : a.fuse.prior_b;
            pk[D + e] = sp;
            pk[2 * D + e] = sgm;
            pk[3 * D + e] = a.inv_obs * sgm * __builtin_amdgcn_rcpf(sp);
            pk[4 * D + e] = __logf(ps) - __logf(sp);
        }
    };
    bool prologue_done = false;
    bool flush_params_bad = false;   // flush launch: a parameter the pending step ran with is not finite (set in front of finish_prologue)
    // returns true when the run was aborted and this WHOLE workgroup knows it (pipelined form); the 16-wave form learns it
    // behind the staging barrier (stage_aborted)
    auto finish_prologue = [&]() -> bool {
        if (!FUSE || prologue_done) return false;
        prologue_done = true;
        const StepFuse& f = a.fuse;
        const int PA = D3P_ACC_COLS(P);
        const bool all_waves = CHAIN && a.chain.pregen && !f.flush_only && !(a.dbg & 128);
        // ---- run status first: an aborted run (a bounded wait ran out somewhere) must not advance any further
        if (CHAIN) {
            // every workgroup of the previous step must have added its sums (and published the state) before the
            // prologue reads them; the first step of a launch follows a kernel boundary instead (then only the abort flag
            // is looked at).  Pipelined form: every wave has prepared its examples already, so ONE wave polls and, behind a
            // workgroup barrier, all waves share the prologue; 16-wave form: each of the (up to 4) prologue waves polls for
            // itself.  The last arriver of each group sets that group's flag; a waiter polls all of them at once.
            const int PWc = all_waves ? W : (W < 4 ? W : 4);
            if (all_waves ? wave == 0 : wave < PWc) {
                const bool ok = chain_wait_groups(
                    step_t > 0 ? a.chain.bar + (size_t)(step_t - 1) * D3P_BAR_WORDS + D3P_BAR_LINE * (1 + D3P_BAR_GROUPS) : nullptr,
                    (uint32_t)a.chain.nw < D3P_BAR_GROUPS ? (uint32_t)a.chain.nw : D3P_BAR_GROUPS, a.chain.abort_flag);
                if (lane == 0) okw()[wave] = ok ? 0u : 1u;
                if (!ok && !all_waves) return false;  // (the other waves learn it behind the staging barrier, see stage_aborted)
            }
            if (all_waves) {
                __syncthreads();
                if (okw()[0] != 0u) return true;
            }
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            D3P_STAMP(7)  // release seen
        } else if (f.status && __hip_atomic_load(f.status, __ATOMIC_RELAXED, D3P_AGENT) != 0u) {
            // (flush launch behind an aborted chained launch: the pending sums are incomplete)
            if (bid == 0 && threadIdx.x == 0 && f.prev_loss_out) *f.prev_loss_out = __builtin_nanf("");
            return true;
        }
        if (!f.apply_prev) {  // no pending update: derive the columns from the parameters as they are
            if (CHAIN)  // (first step of a run) nobody reads the next accumulator yet
                for (int i = bid * blockDim.x + threadIdx.x; i < D3P_ACC_R * PA; i += a.chain.nw * blockDim.x)
                    st_x<true>(f.acc_next + i, 0ll);
            for (int col = threadIdx.x; col < P; col += blockDim.x) pack_column(col, f.params_in[col]);
            return false;
        }
        if (CHAIN) {
            // now nobody reads the next accumulator any more (the previous step's prologues are over): zero it
            const int PWc = all_waves ? W : (W < 4 ? W : 4);
            if (wave < PWc)
                for (int i = bid * (64 * PWc) + (int)threadIdx.x; i < D3P_ACC_R * PA; i += a.chain.nw * 64 * PWc)
                    st_x<true>(f.acc_next + i, 0ll);
        }
        // n (the count column) is consumed only AFTER the column loads below have been issued: its load was issued
        // at kernel entry, and summing it here first would put a second memory round trip in front of those loads
        const float Bf = (float)a.B;
        // the count column is loaded in the same batch as the gradient columns (one memory round trip, not two)
        // (wave-uniform values: the compiler moves them to SGPRs with a wait right after their loads, so they are issued
        // AFTER the per-lane column loads -- otherwise every uniform load costs its own memory round trip first)
        auto load_n = [&]() {
#pragma unroll
            for (int r = 0; r < D3P_ACC_R; ++r) pend_n8[r] = ld_x<CHAIN>(f.acc_prev + (size_t)r * PA + P + 1);
            pend_bc1 = f.prev_meta->bc1;
            pend_bc2 = f.prev_meta->bc2;
        };
        auto count_n = [&]() {
            long long nll = 0;
#pragma unroll
            for (int r = 0; r < D3P_ACC_R; ++r) nll += pend_n8[r];
            // a workgroup that saw a non-finite partial also added 2^44 to the count column: NaN count -> NaN gradient,
            // parameters and loss (what the reference's float sums give), never finite garbage
            return nll >= (1ll << 40) ? __builtin_nanf("") : (float)nll;
        };
        {
            // Only the first PW waves (one per SIMD) run the prologue, CB columns per thread with all loads in
            // flight together; the other waves go straight to their row loads and eps generation, so the
            // prologue's memory latency hides behind their VALU work.  Flush launches use every wave.
            const int PW = (f.flush_only || all_waves) ? W : (W < 4 ? W : 4);
            // Chained form, P a multiple of 8 and 16-byte aligned state: a thread takes 4 ADJACENT columns and fetches them
            // with 12 sixteen-byte agent-scope loads instead of 32 four/eight-byte ones (same bytes, fewer and larger
            // memory-side requests), and workgroup 0 publishes the state with 16-byte stores.  Measured A/B on one box
            // (D3P_DBG=64 selects the narrow loads): 16-wave workgroups 11.89 -> 11.59 us per step; in the pipelined form
            // (8-wave workgroups, `pregen`) the narrow loads are the faster ones (10.93 vs 11.08), so it keeps them.
            const bool wide16 =
                CHAIN && !a.chain.pregen && !f.flush_only && (P & 7) == 0 && !(a.dbg & 64) &&
                ((((uintptr_t)f.acc_prev | (uintptr_t)f.params_in | (uintptr_t)f.m_in | (uintptr_t)f.v_in | (uintptr_t)f.params_out |
                   (uintptr_t)f.m_out | (uintptr_t)f.v_out | (uintptr_t)f.prev_noise) & 15) == 0);
            if (wide16) {
                if (wave < PW) {
                    const auto r_acc = __builtin_amdgcn_make_buffer_rsrc((void*)f.acc_prev, 0, D3P_ACC_R * PA * 8, 0x00020000);
                    const auto r_x = __builtin_amdgcn_make_buffer_rsrc((void*)f.params_in, 0, P * 4, 0x00020000);
                    const auto r_m = __builtin_amdgcn_make_buffer_rsrc((void*)f.m_in, 0, P * 4, 0x00020000);
                    const auto r_v = __builtin_amdgcn_make_buffer_rsrc((void*)f.v_in, 0, P * 4, 0x00020000);
                    for (int col = 4 * (int)threadIdx.x; col < P; col += 4 * 64 * PW) {
                        d3p_u32x4 a8[D3P_ACC_R][2];
#pragma unroll
                        for (int r = 0; r < D3P_ACC_R; ++r) {
                            a8[r][0] = __builtin_amdgcn_raw_buffer_load_b128(r_acc, (r * PA + col) * 8, 0, 16);
                            a8[r][1] = __builtin_amdgcn_raw_buffer_load_b128(r_acc, (r * PA + col) * 8 + 16, 0, 16);
                        }
                        const d3p_u32x4 xv = __builtin_amdgcn_raw_buffer_load_b128(r_x, col * 4, 0, 16);
                        const d3p_u32x4 mv4 = __builtin_amdgcn_raw_buffer_load_b128(r_m, col * 4, 0, 16);
                        const d3p_u32x4 vv4 = __builtin_amdgcn_raw_buffer_load_b128(r_v, col * 4, 0, 16);
                        const float4 zv = *reinterpret_cast<const float4*>(f.prev_noise + col);
                        __builtin_amdgcn_sched_barrier(0);  // keep the uniform loads behind the column loads
                        load_n();
                        const float n = count_n();
                        const float factor = (n == 0.0f) ? 0.0f : Bf / n;  // svi.py:305
                        const float inv_B = 1.0f / Bf, inv_bc1 = 1.0f / pend_bc1, inv_bc2 = 1.0f / pend_bc2;
                        const float noise_scale = f.dp_scale * (a.clip / n), out_scale = a.obs_scale * factor;
                        const float z[4] = {zv.x, zv.y, zv.z, zv.w};
                        d3p_u32x4 xo, mo, vo;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            long long sll = 0;
#pragma unroll
                            for (int r = 0; r < D3P_ACC_R; ++r)
                                sll += (long long)(((unsigned long long)a8[r][j >> 1][2 * (j & 1) + 1] << 32) |
                                                   a8[r][j >> 1][2 * (j & 1)]);
                            const float tot = (float)((double)sll * f.inv_sg);
                            const float g = __fmaf_rn(z[j], noise_scale, tot * inv_B) * out_scale;
                            const float mm = (1.0f - f.b1) * g + f.b1 * __uint_as_float(mv4[j]);
                            const float vv = (1.0f - f.b2) * g * g + f.b2 * __uint_as_float(vv4[j]);
                            const float xx = __uint_as_float(xv[j]) - f.lr * (mm * inv_bc1) *
                                                                          __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv * inv_bc2) + f.adam_eps);
                            xo[j] = __float_as_uint(xx);
                            mo[j] = __float_as_uint(mm);
                            vo[j] = __float_as_uint(vv);
                        }
                        {  // derived columns, 16-byte LDS writes (4 adjacent columns never straddle the two sites: D % 4 == 0)
                            const int site = col >= D, e = col - site * D;
                            if (site == 0) {
                                *reinterpret_cast<float4*>(pk + e) = make_float4(__uint_as_float(xo[0]), __uint_as_float(xo[1]),
                                                                                 __uint_as_float(xo[2]), __uint_as_float(xo[3]));
                            } else {
                                float sp[4], sgm[4], qq[4], lc[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    guide_scale(a.gexp, __uint_as_float(xo[j]), sp[j], sgm[j]);
                                    const float ps = (e + j < a.d) ? a.fuse.prior_w : a.fuse.prior_b;
                                    qq[j] = a.inv_obs * sgm[j] * __builtin_amdgcn_rcpf(sp[j]);
                                    lc[j] = __logf(ps) - __logf(sp[j]);
                                }
                                *reinterpret_cast<float4*>(pk + D + e) = make_float4(sp[0], sp[1], sp[2], sp[3]);
                                *reinterpret_cast<float4*>(pk + 2 * D + e) = make_float4(sgm[0], sgm[1], sgm[2], sgm[3]);
                                *reinterpret_cast<float4*>(pk + 3 * D + e) = make_float4(qq[0], qq[1], qq[2], qq[3]);
                                *reinterpret_cast<float4*>(pk + 4 * D + e) = make_float4(lc[0], lc[1], lc[2], lc[3]);
                            }
                        }
                        if (bid == 0) {  // one workgroup publishes the state
                            __builtin_amdgcn_raw_buffer_store_b128(
                                xo, __builtin_amdgcn_make_buffer_rsrc((void*)f.params_out, 0, P * 4, 0x00020000), col * 4, 0, 16);
                            __builtin_amdgcn_raw_buffer_store_b128(
                                mo, __builtin_amdgcn_make_buffer_rsrc((void*)f.m_out, 0, P * 4, 0x00020000), col * 4, 0, 16);
                            __builtin_amdgcn_raw_buffer_store_b128(
                                vo, __builtin_amdgcn_make_buffer_rsrc((void*)f.v_out, 0, P * 4, 0x00020000), col * 4, 0, 16);
                        }
                    }
                }
            } else if (wave < PW) {
              auto narrow_cols = [&](auto cb_tag) {
                constexpr int CB = decltype(cb_tag)::value;
                const int stride = 64 * PW;
                for (int col0 = threadIdx.x; col0 < P; col0 += stride * CB) {
                    long long r8[CB][D3P_ACC_R];
                    float x[CB], m[CB], v[CB], z[CB];
#pragma unroll
                    for (int j = 0; j < CB; ++j) {
                        const int col = col0 + j * stride;
                        const int cc = col < P ? col : 0;
#pragma unroll
                        for (int r = 0; r < D3P_ACC_R; ++r) r8[j][r] = ld_x<CHAIN>(f.acc_prev + (size_t)r * PA + cc);
                        x[j] = ld_x<CHAIN>(f.params_in + cc);
                        m[j] = ld_x<CHAIN>(f.m_in + cc);
                        v[j] = ld_x<CHAIN>(f.v_in + cc);
                        z[j] = f.prev_noise[cc];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the uniform loads behind the column loads
                    load_n();
                    const float n = count_n();
                    const float factor = (n == 0.0f) ? 0.0f : Bf / n;  // svi.py:305
                    // This arithmetic sits on the critical path of every step (all waves wait for the derived columns), so
                    // the per-step constants are inverted once and the per-column divisions / square root use the 1-ulp
                    // hardware approximations (the two-kernel path keeps IEEE division; they agree to ~1e-7 relative).
                    const float inv_B = 1.0f / Bf, inv_bc1 = 1.0f / pend_bc1, inv_bc2 = 1.0f / pend_bc2;
                    const float noise_scale = f.dp_scale * (a.clip / n), out_scale = a.obs_scale * factor;
#pragma unroll
                    for (int j = 0; j < CB; ++j) {
                        const int col = col0 + j * stride;
                        if (col < P) {
                            long long sll = 0;
#pragma unroll
                            for (int r = 0; r < D3P_ACC_R; ++r) sll += r8[j][r];
                            const float tot = (float)((double)sll * f.inv_sg);
                            const float g = __fmaf_rn(z[j], noise_scale, tot * inv_B) * out_scale;
                            const float mm = (1.0f - f.b1) * g + f.b1 * m[j];
                            const float vv = (1.0f - f.b2) * g * g + f.b2 * v[j];
                            const float xx = x[j] - f.lr * (mm * inv_bc1) *
                                                        __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv * inv_bc2) + f.adam_eps);
                            if (bid == 0) {  // one workgroup publishes the state
                                st_x<CHAIN>(f.params_out + col, xx);
                                st_x<CHAIN>(f.m_out + col, mm);
                                st_x<CHAIN>(f.v_out + col, vv);
                            }
                            pack_column(col, xx);
                        }
                    }
                }
              };
              if (P <= 2 * 64 * PW) narrow_cols(std::integral_constant<int, 2>{});
              else narrow_cols(std::integral_constant<int, 4>{});
            }
        }
        if (bid == 0 && threadIdx.x == 0) {
            load_n();
            const float n = count_n();
            const float factor = (n == 0.0f) ? 0.0f : Bf / n;
            long long lll = 0, lhh = 0;
#pragma unroll
            for (int r = 0; r < D3P_ACC_R; ++r) {
                lll += ld_x<CHAIN>(f.acc_prev + (size_t)r * PA + P);
                lhh += ld_x<CHAIN>(f.acc_prev + (size_t)r * PA + P + 2);
            }
            if (f.prev_loss_out) {
                float lv = ((float)loss_join(lhh, lll) / Bf) * a.obs_scale * factor;
                if (n == 0.0f)
                    lv = (!CHAIN && f.flush_only) ? (flush_params_bad ? __builtin_nanf("") : 0.0f)
                                                  : empty_batch_loss(P, [&](int c) { return ld_x<CHAIN>(f.params_in + c); });
                *f.prev_loss_out = lv;
            }
            // (chained launch: the run's counters are k_flush's -- a word that workgroup 0 of a different step, i.e. a different XCD, plain-
            // stores every step keeps the value of whichever L2 is written back last; one launch per step: this launch is the only writer)
            if (!CHAIN) {
                *f.adam_step = f.prev_meta->adam_i + 1;
                if (f.batch_index) *f.batch_index = f.prev_meta->batch_i + 1u;
            }
        }
        return false;
    };
    // the 4 waves (one per SIMD) that apply the pending update in finish_prologue()  (s_setprio on them was measured to
    // change nothing: the SIMD shares its issue slots evenly whatever the priority)
    const bool prologue_wave = FUSE && !a.fuse.flush_only && a.fuse.apply_prev && wave < 4;
    // (16-wave chained form, D3P_DBG=32 stamps: release seen 1.8 us after entry, prologue 3.2 us, staging barrier passed
    // 2.4-3.5 us later -- it waits for the 12 other waves, whose noise generation starts behind their index -> key load chain --
    // then dot / gradient / reduction 2.0 us, atomics + arrival 2.15 us.  Generating every wave's noise BEFORE the prologue
    // inside ONE workgroup does not help (12.2 -> 18.4 us per step); starting it a whole step earlier does: `pregen` below.)
    const bool gauss = !FULL && a.family == D3P_FAMILY_GAUSS_MEAN;
    int c0[NC], c1[NC];
    bool ok0[NC], ok1[NC];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int n = k * V + i;
            c0[n] = 64 * V * k + V * lane + i;
            c1[n] = c0[n] + half;
            ok0[n] = FULL || (c0[n] < half);
            ok1[n] = FULL || (ok0[n] && (c1[n] < D));
        }

    float accg0[NC], acch0[NC], accg1[NC], acch1[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) accg0[n] = acch0[n] = accg1[n] = acch1[n] = 0.f;
    float loss_acc = 0.f, n_acc = 0.f;
    float acc_gt = 0.f, acc_ht = 0.f;  // TAIL: the tail column's two accumulators (identical in every lane)
    const int ct = half - 1;           // TAIL: the tail column
    const uint32_t n_valid = a.counts ? a.counts[1] : a.B;

    // issue every global load of example p (index -> row -> features, label, sample key)
    auto issue = [&](uint32_t p, ExLoad<NC>& L) {
        const uint32_t row_g = a.idx ? a.idx[p] : p;
        const bool valid = (p < n_valid) && (a.mask ? a.mask[p] != 0 : true);
        const bool mine = (uint64_t)row_g >= a.row_lo && (uint64_t)row_g < a.row_hi;
        L.live = valid && mine;
        L.k0 = L.k1 = 0u;
        L.y = 0.f;
        L.xt = L.et = 0.f;
#pragma unroll
        for (int n = 0; n < NC; ++n) L.x0[n] = L.x1[n] = L.e0[n] = L.e1[n] = 0.f;
        if (!L.live && MODE != 1) return;
        if (!mine) return;  // MODE 1 writes zeros for rows it cannot read
        const size_t row = (size_t)((uint64_t)row_g - a.row_lo);
        const float* xrow = a.X + row * (size_t)a.d;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (V == 4) {
                const int n = k * 4;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (ok0[n]) v0 = *reinterpret_cast<const float4*>(xrow + c0[n]);
                L.x0[n] = v0.x; L.x0[n + 1] = v0.y; L.x0[n + 2] = v0.z; L.x0[n + 3] = v0.w;
                if (TAIL) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) L.x1[n + i] = c1[n + i] < a.d ? xrow[c1[n + i]] : 1.0f;  // column d = intercept
                } else {
                    if (ok1[n]) v1 = *reinterpret_cast<const float4*>(xrow + c1[n]);
                    L.x1[n] = v1.x; L.x1[n + 1] = v1.y; L.x1[n + 2] = v1.z; L.x1[n + 3] = v1.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const int n = k * V + i;
                    L.x0[n] = ok0[n] ? (c0[n] < a.d ? xrow[c0[n]] : 1.0f) : 0.f;  // column d = intercept
                    L.x1[n] = ok1[n] ? (c1[n] < a.d ? xrow[c1[n]] : 1.0f) : 0.f;
                }
            }
        }
        L.y = a.y ? a.y[row] : 0.f;
        if (TAIL) L.xt = xrow[half - 1];
        if (!eps_from_mem) {
            L.k0 = a.skeys[2 * p];
            L.k1 = a.skeys[2 * p + 1];
        } else {
            const float* er = a.eps_ext + (size_t)p * D;
            if (TAIL) L.et = er[half - 1];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if (V == 4 && !TAIL) {  // (TAIL: rows of D = odd floats are not 16-byte aligned)
                    const int n = k * 4;
                    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                    if (ok0[n]) v0 = *reinterpret_cast<const float4*>(er + c0[n]);
                    if (ok1[n]) v1 = *reinterpret_cast<const float4*>(er + c1[n]);
                    L.e0[n] = v0.x; L.e0[n + 1] = v0.y; L.e0[n + 2] = v0.z; L.e0[n + 3] = v0.w;
                    L.e1[n] = v1.x; L.e1[n + 1] = v1.y; L.e1[n + 2] = v1.z; L.e1[n + 3] = v1.w;
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const int n = k * V + i;
                        L.e0[n] = ok0[n] ? er[c0[n]] : 0.f;
                        L.e1[n] = ok1[n] ? er[c1[n]] : 0.f;
                    }
                }
            }
        }
    };

    // items: either every batch position (k == p) or the entries of the dense owned-position list
    const uint32_t n_items = a.plist ? *a.n_list : a.B;
    auto pos = [&](uint32_t k) { return a.plist ? a.plist[k] : k; };
    ExLoad<NC> cur;
    uint32_t p = gw;  // item index
    // The loads of the first example are issued BEFORE the update prologue also by the prologue waves: their index -> row ->
    // features chain then completes while those waves wait for the previous step's release, instead of after the prologue.
    if (!(FUSE && a.fuse.flush_only) && p < n_items) issue(pos(p), cur);
    // Pipelined chained form (8-wave workgroups, two resident per CU: the workgroups of step t + 1 arrive while step t is
    // still in its exchange).  Every wave -- the prologue waves too -- generates the noise of its first TWO examples into
    // its own (still unused) row of the reduction buffer BEFORE waiting for the previous step's release, so between that
    // release and this workgroup's arrival there is only the prologue, the gradient arithmetic and the exchange.
    const bool pregen = CHAIN && FULL && V == 4 && NK == 1 && !eps_from_mem && a.chain.pregen && !a.fuse.flush_only && !(a.dbg & 1);
    int it = 0;  // examples this wave has gone through
    ExLoad<NC> pre;         // second example, loaded ahead of the prologue as well (its index -> row chain would otherwise sit
    bool have_pre = false;  // between the release and the first gradient)
    if (NC == 4 && pregen && p < n_items) {
        float* er = red + (size_t)wave * P;
        const uint32_t p2 = p + total_waves;
        if (p2 < n_items) {
            issue(pos(p2), pre);
            have_pre = true;
        }
        auto gen = [&](uint32_t k0, uint32_t k1, float* dst) {
            float v0[4], v1[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                uint32_t b0, b1;
                threefry2x32(k0, k1, (uint32_t)c0[n], (uint32_t)c1[n], b0, b1);
                v0[n] = bits_to_normal_wu(b0);
                v1[n] = bits_to_normal_wu(b1);
            }
            *reinterpret_cast<float4*>(dst + c0[0]) = make_float4(v0[0], v0[1], v0[2], v0[3]);
            if (TAIL) {  // odd half: the second-half columns are not 16-byte aligned; + the tail column (every lane, same value)
#pragma unroll
                for (int n = 0; n < 4; ++n) dst[c1[n]] = v1[n];
                uint32_t b0, b1;
                threefry2x32(k0, k1, (uint32_t)ct, 0u, b0, b1);
                if (lane == 0) dst[ct] = bits_to_normal_wu(b0);
            } else {
                *reinterpret_cast<float4*>(dst + c1[0]) = make_float4(v1[0], v1[1], v1[2], v1[3]);
            }
        };
        if (cur.live) gen(cur.k0, cur.k1, er);
        if (have_pre && pre.live) gen(pre.k0, pre.k1, er + D);
    }
    // After a staging barrier every wave looks at the verdicts of the polling waves: an aborted run leaves here, as a whole
    // workgroup, before it adds or arrives anywhere.
    const bool all_waves_k = CHAIN && a.chain.pregen && !a.fuse.flush_only && !(a.dbg & 128);
    auto stage_aborted = [&]() {
        uint32_t bad = okw()[0];
        if (!all_waves_k)
            for (int w = 1; w < (W < 4 ? W : 4); ++w) bad |= okw()[w];
        return bad != 0u;
    };
#define D3P_STAGE_SYNC()                          \
    do {                                          \
        __syncthreads();                          \
        staged = true;                            \
        if (CHAIN && stage_aborted()) return;     \
    } while (0)
    D3P_STAMP(8)  // parameter-independent work done (rows loaded, noise of the first two examples generated)
    // Pipelined form: from here on this workgroup is on the critical path of the step (wait for the release, update, gradients,
    // exchange), while the co-resident workgroup of the NEXT step generates its noise on the same SIMDs -- integer VALU work
    // that issues at 4 cycles per wave64 instruction.  Raising the wave priority lets the critical instructions go first.
    if (CHAIN && (a.dbg & 2)) __builtin_amdgcn_s_setprio(3);
    if (FUSE) {
        // Flush launch (one workgroup): it publishes to the caller's arrays, which after an odd number of launches are also the
        // arrays it READS -- the reporter of an empty batch's loss (thread 0, at the end of the prologue) would scan parameters the
        // other threads have already replaced.  Every thread looks at its share first; the vote goes through the reduction buffer.
        if (!CHAIN && a.fuse.flush_only && a.fuse.apply_prev && a.fuse.prev_loss_out) {
            int bad = 0;
            for (int c = threadIdx.x; c < P; c += blockDim.x) bad |= !(fabsf(a.fuse.params_in[c]) <= 3.402823466e38f);
            const bool wave_bad = __any(bad);
            if (lane == 0) red[wave] = wave_bad ? 1.0f : 0.0f;
            __syncthreads();
            bad = 0;
            for (int w = 0; w < W; ++w) bad |= red[w] != 0.0f;
            flush_params_bad = bad != 0;
            __syncthreads();
        }
        const bool aborted = finish_prologue();
        D3P_STAMP(2)
        if (a.fuse.flush_only || aborted) return;  // (aborted: uniform, decided behind the workgroup barrier of the poll)
    }
    // The derived columns are first needed AFTER the noise of the first example has been generated, so for the waves that
    // do not run the prologue the staging barrier sits behind that phase.  The prologue waves arrive at the barrier as
    // soon as the columns are in LDS -- BEFORE generating their own noise: the other 12 waves then start their arithmetic
    // ~1.8 us earlier and the prologue waves' noise generation overlaps with it (every wave passes exactly one barrier).
    bool staged = false;
    if ((prologue_wave || pregen) && p < n_items) D3P_STAGE_SYNC();
    if (!(p < n_items)) { finish_prologue(); D3P_STAGE_SYNC(); }

    while (p < n_items) {
        const uint32_t pn = p + total_waves;
        ExLoad<NC> nxt;
        if (it == 0 && have_pre) nxt = pre;
        else if (pn < n_items) issue(pos(pn), nxt);  // prefetch: in flight while the current example computes

        if (cur.live || MODE == 1) {
            // ---- guide noise eps_i (svi.py:289-290): parity mode reads it, otherwise threefry on chip
            float e0[NC], e1[NC];
            if (eps_from_mem) {
#pragma unroll
                for (int n = 0; n < NC; ++n) { e0[n] = cur.e0[n]; e1[n] = cur.e1[n]; }
            } else if (a.dbg & 1) {
#pragma unroll
                for (int n = 0; n < NC; ++n) { e0[n] = __uint_as_float(cur.k0 & 0x3fffffffu); e1[n] = e0[n] * 0.5f; }
            } else {
                if (pregen && it < 2) {  // generated before the update prologue (see `pregen`)
                    const float* er = red + (size_t)wave * P + it * D;
                    float t0[4], t1[4];
                    *reinterpret_cast<float4*>(t0) = *reinterpret_cast<const float4*>(er + c0[0]);
                    if (TAIL) {
#pragma unroll
                        for (int n = 0; n < NC && n < 4; ++n) t1[n] = er[c1[n]];
                    } else {
                        *reinterpret_cast<float4*>(t1) = *reinterpret_cast<const float4*>(er + c1[0]);
                    }
#pragma unroll
                    for (int n = 0; n < NC && n < 4; ++n) { e0[n] = t0[n]; e1[n] = t1[n]; }
                } else {
#pragma unroll
                    for (int n = 0; n < NC; ++n) {
                        uint32_t b0, b1;
                        threefry2x32(cur.k0, cur.k1, (uint32_t)c0[n], ok1[n] ? (uint32_t)c1[n] : 0u, b0, b1);
                        const float v0 = bits_to_normal_wu(b0), v1 = bits_to_normal_wu(b1);
                        e0[n] = ok0[n] ? v0 : 0.f;
                        e1[n] = ok1[n] ? v1 : 0.f;
                    }
                }
            }

            float et = 0.f;
            if (TAIL) {
                if (eps_from_mem) {
                    et = cur.et;
                } else if (pregen && it < 2) {
                    et = red[(size_t)wave * P + it * D + ct];
                } else {
                    uint32_t b0, b1;
                    threefry2x32(cur.k0, cur.k1, (uint32_t)ct, 0u, b0, b1);
                    et = bits_to_normal_wu(b0);
                }
            }
            D3P_STAMP(3)
            if (!staged) { finish_prologue(); D3P_STAGE_SYNC(); }
            // ---- z = loc + s * eps, logit t = x . z   (derived columns come from LDS)
            float z0[NC], z1[NC];
            float tp = 0.f;
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                const float l0 = ok0[n] ? pk[c0[n]] : 0.f, l1 = ok1[n] ? pk[c1[n]] : 0.f;
                const float s0 = ok0[n] ? pk[D + c0[n]] : 0.f, s1 = ok1[n] ? pk[D + c1[n]] : 0.f;
                z0[n] = __fmaf_rn(s0, e0[n], l0);
                z1[n] = __fmaf_rn(s1, e1[n], l1);
                if (gauss) {  // residuals take the place of the features: dloglik/dz = (x - z) / sigma^2
                    cur.x0[n] = ok0[n] ? cur.x0[n] - z0[n] : 0.f;
                    cur.x1[n] = ok1[n] ? cur.x1[n] - z1[n] : 0.f;
                    tp = __fmaf_rn(cur.x0[n], cur.x0[n], tp);
                    tp = __fmaf_rn(cur.x1[n], cur.x1[n], tp);
                } else {
                    tp = __fmaf_rn(cur.x0[n], z0[n], tp);
                    tp = __fmaf_rn(cur.x1[n], z1[n], tp);
                }
            }
            float t = wave_sum(tp);  // logit x.z, or the squared residual norm
            float zt = 0.f;
            if (TAIL) {
                zt = __fmaf_rn(pk[D + ct], et, pk[ct]);
                t = __fmaf_rn(cur.xt, zt, t);
            }
            D3P_STAMP(4)
            // A = d(-lik_scale * inv_obs * loglik)/dt up to the per-column factor; loglik itself
            float A, loglik;
            if (gauss) {
                A = 2.0f * a.A_scale * a.nh_inv_var;
                loglik = __fmaf_rn(a.nh_inv_var, t, -a.ll_const);
            } else {
                A = a.A_scale * (sigmoid_f(t) - cur.y);
                loglik = cur.y * t - softplus_f(t);
            }

            // ---- per-example gradient, its squared norm and the latent part of the loss
            float g0[NC], h0[NC], g1[NC], h1[NC];
            float n2 = 0.f, lp = 0.f;
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                const bool ic0 = !FULL && a.icpt && (c0[n] == a.d), ic1 = (!FULL || TAIL) && a.icpt && (c1[n] == a.d);
                const float sg0 = ok0[n] ? pk[2 * D + c0[n]] : 0.f, sg1 = ok1[n] ? pk[2 * D + c1[n]] : 0.f;
                const float q0 = ok0[n] ? pk[3 * D + c0[n]] : 0.f, q1 = ok1[n] ? pk[3 * D + c1[n]] : 0.f;
                const float lc0 = ok0[n] ? pk[4 * D + c0[n]] : 0.f, lc1 = ok1[n] ? pk[4 * D + c1[n]] : 0.f;
                g0[n] = __fmaf_rn(ic0 ? a.c1_b : a.c1_w, z0[n], A * cur.x0[n]);
                g1[n] = __fmaf_rn(ic1 ? a.c1_b : a.c1_w, z1[n], A * cur.x1[n]);
                h0[n] = __fmaf_rn(g0[n] * e0[n], sg0, -q0);
                h1[n] = __fmaf_rn(g1[n] * e1[n], sg1, -q1);
                n2 = __fmaf_rn(g0[n], g0[n], n2);
                n2 = __fmaf_rn(h0[n], h0[n], n2);
                n2 = __fmaf_rn(g1[n], g1[n], n2);
                n2 = __fmaf_rn(h1[n], h1[n], n2);
                lp += __fmaf_rn((ic0 ? a.hz_b : a.hz_w) * z0[n], z0[n], __fmaf_rn(-0.5f * e0[n], e0[n], lc0));
                lp += __fmaf_rn((ic1 ? a.hz_b : a.hz_w) * z1[n], z1[n], __fmaf_rn(-0.5f * e1[n], e1[n], lc1));
            }
            n2 = wave_sum(n2);
            lp = wave_sum(lp);
            float gt = 0.f, ht = 0.f;
            if (TAIL) {  // the tail column is a feature column (prior of the weights)
                gt = __fmaf_rn(a.c1_w, zt, A * cur.xt);
                ht = __fmaf_rn(gt * et, pk[2 * D + ct], -pk[3 * D + ct]);
                n2 = __fmaf_rn(gt, gt, __fmaf_rn(ht, ht, n2));
                lp += __fmaf_rn(a.hz_w * zt, zt, __fmaf_rn(-0.5f * et, et, pk[4 * D + ct]));
            }
            // L_i = inv_obs * ((logq - logp) - lik_scale * loglik)   (svi.py:278-281)
            const float L = a.inv_obs * (lp - a.lik_scale * loglik);

            if (MODE != 1) {
                // clip factor 1/max(1, ||g||/C) (svi.py:121-122) folded into the running sum (svi.py:343-346)
                // = min(1, C / ||g||) with the hardware reciprocal square root (1 ulp; ||g|| = 0 gives min(1, inf) = 1)
                const float cf = fminf(1.0f, a.clip * __builtin_amdgcn_rsqf(n2));
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    accg0[n] = __fmaf_rn(cf, g0[n], accg0[n]);
                    acch0[n] = __fmaf_rn(cf, h0[n], acch0[n]);
                    accg1[n] = __fmaf_rn(cf, g1[n], accg1[n]);
                    acch1[n] = __fmaf_rn(cf, h1[n], acch1[n]);
                }
                if (TAIL) {
                    acc_gt = __fmaf_rn(cf, gt, acc_gt);
                    acc_ht = __fmaf_rn(cf, ht, acc_ht);
                }
                loss_acc += L;
                n_acc += 1.0f;
            } else {
                const float m = cur.live ? 1.0f : 0.0f;  // loss * mask => zero loss and gradient (svi.py:281)
                float* gr = a.px_grads + (size_t)p * P;
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    if (ok0[n]) { gr[c0[n]] = g0[n] * m; gr[D + c0[n]] = h0[n] * m; }
                    if (ok1[n]) { gr[c1[n]] = g1[n] * m; gr[D + c1[n]] = h1[n] * m; }
                }
                if (lane == 0) a.px_loss[p] = L * m * a.obs_scale * a.meta[1];  // svi.py:306
            }
        }
        if (!staged) { finish_prologue(); D3P_STAGE_SYNC(); }  // example skipped before reaching the barrier
        cur = nxt;
        p = pn;
        ++it;
    }

    D3P_STAMP(5)
    if (MODE != 1) {
        // ---- workgroup reduction through LDS, one partial row per workgroup (fixed order)
        float* mine = red + (size_t)wave * P;
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            if (ok0[n]) { mine[c0[n]] = accg0[n]; mine[D + c0[n]] = acch0[n]; }
            if (ok1[n]) { mine[c1[n]] = accg1[n]; mine[D + c1[n]] = acch1[n]; }
        }
        if (TAIL && lane == 0) { mine[ct] = acc_gt; mine[D + ct] = acc_ht; }
        float* tail = red + (size_t)W * P;
        if (lane == 0) { tail[2 * wave] = loss_acc; tail[2 * wave + 1] = n_acc; }
        __syncthreads();
        D3P_STAMP(6)
        if (FUSE) {
            // fixed-point integer atomics: exact, order-independent sum of the workgroups' fp32 partials
            // (a partial that is not finite, or whose fixed-point image leaves +-2^52 -- 2048 of those still fit int64 --,
            // raises status[1]: the next prologue then yields NaN like the reference's float sums, instead of finite garbage)
            long long* out = a.fuse.acc_cur + (size_t)(bid % D3P_ACC_R) * D3P_ACC_COLS(P);
            bool bad = false;
            for (int c = threadIdx.x; c < P; c += blockDim.x) {
                float s = 0.f;
                for (int w = 0; w < W; ++w) s += red[(size_t)w * P + c];
                const double sd = (double)s * a.fuse.sg;
                bad |= !(fabs(sd) < 4503599627370496.0);
                if (STAMPS && (a.dbg & 4)) continue;  // ablation (diagnostic instantiation only): no gradient atomics
                atomicAdd(reinterpret_cast<unsigned long long*>(out + c), (unsigned long long)__double2ll_rn(sd));
            }
            if (threadIdx.x < 3) {  // thread 0: loss, fine part; 1: example count; 2: loss, coarse part
                float s = 0.f;
                for (int w = 0; w < W; ++w) s += tail[2 * w + (threadIdx.x & 1)];
                long long hi, lo;
                const bool ok = loss_split(s, hi, lo);
                if (threadIdx.x != 1) bad |= !ok;
                const long long v = threadIdx.x == 0 ? lo : threadIdx.x == 1 ? (long long)s : hi;
                atomicAdd(reinterpret_cast<unsigned long long*>(out + P + threadIdx.x), (unsigned long long)v);
            }
            if (bad) {  // (rare path) poison the count column, see count_n; at most 2^18 of these per step -> still inside int64
                atomicAdd(reinterpret_cast<unsigned long long*>(out + P + 1), 1ull << 44);
                if (a.fuse.status) __hip_atomic_store(a.fuse.status + 1, 1u, __ATOMIC_RELAXED, D3P_AGENT);
            }
            if (CHAIN) {
                // arrive: this workgroup's atomics (and, for workgroup 0, the published state and the zeroed accumulator)
                // are complete at the memory side before the counters move; one counter per group keeps the contention low
                D3P_STAMP(9)   // atomics issued
                __builtin_amdgcn_s_waitcnt(0);
                D3P_STAMP(10)  // ... and acknowledged (wave 0)
                __syncthreads();
                D3P_STAMP(11)
                if (threadIdx.x == 0) {
                    uint32_t* bar = a.chain.bar + (size_t)step_t * D3P_BAR_WORDS;
                    const uint32_t nw = (uint32_t)a.chain.nw, grp = bid % D3P_BAR_GROUPS, gsize = (nw + D3P_BAR_GROUPS - 1u - grp) / D3P_BAR_GROUPS;
                    const uint32_t prev = __hip_atomic_fetch_add(bar + D3P_BAR_LINE * (1 + grp), 1u, __ATOMIC_RELAXED, D3P_AGENT);
                    if (prev + 1u == gsize)  // this group's flag; the waiters poll the flags of all groups
                        __hip_atomic_store(bar + D3P_BAR_LINE * (1 + D3P_BAR_GROUPS + grp), 1u, __ATOMIC_RELAXED, D3P_AGENT);
                    if (STAMPS && a.stamps) {  // arrival returned; copy the stamps of the last two steps out
                        unsigned long long* st = reinterpret_cast<unsigned long long*>(lds) + stamp_off;
                        st[12] = wall_clock64();
                        st[13] = (unsigned long long)prev;
                        const int rec = step_t - (a.chain.K - 2);
                        if (rec >= 0 && bid < 256u)
                            for (int k = 0; k < 16; ++k) a.stamps[((size_t)rec * 256 + bid) * 16 + k] = st[k];
                    }
                }
            }
        } else {
            float* out = a.partials + (size_t)bid * (P + 2);
            for (int c = threadIdx.x; c < P; c += blockDim.x) {
                float s = 0.f;
                for (int w = 0; w < W; ++w) s += red[(size_t)w * P + c];
                out[c] = s;
            }
            // A parameter that is not finite makes the step's loss NaN even when NO example is valid: the reference multiplies every
            // example's (NaN) loss by its mask, NaN * 0 = NaN (svi.py:271-281), while these kernels evaluate no masked example.
            // Workgroup 0's loss partial carries it (the derived columns of the step's parameters are all in LDS: loc, s, sg, q, lc).
            // (the vote goes through the reduction buffer, which has been read out by now: __syncthreads_or would add static LDS to a
            //  kernel whose dynamic LDS may already be the CU's 160 KB)
            int p_bad = 0;
            if (bid == 0) {
                for (int c = threadIdx.x; c < 5 * D; c += blockDim.x) p_bad |= !(fabsf(pk[c]) <= 3.402823466e38f);
                const bool wave_bad = __any(p_bad);
                __syncthreads();
                if (lane == 0) red[wave] = wave_bad ? 1.0f : 0.0f;
                __syncthreads();
                p_bad = 0;
                if (threadIdx.x == 0)
                    for (int w = 0; w < W; ++w) p_bad |= red[w] != 0.0f;
            }
            if (threadIdx.x < 2) {
                float s = 0.f;
                for (int w = 0; w < W; ++w) s += tail[2 * w + threadIdx.x];
                if (threadIdx.x == 0 && p_bad) s = __builtin_nanf("");
                out[P + threadIdx.x] = s;
            }
        }
    }
    if (!CHAIN && a.stamps) {
        __syncthreads();
        if (threadIdx.x == 0) a.stamps[SS * bid + 1] = wall_clock64();
        if ((a.dbg & 32) && !CHAIN && threadIdx.x == 0) a.stamps[8 * bid + 7] = (unsigned long long)(clock64() - clk0);
    }
}

#undef D3P_STAGE_SYNC
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

struct MainGeom {
    int V, NK, W;
    bool full;
    bool tail;  // full tile of features + the intercept column (k_logreg_main's TAIL form)
    bool wide;  // 2048 < d <= 4096, logistic regression: the column-chunked kernel of d3p_logreg_wide.h (two-kernel steps)
    uint32_t blocks;
    size_t lds;
};

// allow_wide: the caller launches the clip-and-accumulate stage (MODE 0) and may be given the column-chunked geometry; the
// materialising stage (MODE 1) always uses the register-tiled kernel
// allow_tail: the caller launches a stage that has the TAIL form (MODE 0 / 2 / 3; the materialising stage has not)
static int main_geometry(const d3p_logreg_model* m, uint32_t B, MainGeom* g, bool allow_wide = true, bool allow_tail = true)
{
    const int D = m->d + (m->intercept ? 1 : 0), half = (D + 1) / 2, P = 2 * D;
    // V = 4: a lane owns 4 adjacent columns of each half and fetches them with 16-byte loads -- needs both halves of a row
    // 16-byte aligned.  (Measured: extending it to rows whose second half is not -- odd halves, the intercept column -- with
    // scalar second-half loads gains little there (d = 512 + intercept 32 -> 29 us/step) and costs the aligned shapes 2 us
    // of extra spills (d = 256: 14.0 -> 16.1), so those shapes stay on the scalar-load form.)
    // (and only pays as a FULL tile -- D = 512 or 1024, logistic regression: every lane / slot predicate folds away.  A partly filled
    // V = 4 tile carries per-element predicates and loses to the scalar-load form at every width measured, batch 4096: d = 16 .. 128:
    // 18 - 19 us per step against 8.5; d = 192, 256: 20 against 10.8; d = 320 .. 448: 21 against 21; d = 640, 768: 36 - 37 against 34 - 35)
    // Beyond D = 1024 the scalar-load form has no tile (8 x 64 columns per half): those rows keep V = 4.
    const bool vec = !m->intercept && (m->d % 8 == 0) && ((m->family == D3P_FAMILY_LOGREG && (m->d == 512 || m->d == 1024)) || half > 512);
    g->V = vec ? 4 : 1;
    const int need = (half + 64 * g->V - 1) / (64 * g->V);
    g->NK = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 0;
    g->full = vec && g->NK > 0 && (D == 2 * 64 * g->V * g->NK) && m->family == D3P_FAMILY_LOGREG;
    // full tile of features + intercept column (d = 512 or 1024 with an intercept: examples/logistic_regression.py)
    static const bool no_tail = getenv("D3P_NO_TAIL_TILE") != nullptr;  // developer switch: the scalar-load form instead
    g->tail = allow_tail && !no_tail && m->intercept && (m->d == 512 || m->d == 1024) && m->family == D3P_FAMILY_LOGREG;
    if (g->tail) {
        g->V = 4;
        g->NK = m->d / 512;
        g->full = true;
    }
    // rows too wide for the register-tiled kernel (NK == 0), or its spilling NK == 8 form: the column-chunked kernel of
    // d3p_logreg_wide.h takes them when the caller runs the clip-and-accumulate stage and its 4 accumulator rows fit the LDS
    const bool wide_ok = allow_wide && (size_t)(4 * P + 8) * sizeof(float) <= 160 * 1024 && getenv("D3P_NO_WIDE_KERNEL") == nullptr;
    if (g->NK == 0 && !wide_ok)
        return fail(D3P_E_UNSUPPORTED, "step kernel: latent dimension %d: the register-tiled kernel holds at most %d columns for rows of this "
                    "alignment%s", D, 2 * 64 * g->V * 8, allow_wide ? " and the column-chunked kernel's accumulator rows do not fit the LDS" : "");
    const bool too_wide = g->NK == 0;
    if (too_wide && wide_ok) g->NK = 8;  // placeholder: only the chunked kernel is launched with this geometry
    // waves per workgroup (default 16 = one 1024-thread workgroup per CU at 4 waves/SIMD), reduced until
    // pack (5D) + reduction buffer (W x P) fit 64 KiB of LDS; one example per wave per pass.
    const int W_max = g->NK == 1 ? 16 : g->NK == 2 ? (g->V == 1 ? 16 : 8) : (g->NK == 4 && g->V == 1) ? 8 : 4;   // D3P_MAIN_MAX_WAVES
    int W = W_max, epw = 1;
    {
        // One workgroup per CU and ceil(B / (W x CUs)) examples per wave: every workgroup pays the update prologue (48 KB
        // of replica / state reads) and P + 2 accumulator atomics per step, so more workgroups than CUs only multiplies
        // that (B = 32768: 2048 workgroups 80.5 us per step, 256 workgroups x 8 examples per wave 41.8 us = 1.61 TB/s of
        // algorithmic traffic, the VALU ceiling of this kernel; B = 8192: 23.0 -> 16.9 us).
        static const int cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
                n = 256;
            (void)hipGetLastError();
            return n;
        }();
        const uint64_t per_pass = (uint64_t)W * (uint64_t)cus;
        epw = (int)(((uint64_t)B + per_pass - 1) / per_pass);
        if (epw < 1) epw = 1;
        if (epw > 64) epw = 64;
    }
    {  // developer overrides of the geometry (tuning sweeps), read once per process
        static const int env_w = [] { const char* e = getenv("D3P_MAIN_W"); return e ? atoi(e) : 0; }();
        static const int env_epw = [] { const char* e = getenv("D3P_MAIN_EPW"); return e ? atoi(e) : 0; }();
        if (env_w >= 1 && env_w <= W_max) W = env_w;
        if (env_epw >= 1 && env_epw <= 64) epw = env_epw;
    }
    auto lds_bytes = [&](int w) { return main_lds_bytes(D, w); };
    // (the scalar-load V = 1, NK = 8 form does not spill and beats the chunked kernel: d = 512 + intercept 32 vs 94 us/step)
    g->wide = wide_ok && (too_wide || (g->V == 4 && g->NK == 8));  // accumulator rows only (derived columns come from the global pack): 4 waves
    if (!g->wide) {
        while (W > 1 && lds_bytes(W) > 96 * 1024) W >>= 1;
        if (lds_bytes(W) > 160 * 1024)
            return fail(D3P_E_UNSUPPORTED, "logreg kernel: P = %d does not fit the LDS reduction buffer", P);
    }
    if (g->wide) W = 4;
    g->W = W;
    uint64_t waves = ((uint64_t)B + epw - 1) / epw;
    uint64_t blocks = (waves + W - 1) / W;
    if (blocks < 1) blocks = 1;
    if (blocks > D3P_MAIN_MAX_BLOCKS) blocks = D3P_MAIN_MAX_BLOCKS;
    g->blocks = (uint32_t)blocks;
    g->lds = g->wide ? (size_t)(4 * P + 8) * sizeof(float) : lds_bytes(W);
    return D3P_OK;
}

template <int MODE>
static int launch_main(hipStream_t s, const MainGeom& g, const MainArgs& a, hipEvent_t e0 = nullptr,
                       hipEvent_t e1 = nullptr)
{
    // hipExtLaunchKernelGGL records e0/e1 tightly around the kernel (used by the timing entry point)
#define D3P_LAUNCH_F(V_, NK_, F_, E_)                                                                                     \
    if (e0)                                                                                                               \
        hipExtLaunchKernelGGL((k_logreg_main<V_, NK_, MODE, F_, E_>), dim3(g.blocks), dim3(64 * g.W), g.lds, s, e0, e1, 0, \
                              a);                                                                                         \
    else                                                                                                                  \
        hipLaunchKernelGGL((k_logreg_main<V_, NK_, MODE, F_, E_>), dim3(g.blocks), dim3(64 * g.W), g.lds, s, a);          \
    return check_launch("k_logreg_main")
#define D3P_LAUNCH(V_, NK_) D3P_LAUNCH_F(V_, NK_, false, -1)
#define D3P_LAUNCH_T(NK_, E_)                                                                                               \
    if (e0)                                                                                                               \
        hipExtLaunchKernelGGL((k_logreg_main<4, NK_, MODE, true, E_, true>), dim3(g.blocks), dim3(64 * g.W), g.lds, s, e0,  \
                              e1, 0, a);                                                                                  \
    else                                                                                                                  \
        hipLaunchKernelGGL((k_logreg_main<4, NK_, MODE, true, E_, true>), dim3(g.blocks), dim3(64 * g.W), g.lds, s, a);     \
    return check_launch("k_logreg_main")
    if constexpr (MODE == 3) {  // D3P_DBG=32: the stamped instantiation of the headline tile
        if ((a.dbg & 32) && a.stamps && g.V == 4 && g.full && !g.tail && g.NK == 1 && !a.eps_ext) {
            hipLaunchKernelGGL((k_logreg_main<4, 1, 3, true, 0, false, true>), dim3(g.blocks), dim3(64 * g.W), g.lds, s, a);
            return check_launch("k_logreg_main");
        }
    }
    if (g.tail) {
        if (MODE == 1) return fail(D3P_E_UNSUPPORTED, "logreg kernel: the materialising stage has no tail-column form");
        if (MODE != 1) {
            if (a.eps_ext) {
                if (g.NK == 1) { D3P_LAUNCH_T(1, 1); } else { D3P_LAUNCH_T(2, 1); }
            } else {
                if (g.NK == 1) { D3P_LAUNCH_T(1, 0); } else { D3P_LAUNCH_T(2, 0); }
            }
        }
    }
    if (g.V == 4 && g.full && MODE != 1) {
        if (a.eps_ext) {
            switch (g.NK) {
            case 1: D3P_LAUNCH_F(4, 1, true, 1);
            case 2: D3P_LAUNCH_F(4, 2, true, 1);
            default: break;
            }
        } else {
            switch (g.NK) {
            case 1: D3P_LAUNCH_F(4, 1, true, 0);
            case 2: D3P_LAUNCH_F(4, 2, true, 0);
            default: break;
            }
        }
    }
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
#undef D3P_LAUNCH_F
#undef D3P_LAUNCH_T
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
    a->family = m->family;
    a->gexp = m->guide_transform;
    a->nh_inv_var = m->family == D3P_FAMILY_GAUSS_MEAN ? -0.5f / (m->lik_sigma * m->lik_sigma) : 0.f;
    a->ll_const = m->family == D3P_FAMILY_GAUSS_MEAN ? (float)D * (logf(m->lik_sigma) + 0.91893853320467267f) : 0.f;
}

}  // namespace d3p
