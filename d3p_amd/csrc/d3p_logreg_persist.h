// Persistent form of the logistic-regression DP-VI step (MODE 4): the K <= 128 steps of a prepared batch run in ONE launch
// of 256 RESIDENT workgroups (one per CU) that loop over the steps.  Wave w of workgroup b owns batch position 16 b + w in
// every step.  The step of DPSVI.update (svi.py:395-434) has two kinds of work:
//
//   * parameter-INDEPENDENT: the Feistel index, the gathered row, the per-example threefry key and the guide noise eps
//     (svi.py:289-290) -- 60 % of the VALU instructions of a step;
//   * parameter-DEPENDENT, a serial chain across the chip: z = loc + s eps, logit, gradient, norm, clip (svi.py:238-325)
//     -> sum over the batch -> mean, Gaussian mechanism, Adam (svi.py:343-393) -> the next step's parameters.
//
// Measured on MI355X (tools/probes/exchange_probe.hip, barrier_probe.hip): what MODE 2/3 pay for the chain is not its
// arithmetic (0.7 us) but the exchange: 262 k contended int64 atomics per step serialise at the memory side (~7 us when all
// workgroups fire together), every workgroup re-reads the 4 accumulator replicas + optimiser state (12.6 MB per step,
// ~1 us per replica) and a grid barrier with 4 arrivals + 4 pollers per workgroup on counters 64 B apart costs 15 us, one
// with ONE arrival + ONE poller per workgroup, counters 128 B apart and per-group release words 2.2 us.  So here:
//
//   1. every workgroup stores its clipped partial row (P + 2 floats, coalesced agent-scope stores, acknowledged);  barrier 1
//   2. workgroup b OWNS columns 4 b .. 4 b + 3: its waves 0..3 read those columns (+ count, + loss in workgroup 0) of the
//      256 rows, one row per lane, sum them in 64-bit fixed point (exact and order independent => bitwise identical to the
//      atomics of MODE 2/3), wave 0 applies mean / noise / Adam ONCE, with the optimiser state of its columns held in
//      registers for the whole launch, and publishes the 4 new parameters;                                       barrier 2
//   3. every thread reads the parameter of its own column and derives the LDS columns.
//   Also measured and rejected for this exchange: tagged 64-bit words polled by the consumers instead of barriers ("LL"):
//   write-through stores take 3.5 - 5 us to become visible to another XCD, 262 k memory-side swaps per step are slower
//   still, and tight polling by every wave floods the fabric (14.5 - 19 us per step against 12.7 with two barriers).
//
// Meanwhile the waves generate the noise of the NEXT step (index / key prefetched one step ahead, so no memory round trip
// sits in front of the arithmetic) and only THEN issue its row gather: the partial-row stores of step 1 above find the
// memory pipeline empty (issued together with 8 MB of gathers they took 3.5 us to be acknowledged).  The VALU-bound and the
// latency-bound halves of consecutive steps overlap.
// Nothing is pending at the end of a launch (step t's update is applied inside step t), so there is no ping-pong state and
// no flush launch.  Waits are bounded and raise the abort flag instead of hanging; the host launches this kernel only when
// all 256 workgroups are resident at once (d3p_dpvi.hip: use_persistent_steps).
// Specialised for the benchmark geometry: d = 512 without intercept (D = 512, P = 1024), noise on chip, 16 waves, 256
// workgroups (3840 < B <= 4096).
#pragma once
#include "d3p_logreg_kernel.h"

namespace d3p {

#define D3P_PERSIST_D 512
#define D3P_PERSIST_W 16
#define D3P_PERSIST_NW 256   // workgroups = column blocks of 4

static inline size_t persist_lds_bytes()
{
    return (size_t)(5 * D3P_PERSIST_D + D3P_PERSIST_W * 2 * D3P_PERSIST_D + 2 * D3P_PERSIST_W) * sizeof(float) +
           (4 * 6 + 2) * sizeof(long long);  // + the owner waves' partial integer sums and their arrival counter
}

#define D3P_PBAR_LINE 32                     // words per 128-byte line: every counter of the grid barrier has its own line
#define D3P_PBAR_WORDS (17 * D3P_PBAR_LINE)  // top, 8 group arrival counters, 8 group release words

// Grid barrier, epoch-counted (counters only grow within a launch): ONE arrival per workgroup on the counter of its group
// (blockIdx % 8: the workgroups of one XCD), the last arriver of a group bumps the top counter, the last of those
// publishes the epoch to the 8 release words, and ONE lane per workgroup polls the release word of its group (2.2 us per
// barrier; 4 arrivals + 4 pollers per workgroup on counters 64 B apart: 15 us -- tools/probes/barrier_probe.hip).
__device__ __forceinline__ void pbar_arrive(uint32_t* bar, uint32_t bid, uint32_t nw, uint32_t epoch)
{
    const uint32_t grp = bid & 7u, gsize = (nw + 7u - grp) / 8u, ngroups = nw < 8u ? nw : 8u;
    const uint32_t prev = __hip_atomic_fetch_add(bar + D3P_PBAR_LINE * (1 + grp), 1u, __ATOMIC_RELAXED, D3P_AGENT);
    if (prev + 1u == gsize * epoch) {
        const uint32_t top = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, D3P_AGENT);
        if (top + 1u == ngroups * epoch)
            for (uint32_t g = 0; g < 8u; ++g) __hip_atomic_store(bar + D3P_PBAR_LINE * (9 + g), epoch, __ATOMIC_RELAXED, D3P_AGENT);
    }
}

__device__ __forceinline__ bool pbar_wait(const uint32_t* bar, uint32_t bid, uint32_t epoch, uint32_t* abort_flag)
{
    const uint32_t* p = bar + D3P_PBAR_LINE * (9 + (bid & 7u));
    for (uint32_t spins = 0;; ++spins) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, D3P_AGENT) >= epoch) return true;
        if (spins > (1u << 21) || ((spins & 63u) == 63u && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, D3P_AGENT) != 0u)) {
            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, D3P_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// wave64 sum of 64-bit integers with DPP (the same lane pattern as wave_sum): __shfl_xor on a long long compiles to two
// ds_bpermute round trips per step, ~0.4 us per sum when six of them sit on the critical path of every step.
template <int CTRL>
__device__ __forceinline__ long long dpp_mov_i64(long long v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, 0xF, 0xF, false);
    return (long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
}

__device__ __forceinline__ long long readlane_i64(long long v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
    v += dpp_mov_i64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov_i64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov_i64<0x141>(v);  // row_half_mirror
    v += dpp_mov_i64<0x140>(v);  // row_mirror
    return (readlane_i64(v, 0) + readlane_i64(v, 16)) + (readlane_i64(v, 32) + readlane_i64(v, 48));
}

struct PersistPre {  // index / validity / threefry key of this wave's example in a later step (wave-uniform)
    uint32_t row_g, n_valid, k0, k1;
};

struct PersistEx {   // one example, parameter-independent part
    float4 x0, x1;   // columns 4 lane .. 4 lane + 3 and 256 + the same
    float e0[4], e1[4];
    float y;
    bool live;
};

// STAMPS: developer diagnostic (D3P_PERSIST_STAMPS=1) -- waves 0 and 5 of workgroups 0, 85, 170, 255 record the 100 MHz wall
// clock at the phase boundaries of every step: stamps[((wg / 85) * 32 + t) * 16 + k].
template <bool STAMPS>
__global__ void __launch_bounds__(64 * D3P_PERSIST_W) k_logreg_persist(MainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = D3P_PERSIST_D, HALF = D / 2, P = 2 * D, PA = P + 2, W = D3P_PERSIST_W;
    const ChainFuse& cf = a.chain;
    const StepFuse& f = a.fuse;  // per-run constants only (hyper-parameters, fixed-point scales, schedule)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t bid = blockIdx.x, nw = (uint32_t)cf.nw;  // nw == D3P_PERSIST_NW
    float* pk = lds;                  // [loc | s | sg | q | lc]
    float* red = lds + 5 * D;         // W x P
    float* tail = red + (size_t)W * P;
    long long* lpart = reinterpret_cast<long long*>(tail + 2 * W);  // [4 owner waves][6] partial sums
    uint32_t* lflag = reinterpret_cast<uint32_t*>(lpart + 24);       // owner waves 1..3 that delivered (monotonic)
    uint32_t* lgo = lflag + 2;                                        // barrier 1 seen by wave 0 (epoch)
    uint32_t* lcnt = lflag + 1;                                       // waves whose share of the partial row is acknowledged
    const uint32_t p = bid * W + (uint32_t)wave;  // this wave's batch position, the same in every step
    const bool has_item = p < a.B;
    const int c0 = 4 * lane, c1 = c0 + HALF;
    float* rows = a.partials;                  // nw x PA: the workgroups' partial rows of the current step
    uint32_t* bar1 = cf.bar;                   // rows stored
    uint32_t* bar2 = cf.bar + D3P_PBAR_WORDS;  // parameters published
    float* const params = cf.state[0][0];
    float* const adam_m = cf.state[0][1];
    float* const adam_v = cf.state[0][2];
    const bool st_on = STAMPS && a.stamps && (bid % 85u == 0u) && (bid / 85u < 4u) && lane == 0 && (wave == 0 || wave == 5);
    int st_t = 0;
#define D3P_PST(k) if (STAMPS && st_on && st_t < 32) a.stamps[((size_t)(bid / 85u) * 32 + st_t) * 16 + (wave == 0 ? 0 : 8) + (k)] = wall_clock64();

    // ---- parameter-independent part of a step for this wave's example.  Index, validity and threefry key are fetched ONE
    // STEP AHEAD (PersistPre), so the row gather can be issued at once and the noise generation never waits on memory:
    // fetching them where they are used costs three serial round trips (index -> row, key) in front of the arithmetic.
    // Every load is UNCONDITIONAL (clamped addresses, masked afterwards): a conditional load ends in a phi, and the
    // compiler then waits for the freshly issued row gather before the arithmetic that follows the join.
    const uint32_t p_ld = has_item ? p : 0u;
    auto prefetch = [&](int t, PersistPre& N) {
        const int tc = t < cf.K ? t : cf.K - 1;
        N.row_g = cf.idx_base ? cf.idx_base[(size_t)tc * a.B + p_ld] : p_ld;
        N.n_valid = cf.slots[tc].counts[1];
        const uint32_t* sk = cf.skeys_base + (size_t)tc * 2 * a.B + 2 * (size_t)p_ld;
        N.k0 = sk[0];
        N.k1 = sk[1];
    };
    // C: the step's index / validity / key, already in SGPRs (moved there at the top of the step, before any store of the
    // exchange is in flight: the vmcnt counter is in order, so consuming a prefetched value later would wait for the
    // write-through stores of the partial row as well)
    auto phase_a = [&](int t, const PersistPre& C, PersistPre& N, PersistEx& E) {
        const bool mine = (uint64_t)C.row_g >= a.row_lo && (uint64_t)C.row_g < a.row_hi;
        E.live = has_item && p < C.n_valid && mine;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            uint32_t b0, b1;
            threefry2x32(C.k0, C.k1, (uint32_t)(c0 + n), (uint32_t)(c1 + n), b0, b1);
            E.e0[n] = bits_to_normal_wu(b0);
            E.e1[n] = bits_to_normal_wu(b1);
        }
        __builtin_amdgcn_sched_barrier(0);  // the gather stays BEHIND the noise generation (see the header)
        const size_t row = E.live ? (size_t)((uint64_t)C.row_g - a.row_lo) : 0;
        const float* xrow = a.X + row * (size_t)D;
        E.x0 = *reinterpret_cast<const float4*>(xrow + c0);
        E.x1 = *reinterpret_cast<const float4*>(xrow + c1);
        E.y = a.y ? a.y[row] : 0.f;
        prefetch(t + 1, N);  // consumed one step later
    };
    auto to_sgpr = [&](const PersistPre& V) {
        PersistPre S;
        S.row_g = __builtin_amdgcn_readfirstlane(V.row_g);
        S.n_valid = __builtin_amdgcn_readfirstlane(V.n_valid);
        S.k0 = __builtin_amdgcn_readfirstlane(V.k0);
        S.k1 = __builtin_amdgcn_readfirstlane(V.k1);
        return S;
    };
    auto pack_column = [&](int col, float x) {
        if (col < D) {
            pk[col] = x;
        } else {
            const int e = col - D;
            float sp, sgm;
            guide_scale(0, x, sp, sgm);
            pk[D + e] = sp;
            pk[2 * D + e] = sgm;
            pk[3 * D + e] = a.inv_obs * sgm * __builtin_amdgcn_rcpf(sp);
            pk[4 * D + e] = __logf(f.prior_w) - __logf(sp);
        }
    };

    // optimiser state of the owned columns: lanes 0..3 of wave 0 hold column 4 bid + lane for the whole launch
    const int own_col = 4 * (int)bid + (lane & 3);
    float own_x = 0.f, own_m = 0.f, own_v = 0.f;
    if (wave == 0 && lane < 4) {
        own_x = params[own_col];
        own_m = adam_m[own_col];
        own_v = adam_v[own_col];
    }
    if (tid == 0) { *lflag = 0u; *lcnt = 0u; *lgo = 0u; }
    pack_column(tid, params[tid]);  // P == blockDim.x: one column per thread

    PersistEx E;
    PersistPre pre_nxt;
    {
        PersistPre first;
        prefetch(0, first);
        phase_a(0, to_sgpr(first), pre_nxt, E);
    }
    __syncthreads();

    for (int t = 0; t < cf.K; ++t) {
        st_t = t;
        const int g = cf.g0 + t;
        const uint32_t epoch = (uint32_t)t + 1u;
        const PersistPre pre_s = to_sgpr(pre_nxt);  // step t + 1's index / key (prefetched during step t - 1's exchange)
        if (wave == 0) { D3P_PST(2) } else { D3P_PST(0) }

        // ---- parameter-dependent part: z = loc + s eps, logit, gradient, norm, clip (svi.py:238-325)
        float cg0[4], ch0[4], cg1[4], ch1[4];
        float L = 0.f, cnt = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) cg0[n] = ch0[n] = cg1[n] = ch1[n] = 0.f;
        if (E.live) {
            const float x0[4] = {E.x0.x, E.x0.y, E.x0.z, E.x0.w}, x1[4] = {E.x1.x, E.x1.y, E.x1.z, E.x1.w};
            float z0[4], z1[4];
            float tp = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                z0[n] = __fmaf_rn(pk[D + c0 + n], E.e0[n], pk[c0 + n]);
                z1[n] = __fmaf_rn(pk[D + c1 + n], E.e1[n], pk[c1 + n]);
                tp = __fmaf_rn(x0[n], z0[n], tp);
                tp = __fmaf_rn(x1[n], z1[n], tp);
            }
            const float tl = wave_sum(tp);
            const float A = a.A_scale * (sigmoid_f(tl) - E.y);
            const float loglik = E.y * tl - softplus_f(tl);
            float g0[4], h0[4], g1[4], h1[4];
            float n2 = 0.f, lp = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float sg0 = pk[2 * D + c0 + n], sg1 = pk[2 * D + c1 + n];
                const float q0 = pk[3 * D + c0 + n], q1 = pk[3 * D + c1 + n];
                const float lc0 = pk[4 * D + c0 + n], lc1 = pk[4 * D + c1 + n];
                g0[n] = __fmaf_rn(a.c1_w, z0[n], A * x0[n]);
                g1[n] = __fmaf_rn(a.c1_w, z1[n], A * x1[n]);
                h0[n] = __fmaf_rn(g0[n] * E.e0[n], sg0, -q0);
                h1[n] = __fmaf_rn(g1[n] * E.e1[n], sg1, -q1);
                n2 = __fmaf_rn(g0[n], g0[n], n2);
                n2 = __fmaf_rn(h0[n], h0[n], n2);
                n2 = __fmaf_rn(g1[n], g1[n], n2);
                n2 = __fmaf_rn(h1[n], h1[n], n2);
                lp += __fmaf_rn(a.hz_w * z0[n], z0[n], __fmaf_rn(-0.5f * E.e0[n], E.e0[n], lc0));
                lp += __fmaf_rn(a.hz_w * z1[n], z1[n], __fmaf_rn(-0.5f * E.e1[n], E.e1[n], lc1));
            }
            n2 = wave_sum(n2);
            lp = wave_sum(lp);
            L = a.inv_obs * (lp - a.lik_scale * loglik);  // svi.py:278-281
            cnt = 1.0f;
            const float cfac = fminf(1.0f, a.clip * __builtin_amdgcn_rsqf(n2));  // svi.py:121-122
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                cg0[n] = __fmaf_rn(cfac, g0[n], 0.f);
                ch0[n] = __fmaf_rn(cfac, h0[n], 0.f);
                cg1[n] = __fmaf_rn(cfac, g1[n], 0.f);
                ch1[n] = __fmaf_rn(cfac, h1[n], 0.f);
            }
        }
        {
            float* mine = red + (size_t)wave * P;
            *reinterpret_cast<float4*>(mine + c0) = make_float4(cg0[0], cg0[1], cg0[2], cg0[3]);
            *reinterpret_cast<float4*>(mine + D + c0) = make_float4(ch0[0], ch0[1], ch0[2], ch0[3]);
            *reinterpret_cast<float4*>(mine + c1) = make_float4(cg1[0], cg1[1], cg1[2], cg1[3]);
            *reinterpret_cast<float4*>(mine + D + c1) = make_float4(ch1[0], ch1[1], ch1[2], ch1[3]);
            if (lane == 0) { tail[2 * wave] = L; tail[2 * wave + 1] = cnt; }
        }
        if (wave == 0) { D3P_PST(3) } else { D3P_PST(1) }
        __syncthreads();  // R: the workgroup's 16 clipped rows are in LDS
        if (wave == 0) { D3P_PST(4) } else { D3P_PST(2) }

        if (wave < 4) {
            // ---- fixed-order sum of the 16 rows (the workgroup's partial row, exactly the value MODE 2/3 feed to their
            // atomics), stored for the column owners
            const int cb = 4 * tid;  // columns cb .. cb + 3
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int h = 0; h < W; h += 8) {  // 8 LDS reads in flight, summed in wave order
                float4 r[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) r[w] = *reinterpret_cast<const float4*>(red + (size_t)(h + w) * P + cb);
#pragma unroll
                for (int w = 0; w < 8; ++w) { s.x += r[w].x; s.y += r[w].y; s.z += r[w].z; s.w += r[w].w; }
            }
            float st = 0.f;  // tid 0: the workgroup's loss, tid 1: its live-example count
            if (tid < 2)
                for (int w = 0; w < W; ++w) st += tail[2 * w + tid];
            float* myrow = rows + (size_t)bid * PA;
            st_x<true>(reinterpret_cast<unsigned long long*>(myrow + cb),
                       ((unsigned long long)__float_as_uint(s.y) << 32) | __float_as_uint(s.x));
            st_x<true>(reinterpret_cast<unsigned long long*>(myrow + cb + 2),
                       ((unsigned long long)__float_as_uint(s.w) << 32) | __float_as_uint(s.z));
            if (tid < 2) st_x<true>(myrow + P + tid, st);
            __builtin_amdgcn_s_waitcnt(0);  // this wave's share of the row is complete at the memory side
            if (lane == 0) __hip_atomic_fetch_add(lcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wave == 0 && lane == 0) {
                for (uint32_t spins = 0; spins < (1u << 22); ++spins)
                    if (__hip_atomic_load(lcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 4u * epoch) break;
                pbar_arrive(bar1, bid, nw, epoch);
            }
            if (wave == 0) { D3P_PST(5) }
        }

        // ---- next step's parameter-independent part (overlaps the exchange)
        if (t + 1 < cf.K) phase_a(t + 1, pre_s, pre_nxt, E);
        if (wave == 0) { D3P_PST(7) } else { D3P_PST(3) }
        // key chain of the NEXT batch (svi.py:208-211): one step per step, on a wave that has slack
        if (bid == nw - 1 && wave == 4 && t < cf.K_next)
            chain_step<true>(f.chain_sched, cf.chain_slots + t, t, t == cf.K_next - 1);

        if (wave < 4) {
            // ---- column owners: after barrier 1 the 4 waves read columns 4 bid .. 4 bid + 3 (+ count, + loss in workgroup
            // 0) of rows 64 wave + lane and sum them in 64-bit fixed point
            const float* r = rows + (size_t)(64 * wave + lane) * PA;
            const StepMeta* pm = reinterpret_cast<const StepMeta*>(&cf.slots[t].adam_i);
            const float bc1 = pm->bc1, bc2 = pm->bc2;
            const float zn = cf.noise_base[(size_t)t * P + own_col];
            if (wave == 0) { D3P_PST(0) }
            if (wave == 0 && lane == 0) (void)pbar_wait(bar1, bid, epoch, cf.abort_flag);
            if (wave > 0 && lane == 0)  // waves 1..3 follow wave 0's poll through LDS (one poller per workgroup)
                for (uint32_t spins = 0; spins < (1u << 22); ++spins)
                    if (__hip_atomic_load(lgo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= epoch) break;
            if (wave == 0 && lane == 0) __hip_atomic_store(lgo, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            const unsigned long long w01 = ld_x<true>(reinterpret_cast<const unsigned long long*>(r + 4 * bid));
            const unsigned long long w23 = ld_x<true>(reinterpret_cast<const unsigned long long*>(r + 4 * bid + 2));
            const float rc = ld_x<true>(r + P + 1);
            const float rl = bid == 0 ? ld_x<true>(r + P) : 0.f;
            if (wave == 0) { D3P_PST(6) }
            long long s0 = wave_sum_i64(__double2ll_rn((double)__uint_as_float((uint32_t)w01) * f.sg));
            long long s1 = wave_sum_i64(__double2ll_rn((double)__uint_as_float((uint32_t)(w01 >> 32)) * f.sg));
            long long s2 = wave_sum_i64(__double2ll_rn((double)__uint_as_float((uint32_t)w23) * f.sg));
            long long s3 = wave_sum_i64(__double2ll_rn((double)__uint_as_float((uint32_t)(w23 >> 32)) * f.sg));
            long long sn = wave_sum_i64(__double2ll_rn((double)rc));
            // (loss: fixed point at 2^-24 within this one workgroup -- exact and order independent for |row loss| < 2^30)
            long long sl = bid == 0 ? wave_sum_i64(__double2ll_rn((double)rl * D3P_LOSS_LO_SCALE)) : 0ll;
            if (wave > 0) {
                if (lane == 0) {
                    long long* mp = lpart + 6 * wave;
                    mp[0] = s0; mp[1] = s1; mp[2] = s2; mp[3] = s3; mp[4] = sn; mp[5] = sl;
                    __hip_atomic_fetch_add(lflag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                for (uint32_t spins = 0; spins < (1u << 22); ++spins)  // the other three owner waves' partial sums
                    if (__hip_atomic_load(lflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= 3u * ((uint32_t)t + 1u)) break;
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const long long* mp = lpart + 6 * w;
                    s0 += mp[0]; s1 += mp[1]; s2 += mp[2]; s3 += mp[3]; sn += mp[4]; sl += mp[5];
                }
                const long long sll = (lane & 3) == 0 ? s0 : (lane & 3) == 1 ? s1 : (lane & 3) == 2 ? s2 : s3;
                const float n = (float)sn, Bf = (float)a.B;
                const float factor = (n == 0.0f) ? 0.0f : Bf / n;  // svi.py:305
                const float inv_B = 1.0f / Bf, inv_bc1 = 1.0f / bc1, inv_bc2 = 1.0f / bc2;
                const float noise_scale = f.dp_scale * (a.clip / n), out_scale = a.obs_scale * factor;
                if (lane < 4) {
                    const float tot = (float)((double)sll * f.inv_sg);
                    const float gg = __fmaf_rn(zn, noise_scale, tot * inv_B) * out_scale;  // svi.py:343-346, :365-375
                    const float mm = (1.0f - f.b1) * gg + f.b1 * own_m;                     // Adam, svi.py:379-393
                    const float vv = (1.0f - f.b2) * gg * gg + f.b2 * own_v;
                    const float xx = own_x - f.lr * (mm * inv_bc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv * inv_bc2) + f.adam_eps);
                    own_x = xx; own_m = mm; own_v = vv;
                    st_x<true>(params + own_col, xx);
                    adam_m[own_col] = mm;
                    adam_v[own_col] = vv;
                }
                if (bid == 0 && lane == 0) {
                    if (cf.losses) cf.losses[g] = ((float)((double)sl * (1.0 / D3P_LOSS_LO_SCALE)) / Bf) * a.obs_scale * factor;
                    *f.adam_step = pm->adam_i + 1;
                    if (f.batch_index) *f.batch_index = pm->batch_i + 1u;
                }
                D3P_PST(1)
                if (t + 1 < cf.K) {
                    __builtin_amdgcn_s_waitcnt(0);  // the published parameters are complete at the memory side
                    if (lane == 0) {
                        pbar_arrive(bar2, bid, nw, epoch);
                        (void)pbar_wait(bar2, bid, epoch, cf.abort_flag);
                    }
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                }
                D3P_PST(12)
            }
        }
        if (t + 1 >= cf.K) break;
        __syncthreads();  // S1: the parameters of step t + 1 are published (wave 0 arrives after barrier 2)
        pack_column(tid, ld_x<true>(params + tid));  // P == blockDim.x: one column per thread
        __syncthreads();  // S2: the derived columns of step t + 1 are staged
    }
    if (bid == nw - 1 && wave == 4)
        for (int t = cf.K; t < cf.K_next; ++t) chain_step<true>(f.chain_sched, cf.chain_slots + t, t, t == cf.K_next - 1);
#undef D3P_PST
}

}  // namespace d3p
