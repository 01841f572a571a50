// Probe: latency of grid-barrier variants between 256 resident 1024-thread workgroups on MI355X (8 XCDs, memory-side
// coherence for agent-scope accesses).  us per barrier, 400 barriers per launch.
//   variant 0: two-level counters 64 B apart, 4 arrivals + 4 pollers per workgroup (what MODE 3/4 do)
//   variant 1: two-level counters 128 B apart, 1 arrival + 1 poller per workgroup
//   variant 2: variant 1 + the last arriver publishes 8 per-group release words (pollers of a group share one line)
//   variant 3: flag array: workgroup b stores epoch to flags[b]; one wave per workgroup polls all 256 flags (4 per lane)
//   variant 4: flag array, flags 64 B apart... (one flag per 64 B)
//   variant 5: single counter, 1 arrival + 1 poller per workgroup
//   sleep: s_sleep argument between polls
// build: hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define AGENT __HIP_MEMORY_SCOPE_AGENT

template <int SLEEP>
__device__ __forceinline__ bool spin_ge(const unsigned* p, unsigned target, unsigned* abort_flag)
{
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, AGENT) >= target) return true;
        if (spins > (1u << 20)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, AGENT); return false; }
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
    }
}

template <int SLEEP>
__global__ void __launch_bounds__(1024) k_bar(unsigned* bar, unsigned* abort_flag, int iters, int variant, float* out)
{
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ unsigned go;
    float x = (float)tid;
    for (int t = 0; t < iters; ++t) {
        const unsigned epoch = (unsigned)(t + 1);
        x = __fmaf_rn(x, 1.0001f, 0.5f);
        if (variant == 0) {
            if (wave < 4 && lane == 0) {
                const unsigned grp = bid & 7u, gsize = (nwg + 7u - grp) / 8u;
                const unsigned prev = __hip_atomic_fetch_add(bar + 16 * (1 + grp), 1u, __ATOMIC_RELAXED, AGENT);
                if (prev + 1u == gsize * 4u * epoch) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, AGENT);
                (void)spin_ge<SLEEP>(bar, 8u * epoch, abort_flag);
            }
            __syncthreads();
        } else if (variant == 1 || variant == 2 || variant == 5) {
            __syncthreads();
            if (tid == 0) {
                if (variant == 5) {
                    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, AGENT);
                    (void)spin_ge<SLEEP>(bar, nwg * epoch, abort_flag);
                } else {
                    const unsigned grp = bid & 7u, gsize = (nwg + 7u - grp) / 8u;
                    const unsigned prev = __hip_atomic_fetch_add(bar + 32 * (1 + grp), 1u, __ATOMIC_RELAXED, AGENT);
                    if (prev + 1u == gsize * epoch) {
                        const unsigned top = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, AGENT);
                        if (variant == 2 && top + 1u == 8u * epoch)
                            for (int g = 0; g < 8; ++g) __hip_atomic_store(bar + 32 * (16 + g), epoch, __ATOMIC_RELAXED, AGENT);
                    }
                    if (variant == 2) (void)spin_ge<SLEEP>(bar + 32 * (16 + grp), epoch, abort_flag);
                    else (void)spin_ge<SLEEP>(bar, 8u * epoch, abort_flag);
                }
            }
            __syncthreads();
        } else {
            const int stride = variant == 3 ? 1 : 16;
            __syncthreads();
            if (tid == 0) __hip_atomic_store(bar + (size_t)bid * stride, epoch, __ATOMIC_RELAXED, AGENT);
            if (wave == 0) {
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
                    for (unsigned j = lane; j < nwg; j += 64)
                        ok = ok && __hip_atomic_load(bar + (size_t)j * stride, __ATOMIC_RELAXED, AGENT) >= epoch;
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    if (spins > (1u << 20)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, AGENT); break; }
                    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
                }
            }
            __syncthreads();
        }
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, AGENT)) return;
    }
    if (x == 1.2345f) out[0] = x;
    (void)go;
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount, iters = 400;
    unsigned* bar; float* out;
    (void)hipMalloc(&bar, 1 << 16); (void)hipMalloc(&out, 64);
    unsigned* abort_flag = bar + (1 << 13);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int sleep = 0; sleep < 3; ++sleep)
        for (int v = 0; v < 6; ++v) {
            (void)hipMemset(bar, 0, 1 << 16);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            if (sleep == 0) hipLaunchKernelGGL(k_bar<0>, dim3(grid), dim3(1024), 0, 0, bar, abort_flag, iters, v, out);
            else if (sleep == 1) hipLaunchKernelGGL(k_bar<1>, dim3(grid), dim3(1024), 0, 0, bar, abort_flag, iters, v, out);
            else hipLaunchKernelGGL(k_bar<8>, dim3(grid), dim3(1024), 0, 0, bar, abort_flag, iters, v, out);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned ab = 0; (void)hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost);
            printf("variant %d  sleep %d : %.3f us per barrier (abort=%u)\n", v, sleep == 2 ? 8 : sleep, ms * 1000 / iters, ab);
        }
    return 0;
}
