// Probe: latency of hand-rolled grid barriers between 256 co-resident 1024-thread workgroups (one per CU), the
// synchronisation a persistent multi-step DP-VI kernel would need instead of one launch per step.
// mode bits: 1 = also do the step's int64 accumulator atomics + read-back; 2 = flag-array barrier (each workgroup
// stores its epoch, everybody polls all flags) instead of one shared counter; 4 = no release/acquire fences (timing
// only); 8 = two-level counter (8 groups by blockIdx % 8, last arriver of a group bumps the top counter).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define AGENT __HIP_MEMORY_SCOPE_AGENT

__global__ void __launch_bounds__(1024) k_persist(unsigned long long* acc, unsigned int* bar, unsigned int* flags,
                                                   unsigned int* abort_flag, int iters, int mode, float* out)
{
    __shared__ float lds[1024];
    __shared__ int ok;
    const unsigned nwg = gridDim.x;
    float carry = 0.f;
    for (int t = 0; t < iters; ++t) {
        float v = (float)threadIdx.x * 1e-3f + carry;
        lds[threadIdx.x] = v;
        __syncthreads();
        if (mode & 1) {
            unsigned long long* a = acc + (size_t)(t % 3) * 4 * 1024 + (size_t)(blockIdx.x % 4) * 1024;
            atomicAdd(a + threadIdx.x, (unsigned long long)__float2ll_rn(lds[threadIdx.x ^ 1] * 1048576.0f));
            if (blockIdx.x == 0) {
                unsigned long long* z = acc + (size_t)((t + 1) % 3) * 4 * 1024;
                for (int i = threadIdx.x; i < 4 * 1024; i += 1024) z[i] = 0;
            }
            __builtin_amdgcn_s_waitcnt(0);
        }
        __syncthreads();
        const unsigned epoch = (unsigned)(t + 1);
        const int order = (mode & 4) ? __ATOMIC_RELAXED : __ATOMIC_RELEASE;
        if (mode & 2) {
            if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, order, AGENT);
            unsigned spins = 0;
            for (;;) {
                bool mine = true;
                if (threadIdx.x < nwg) mine = __hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, AGENT) >= epoch;
                if (__syncthreads_and(mine)) break;
                if (++spins > (1u << 20)) { if (threadIdx.x == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, AGENT); break; }
            }
            if (!(mode & 4)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            if (threadIdx.x == 0) {
                unsigned target;
                unsigned int* watch;
                if (mode & 8) {
                    const unsigned grp = blockIdx.x & 7u, gsize = (nwg + 7u - grp) / 8u;
                    const unsigned prev = __hip_atomic_fetch_add(bar + 32 * (1 + grp), 1u, order, AGENT);
                    if (prev + 1 == gsize * epoch) __hip_atomic_fetch_add(bar, 1u, order, AGENT);
                    watch = bar;
                    target = 8u * epoch;
                } else {
                    __hip_atomic_fetch_add(bar, 1u, order, AGENT);
                    watch = bar;
                    target = nwg * epoch;
                }
                unsigned spins = 0;
                while (__hip_atomic_load(watch, __ATOMIC_RELAXED, AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, AGENT); break; }
                }
                if (!(mode & 4)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, AGENT)) return;
        if (mode & 1) {
            long long s = 0;
            for (int r = 0; r < 4; ++r)
                s += (long long)__hip_atomic_load(acc + (size_t)(t % 3) * 4 * 1024 + (size_t)r * 1024 + threadIdx.x, __ATOMIC_RELAXED, AGENT);
            carry = (float)s * 1e-12f;
        }
    }
    if (blockIdx.x == 0) out[threadIdx.x] = carry;
    (void)ok;
}

int main()
{
    unsigned long long* acc; unsigned int* bar; float* out;
    (void)hipMalloc(&acc, 3 * 4 * 1024 * 8); (void)hipMalloc(&bar, 8192); (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount;
    const int iters = 501;
    for (int mode : {0, 4, 8, 12, 2, 6, 1, 9, 3, 5}) {
        (void)hipMemset(acc, 0, 3 * 4 * 1024 * 8); (void)hipMemset(bar, 0, 8192);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_persist, dim3(grid), dim3(1024), 0, 0, acc, bar, bar + 1024, bar + 2000, iters, mode, out);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned h[2048]; (void)hipMemcpy(h, bar, 8192, hipMemcpyDeviceToHost);
        printf("mode=%2d [%s%s%s%s]: %.3f us per iteration (abort=%u)\n", mode, (mode & 1) ? "acc " : "", (mode & 2) ? "flags " : "",
               (mode & 4) ? "nofence " : "", (mode & 8) ? "2level " : "", ms * 1000 / iters, h[2000]);
    }
    return 0;
}
