// Probe: K dependent "steps" of 256 x 1024-thread workgroups in ONE launch of K x 256 workgroups, against one launch per
// step.  Workgroups are dispatched in linear-id order, so every workgroup of step t is resident before any of step t + 1
// is placed; a step-(t+1) workgroup does its independent work first (stand-in for the eps generation), then waits until
// all step-t workgroups have arrived (two-level counter, relaxed agent-scope atomics only: no L2-wide flushes), reads the
// step-t sums with agent-scope loads, does the dependent work, adds its own contribution and arrives.
// Every read-back is verified (each thread adds 1 per step -> each column sums to the number of workgroups).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define AGENT __HIP_MEMORY_SCOPE_AGENT
#define R 4

__device__ __forceinline__ float spin_fma(float v, int n)
{
    for (int i = 0; i < n; ++i) v = __fmaf_rn(v, 1.0000001f, 1e-7f);
    return v;
}

// chained != 0: grid = steps * nwg, waits inside.  chained == 0: one launch per step (step0 = the step).
__global__ void __launch_bounds__(1024) k_steps(unsigned long long* acc, unsigned int* cnt, unsigned int* err, unsigned int* abort_flag,
                                                int nwg, int step0, int chained, int pre, int post, float* out)
{
    const int step = chained ? (int)(blockIdx.x / nwg) : step0;
    const int bid = chained ? (int)(blockIdx.x % nwg) : (int)blockIdx.x;
    float v = spin_fma((float)threadIdx.x * 1e-3f, pre);
    if (chained && step > 0) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(cnt + 256 * (step - 1), __ATOMIC_RELAXED, AGENT) < 8u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, AGENT); break; }
            }
        }
        __syncthreads();
    }
    if (bid == 0)  // zero the accumulator of step + 1 (last read by the workgroups of step - 1, which have all arrived)
        for (int i = threadIdx.x; i < R * 1024; i += 1024)
            __hip_atomic_store(acc + (size_t)((step + 1) % 3) * R * 1024 + i, 0ull, __ATOMIC_RELAXED, AGENT);
    if (step > 0) {
        long long s = 0;
        for (int r = 0; r < R; ++r)
            s += (long long)__hip_atomic_load(acc + (size_t)((step + 2) % 3) * R * 1024 + (size_t)r * 1024 + threadIdx.x, __ATOMIC_RELAXED, AGENT);
        if (s != (long long)nwg) atomicAdd(err, 1u);
        v += (float)s * 1e-9f;
    }
    v = spin_fma(v, post);
    atomicAdd(acc + (size_t)(step % 3) * R * 1024 + (size_t)(bid % R) * 1024 + threadIdx.x, 1ull + (v > 1e30f ? 1ull : 0ull));
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (chained && threadIdx.x == 0) {
        const unsigned grp = (unsigned)bid & 7u, gsize = ((unsigned)nwg + 7u - grp) / 8u;
        const unsigned prev = __hip_atomic_fetch_add(cnt + 256 * step + 16 * (1 + grp), 1u, __ATOMIC_RELAXED, AGENT);
        if (prev + 1 == gsize) __hip_atomic_fetch_add(cnt + 256 * step, 1u, __ATOMIC_RELAXED, AGENT);
    }
    if (bid == 0 && threadIdx.x == 0) out[0] = v;
}

int main()
{
    const int K = 400;
    unsigned long long* acc; unsigned int* cnt; float* out;
    (void)hipMalloc(&acc, 3 * R * 1024 * 8); (void)hipMalloc(&cnt, (K + 2) * 256 * 4 + 4096); (void)hipMalloc(&out, 4096);
    unsigned int* err = cnt + (K + 1) * 256; unsigned int* abort_flag = err + 16;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int nwg = prop.multiProcessorCount;
    for (int pre : {0, 100, 200}) for (int post : {0, 100, 200}) {
        float t[2];
        unsigned errs[2], ab = 0;
        for (int chained = 0; chained < 2; ++chained) {
            (void)hipMemset(acc, 0, 3 * R * 1024 * 8); (void)hipMemset(cnt, 0, (K + 2) * 256 * 4 + 4096);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            if (chained) {
                hipLaunchKernelGGL(k_steps, dim3(K * nwg), dim3(1024), 0, 0, acc, cnt, err, abort_flag, nwg, 0, 1, pre, post, out);
            } else {
                for (int s = 0; s < K; ++s)
                    hipLaunchKernelGGL(k_steps, dim3(nwg), dim3(1024), 0, 0, acc, cnt, err, abort_flag, nwg, s, 0, pre, post, out);
            }
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            t[chained] = ms * 1000 / K;
            unsigned h[32]; (void)hipMemcpy(h, err, 128, hipMemcpyDeviceToHost);
            errs[chained] = h[0]; ab |= h[16];
        }
        printf("pre=%3d post=%3d: per-step launches %.2f us/step (errors %u), chained single launch %.2f us/step (errors %u, abort %u)\n",
               pre, post, t[0], errs[0], t[1], errs[1], ab);
    }
    return 0;
}
