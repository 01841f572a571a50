// bf16 MFMA issue-rate probe (round 6): what does v_mfma_f32_32x32x16_bf16 sustain (a) as ONE dependent chain per wave -- the shape of
// k_gemm_bf16x3's inner loop, where all six products of a k-step accumulate into the wave's single 32 x 32 accumulator -- and (b) with
// 2 / 4 independent accumulators per wave, at 1, 2 and 4 waves per SIMD?  Register operands only, no memory in the loop.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_probe mfma_bf16_probe.hip && ./mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma_bf16(int iters, float* out, unsigned seed)
{
    float16v acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // (non-trivial bit patterns: the pipe's power depends on the data)
        a[i] = (__bf16)(1.0f + 0.001f * (float)((threadIdx.x * 7 + i * 13 + seed) & 255));
        b[i] = (__bf16)(1.0f - 0.001f * (float)((threadIdx.x * 5 + i * 11 + seed) & 255));
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[j][v];
    if (s == 12345.678f) out[0] = s;
}

// the product kernel's shape: 8 waves per workgroup (two per SIMD, one workgroup per CU), ONE accumulator per wave, PER MFMAs between
// two workgroup barriers (k_gemm_bf16x3: 12 per K slice of 32); SYNC 0: none, 1: __syncthreads(), 2: s_barrier alone.  RANDOM: operand
// bits from a hash (random mantissas and signs, exponents near 1): the clock the chip holds depends on the data (DVFS).
template <int PER, int SYNC, bool RANDOM>
__global__ void __launch_bounds__(512) k_mfma_bf16_wg(int iters, float* out, unsigned seed)
{
    float16v acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (RANDOM) {
            unsigned h = (threadIdx.x * 2654435761u) ^ (i * 40503u) ^ (seed * 2246822519u) ^ (blockIdx.x * 3266489917u);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const unsigned short ba = (unsigned short)((h & 0x807fu) | (((124u + ((h >> 8) & 7u)) & 0xffu) << 7));
            const unsigned short bb = (unsigned short)(((h >> 16) & 0x807fu) | (((124u + ((h >> 27) & 7u)) & 0xffu) << 7));
            a[i] = __builtin_bit_cast(__bf16, ba);
            b[i] = __builtin_bit_cast(__bf16, bb);
        } else {
            a[i] = (__bf16)(1.0f + 0.001f * (float)((threadIdx.x * 7 + i * 13 + seed) & 255));
            b[i] = (__bf16)(1.0f - 0.001f * (float)((threadIdx.x * 5 + i * 11 + seed) & 255));
        }
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < PER; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        if (SYNC == 1) __syncthreads();
        if (SYNC == 2) __builtin_amdgcn_s_barrier();
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) s += acc[v];
    if (s == 12345.678f) out[0] = s;
}

template <int PER, int SYNC, bool RANDOM>
static void run_wg(const char* what, int iters, int reps)
{
    float* out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mfma_bf16_wg<PER, SYNC, RANDOM>), dim3(256), dim3(512), 0, 0, iters, out, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_mfma_bf16_wg<PER, SYNC, RANDOM>), dim3(256), dim3(512), 0, 0, iters, out, 2u + r);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double n_mfma = (double)iters * PER;   // per wave; two waves per SIMD
    printf("{\"case\": \"%s\", \"mfmas_between_barriers\": %d, \"ms_per_launch\": %.3f, \"launches\": %d, \"ns_per_mfma_and_simd\": %.2f, \"pflops\": %.3f}\n", what, PER, ms, reps,
           ms * 1e6 / (n_mfma * 2.0), 256.0 * 8 * n_mfma * 32768.0 / ms / 1e12);
    (void)hipFree(out);
}

template <int NACC>
static void run(const char* what, int blocks, int iters)
{
    float* out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma_bf16<NACC>, dim3(blocks), dim3(256), 0, 0, iters, out, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma_bf16<NACC>, dim3(blocks), dim3(256), 0, 0, iters, out, 2u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)iters * 24;   // per wave
    const double flops = (double)blocks * 4 * n_mfma * 32768.0;
    const double waves_per_simd = blocks / 256.0;
    printf("{\"case\": \"%s\", \"accumulators\": %d, \"workgroups\": %d, \"ms\": %.3f, \"ns_per_mfma_and_simd\": %.2f, \"pflops\": %.3f}\n", what, NACC, blocks, ms,
           ms * 1e6 / (n_mfma * waves_per_simd), flops / ms / 1e12);
    (void)hipFree(out);
}

int main()
{
    const int iters = 20000;
    run<1>("1 wave per SIMD, ONE dependent chain", 256, iters);
    run<1>("2 waves per SIMD, ONE dependent chain each", 512, iters);
    run<1>("4 waves per SIMD, ONE dependent chain each", 1024, iters);
    run<2>("1 wave per SIMD, 2 accumulators", 256, iters);
    run<2>("2 waves per SIMD, 2 accumulators", 512, iters);
    run<4>("1 wave per SIMD, 4 accumulators", 256, iters);
    run<4>("2 waves per SIMD, 4 accumulators", 512, iters);
    // the product kernel's shape (8 waves per workgroup, one accumulator per wave)
    run_wg<12, 0, false>("8-wave workgroups, no barrier, benign data", 40000, 1);
    run_wg<12, 1, false>("8-wave workgroups, __syncthreads every 12 MFMAs, benign data", 40000, 1);
    run_wg<12, 2, false>("8-wave workgroups, s_barrier every 12 MFMAs, benign data", 40000, 1);
    run_wg<6, 1, false>("8-wave workgroups, __syncthreads every 6 MFMAs, benign data", 80000, 1);
    run_wg<12, 0, true>("8-wave workgroups, no barrier, RANDOM data, one 15 ms launch", 40000, 1);
    run_wg<12, 1, true>("8-wave workgroups, __syncthreads every 12 MFMAs, RANDOM data, one launch", 40000, 1);
    run_wg<12, 1, true>("8-wave workgroups, __syncthreads every 12 MFMAs, RANDOM data, 200 launches back to back (~3 s)", 40000, 200);
    // short launches, like the products of a step: 600 MFMAs per SIMD pair
    run_wg<12, 1, true>("8-wave workgroups, __syncthreads every 12 MFMAs, RANDOM data, 25 slices per launch x 2000 launches", 25, 2000);
    return 0;
}
