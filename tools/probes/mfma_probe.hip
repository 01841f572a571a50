// fp32 MFMA issue-rate probe: what one wavefront per SIMD can sustain with v_mfma_f32_32x32x2_f32.
// Each wavefront runs `iters` x 32 MFMAs on two (or four) independent accumulators with register operands only (no memory
// in the loop); reports ns per MFMA and the TFLOP/s of the whole chip for 1, 2 and 4 wavefronts per SIMD, on all CUs or on one.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(int iters, float* out)
{
    float16v acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[j][v];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
static void run(const char* what, int blocks, int iters)
{
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)iters * 32, ns = ms * 1e6 / n_mfma;  // per wavefront (all run concurrently when resident)
    const double flops = (double)blocks * 4 * n_mfma * 4096;
    printf("{\"case\": \"%s\", \"accumulators\": %d, \"workgroups\": %d, \"ms\": %.3f, \"ns_per_mfma_per_wave\": %.2f, \"tflops\": %.1f}\n", what,
           NACC, blocks, ms, ns, flops / ms / 1e9);
    hipFree(out);
}

int main()
{
    const int iters = 20000;
    run<1>("1 wave per SIMD, all CUs", 256, iters);
    run<1>("2 waves per SIMD, all CUs", 512, iters);
    run<2>("1 wave per SIMD, all CUs", 256, iters);
    run<4>("1 wave per SIMD, all CUs", 256, iters);
    run<2>("2 waves per SIMD, all CUs", 512, iters);
    run<2>("4 waves per SIMD, all CUs", 1024, iters);
    run<2>("1 wave per SIMD, one CU", 1, iters);
    run<2>("1 wave per SIMD, 32 CUs", 32, iters);
    return 0;
}
