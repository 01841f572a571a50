// Probe: do the four wavefronts that share a SIMD interleave after a workgroup barrier, or run one after the other?
//
// Round 3's per-wave stamps of the step kernel (D3P_DBG=288) show the 16 waves of a 1024-thread workgroup finishing the
// per-example arithmetic in four groups 0.5 us apart -- waves {0-3}, {4-7}, {8-11}, {12-15}, i.e. the four waves of a SIMD one
// after the other at the single-wave issue rate -- although valu_opcode_probe shows four waves per SIMD sharing the issue port
// at 1.7 - 2.7 cycles per instruction.  This probe isolates the cause: one 1024-thread workgroup per CU; after a barrier every
// wave executes N instructions of a chosen shape and stamps its finishing time.
//   variant 0: straight-line v_fma_f32, 8 independent chains          variant 1: the same in a loop of 64
//   variant 2: straight-line, ONE dependent chain                      variant 3: preceded by 8 ds_read_b128 per wave
//   variant 4: fma interleaved with DPP wave-sum steps + v_readlane    variant 5: variant 0 with s_setprio (3 - wave / 4)
//   variants 6 / 7 (round 3, for "would 8 waves with twice the work each finish sooner than 16?"): the workgroup has 512 threads -- two
//   waves per SIMD -- and every wave executes 512 instructions: 6 = variant 0's shape (8 independent chains), 7 = variant 4's mix
// Output: per variant the finishing times (us after the first wave) of the 16 waves of workgroup 0, and the kernel time.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wave_order_probe wave_order_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define N_INSTR 256

#define FMA8 asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t" \
                          "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"       \
                          : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c))
#define FMA1 asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t" \
                          "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"       \
                          : "+v"(r0) : "v"(b), "v"(c))
#define R4(X) X X X X
#define R32(X) R4(R4(X X))

template <int VARIANT>
__global__ void __launch_bounds__(1024) k_probe(float seed, float* __restrict__ out, unsigned long long* __restrict__ stamps)
{
    __shared__ __attribute__((aligned(16))) float lds[16 * 1024];
    __shared__ unsigned long long st[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float r0 = seed + tid, r1 = r0 * 1.01f, r2 = r0 * 1.02f, r3 = r0 * 1.03f, r4 = r0 * 1.04f, r5 = r0 * 1.05f, r6 = r0 * 1.06f, r7 = r0 * 1.07f;
    float b = 0.999f + 1e-6f * seed, c = 1e-3f;
    asm volatile("" : "+v"(b), "+v"(c));
    for (int i = tid; i < 16 * 1024; i += blockDim.x) lds[i] = seed * i;
    if (VARIANT == 5) {
        if ((wave >> 2) == 0) __builtin_amdgcn_s_setprio(3);
        else if ((wave >> 2) == 1) __builtin_amdgcn_s_setprio(2);
        else if ((wave >> 2) == 2) __builtin_amdgcn_s_setprio(1);
    }
    unsigned long long t0 = 0, t1 = 0;
    long long c0 = 0, c1 = 0;
    for (int rep = 0; rep < 6; ++rep) {   // the stamped pass is the last one: instruction cache warm, as in a chained launch
    __syncthreads();
    t0 = wall_clock64();
    c0 = clock64();
    if (VARIANT == 3) {
        const float4* p = reinterpret_cast<const float4*>(lds) + lane;
        float4 a0 = p[0], a1 = p[64], a2 = p[128], a3 = p[192], a4 = p[256], a5 = p[320], a6 = p[384], a7 = p[448];
        r0 += a0.x + a0.y; r1 += a1.x + a1.z; r2 += a2.x; r3 += a3.y; r4 += a4.x; r5 += a5.w; r6 += a6.x; r7 += a7.x;
    }
    if (VARIANT == 6) {
        R32(FMA8;) R32(FMA8;)
    } else if (VARIANT == 7) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            FMA8; FMA8; FMA8;
            float v = r0 + r1;
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            r2 += __builtin_amdgcn_readlane(__float_as_int(v), 0) * 1e-9f;
            r3 += __expf(v) * 1e-9f;
        }
    } else if (VARIANT == 0 || VARIANT == 3 || VARIANT == 5) {
        R32(FMA8;)
    } else if (VARIANT == 1) {
        for (int i = 0; i < N_INSTR / 64; ++i) { R4(FMA8; FMA8;) }
    } else if (VARIANT == 2) {
        R32(FMA1;)
    } else if (VARIANT == 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            FMA8; FMA8; FMA8;
            float v = r0 + r1;
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            r2 += __builtin_amdgcn_readlane(__float_as_int(v), 0) * 1e-9f;
            r3 += __expf(v) * 1e-9f;
        }
    }
    t1 = wall_clock64();
    c1 = clock64();
    }
    if (lane == 0) st[wave] = t1;
    out[(size_t)blockIdx.x * 1024 + tid] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (lane == 0 && blockDim.x < 1024 && wave + 8 < 16) st[wave + 8] = t1;   // (8-wave variants: the upper half repeats the lower)
    __syncthreads();
    if (blockIdx.x == 0 && tid < 16) stamps[tid] = st[tid];
    if (blockIdx.x == 0 && tid == 0) { stamps[16] = t0; stamps[17] = (unsigned long long)(c1 - c0); stamps[18] = t1 - t0; }
}

template <int V>
static void run(const char* what, int cus, float* out, unsigned long long* stamps, int threads = 1024)
{
    unsigned long long h[19];
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_probe<V>, dim3(cus), dim3(threads), 0, 0, 1.0f, out, stamps);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_probe<V>, dim3(cus), dim3(threads), 0, 0, 2.0f, out, stamps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"variant\": %d, \"what\": \"%s\", \"kernel_us\": %.2f, \"wave_done_us_after_barrier\": [", V, what, 1000.0 * ms);
    for (int w = 0; w < 16; ++w) printf("%s%.2f", w ? ", " : "", (double)(h[w] - h[16]) * 0.01);
    printf("], \"wave0_shader_cycles\": %llu, \"wave0_wall_us\": %.2f, \"wave0_clock_ghz\": %.2f}\n", h[17], (double)h[18] * 0.01, (double)h[17] / ((double)h[18] * 10.0));
}

int main()
{
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    float* out;
    unsigned long long* stamps;
    hipMalloc(&out, (size_t)cus * 1024 * sizeof(float));
    hipMalloc(&stamps, 19 * sizeof(unsigned long long));
    run<0>("256 straight-line v_fma_f32, 8 independent chains", cus, out, stamps);
    run<1>("the same in a loop of 64", cus, out, stamps);
    run<2>("256 straight-line v_fma_f32, ONE dependent chain", cus, out, stamps);
    run<3>("8 ds_read_b128 per wave, then variant 0", cus, out, stamps);
    run<4>("192 fma + 8 x (4 DPP steps, readlane, exp)", cus, out, stamps);
    run<5>("variant 0 with s_setprio 3 - wave / 4", cus, out, stamps);
    run<6>("8 waves (2 per SIMD) x 512 straight-line v_fma_f32, 8 independent chains", cus, out, stamps, 512);
    run<7>("8 waves (2 per SIMD) x [384 fma + 16 x (4 DPP steps, readlane, exp)]", cus, out, stamps, 512);
    return 0;
}
