// Does VALU / LDS work of one wavefront run in the shadow of another wavefront's MFMAs on the same SIMD?  
// One workgroup of 8 wavefronts per CU: wavefronts 0-3 (one per SIMD) run a chain of v_mfma_f32_32x32x2_f32, wavefronts 4-7 (their
// SIMD siblings; with and without s_setprio 3) run `mode`: 1 = independent v_fma_f32, 2 = ds_read_b128 + ds_write_b128 on private LDS rows, 3 = global loads.
// Each wavefront reports its own duration (100 MHz wall clock); printed: MFMA alone, the other alone, both together.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip && ./mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512) k_mix(int n_mfma, int n_other, int mode, int prio, const float* src, unsigned long long* out, float* sink)
{
    __shared__ float4 lds[512 * 4];
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    float r = 0.f;
    if (prio == 1 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    if (wave < 4) {
        float16v acc;
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f;
        for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                if (prio == 2) __builtin_amdgcn_s_sleep(1);  // yield the issue port while the MFMA runs (64 cycles)
                if (prio == 3) { asm volatile("s_nop 15"); }
            }
            asm volatile("" : "+v"(a), "+v"(b));
        }
        for (int v = 0; v < 16; ++v) r += acc[v];
    } else if (mode == 1) {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
        for (int i = 0; i < n_other; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __fmaf_rn(x[j], 1.0001f, 0.5f);  // 32 independent-ish v_fma per iteration
        }
        for (int j = 0; j < 8; ++j) r += x[j];
    } else if (mode == 2) {
        float4 v = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
        for (int i = 0; i < n_other; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                lds[threadIdx.x * 4 + u] = v;
                v.x += lds[threadIdx.x * 4 + ((u + 1) & 3)].y;
            }
        }
        r = v.x;
    } else {
        const float4* p = reinterpret_cast<const float4*>(src) + threadIdx.x;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < n_other; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 w = p[(size_t)(i * 4 + u) * 512 + blockIdx.x * 64];
                v.x += w.x; v.y += w.y;
            }
        }
        r = v.x + v.y;
    }
    const unsigned long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    if (r == 12345.678f) sink[0] = r;
}

int main()
{
    unsigned long long* out;
    float *sink, *src;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 4);
    hipMalloc(&src, 512u << 20);
    hipMemset(src, 0, 512u << 20);
    const char* names[4] = {"", "v_fma_f32 x 32 per iteration", "LDS b128 write + read x 4 per iteration", "global_load_dwordx4 x 4 per iteration"};
    for (int prio = 0; prio <= 3; ++prio)  // 0: plain, 1: s_setprio 3 on the other wave, 2: s_sleep 1 after every MFMA, 3: s_nop 15 after every MFMA
    for (int mode = 1; mode <= 3; ++mode) {
        const int n_mfma = 2000, n_other = mode == 1 ? 2000 : mode == 2 ? 4000 : 2000;
        unsigned long long h[3][8];
        for (int c = 0; c < 3; ++c) {  // 0: MFMA alone, 1: other alone, 2: both
            hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, c == 1 ? 0 : n_mfma, c == 0 ? 0 : n_other, mode, prio, src, out, sink);
            hipDeviceSynchronize();
            hipMemcpy(h[c], out, 64, hipMemcpyDeviceToHost);
        }
        printf("{\"variant\": %d, \"other\": \"%s\", \"mfma_alone_us\": %.1f, \"other_alone_us\": %.1f, \"together_mfma_us\": %.1f, \"together_other_us\": %.1f}\n",
               prio, names[mode], h[0][0] * 0.01, h[1][4] * 0.01, h[2][0] * 0.01, h[2][4] * 0.01);
    }
    return 0;
}
