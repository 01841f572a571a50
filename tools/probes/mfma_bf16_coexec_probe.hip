// Does the bf16 matrix pipe run BESIDE the vector unit and the LDS in the product kernel's shape?  8-wave workgroups (two waves per
// SIMD), each wave one dependent chain of v_mfma_f32_32x32x16_bf16; per MFMA, NV vector instructions of the splitting mix
// (v_perm / v_and / v_sub_f32) on registers of their own and NL ds_write_b64 + ds_read_b128; a workgroup barrier every 12 MFMAs.
// Three launches per case: the MFMAs alone, the other work alone, both interleaved one MFMA : NV : NL (sched_group_barrier, as in
// k_gemm_bf16x3).  Random operand bits.   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_coexec_probe mfma_bf16_coexec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

template <bool MFMA, int NV, int NL>
__global__ void __launch_bounds__(512) k_coexec(int iters, float* out, unsigned seed)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[8][64][40];
    float16v acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        unsigned h = (threadIdx.x * 2654435761u) ^ (i * 40503u) ^ (seed * 2246822519u) ^ (blockIdx.x * 3266489917u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        a[i] = __builtin_bit_cast(__bf16, (unsigned short)((h & 0x807fu) | (((124u + ((h >> 8) & 7u)) & 0xffu) << 7)));
        b[i] = __builtin_bit_cast(__bf16, (unsigned short)(((h >> 16) & 0x807fu) | (((124u + ((h >> 27) & 7u)) & 0xffu) << 7)));
    }
    float x[4];
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = 1.0f + 0.37f * (float)((threadIdx.x + 13 * i) & 63); w[i] = threadIdx.x * 97u + i; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV / 4; ++q) {   // the splitting mix: and, sub, perm, (and the next value feeds on the result)
                const int r = (u + q) & 3;
                const uint32_t ux = __float_as_uint(x[r]);
                const float rem = x[r] - __uint_as_float(ux & 0xffff0000u);
                w[r] = __builtin_amdgcn_perm(__float_as_uint(rem), ux, 0x07060302u) ^ w[r];
                x[r] = rem + 1.5f;
            }
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                *reinterpret_cast<uint2*>(&lds[wave][lane][8 * ((u + q) & 3)]) = make_uint2(w[q & 3], w[(q + 1) & 3]);
                const u32x4v rd = *reinterpret_cast<const u32x4v*>(&lds[wave][(lane + 1) & 63][8 * ((u + q + 1) & 3)]);
                w[(q + 2) & 3] ^= rd[0] ^ rd[3];
            }
            if (MFMA && (NV || NL)) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
                if (NL) __builtin_amdgcn_sched_group_barrier(0x200, NL, 0);
                if (NL) __builtin_amdgcn_sched_group_barrier(0x100, NL, 0);
            }
        }
        __syncthreads();
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) s += acc[v];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += x[i] + (float)w[i];
    if (s == 12345.678f) out[0] = s;
}

template <bool MFMA, int NV, int NL>
static double run(int iters)
{
    float* out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_coexec<MFMA, NV, NL>), dim3(256), dim3(512), 0, 0, iters, out, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((k_coexec<MFMA, NV, NL>), dim3(256), dim3(512), 0, 0, iters, out, 2u + r);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms / 4 * 1e6 / ((double)iters * 12 * 2);   // ns per MFMA slot and SIMD (two waves per SIMD)
}

template <int NV, int NL>
static void report(const char* what)
{
    const int iters = 20000;
    const double m = run<true, 0, 0>(iters), o = run<false, NV, NL>(iters), t = run<true, NV, NL>(iters);
    printf("{\"other_work_per_mfma\": \"%s\", \"ns_per_mfma_slot_and_simd\": {\"mfma_alone\": %.2f, \"other_alone\": %.2f, \"together\": %.2f}, "
           "\"together_over_max\": %.2f, \"together_over_sum\": %.2f}\n", what, m, o, t, t / (m > o ? m : o), t / (m + o));
}

int main()
{
    report<4, 0>("4 VALU");
    report<8, 0>("8 VALU");
    report<12, 0>("12 VALU (the product kernel's staging phase)");
    report<16, 0>("16 VALU");
    report<0, 1>("1 ds_write_b64 + 1 ds_read_b128");
    report<0, 2>("2 ds_write_b64 + 2 ds_read_b128");
    report<12, 2>("12 VALU + 2 ds_write_b64 + 2 ds_read_b128");
    return 0;
}
