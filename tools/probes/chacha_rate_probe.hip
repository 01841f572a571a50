// Probe: at what rate does the chip make ChaCha20 blocks in the one-block-per-lane form k_poisson_flags uses -- and what limits it?
// (round 3: the kernel makes 30 G blocks/s = 3.9 cycles per VALU instruction and SIMD at 8 waves per SIMD, where the per-opcode
// table of valu_opcode_probe.hip predicts ~2.0 for its add / xor / v_alignbit_b32 mix)
//   variant 0: one block per thread, 256-thread workgroups, natural occupancy (25 VGPRs: 8 waves per SIMD)
//   variant 1 / 2: the same limited to 4 / 2 waves per SIMD by a dynamic LDS request
//   variant 3: TWO blocks per thread, interleaved by the compiler (8 independent quarter rounds per round)
//   variant 4: one block per thread, 1024-thread workgroups
// Output per variant: blocks per second and cycles per VALU instruction and SIMD (1090 instructions per block and wave, 2.34 GHz).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../d3p_amd/csrc -I../../include -o chacha_rate_probe chacha_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "d3p_device.h"

using namespace d3p;

template <int NB>
__global__ void k_blocks(const uint32_t* __restrict__ key, uint32_t n_chunks, uint32_t thr, uint16_t* __restrict__ flags)
{
    extern __shared__ uint32_t pad[];
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) * NB;
    if (t >= n_chunks) return;
    uint32_t k[16];
    load_key(key + 16 * blockIdx.y, k);
    uint32_t o[NB][16];
#pragma unroll
    for (int j = 0; j < NB; ++j) keystream_block(k, t + j, o[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        uint32_t m = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) m |= (((o[j][w] >> 9) - thr - 1u) >> 31) << w;
        flags[(size_t)blockIdx.y * n_chunks + t + j] = (uint16_t)m;
    }
    if (threadIdx.x == 9999) pad[0] = 1;
}

template <int NB>
static void run(const char* what, int threads, size_t lds, const uint32_t* key, uint16_t* flags)
{
    const uint32_t n_chunks = 625000, steps = 128;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_blocks<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const dim3 grid((n_chunks / NB + threads - 1) / threads, steps);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_blocks<NB>, grid, dim3(threads), lds, 0, key, n_chunks, 3435u, flags);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_blocks<NB>, grid, dim3(threads), lds, 0, key, n_chunks, 3435u, flags);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double blocks = 3.0 * n_chunks * steps, sec = ms * 1e-3;
    const double instr_per_simd = blocks / 64.0 * 1090.0 / 1024.0;
    printf("{\"what\": \"%s\", \"us_per_step\": %.2f, \"gblocks_per_s\": %.1f, \"cycles_per_valu_instr_and_simd\": %.2f}\n", what,
           ms * 1e3 / (3.0 * steps), blocks / sec * 1e-9, sec * 2.34e9 / instr_per_simd);
}

int main()
{
    uint32_t* key;
    uint16_t* flags;
    hipMalloc(&key, 128 * 16 * 4);
    hipMemset(key, 7, 128 * 16 * 4);
    hipMalloc(&flags, (size_t)128 * 625000 * 2 + 64);
    run<1>("1 block per thread, 256 threads, natural occupancy", 256, 0, key, flags);
    run<1>("... limited to 4 waves per SIMD", 256, 40 * 1024, key, flags);
    run<1>("... limited to 2 waves per SIMD", 256, 80 * 1024, key, flags);
    run<2>("2 blocks per thread, 256 threads", 256, 0, key, flags);
    run<2>("2 blocks per thread, 4 waves per SIMD", 256, 40 * 1024, key, flags);
    run<1>("1 block per thread, 1024 threads", 1024, 0, key, flags);
    return 0;
}
