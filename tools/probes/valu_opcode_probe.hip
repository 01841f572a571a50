// Probe: issue cost of single VALU opcodes on a CDNA4 SIMD, alone, at 1 / 2 / 4 waves per SIMD.
//
// Round 2 measured 3.3 - 3.6 cycles per wave64 VALU instruction for the *mix* of the noise generation (threefry + erf_inv) and
// 3.9 - 4.5 for the mixture-model gradient kernel; MI355X_MICROARCH.md says 2 cycles for v_fma_f32 on the SIMD-32.  This probe
// says which opcode classes run at which rate: every kernel executes ONE opcode in a loop of 64 instructions (8 independent
// register chains x 8, so no instruction waits for its predecessor; a second form with ONE chain gives the dependent latency)
// on every SIMD of every CU, with 256 / 512 / 1024-thread workgroups (1 / 2 / 4 waves per SIMD).
//
// Reported per (opcode, waves per SIMD): shader cycles per instruction per SIMD = clock64 ticks of a wave / (its instructions x
// waves per SIMD), the shader clock during the run (clock64 / wall_clock64), and the wall-time rate.
// rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU on the same binary cross-checks the instruction counts.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o valu_opcode_probe valu_opcode_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LOOPS 512
#define PER_LOOP 64

// one asm statement = 8 independent instructions on registers r0 .. r7 (operands b, c are loop-invariant registers)
#define OP8(INS, A3)                                                                                            \
    asm volatile(INS " %0, %0, %8" A3 "\n\t" INS " %1, %1, %8" A3 "\n\t" INS " %2, %2, %8" A3 "\n\t" INS " %3, %3, %8" A3 "\n\t"      \
                 INS " %4, %4, %8" A3 "\n\t" INS " %5, %5, %8" A3 "\n\t" INS " %6, %6, %8" A3 "\n\t" INS " %7, %7, %8" A3             \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                                      \
                 : "v"(b), "v"(c))
// one chain: 8 dependent instructions
#define OP1(INS, A3)                                                                                            \
    asm volatile(INS " %0, %0, %1" A3 "\n\t" INS " %0, %0, %1" A3 "\n\t" INS " %0, %0, %1" A3 "\n\t" INS " %0, %0, %1" A3 "\n\t"      \
                 INS " %0, %0, %1" A3 "\n\t" INS " %0, %0, %1" A3 "\n\t" INS " %0, %0, %1" A3 "\n\t" INS " %0, %0, %1" A3             \
                 : "+v"(r0)                                                                                                            \
                 : "v"(b), "v"(c))
// unary forms (v_exp_f32 d, s)
#define UN8(INS)                                                                                                \
    asm volatile(INS " %0, %0\n\t" INS " %1, %1\n\t" INS " %2, %2\n\t" INS " %3, %3\n\t" INS " %4, %4\n\t" INS " %5, %5\n\t"            \
                 INS " %6, %6\n\t" INS " %7, %7"                                                                                        \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))
#define UN1(INS)                                                                                                \
    asm volatile(INS " %0, %0\n\t" INS " %0, %0\n\t" INS " %0, %0\n\t" INS " %0, %0\n\t" INS " %0, %0\n\t" INS " %0, %0\n\t"            \
                 INS " %0, %0\n\t" INS " %0, %0"                                                                                        \
                 : "+v"(r0))

#define KERNEL(NAME, BODY)                                                                                      \
    template <int THREADS>                                                                                      \
    __global__ void __launch_bounds__(THREADS) NAME(uint32_t seed, uint32_t* __restrict__ out, unsigned long long* __restrict__ clocks) \
    {                                                                                                           \
        uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3u + 1u, r2 = r0 * 5u + 2u, r3 = r0 * 7u + 3u, r4 = r0 * 11u + 4u, r5 = r0 * 13u + 5u, \
                 r6 = r0 * 17u + 6u, r7 = r0 * 19u + 7u;                                                        \
        uint32_t b = seed * 2654435761u + 12345u, c = 7u;                                                       \
        asm volatile("" : "+v"(b), "+v"(c));                                                                    \
        const long long c0 = clock64();                                                                         \
        const unsigned long long w0 = wall_clock64();                                                           \
        for (int i = 0; i < LOOPS; ++i) {                                                                       \
            BODY BODY BODY BODY BODY BODY BODY BODY                                                             \
        }                                                                                                       \
        const long long c1 = clock64();                                                                         \
        const unsigned long long w1 = wall_clock64();                                                           \
        out[(size_t)blockIdx.x * THREADS + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                                       \
            clocks[2 * (threadIdx.x >> 6)] = (unsigned long long)(c1 - c0);                                     \
            clocks[2 * (threadIdx.x >> 6) + 1] = w1 - w0;                                                       \
        }                                                                                                       \
    }

// 64-bit forms need register pairs
#define KERNEL64(NAME, BODY)                                                                                    \
    template <int THREADS>                                                                                      \
    __global__ void __launch_bounds__(THREADS) NAME(uint32_t seed, uint32_t* __restrict__ out, unsigned long long* __restrict__ clocks) \
    {                                                                                                           \
        double r0 = 1.0 + 1e-9 * (seed + threadIdx.x), r1 = r0 * 1.01, r2 = r0 * 1.02, r3 = r0 * 1.03, r4 = r0 * 1.04, r5 = r0 * 1.05,   \
               r6 = r0 * 1.06, r7 = r0 * 1.07;                                                                  \
        double b = 1.0 - 1e-12 * seed, c = 1e-13;                                                               \
        asm volatile("" : "+v"(b), "+v"(c));                                                                    \
        const long long c0 = clock64();                                                                         \
        const unsigned long long w0 = wall_clock64();                                                           \
        for (int i = 0; i < LOOPS; ++i) {                                                                       \
            BODY BODY BODY BODY BODY BODY BODY BODY                                                             \
        }                                                                                                       \
        const long long c1 = clock64();                                                                         \
        const unsigned long long w1 = wall_clock64();                                                           \
        out[(size_t)blockIdx.x * THREADS + threadIdx.x] = (uint32_t)__double_as_longlong(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7); \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                                       \
            clocks[2 * (threadIdx.x >> 6)] = (unsigned long long)(c1 - c0);                                     \
            clocks[2 * (threadIdx.x >> 6) + 1] = w1 - w0;                                                       \
        }                                                                                                       \
    }

KERNEL(k_add_u32, OP8("v_add_u32", "");)
KERNEL(k_add_u32_dep, OP1("v_add_u32", "");)
KERNEL(k_xor_b32, OP8("v_xor_b32", "");)
KERNEL(k_alignbit_b32, OP8("v_alignbit_b32", ", %9");)
KERNEL(k_alignbit_imm, OP8("v_alignbit_b32", ", 13");)
KERNEL(k_add3_u32, OP8("v_add3_u32", ", %9");)
KERNEL(k_xad_u32, OP8("v_xad_u32", ", %9");)
KERNEL(k_lshl_add_u32, OP8("v_lshl_add_u32", ", %9");)
KERNEL(k_mul_lo_u32, OP8("v_mul_lo_u32", "");)
KERNEL(k_mul_u32_u24, OP8("v_mul_u32_u24", "");)
KERNEL(k_and_or_b32, OP8("v_and_or_b32", ", %9");)
KERNEL(k_add_f32, OP8("v_add_f32", "");)
KERNEL(k_mul_f32, OP8("v_mul_f32", "");)
KERNEL(k_fma_f32, OP8("v_fma_f32", ", %9");)
KERNEL(k_fma_f32_dep, OP1("v_fma_f32", ", %2");)
KERNEL(k_fmac_f32, OP8("v_fmac_f32", "");)
KERNEL(k_max_f32, OP8("v_max_f32", "");)
KERNEL(k_cndmask_b32, OP8("v_cndmask_b32", ", vcc");)
KERNEL(k_exp_f32, UN8("v_exp_f32");)
KERNEL(k_log_f32, UN8("v_log_f32");)
KERNEL(k_rcp_f32, UN8("v_rcp_f32");)
KERNEL(k_rsq_f32, UN8("v_rsq_f32");)
KERNEL(k_sqrt_f32, UN8("v_sqrt_f32");)
KERNEL(k_cvt_f32_u32, UN8("v_cvt_f32_u32");)
#define DPPMOV(R) "v_mov_b32_dpp " R ", " R " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
KERNEL(k_mov_dpp_row_shr, asm volatile(DPPMOV("%0") DPPMOV("%1") DPPMOV("%2") DPPMOV("%3") DPPMOV("%4") DPPMOV("%5") DPPMOV("%6") DPPMOV("%7")
                                       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));)
KERNEL(k_add_f32_dpp, OP8("v_add_f32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf");)
KERNEL(k_add_f32_dpp_dep, OP1("v_add_f32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf");)
KERNEL64(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                                 "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                                 : "v"(b), "v"(c));)
KERNEL64(k_add_f64, asm volatile("v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8\n\t"
                                 "v_add_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %8\n\tv_add_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %8"
                                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                                 : "v"(b), "v"(c));)
KERNEL64(k_pk_fma_f32, asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                                    "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9"
                                    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                                    : "v"(b), "v"(c));)

struct Entry {
    const char* name;
    const char* note;
    void (*k256)(uint32_t, uint32_t*, unsigned long long*);
    void (*k512)(uint32_t, uint32_t*, unsigned long long*);
    void (*k1024)(uint32_t, uint32_t*, unsigned long long*);
};
#define E(N, NOTE) {#N, NOTE, N<256>, N<512>, N<1024>}

int main(int argc, char** argv)
{
    const char* only = argc > 1 ? argv[1] : nullptr;
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    uint32_t* out;
    unsigned long long *clocks, hclocks[32];
    hipMalloc(&out, (size_t)cus * 1024 * sizeof(uint32_t));
    hipMalloc(&clocks, sizeof(hclocks));
    const Entry table[] = {
        E(k_add_u32, "8 independent chains"), E(k_add_u32_dep, "ONE dependent chain"), E(k_xor_b32, ""), E(k_alignbit_b32, "VOP3, shift in a register"),
        E(k_alignbit_imm, "VOP3, inline-constant shift (threefry's rotate)"), E(k_add3_u32, "VOP3"), E(k_xad_u32, "VOP3"), E(k_lshl_add_u32, "VOP3"),
        E(k_mul_lo_u32, "32-bit integer multiply"), E(k_mul_u32_u24, ""), E(k_and_or_b32, "VOP3"), E(k_add_f32, ""), E(k_mul_f32, ""),
        E(k_fma_f32, "VOP3"), E(k_fma_f32_dep, "ONE dependent chain"), E(k_fmac_f32, "VOP2"), E(k_max_f32, ""), E(k_cndmask_b32, ""),
        E(k_exp_f32, "transcendental"), E(k_log_f32, "transcendental"), E(k_rcp_f32, "transcendental"), E(k_rsq_f32, "transcendental"),
        E(k_sqrt_f32, "transcendental"), E(k_cvt_f32_u32, ""), E(k_mov_dpp_row_shr, "DPP row_shr:1"),
        E(k_add_f32_dpp, "DPP row_shr:1"), E(k_add_f32_dpp_dep, "DPP row_shr:1, ONE dependent chain (the wave-sum pattern)"),
        E(k_fma_f64, "64-bit"), E(k_add_f64, "64-bit"), E(k_pk_fma_f32, "packed: two fp32 fma per lane"),
    };
    const double instr_per_wave = (double)LOOPS * PER_LOOP;
    for (const Entry& e : table) {
        if (only && !strstr(e.name, only)) continue;
        for (int wps = 1; wps <= 4; wps *= 2) {
            auto k = wps == 1 ? e.k256 : wps == 2 ? e.k512 : e.k1024;
            const int threads = 256 * wps;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 0, 0, 1u, out, clocks);
            hipDeviceSynchronize();
            const int reps = 10;
            hipEventRecord(e0, 0);
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 0, 0, 2u + i, out, clocks);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(hclocks, clocks, sizeof(hclocks), hipMemcpyDeviceToHost);
            if (hipGetLastError() != hipSuccess) { fprintf(stderr, "%s failed\n", e.name); return 1; }
            double cyc = 0.0, wall = 0.0;  // mean over the waves of workgroup 0
            const int nwaves = threads / 64;
            for (int w = 0; w < nwaves; ++w) { cyc += (double)hclocks[2 * w]; wall += (double)hclocks[2 * w + 1]; }
            cyc /= nwaves; wall /= nwaves;
            const double ghz = cyc / (wall * 10.0);  // wall_clock64 ticks at 100 MHz
            printf("{\"opcode\": \"%s\", \"note\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_simd\": %.3f, "
                   "\"cycles_per_instr_one_wave\": %.3f, \"shader_clock_ghz\": %.3f, \"us_per_launch\": %.2f, "
                   "\"instr_per_wave\": %.0f, \"cus\": %d}\n",
                   e.name + 2, e.note, wps, cyc / (instr_per_wave * wps), cyc / instr_per_wave, ghz, 1000.0 * ms / reps, instr_per_wave, cus);
            fflush(stdout);
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
    }
    hipFree(out);
    hipFree(clocks);
    return 0;
}
