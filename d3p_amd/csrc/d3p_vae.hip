// DP-VI step for the variational auto-encoder of BASELINE config 5 (examples/vae.py:65-153): the one place on the
// path where the model forward is a dense layer stack, so the per-example gradient work is GEMM-shaped and runs on
// the matrix cores (fp32 MFMA, v_mfma_f32_32x32x2_f32; the reference computes in float32 throughout).
//
// The B x P per-example gradient tensor (P = 688 884) is never formed:
//   * a dense layer y = a W + b has per-example gradients  dW_i = a_i d_i^T, db_i = d_i, so
//     ||dW_i||_F^2 + ||db_i||^2 = (||a_i||^2 + 1) ||d_i||^2   -> the joint L2 norm needs only row norms;
//   * the clipped sum  sum_i c_i a_i d_i^T = A^T (diag(c) Delta)  is one GEMM per layer.
// Forward and backward-data passes are batched GEMMs over the B examples; activations, the reparametrised latent,
// the Bernoulli likelihood and the clip factors are small row-wise kernels in between.
// Formulas: oracle/d3p_oracle.c (d3po_vae_step_sums), which materialises every per-example gradient as the check.
#include <type_traits>
#include "d3p_device.h"
#include "d3p_host.h"
#include "d3p_fmesh.h"
#include <mutex>
#include <unordered_map>
#include <map>
#include <utility>
#include "d3p_logreg_kernel.h"  // px_sample_key

namespace d3p {

static inline size_t align_up_v(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------------------------------
// C[M x N] = alpha * sum_k A(m, k) B(k, n) (+ bias[n]) (+ C),  A(m, k) = A[m * a_sm + k * a_sk], B(k, n) = B[k * b_sk + n * b_sn]
// (element strides cover NN / NT / TN), C row-major with leading dimension ldc.
// Workgroup = 4 wavefronts (2 x 2), tile 64 x 64, K in slices of 16 staged through LDS; every wavefront owns a 32 x 32
// accumulator = 16 VGPRs of v_mfma_f32_32x32x2_f32 (lane l: A row / B column l % 32, k = l / 32; D[i][j] with
// j = l % 32, i = 8 (v / 4) + 4 (l / 32) + v % 4).
// ------------------------------------------------------------------------------------------------------------------
typedef float float16v __attribute__((ext_vector_type(16)));

struct GemmJumps {  // see GemmArgs::n_seg
    int n_seg, k_seg;
    long long b_njump, b_kjump, bias_njump, c_njump;
};

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;  // nullable, length N
    int M, N, K;
    long long a_sm, a_sk, b_sk, b_sn;
    int ldc;
    float alpha;
    int accumulate;
    int a_last_one;   // row M - 1 of op(A) is all ones (bias gradients ride along with the weight gradients)
    int k_per;        // K range of one split (multiple of D3P_GK); gridDim.z splits
    float* part;      // split-K partial tiles [gridDim.z][M][N] (nullable when gridDim.z == 1)
    int epi;          // epilogue: 0 store; 1 softplus (C = softplus(o), C2 = sigmoid(o) = its derivative); 2 C = o * C2;
                      // 3 (the dz product, C = [dz | du] of row stride 2 Z): dz = o + sc z, du = dz sd eps - sc with [z | sd] = ex_zu (same
                      // layout as C), eps dense B x Z (the backward pass through the reparametrised latent, svi.py:289-290 draw)
                      // 4 (k_gemm_bf16x3<.., EPI4> only; the decoder's output layer, C = logits B x D): what k_vae_out does in a launch of
                      // its own -- C = sc (sigmoid(o) - x) with x = ep_x (the batch, laid out like C; requested BEFORE the K loop: at
                      // the end the tile's 32 KB would arrive in one burst with nothing left to overlap), and per row and group of 32
                      // columns (one wave's share) the partial sums of x o - softplus(o) and of x^2 in ep_ll / ep_xx
                      // ([ceil(N / 32)][M]), summed in fixed order (DPP) -- the norm kernel adds the groups (vae_out_finish_row)
    const float* ex_zu;
    const float* ex_eps;
    int ex_Z;
    float ex_sc;
    const float* ep_x;
    float* ep_ll;
    float* ep_xx;
    // Two operands side by side that are not adjacent in memory (the pairs (Wl, Ws) / (bl, bs) of the flat parameter layout lie
    // H Z apart): columns n >= n_seg of B, of the bias and of C, and rows k >= k_seg of B, are displaced by a constant.  0 / 0 /
    // INT_MAX segments = plain GEMM.  Honoured by the scalar B fetch and by every store path.
    int n_seg, k_seg;
    long long b_njump, b_kjump, bias_njump, c_njump;
    float* C2;        // second operand / output of the epilogue, same shape and leading dimension as C
    // k_gemm_bf16x3 only, nullable: a device word that is non-zero when EVERY element of A is exactly a bf16 number (low 16 bits of
    // the fp32 pattern zero) -- binarised images (examples/vae.py:157-168), one-hot rows, small integers.  The second and third
    // plane of A are then exactly zero: the kernel stages plane 0 alone (no splitting of A) and issues the three products with a_0
    // (of six); the sums are those of the general path (the dropped products are exact zeros, the others come in the same order).
    const uint32_t* a_exact16;
    // k_gemm_bf16x3 with an n-fast B only, nullable: row k of B is multiplied by b_row_scale[k] while it is staged (the clip factors
    // of the weight-gradient products: B = a delta array, k = the example -- svi.py:121-122 folded into the sum; the fp32 product
    // c_k b is rounded once, as if the scaled rows had been written to memory and read back)
    const float* b_row_scale;
    uint32_t a_exact_nonce;   // A is exact iff *a_exact16 != a_exact_nonce: the checking pass stores the nonce of ITS pass when it finds an
                              // inexact element, so the word needs no reset between passes (whatever it held before, a stale match can
                              // only send an exact batch down the general path)
};

// the flag of GemmArgs::a_exact16 for an array of n floats (n a multiple of 4, 16-byte aligned): 1 unless some element has low bits
__device__ __forceinline__ void exact16_pass(const float* __restrict__ x, size_t n4, uint32_t* __restrict__ flag, uint32_t nonce, unsigned block,
                                             unsigned n_blocks)
{
    uint32_t bad = 0u;
    for (size_t i = (size_t)block * blockDim.x + threadIdx.x; i < n4; i += (size_t)n_blocks * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(x)[i];
        bad |= (v.x | v.y | v.z | v.w) & 0xffffu;
    }
    if (__ballot(bad != 0u) != 0ull && (threadIdx.x & 63) == 0) *flag = nonce;   // (every writer writes the same value)
}

__global__ void __launch_bounds__(256) k_exact16_flag(const float* __restrict__ x, size_t n4, uint32_t* __restrict__ flag, uint32_t nonce)
{
    exact16_pass(x, n4, flag, nonce, blockIdx.x, gridDim.x);
}

// The run loop's batch: rows idx[0 .. B - 1] of the resident table gathered into xb (what d3p_take_rows does) AND the exactness
// pass over them in the same sweep (the batch is read once instead of twice).  d4 = row length in 16-byte words.
__global__ void __launch_bounds__(256) k_vae_gather_check(const float* __restrict__ table, const uint32_t* __restrict__ idx, uint32_t B, uint32_t d4,
                                                          float* __restrict__ xb, uint32_t* __restrict__ flag, uint32_t nonce)
{
    uint32_t bad = 0u;
    const size_t n = (size_t)B * d4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(e / d4), c = (uint32_t)(e % d4);
        const uint4 v = reinterpret_cast<const uint4*>(table)[(size_t)idx[row] * d4 + c];
        reinterpret_cast<uint4*>(xb)[e] = v;
        bad |= (v.x | v.y | v.z | v.w) & 0xffffu;
    }
    if (flag && __ballot(bad != 0u) != 0ull && (threadIdx.x & 63) == 0) *flag = nonce;
}

// o = alpha * acc + bias (+ C); then the epilogue
__device__ __forceinline__ void gemm_store(const GemmArgs& g, int row, int col, float acc, float bv)
{
    const size_t e = (size_t)row * g.ldc + col + (col >= g.n_seg ? g.c_njump : 0ll);
    float o = __fmaf_rn(g.alpha, acc, bv);
    if (g.accumulate) o += g.C[e];
    if (g.epi == 1) {  // one exponential: en = exp(-|o|); sigmoid = 1 / (1 + en) or en / (1 + en); softplus = max(o, 0) + log(1 + en)
        const float en = __expf(-fabsf(o)), r = __builtin_amdgcn_rcpf(1.0f + en);
        g.C2[e] = o >= 0.0f ? r : en * r;
        o = fmaxf(o, 0.0f) + __logf(1.0f + en);
    } else if (g.epi == 2) {
        o *= g.C2[e];
    } else if (g.epi == 3) {
        o = __fmaf_rn(g.ex_sc, g.ex_zu[e], o);
        g.C[e + g.ex_Z] = o * g.ex_zu[e + g.ex_Z] * g.ex_eps[(size_t)row * g.ex_Z + col] - g.ex_sc;
    }
    g.C[e] = o;
}

// XCD-aware tile order.  The chip's 8 XCDs have an L2 each and workgroups are dealt to them round-robin by linear id, so with the
// natural blockIdx -> tile mapping the tiles that SHARE operand data (the N-tiles of one M row block; the tiles of one split-K
// slab) land on 8 different L2s and every XCD streams (nearly) the whole of A and B from memory: 134 MB for the 14 MB h1 product.
// Here workgroup L (XCD L % 8, the j = L / 8-th workgroup of that XCD) takes tile R = start(XCD) + j of the reuse order
// n-fastest, then m, then the K slab: every XCD works through ONE contiguous run of that order, so an A row block is fetched into
// one L2 once and hit by the other N-tiles.  (Only locality depends on the round-robin assumption; the mapping is a bijection.)
__device__ __forceinline__ void xcd_tile(int& bx, int& by, int& bz)
{
    const unsigned gx = gridDim.x, gy = gridDim.y, T = gx * gy * gridDim.z;
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned c = L & 7u, j = L >> 3, base = T >> 3, r = T & 7u;
    const unsigned R = c * base + (c < r ? c : r) + j;
    bx = (int)(R % gx);
    by = (int)((R / gx) % gy);
    bz = (int)(R / (gx * gy));
}

#define D3P_GT 64  // tile edge
#define D3P_GK 16  // K slice

// VA / VB: the operand is fetched with ONE 16-byte load per thread and slice along its unit-stride dimension (the host
// checks alignment and that the other stride and the extents are multiples of 4); otherwise 4 scalar loads with per-element
// guards.  The scalar form issues 8 loads per thread and slice, each behind 64-bit index arithmetic, and was the
// bottleneck of these GEMMs (adding one more dependent load per element cost +13 % on the whole step).
template <bool VA, bool VB>
__global__ void __launch_bounds__(256) k_gemm_f32(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) float As[D3P_GK][D3P_GT + 4];  // [k][m]
    __shared__ __attribute__((aligned(16))) float Bs[D3P_GK][D3P_GT + 4];  // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tx, ty, tz;
    xcd_tile(tx, ty, tz);
    const int m0 = ty * D3P_GT, n0 = tx * D3P_GT;
    const int kbeg = tz * g.k_per, kend = (kbeg + g.k_per < g.K) ? kbeg + g.k_per : g.K;
    float16v acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    // staging maps: the unit-stride dimension of each operand runs along consecutive threads
    const bool a_kfast = g.a_sk == 1, b_nfast = g.b_sn == 1;
    int am[4], ak[4], bk[4], bn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = tid + 256 * r;  // 0 .. 1023
        if (a_kfast) { ak[r] = e & 15; am[r] = e >> 4; } else { am[r] = e & 63; ak[r] = e >> 6; }
        if (b_nfast) { bn[r] = e & 63; bk[r] = e >> 6; } else { bk[r] = e & 15; bn[r] = e >> 4; }
    }
    // vector maps: 4 consecutive elements of the fast dimension per thread
    const int vam = a_kfast ? (tid >> 2) : 4 * (tid & 15), vak = a_kfast ? 4 * (tid & 3) : (tid >> 4);
    const int vbn = b_nfast ? 4 * (tid & 15) : (tid >> 2), vbk = b_nfast ? (tid >> 4) : 4 * (tid & 3);
    const int m_real = g.a_last_one ? g.M - 1 : g.M;  // rows of A that exist in memory
    float ra[4], rb[4];
    auto fetch = [&](int k0) {  // global -> registers for the slice starting at k0
        if (VA) {
            const int gm = m0 + vam, gk = k0 + vak;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_kfast) {  // 4 k of one row
                if (gm < m_real && gk < kend) v = *reinterpret_cast<const float4*>(g.A + (long long)gm * g.a_sm + gk);
                else if (gm < g.M && gk < kend) v = make_float4(1.f, 1.f, 1.f, 1.f);  // the virtual row of ones
            } else {        // 4 rows of one k
                if (gk < kend) {
                    if (gm + 3 < m_real) {
                        v = *reinterpret_cast<const float4*>(g.A + (long long)gk * g.a_sk + gm);
                    } else {
                        float t[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            t[i] = (gm + i < m_real) ? g.A[(long long)gk * g.a_sk + gm + i] : ((gm + i < g.M) ? 1.0f : 0.f);
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
            }
            ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + am[r], gka = k0 + ak[r];
                float va = 0.f;
                if (gm < g.M && gka < kend)
                    va = (g.a_last_one && gm == g.M - 1) ? 1.0f : g.A[(long long)gm * g.a_sm + (long long)gka * g.a_sk];
                ra[r] = va;
            }
        }
        if (VB) {
            const int gn = n0 + vbn, gk = k0 + vbk;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_nfast) {  // 4 n of one k
                if (gk < kend) {
                    if (gn + 3 < g.N) {
                        v = *reinterpret_cast<const float4*>(g.B + (long long)gk * g.b_sk + gn);
                    } else {
                        float t[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) t[i] = (gn + i < g.N) ? g.B[(long long)gk * g.b_sk + gn + i] : 0.f;
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
            } else {        // 4 k of one column
                if (gn < g.N && gk < kend) v = *reinterpret_cast<const float4*>(g.B + (long long)gn * g.b_sn + gk);
            }
            rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gkb = k0 + bk[r], gn = n0 + bn[r];
                rb[r] = (gkb < kend && gn < g.N)
                            ? g.B[(long long)gkb * g.b_sk + (long long)gn * g.b_sn + (gn >= g.n_seg ? g.b_njump : 0ll) + (gkb >= g.k_seg ? g.b_kjump : 0ll)]
                            : 0.f;
            }
        }
    };
    auto stage = [&]() {  // registers -> LDS
        if (VA) {
            if (a_kfast) {
#pragma unroll
                for (int i = 0; i < 4; ++i) As[vak + i][vam] = ra[i];
            } else {
                *reinterpret_cast<float4*>(&As[vak][vam]) = make_float4(ra[0], ra[1], ra[2], ra[3]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) As[ak[r]][am[r]] = ra[r];
        }
        if (VB) {
            if (b_nfast) {
                *reinterpret_cast<float4*>(&Bs[vbk][vbn]) = make_float4(rb[0], rb[1], rb[2], rb[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) Bs[vbk + i][vbn] = rb[i];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) Bs[bk[r]][bn[r]] = rb[r];
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += D3P_GK) {
        stage();
        __syncthreads();
        if (k0 + D3P_GK < kend) fetch(k0 + D3P_GK);  // next slice in flight while this one multiplies
#pragma unroll
        for (int kk = 0; kk < D3P_GK; kk += 2) {
            const float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= g.N) return;
    if (g.part) {  // split-K: raw partial tile, combined in fixed order by k_gemm_reduce
        float* out = g.part + (size_t)tz * g.M * g.N;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + wm * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
            if (row < g.M) out[(size_t)row * g.N + col] = acc[v];
        }
        return;
    }
    const float bv = g.bias ? g.bias[col + (col >= g.n_seg ? g.bias_njump : 0ll)] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int row = m0 + wm * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
        if (row < g.M) gemm_store(g, row, col, acc[v], bv);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// GEMM for the large products (operands 16-byte loadable): tile 128 (M) x 64 (N), K slices of 32, EIGHT wavefronts, each
// owning one 32 x 32 block of the tile (wavefront w: rows 32 (w & 3), columns 32 (w >> 2)).
//
// What bounds an fp32 GEMM on this chip (measured: tools/probes/mfma_probe.hip, mfma_valu_probe.hip, profiles/r02_mfma_*):
//   * v_mfma_f32_32x32x2_f32 sustains 155 TFLOP/s (27.1 ns per instruction and SIMD) from one wavefront per SIMD and one
//     accumulator chain;
//   * while an fp32 MFMA runs NOTHING else issues on its SIMD -- not a later instruction of the same wavefront, not a sibling
//     wavefront's VALU, LDS or memory instruction, whatever its priority.  The efficiency of a GEMM is therefore
//     MFMA cycles / (MFMA cycles + issue cycles of every other instruction on the SIMD + stalls nobody covers).
// Hence: (a) few instructions per MFMA -- fragments are read with ds_read_b128 (4 MFMA steps per read), operands are fetched
// with 16-byte loads whose addresses are a per-thread base + slice offset, edges are applied arithmetically at staging time;
// (b) two wavefronts per SIMD even when the tile count gives one workgroup per CU (224 tiles for 4096 x 400), so that one's
// waits (LDS round trip after the barrier, the barrier itself) are covered by the other's MFMAs; (c) one barrier per slice,
// placed in the MIDDLE of the slice's MFMAs: 8 MFMAs | stage slice i + 1, fetch slice i + 3 | barrier | read the fragments
// of slice i + 1 | 8 MFMAs -- the fragment reads land during the second half.
// (Tried and measured slower: 4 wavefronts of 32 x 64 with one workgroup per CU, 40-57 TFLOP/s; a ping-pong of two wavefront
// groups alternating between MFMA and memory phases, 35-44: the memory phase cannot run beside the MFMA phase.)
//
// LDS layout [row][k], k fastest, row stride 36 floats.  The MFMA step t of a slice multiplies columns {t, 16 + t} (any pairing
// of the 32 k's gives the same sum), so lane (r, h) needs k = 16 h .. 16 h + 15 of its row: four ds_read_b128 per operand.
// An operand that is k-fast in memory is stored as loaded (b128).  A row-fast one (float4 = 4 rows at one k) is transposed in
// registers where the thread holds two k's (A: 4 x b64 stores) and by four b32 stores otherwise (B).
// Loads are straight-line from CLAMPED addresses, the edges are applied arithmetically at staging time (guarded loads -- also
// "valid ? load : fill", which LLVM turns back into a branch -- make the compiler wait with vmcnt(0): no prefetch):
//   * k beyond the split's range: A is multiplied by 0 there (B may hold anything finite);
//   * rows of A beyond m_real: multiplied by 0, plus 1 for the virtual row of ones; rows >= M / columns >= N only feed
//     accumulator entries that are never stored.
// The host guarantees N % 4 == 0 for an n-fast B (a float4 is inside or outside whole) and, for an m-fast A, a row stride of at
// least roundup4(m_real) (the last float4 may hold up to three rows past m_real, masked one by one).
// AK: A is k-fast (a_sk == 1), otherwise m-fast (a_sm == 1); BN: B is n-fast (b_sn == 1), otherwise k-fast (b_sk == 1).
// ------------------------------------------------------------------------------------------------------------------
#define D3P_GTM 128
#define D3P_GKB 32
#define D3P_GLD (D3P_GKB + 4)
template <bool AK, bool BN>
__global__ void __launch_bounds__(512) k_gemm_f32_w8(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) float As[2][D3P_GTM][D3P_GLD];  // [m][k]
    __shared__ __attribute__((aligned(16))) float Bs[2][D3P_GT][D3P_GLD];   // [n][k]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2, wr = wave & 3;
    int tx, ty, tz;
    xcd_tile(tx, ty, tz);
    const int m0 = ty * D3P_GTM, n0 = tx * D3P_GT;
    const int kbeg = tz * g.k_per, kend = (kbeg + g.k_per < g.K) ? kbeg + g.k_per : g.K;
    const int m_real = g.a_last_one ? g.M - 1 : g.M;  // rows of A that exist in memory
    float16v acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;

    // ---- this thread's share of a slice: two float4 of A, one of B (row / k offsets inside the tile and slice)
    int a_m[2], a_k[2], b_n, b_k;
    if (AK) {
#pragma unroll
        for (int r = 0; r < 2; ++r) { const int f = tid + 512 * r; a_m[r] = f >> 3; a_k[r] = 4 * (f & 7); }
    } else {  // rows 4 mg .. 4 mg + 3 at k = 2 kp, 2 kp + 1
        const int mg = (tid & 7) + 8 * (tid >> 7), kp = (tid >> 3) & 15;
        a_m[0] = a_m[1] = 4 * mg;
        a_k[0] = 2 * kp;
        a_k[1] = 2 * kp + 1;
    }
    if (BN) { b_n = 4 * ((tid & 7) + 8 * (tid >> 8)); b_k = (tid >> 3) & 31; } else { b_n = tid >> 3; b_k = 4 * (tid & 7); }
    // Source addresses: per-thread 32-bit element offsets inside (tile rows, slice) -- rows clamped into the matrix -- on top of a
    // wavefront-uniform slice base that the scalar unit advances.  The two uniform conditions below select the cheap paths: a
    // slice that lies inside [0, K) whole is fetched from base + offset (no per-thread arithmetic; the few slices that straddle
    // K or lie beyond it -- the tail and the prefetches past the end -- clamp k per thread), and a tile that touches neither the
    // last rows of A nor the end of the split's K range is staged as loaded.
    // (K need not be a multiple of 4: the last float4 of a k-fast operand then runs past K inside its row -- the host checks the row
    // stride -- and A is masked element by element there; what B holds beyond K multiplies zeros)
    const int K4 = (g.K + 3) & ~3;
    const int ka_last = AK ? K4 - 4 : g.K - 1, kb_last = BN ? g.K - 1 : K4 - 4;
    // (m-fast A: the last float4 that holds real rows starts at roundup4(m_real) - 4; it may run up to 3 rows past m_real inside the
    // row stride -- the host checks a_sk >= roundup4(m_real) -- and those rows are masked one by one)
    const int m_last = AK ? m_real - 1 : ((m_real + 3) & ~3) - 4, n_last = g.N - (BN ? 4 : 1);
    const long long a_kstride = AK ? 1 : g.a_sk, b_kstride = BN ? g.b_sk : 1;
    long long a_row[2];  // element offset of this thread's (clamped) row(s), k = 0
    unsigned a_off[2];   // ... + its k offset inside a slice, relative to the slice base
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int gm = m0 + a_m[r], cm = gm < m_last ? gm : m_last;
        a_row[r] = AK ? (long long)cm * g.a_sm : (long long)cm;
        a_off[r] = (unsigned)(a_row[r] - (AK ? (long long)m0 * g.a_sm : (long long)m0) + (long long)a_k[r] * a_kstride);
    }
    const float* a_tile = g.A + (AK ? (long long)m0 * g.a_sm : (long long)m0);  // uniform; rows clamp downwards only: offsets stay >= 0
    const int gn_c = (n0 + b_n) < n_last ? (n0 + b_n) : n_last;
    const long long b_row = BN ? (long long)gn_c : (long long)gn_c * g.b_sn;
    // (a clamped column can lie left of the tile's first column only when the tile starts within 3 of N: then the base is moved)
    const int n_base = n0 < n_last ? n0 : n_last;
    const float* b_tile = g.B + (BN ? (long long)n_base : (long long)n_base * g.b_sn);
    const unsigned b_off = (unsigned)(b_row - (BN ? (long long)n_base : (long long)n_base * g.b_sn) + (long long)b_k * b_kstride);
    const bool m_edge = m0 + D3P_GTM > m_real;  // uniform: some rows of the tile are virtual (ones row) or absent
    // edge factors (constant over the slices): keep = 1 for a row that exists in memory, one = 1 for the virtual row of ones (row
    // m_real when a_last_one).  k-fast A: a float4 is one row (two rows per thread); m-fast A: four rows gm .. gm + 3, the same for
    // both of the thread's float4s.
    const int gm_a0 = m0 + a_m[0], gm_a1 = m0 + a_m[1];
    auto keep_of = [&](int row) { return row < m_real ? 1.f : 0.f; };
    auto one_of = [&](int row) { return (g.a_last_one && row == m_real) ? 1.f : 0.f; };
    const float kp0 = keep_of(gm_a0), kp1 = AK ? keep_of(gm_a1) : keep_of(gm_a0 + 1), kp2 = keep_of(gm_a0 + 2), kp3 = keep_of(gm_a0 + 3);
    const float on0 = one_of(gm_a0), on1 = AK ? one_of(gm_a1) : one_of(gm_a0 + 1), on2 = one_of(gm_a0 + 2), on3 = one_of(gm_a0 + 3);

    // (separate variables, not arrays: an array of float4 written on two paths was promoted to LDS by the compiler)
    float4 ra00, ra01, rb0, ra10, ra11, rb1;
    // Register set S fetches the slices S, S + 2, S + 4, ... of the split: its slice base pointers and k advance by two slices per
    // use (two scalar adds each -- a 64-bit k0 * stride product per fetch cost a dozen scalar instructions, and on this chip
    // they too wait for the MFMA in flight).
    const long long a_step2 = 2ll * D3P_GKB * a_kstride, b_step2 = 2ll * D3P_GKB * b_kstride;
    int fk[2] = {kbeg, kbeg + D3P_GKB};
    const float* fa[2] = {a_tile + (long long)kbeg * a_kstride, a_tile + (long long)(kbeg + D3P_GKB) * a_kstride};
    const float* fb[2] = {b_tile + (long long)kbeg * b_kstride, b_tile + (long long)(kbeg + D3P_GKB) * b_kstride};
    auto fetch = [&](auto S) {  // the set's next slice -> register set S; exactly three loads on either path
        constexpr int s = decltype(S)::value;
        float4 &a0 = s ? ra10 : ra00, &a1 = s ? ra11 : ra01, &bb = s ? rb1 : rb0;
        const int k0 = fk[s];
        if (k0 + D3P_GKB <= g.K) {
            a0 = *reinterpret_cast<const float4*>(fa[s] + a_off[0]);
            a1 = *reinterpret_cast<const float4*>(fa[s] + a_off[1]);
            bb = *reinterpret_cast<const float4*>(fb[s] + b_off);
        } else {
            const int gk0 = k0 + a_k[0], gk1 = k0 + a_k[1], gkb = k0 + b_k;
            a0 = *reinterpret_cast<const float4*>(g.A + a_row[0] + (long long)(gk0 < ka_last ? gk0 : ka_last) * a_kstride);
            a1 = *reinterpret_cast<const float4*>(g.A + a_row[1] + (long long)(gk1 < ka_last ? gk1 : ka_last) * a_kstride);
            bb = *reinterpret_cast<const float4*>(g.B + b_row + (long long)(gkb < kb_last ? gkb : kb_last) * b_kstride);
        }
        fk[s] = k0 + 2 * D3P_GKB;
        fa[s] += a_step2;
        fb[s] += b_step2;
    };
    auto stage = [&](auto S, int buf, int k0) {  // register set S (slice starting at k0) -> LDS buffer, edges applied
        constexpr int s = decltype(S)::value;
        const float4 bb = s ? rb1 : rb0;
        float4 o[2] = {s ? ra10 : ra00, s ? ra11 : ra01};
        if (m_edge || k0 + D3P_GKB > kend) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float in_k = (k0 + a_k[r] < kend) ? 1.f : 0.f;
                const float4 v = o[r];
                if (AK) {  // one row, four k's per float4: thread rows gm_a0 (r = 0), gm_a1 (r = 1)
                    const float keep = r ? kp1 : kp0, one = r ? on1 : on0;
                    const int kq = k0 + a_k[r];
                    const float i0 = kq + 0 < kend ? 1.f : 0.f, i1 = kq + 1 < kend ? 1.f : 0.f, i2 = kq + 2 < kend ? 1.f : 0.f, i3 = kq + 3 < kend ? 1.f : 0.f;
                    o[r] = make_float4(__fmaf_rn(v.x, keep * i0, one * i0), __fmaf_rn(v.y, keep * i1, one * i1), __fmaf_rn(v.z, keep * i2, one * i2),
                                       __fmaf_rn(v.w, keep * i3, one * i3));
                } else {   // four rows per float4
                    o[r] = make_float4(__fmaf_rn(v.x, kp0 * in_k, on0 * in_k), __fmaf_rn(v.y, kp1 * in_k, on1 * in_k),
                                       __fmaf_rn(v.z, kp2 * in_k, on2 * in_k), __fmaf_rn(v.w, kp3 * in_k, on3 * in_k));
                }
            }
        }
        if (AK) {
            *reinterpret_cast<float4*>(&As[buf][a_m[0]][a_k[0]]) = o[0];
            *reinterpret_cast<float4*>(&As[buf][a_m[1]][a_k[1]]) = o[1];
        } else {  // 2 k x 4 rows, transposed in registers
            *reinterpret_cast<float2*>(&As[buf][a_m[0] + 0][a_k[0]]) = make_float2(o[0].x, o[1].x);
            *reinterpret_cast<float2*>(&As[buf][a_m[0] + 1][a_k[0]]) = make_float2(o[0].y, o[1].y);
            *reinterpret_cast<float2*>(&As[buf][a_m[0] + 2][a_k[0]]) = make_float2(o[0].z, o[1].z);
            *reinterpret_cast<float2*>(&As[buf][a_m[0] + 3][a_k[0]]) = make_float2(o[0].w, o[1].w);
        }
        if (BN) {
            Bs[buf][b_n + 0][b_k] = bb.x;
            Bs[buf][b_n + 1][b_k] = bb.y;
            Bs[buf][b_n + 2][b_k] = bb.z;
            Bs[buf][b_n + 3][b_k] = bb.w;
        } else {
            *reinterpret_cast<float4*>(&Bs[buf][b_n][b_k]) = bb;
        }
    };
    struct Frag { float4 a[4], b[4]; };
    const int fr = lane & 31, fh = lane >> 5;
    auto read_frags = [&](int buf, Frag& f) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f.a[q] = *reinterpret_cast<const float4*>(&As[buf][32 * wr + fr][16 * fh + 4 * q]);
            f.b[q] = *reinterpret_cast<const float4*>(&Bs[buf][32 * grp + fr][16 * fh + 4 * q]);
        }
    };
    auto mma_half = [&](const Frag& f, int half) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 a = f.a[2 * half + q], b = f.b[2 * half + q];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int KB = D3P_GKB;
    // Slice i lives in LDS buffer i & 1; at the top of iteration i the register set (i + 1) & 1 holds slice i + 1 and the set i & 1
    // slice i + 2 (both possibly still in flight) and f0 the fragments of slice i.  A buffer is rewritten one barrier after its last
    // reader issued (and completed) its reads, and read after the barrier that follows its writes.
    // (the slice count is rounded up to even: the register-set indices stay compile-time constants; empty slices multiply zeros)
    Frag f0, f1;
    fetch(S0{});
    fetch(S1{});
    stage(S0{}, 0, kbeg);
    fetch(S0{});
    __syncthreads();
    read_frags(0, f0);
    const int ns = (kend - kbeg + KB - 1) / KB;
    for (int i = 0; i < ns; i += 2) {
        const int k_i = kbeg + i * KB;
        mma_half(f0, 0);
        stage(S1{}, 1, k_i + KB);
        fetch(S1{});
        __syncthreads();
        read_frags(1, f1);
        mma_half(f0, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_half(f1, 0);
        stage(S0{}, 0, k_i + 2 * KB);
        fetch(S0{});
        __syncthreads();
        read_frags(0, f0);
        mma_half(f1, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int col = n0 + 32 * grp + (lane & 31);
    if (col >= g.N) return;
    if (g.part) {  // split-K: raw partial tile, combined in fixed order by the consumer
        float* out = g.part + (size_t)tz * g.M * g.N;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + wr * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
            if (row < g.M) out[(size_t)row * g.N + col] = acc[v];
        }
        return;
    }
    const float bv = g.bias ? g.bias[col + (col >= g.n_seg ? g.bias_njump : 0ll)] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int row = m0 + wr * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
        if (row < g.M) gemm_store(g, row, col, acc[v], bv);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix pipe at fp32 accuracy (round 3): every fp32 operand element is split EXACTLY into three
// bf16 parts (x = x0 + x1 + x2: the top 8, the next 8 and the last 8 bits of its 24-bit significand -- truncations, so each
// remainder is exact), and the product a b is accumulated in fp32 as
//     a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0
// -- six v_mfma_f32_32x32x16_bf16 per 16 k's.  Every partial product is exact (8 x 8 bits), the three dropped terms are below
// 2^-23 |a b| -- the size of ONE fp32 rounding of the product -- and the accumulation is the fp32 accumulation of the fp32 MFMA.
// Why: v_mfma_f32_32x32x2_f32 runs on the VECTOR pipe at 1/16 of the bf16 rate and nothing issues beside it (the 41 % ceiling of
// k_gemm_f32_w8); the bf16 matrix pipe does six of these MFMAs in 6/16 of the time AND lets the vector unit
// (which does the splitting: 5.5 instructions per element) and the LDS run beside it.
// Same tile (128 x 64, eight waves of 32 x 32, K slices of 32), same fetch path, edge handling and epilogue as k_gemm_f32_w8; the
// LDS holds three bf16 planes per operand, [plane][row][k] with k fastest and a row stride of 40 bf16 (80 bytes: the b128
// fragment reads of 16 consecutive rows fall into distinct 16-byte slots of a 256-byte bank row).
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
#define D3P_BLD 40   // row stride of a plane in bf16 elements
// Chunk swizzle of the planes (round 6): a row's K slice is four 16-byte chunks (8 bf16 each); row r keeps chunk c at position
// c ^ parity(r & 0x1c).  The fragment reads stay conflict-free (the ds_read_b128 lane groups still meet 16 distinct slots of the
// 256-byte bank row), and the staging stores of the m-fast A and n-fast B forms -- lanes 4 rows apart writing the same word of
// their rows, banks 16 r mod 32: 4-way, twice the LDS-array cycles (SQ_LDS_BANK_CONFLICT was 55 % of SQ_LDS_IDX_ACTIVE in the
// grouped weight-gradient launch, profiles/r06_vae_gemm_pmc.json) -- become 2-way, which a ds_write_b32 hides
// (MI355X_MICROARCH.md, LDS; the layouts were searched with the guide's bank model).
#ifdef D3P_NO_PLANE_SWZ   // (A/B builds: the unswizzled layout of rounds 3 - 5)
__device__ __forceinline__ int plane_swz(int) { return 0; }
#else
__device__ __forceinline__ int plane_swz(int row) { return (__builtin_popcount((unsigned)row & 0x1cu) & 1) << 3; }   // XOR into the k index
#endif
// DIAGNOSTIC builds only (tools/gemm_diag.sh compiles this file with -DD3P_GEMM_DIAG=<bits> into a library of its own; results are
// WRONG, only the time matters): 1 no splitting arithmetic (all planes = the top halves), 2 no global loads behind the first two
// slices, 4 no staging at all (no splitting, no LDS writes), 8 fragments read once, 16 no MFMAs.  0 = the product kernel.
#ifndef D3P_GEMM_DIAG
#define D3P_GEMM_DIAG 0
#endif
// the three bf16 parts of two fp32 values, as three words {part of x1 (high half) | part of x0 (low half)}: truncations (the top 8,
// the next 8 and the last 8 bits of the 24-bit significand), so each remainder is exact.  11 vector instructions per pair:
// 4 v_and, 4 v_sub (VOP2: 2.4 cycles each), 3 v_perm (4.3).
// -DD3P_SPLIT_RNE (round 6, built and measured, NOT adopted): ROUND-TO-NEAREST parts -- x0 = rn(x), r = x - x0, x1 = rn(r),
// s = r - x1, x2 = s: exact too for every |x| > 2^-110, and smaller parts -- as 3 v_cvt_pk_bf16_f32 (both elements at once, the
// packed word the LDS wants) + 2 v_lshl + 2 v_and + 2 v_pk_add_f32 = 9 instructions per pair.  FEWER instructions (60 instead of 72
// per wave and K slice) and SLOWER: h1 32.8 -> 34.3 us, the VAE update 249 -> 253.5 us on one box, alternating
// (profiles/r06_split_rne_ab.txt) -- the converting and the packed instructions issue at the VOP3 / packed rate or worse
// (profiles/r03_valu_opcodes.json: 4.3 cycles against 2.4), and beside the bf16 MFMAs issue time is what counts
// (docs/experiments_r06.md section 3).
typedef float d3p_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 d3p_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& w0, uint32_t& w1, uint32_t& w2)
{
#ifndef D3P_SPLIT_RNE
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    w0 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    if (D3P_GEMM_DIAG & 1) { w1 = w0; w2 = w0; return; }
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    w1 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    w2 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
#else
    // (the packed subtraction by name: left to itself the compiler takes v_pk_add_f32 for half of the pairs and two v_add_f32 for the rest)
    auto pk_sub = [](d3p_f32x2 a, d3p_f32x2 b) {
        d3p_f32x2 d;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        return d;
    };
    const d3p_f32x2 x = {x0, x1};
    w0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, d3p_bf16x2));
    if (D3P_GEMM_DIAG & 1) { w1 = w0; w2 = w0; return; }
    const d3p_f32x2 h0 = {__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)};
    const d3p_f32x2 r = pk_sub(x, h0);
    w1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, d3p_bf16x2));
    const d3p_f32x2 h1 = {__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
    const d3p_f32x2 sres = pk_sub(r, h1);
    w2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(sres, d3p_bf16x2));
#endif
}

// one 128 x 64 tile (tx, ty) of K slab tz of the product g, by one 8-wave workgroup
template <bool AK, bool BN, bool EPI4 = false>
__device__ __forceinline__ void gemm_bf16x3_tile(const GemmArgs& g, const int tx, const int ty, const int tz)
{
    __shared__ __attribute__((aligned(16))) unsigned short Ap[2][3][D3P_GTM][D3P_BLD];  // [buffer][plane][m][k]
    __shared__ __attribute__((aligned(16))) unsigned short Bp[2][3][D3P_GT][D3P_BLD];   // [buffer][plane][n][k]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2, wr = wave & 3;
    const int m0 = ty * D3P_GTM, n0 = tx * D3P_GT;
    const int kbeg = tz * g.k_per, kend = (kbeg + g.k_per < g.K) ? kbeg + g.k_per : g.K;
    const int m_real = g.a_last_one ? g.M - 1 : g.M;
    float16v acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    float xpre[EPI4 ? 16 : 1];   // EPI4: this lane's 16 elements of ep_x, in flight during the K loop
    if (EPI4) {
        const int pcol = n0 + 32 * grp + (lane & 31);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + wr * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
            const bool ok = pcol < g.N && row < g.M;
            xpre[v] = ok ? g.ep_x[(size_t)row * g.ldc + pcol] : 0.f;
        }
    }

    // ---- this thread's share of a slice (as in k_gemm_f32_w8): two float4 of A, one of B
    int a_m[2], a_k[2], b_n, b_k;
    if (AK) {
#pragma unroll
        for (int r = 0; r < 2; ++r) { const int f = tid + 512 * r; a_m[r] = f >> 3; a_k[r] = 4 * (f & 7); }
    } else {  // rows 4 mg .. 4 mg + 3 at k = 2 kp, 2 kp + 1
        const int mg = (tid & 7) + 8 * (tid >> 7), kp = (tid >> 3) & 15;
        a_m[0] = a_m[1] = 4 * mg;
        a_k[0] = 2 * kp;
        a_k[1] = 2 * kp + 1;
    }
    // n-fast B: columns 4 ng .. 4 ng + 3 at ONE k; lanes l and l ^ 8 hold k and k ^ 1 of the same columns (bit 3 of tid = bit 0 of k)
    if (BN) { b_n = 4 * ((tid & 7) + 8 * (tid >> 8)); b_k = (tid >> 3) & 31; } else { b_n = tid >> 3; b_k = 4 * (tid & 7); }
    // where this thread's share goes in the swizzled planes (plane_swz: rows 4 mg .. 4 mg + 3 share one swizzle, b_n likewise)
    const int a_kw[2] = {a_k[0] ^ plane_swz(a_m[0]), a_k[1] ^ plane_swz(a_m[1])};
    const int b_sw = plane_swz(b_n);
    const int K4 = (g.K + 3) & ~3;
    const int ka_last = AK ? K4 - 4 : g.K - 1, kb_last = BN ? g.K - 1 : K4 - 4;
    const int m_last = AK ? m_real - 1 : ((m_real + 3) & ~3) - 4, n_last = g.N - (BN ? 4 : 1);
    const long long a_kstride = AK ? 1 : g.a_sk, b_kstride = BN ? g.b_sk : 1;
    long long a_row[2];
    unsigned a_off[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int gm = m0 + a_m[r], cm = gm < m_last ? gm : m_last;
        a_row[r] = AK ? (long long)cm * g.a_sm : (long long)cm;
        a_off[r] = (unsigned)(a_row[r] - (AK ? (long long)m0 * g.a_sm : (long long)m0) + (long long)a_k[r] * a_kstride);
    }
    const float* a_tile = g.A + (AK ? (long long)m0 * g.a_sm : (long long)m0);
    const int gn_c = (n0 + b_n) < n_last ? (n0 + b_n) : n_last;
    const long long b_row = BN ? (long long)gn_c : (long long)gn_c * g.b_sn;
    const int n_base = n0 < n_last ? n0 : n_last;
    const float* b_tile = g.B + (BN ? (long long)n_base : (long long)n_base * g.b_sn);
    const unsigned b_off = (unsigned)(b_row - (BN ? (long long)n_base : (long long)n_base * g.b_sn) + (long long)b_k * b_kstride);
    const bool m_edge = m0 + D3P_GTM > m_real;
    const int gm_a0 = m0 + a_m[0], gm_a1 = m0 + a_m[1];
    auto keep_of = [&](int row) { return row < m_real ? 1.f : 0.f; };
    auto one_of = [&](int row) { return (g.a_last_one && row == m_real) ? 1.f : 0.f; };
    const float kp0 = keep_of(gm_a0), kp1 = AK ? keep_of(gm_a1) : keep_of(gm_a0 + 1), kp2 = keep_of(gm_a0 + 2), kp3 = keep_of(gm_a0 + 3);
    const float on0 = one_of(gm_a0), on1 = AK ? one_of(gm_a1) : one_of(gm_a0 + 1), on2 = one_of(gm_a0 + 2), on3 = one_of(gm_a0 + 3);

    float4 ra00, ra01, rb0, ra10, ra11, rb1;
    float rs0 = 1.f, rs1 = 1.f;   // b_row_scale of this thread's row of B (BN), per register set
    const bool b_scaled = BN && g.b_row_scale != nullptr;
    const long long a_step2 = 2ll * D3P_GKB * a_kstride, b_step2 = 2ll * D3P_GKB * b_kstride;
    int fk[2] = {kbeg, kbeg + D3P_GKB};
    const float* fa[2] = {a_tile + (long long)kbeg * a_kstride, a_tile + (long long)(kbeg + D3P_GKB) * a_kstride};
    const float* fb[2] = {b_tile + (long long)kbeg * b_kstride, b_tile + (long long)(kbeg + D3P_GKB) * b_kstride};
    auto fetch = [&](auto S) {
        constexpr int s = decltype(S)::value;
        float4 &a0 = s ? ra10 : ra00, &a1 = s ? ra11 : ra01, &bb = s ? rb1 : rb0;
        const int k0 = fk[s];
        if ((D3P_GEMM_DIAG & 2) && k0 >= kbeg + 2 * D3P_GKB) { fk[s] = k0 + 2 * D3P_GKB; return; }
        if (b_scaled) {
            const int gk = k0 + b_k;
            (s ? rs1 : rs0) = g.b_row_scale[gk < g.K ? gk : g.K - 1];
        }
        if (k0 + D3P_GKB <= g.K) {
            a0 = *reinterpret_cast<const float4*>(fa[s] + a_off[0]);
            a1 = *reinterpret_cast<const float4*>(fa[s] + a_off[1]);
            bb = *reinterpret_cast<const float4*>(fb[s] + b_off);
        } else {
            const int gk0 = k0 + a_k[0], gk1 = k0 + a_k[1], gkb = k0 + b_k;
            a0 = *reinterpret_cast<const float4*>(g.A + a_row[0] + (long long)(gk0 < ka_last ? gk0 : ka_last) * a_kstride);
            a1 = *reinterpret_cast<const float4*>(g.A + a_row[1] + (long long)(gk1 < ka_last ? gk1 : ka_last) * a_kstride);
            bb = *reinterpret_cast<const float4*>(g.B + b_row + (long long)(gkb < kb_last ? gkb : kb_last) * b_kstride);
        }
        fk[s] = k0 + 2 * D3P_GKB;
        fa[s] += a_step2;
        fb[s] += b_step2;
    };
    // EDGE = false: the slice lies inside [kbeg, kend) and the tile inside the rows of A that exist -- no edge arithmetic, no branch
    // (the steady state: its body is straight-line code that the scheduler interleaves with the MFMAs of the slice before)
    auto stage = [&](auto S, int buf, int k0, auto EDGE, auto A1P) {  // register set S (slice starting at k0): edges applied, split, three planes -> LDS
        constexpr int s = decltype(S)::value;
        constexpr bool A1 = decltype(A1P)::value;   // A is exactly bf16: plane 0 alone
        if ((D3P_GEMM_DIAG & 4) && k0 >= kbeg + 2 * D3P_GKB) return;
        float4 bb = s ? rb1 : rb0;
        if (BN) { const float c = s ? rs1 : rs0; bb = make_float4(bb.x * c, bb.y * c, bb.z * c, bb.w * c); }
        float4 o[2] = {s ? ra10 : ra00, s ? ra11 : ra01};
        if (decltype(EDGE)::value && (m_edge || k0 + D3P_GKB > kend)) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float in_k = (k0 + a_k[r] < kend) ? 1.f : 0.f;
                const float4 v = o[r];
                if (AK) {
                    const float keep = r ? kp1 : kp0, one = r ? on1 : on0;
                    const int kq = k0 + a_k[r];
                    const float i0 = kq + 0 < kend ? 1.f : 0.f, i1 = kq + 1 < kend ? 1.f : 0.f, i2 = kq + 2 < kend ? 1.f : 0.f, i3 = kq + 3 < kend ? 1.f : 0.f;
                    o[r] = make_float4(__fmaf_rn(v.x, keep * i0, one * i0), __fmaf_rn(v.y, keep * i1, one * i1), __fmaf_rn(v.z, keep * i2, one * i2),
                                       __fmaf_rn(v.w, keep * i3, one * i3));
                } else {
                    o[r] = make_float4(__fmaf_rn(v.x, kp0 * in_k, on0 * in_k), __fmaf_rn(v.y, kp1 * in_k, on1 * in_k),
                                       __fmaf_rn(v.z, kp2 * in_k, on2 * in_k), __fmaf_rn(v.w, kp3 * in_k, on3 * in_k));
                }
            }
        }
        uint32_t w[3];
        auto top16 = [](float x0, float x1) { return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u); };
        if (AK) {  // one row, four consecutive k's per float4: two words per plane, one 8-byte store
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (A1) {
                    *reinterpret_cast<uint2*>(&Ap[buf][0][a_m[r]][a_kw[r]]) = make_uint2(top16(o[r].x, o[r].y), top16(o[r].z, o[r].w));
                } else {
                    uint32_t x[3], y[3];
                    split_pair(o[r].x, o[r].y, x[0], x[1], x[2]);
                    split_pair(o[r].z, o[r].w, y[0], y[1], y[2]);
#pragma unroll
                    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(&Ap[buf][p][a_m[r]][a_kw[r]]) = make_uint2(x[p], y[p]);
                }
            }
        } else {   // four rows at k = 2 kp (o[0]) and 2 kp + 1 (o[1]): one word per row and plane
            const float lo4[4] = {o[0].x, o[0].y, o[0].z, o[0].w}, hi4[4] = {o[1].x, o[1].y, o[1].z, o[1].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (A1) {
                    *reinterpret_cast<uint32_t*>(&Ap[buf][0][a_m[0] + i][a_kw[0]]) = top16(lo4[i], hi4[i]);
                } else {
                    split_pair(lo4[i], hi4[i], w[0], w[1], w[2]);
#pragma unroll
                    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint32_t*>(&Ap[buf][p][a_m[0] + i][a_kw[0]]) = w[p];
                }
            }
        }
        if (BN) {  // four columns at one k; the lane 8 away holds k ^ 1: the even-k lane takes columns 0, 1, the odd-k lane 2, 3
            const bool odd = (b_k & 1) != 0;
            const float give0 = odd ? bb.x : bb.z, give1 = odd ? bb.y : bb.w;   // what the partner needs from this lane
            const float got0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give0), 0x128, 0xF, 0xF, false));  // row_ror:8
            const float got1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give1), 0x128, 0xF, 0xF, false));
            // this lane's two columns at (k even, k odd)
            const float e0 = odd ? got0 : bb.x, o0 = odd ? bb.z : got0, e1 = odd ? got1 : bb.y, o1 = odd ? bb.w : got1;
            const int nn = b_n + (odd ? 2 : 0), kk = (b_k & ~1) ^ b_sw;   // (rows nn, nn + 1 lie in b_n's group of four: one swizzle)
            split_pair(e0, o0, w[0], w[1], w[2]);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint32_t*>(&Bp[buf][p][nn][kk]) = w[p];
            split_pair(e1, o1, w[0], w[1], w[2]);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint32_t*>(&Bp[buf][p][nn + 1][kk]) = w[p];
        } else {
            uint32_t x[3], y[3];
            split_pair(bb.x, bb.y, x[0], x[1], x[2]);
            split_pair(bb.z, bb.w, y[0], y[1], y[2]);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(&Bp[buf][p][b_n][b_k ^ b_sw]) = make_uint2(x[p], y[p]);
        }
    };
    struct Frag { bf16x8 a[2][3], b[2][3]; };   // [k step of 16][plane]
    const int fr = lane & 31, fh = lane >> 5;
    const int f_sw = plane_swz(fr);   // (the wave's rows 32 wr + fr / 32 grp + fr: bits 2 - 4 are fr's)
    bool frags_read = false;
    auto read_frags = [&](int buf, Frag& f, auto A1P) {
        constexpr bool A1 = decltype(A1P)::value;
        if (D3P_GEMM_DIAG & 8) { if (frags_read) return; frags_read = true; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (!A1 || p == 0)
                    f.a[ks][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&Ap[buf][p][32 * wr + fr][(16 * ks + 8 * fh) ^ f_sw]));
                f.b[ks][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&Bp[buf][p][32 * grp + fr][(16 * ks + 8 * fh) ^ f_sw]));
            }
    };
    auto mma_half = [&](const Frag& f, int ks, auto A1P) {   // smallest terms first
        constexpr bool A1 = decltype(A1P)::value;
        if (D3P_GEMM_DIAG & 16) return;
        if (!A1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][2], f.b[ks][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][0], f.b[ks][2], acc, 0, 0, 0);
        if (!A1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][1], f.b[ks][1], acc, 0, 0, 0);
        if (!A1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][1], f.b[ks][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][0], f.b[ks][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][0], f.b[ks][0], acc, 0, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int KB = D3P_GKB;
    using EdgeY = std::true_type;
    using EdgeN = std::false_type;
    auto main_loop = [&](auto A1P) {
    constexpr bool A1 = decltype(A1P)::value;
    Frag f0, f1;
    fetch(S0{});
    fetch(S1{});
    stage(S0{}, 0, kbeg, EdgeY{}, A1P);
    fetch(S0{});
    __syncthreads();
    read_frags(0, f0, A1P);
    if (D3P_GEMM_DIAG & 8) f1 = f0;
    const int ns = (kend - kbeg + KB - 1) / KB;
    // one MFMA, then a share of the other work of the same half slice: the bf16 matrix pipe runs beside the vector unit and the LDS,
    // but only what stands BETWEEN two MFMAs in a wave's instruction stream can run beside them
#ifndef D3P_SPLIT_RNE
#define D3P_MIX_VALU 12
#else
#define D3P_MIX_VALU 10
#endif
#define D3P_MIX_STAGE()                                                                                     \
    _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  /* 1 MFMA */                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, D3P_MIX_VALU, 0); /* the splitting's vector instructions */ \
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);  /* 2 LDS writes */                              \
    }
#define D3P_MIX_READ()                                                                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                                                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  /* 2 LDS reads */                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  /* 1 MFMA */                                    \
    }
#define D3P_MIX_STAGE_A1()                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) {                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  /* 1 MFMA */                                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0); /* 10 VALU (splitting B, packing A) */          \
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);  /* 2 LDS writes */                              \
    }
#define D3P_MIX_READ_A1()                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) {                                                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  /* 3 LDS reads */                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  /* 1 MFMA */                                    \
    }
    for (int i = 0; i < ns; i += 2) {
        const int k_i = kbeg + i * KB;
        if (!m_edge && k_i + 3 * KB <= kend) {   // both slices staged in this iteration are whole: the branch-free body
            mma_half(f0, 0, A1P);
            stage(S1{}, 1, k_i + KB, EdgeN{}, A1P);
            if (A1) { D3P_MIX_STAGE_A1() } else { D3P_MIX_STAGE() }
            fetch(S1{});
            __syncthreads();
            read_frags(1, f1, A1P);
            mma_half(f0, 1, A1P);
            if (A1) { D3P_MIX_READ_A1() } else { D3P_MIX_READ() }
            __builtin_amdgcn_sched_barrier(0);
            mma_half(f1, 0, A1P);
            stage(S0{}, 0, k_i + 2 * KB, EdgeN{}, A1P);
            if (A1) { D3P_MIX_STAGE_A1() } else { D3P_MIX_STAGE() }
            fetch(S0{});
            __syncthreads();
            read_frags(0, f0, A1P);
            mma_half(f1, 1, A1P);
            if (A1) { D3P_MIX_READ_A1() } else { D3P_MIX_READ() }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            mma_half(f0, 0, A1P);
            stage(S1{}, 1, k_i + KB, EdgeY{}, A1P);
            fetch(S1{});
            __syncthreads();
            read_frags(1, f1, A1P);
            mma_half(f0, 1, A1P);
            mma_half(f1, 0, A1P);
            stage(S0{}, 0, k_i + 2 * KB, EdgeY{}, A1P);
            fetch(S0{});
            __syncthreads();
            read_frags(0, f0, A1P);
            mma_half(f1, 1, A1P);
        }
    }
#undef D3P_MIX_STAGE_A1
#undef D3P_MIX_READ_A1
    };
    // (wave-uniform: the flag is one word for the whole product)
    if (g.a_exact16 && (uint32_t)__builtin_amdgcn_readfirstlane((int)*g.a_exact16) != g.a_exact_nonce) main_loop(std::true_type{});
    else main_loop(std::false_type{});
#undef D3P_MIX_STAGE
#undef D3P_MIX_READ
    const int col = n0 + 32 * grp + (lane & 31);
    if (EPI4) {   // (every lane stays for the row sums: a column beyond N contributes zeros)
        const bool col_ok = col < g.N;
        const float bv = (g.bias && col_ok) ? g.bias[col] : 0.f;
        float ll[16], xx[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + wr * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
            const bool ok = col_ok && row < g.M;
            const float o = __fmaf_rn(g.alpha, acc[v], bv), x = xpre[v];
            // one exponential per element, as in k_vae_out: en = exp(-|o|); softplus(o) = max(o, 0) + log(1 + en)
            const float en = __expf(-fabsf(o)), r = __builtin_amdgcn_rcpf(1.0f + en);
            if (ok) g.C[(size_t)row * g.ldc + col] = g.ex_sc * ((o >= 0.0f ? r : en * r) - x);
            ll[v] = ok ? x * o - (fmaxf(o, 0.0f) + __logf(1.0f + en)) : 0.f;
            xx[v] = x * x;
        }
        // sums over the 32 columns of each half wave (lanes 0 - 31: rows + 0, lanes 32 - 63: rows + 4): four DPP steps inside the rows
        // of 16 lanes, row_bcast:15 adds row 0 into row 1 and row 2 into row 3 -- lanes 31 and 63 hold the half sums
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float a_ = ll[v], b_ = xx[v];
            a_ += dpp_mov<0xB1>(a_); b_ += dpp_mov<0xB1>(b_);
            a_ += dpp_mov<0x4E>(a_); b_ += dpp_mov<0x4E>(b_);
            a_ += dpp_mov<0x141>(a_); b_ += dpp_mov<0x141>(b_);
            a_ += dpp_mov<0x140>(a_); b_ += dpp_mov<0x140>(b_);
            asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1"
                : "+v"(a_), "+v"(b_));
            ll[v] = a_; xx[v] = b_;
        }
        if ((lane & 31) == 31 && n0 + 32 * grp < g.N) {
            const int cg = (n0 >> 5) + grp;   // this wave's group of 32 columns
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m0 + wr * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
                if (row < g.M) {
                    g.ep_ll[(size_t)cg * g.M + row] = ll[v];
                    g.ep_xx[(size_t)cg * g.M + row] = xx[v];
                }
            }
        }
        return;
    }
    if (col >= g.N) return;
    if (g.part) {
        float* out = g.part + (size_t)tz * g.M * g.N;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + wr * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
            if (row < g.M) out[(size_t)row * g.N + col] = acc[v];
        }
        return;
    }
    const float bv = g.bias ? g.bias[col + (col >= g.n_seg ? g.bias_njump : 0ll)] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int row = m0 + wr * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
        if (row < g.M) gemm_store(g, row, col, acc[v], bv);
    }
}

template <bool AK, bool BN, bool EPI4 = false>
__global__ void __launch_bounds__(512) k_gemm_bf16x3(GemmArgs g)
{
    int tx, ty, tz;
    xcd_tile(tx, ty, tz);
    gemm_bf16x3_tile<AK, BN, EPI4>(g, tx, ty, tz);
}

// Several products of one operand form in ONE launch (the weight-gradient products of a step: independent of each other, each too
// short to fill the chip for a whole number of rounds -- 112 / 468 / 490 / 128 workgroups on 256 CUs -- and each paying its own
// launch floor): a linear grid over the workgroups of all members, dealt to the XCDs as xcd_tile deals one product's (every XCD
// walks one contiguous run of the order member, K slab, m, n), the host choosing ONE K range per workgroup for all members so that
// the run is a whole number of equal rounds (gemm_group_splits).
#define D3P_GROUP_MAX 6
#define D3P_WPART_SPLITS 16  // most split-K partial tiles a product leaves
struct GemmGroup {
    GemmArgs g[D3P_GROUP_MAX];
    // Every XCD takes one contiguous eighth of EVERY member, member after member (so the XCDs work through the members in step --
    // one run over all members would give some XCDs only the cheap ones): slot[p] = first per-XCD slot of member p (its share is
    // slot[p + 1] - slot[p] = ceil(cnt[p] / 8) workgroups per XCD; the up to 7 surplus workgroups of a member leave at once)
    unsigned slot[D3P_GROUP_MAX + 1];
    unsigned cnt[D3P_GROUP_MAX];         // workgroups of member p = gx gy splits
    unsigned gx[D3P_GROUP_MAX], gy[D3P_GROUP_MAX];
    // one more workgroup behind the members', when asked for: *sum_out = sum of sum_in[0 .. sum_n - 1] in fixed order (the loss sum of
    // a step whose output layer was fused -- NormArgs: the norm kernel in front of this launch completes px_loss, so it cannot sum it)
    const float* sum_in;
    float* sum_out;
    unsigned sum_n;
};

template <bool AK, bool BN>
__global__ void __launch_bounds__(512) k_gemm_bf16x3_group(GemmGroup q)
{
    if (q.sum_in && blockIdx.x == 0) {   // (workgroup-uniform; the FIRST workgroup: dealt out last it would be the launch's tail)
        __shared__ float part[512];
        float sm = 0.f;
        for (unsigned i = threadIdx.x; i < q.sum_n; i += 512) sm += q.sum_in[i];
        part[threadIdx.x] = sm;
        __syncthreads();
        for (int off = 256; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) *q.sum_out = part[0];
        return;
    }
    const unsigned bid = blockIdx.x - (q.sum_in ? 1u : 0u);
    const unsigned c = bid & 7u, j = bid >> 3;
    int p = 0;
#pragma unroll
    for (int k = 1; k < D3P_GROUP_MAX; ++k) p += (j >= q.slot[k]) ? 1 : 0;
    const unsigned R = c * (q.slot[p + 1] - q.slot[p]) + (j - q.slot[p]);
    if (R >= q.cnt[p]) return;
    const unsigned gx = q.gx[p], gy = q.gy[p];
    gemm_bf16x3_tile<AK, BN>(q.g[p], (int)(R % gx), (int)((R / gx) % gy), (int)(R / (gx * gy)));
}

// members collected by gemm() instead of being launched (its `group` argument), then launched together
struct GemmGroupPlan {
    GemmGroup q;
    int n = 0;
    bool ak = false, bn = false;
    const float* sum_in = nullptr;   // GemmGroup::sum_in / sum_out / sum_n
    float* sum_out = nullptr;
    unsigned sum_n = 0;
};

static int gemm_group_launch(hipStream_t s, GemmGroupPlan& G)
{
    if (G.n == 0) return D3P_OK;
    unsigned slots = 0;
    for (int p = 0; p < D3P_GROUP_MAX; ++p) {
        G.q.slot[p] = slots;
        if (p < G.n) slots += (G.q.cnt[p] + 7u) / 8u;
        else { G.q.cnt[p] = 0; G.q.gx[p] = G.q.gy[p] = 1; G.q.g[p] = G.q.g[0]; }
    }
    G.q.slot[D3P_GROUP_MAX] = slots;
    G.q.sum_in = G.sum_in; G.q.sum_out = G.sum_out; G.q.sum_n = G.sum_n;
    const dim3 grid(8u * slots + (G.sum_in ? 1u : 0u));
    if (G.ak && G.bn) hipLaunchKernelGGL((k_gemm_bf16x3_group<true, true>), grid, dim3(512), 0, s, G.q);
    else if (G.ak) hipLaunchKernelGGL((k_gemm_bf16x3_group<true, false>), grid, dim3(512), 0, s, G.q);
    else if (G.bn) hipLaunchKernelGGL((k_gemm_bf16x3_group<false, true>), grid, dim3(512), 0, s, G.q);
    else hipLaunchKernelGGL((k_gemm_bf16x3_group<false, false>), grid, dim3(512), 0, s, G.q);
    G.n = 0;
    return check_launch("k_gemm_bf16x3_group");
}

// Split count of a k_gemm_bf16x3 product, or ONE count for all members of a group (they share K = the batch): the one whose rounds
// of one workgroup per CU (92 KB of LDS each) cost least, a round costing its K range plus about two slices of prologue and
// epilogue.  tiles = 128 x 64 tiles of the product, or of all members together.
static int gemm_group_splits(unsigned tiles, int K)
{
    static const int n_cu = [] {   // (one process drives one device: d3p_amd.dist; initialised once, thread-safely)
        int dev = 0, n = 0;
        const bool ok = hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0;
        if (!ok) (void)hipGetLastError();
        return ok ? n : 256;
    }();
    int best = 1;
    double best_cost = 1e30;
    for (int sp = 1; sp <= D3P_WPART_SPLITS; ++sp) {
        int kp = (K + sp - 1) / sp;
        kp = (kp + D3P_GKB - 1) / D3P_GKB * D3P_GKB;
        const int ns = (K + kp - 1) / kp;
        if (ns != sp || kp < 2 * D3P_GKB) continue;
        const double rounds = std::ceil((double)tiles * ns / n_cu);
        const double cost = rounds * (kp + 2 * D3P_GKB);
        if (cost < best_cost) { best_cost = cost; best = sp; }
    }
    return best;
}

__global__ void k_gemm_reduce(GemmArgs g, int splits)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)g.M * g.N) return;
    const int row = (int)(t / g.N), col = (int)(t % g.N);
    // (all tiles requested at once, like k_vae_finalize: a loop over a run-time count is one memory round trip per tile)
    float tv[D3P_WPART_SPLITS];
#pragma unroll
    for (int z = 0; z < D3P_WPART_SPLITS; ++z) tv[z] = z < splits ? g.part[(size_t)z * g.M * g.N + t] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int z = 0; z < D3P_WPART_SPLITS; ++z) s += z < splits ? tv[z] : 0.f;  // fixed order
    gemm_store(g, row, col, s, g.bias ? g.bias[col + (col >= g.n_seg ? g.bias_njump : 0ll)] : 0.f);
}

// part / part_floats: optional split-K scratch.  At B = 4096 the GEMMs of this model are parallelism-starved on 256 CUs (bigger
// tiles lose), so besides the weight gradients the two N = 400 forward / backward-data GEMMs are split too.
// The split count is chosen so that short grids (the weight-gradient
// GEMMs: K = batch, M x N = a weight matrix) still put a few workgroups on every CU.
// will gemm() send this product (no displaced B segments) to k_gemm_bf16x3 with an n-fast B?  (the conditions of `big` below)
static bool gemm_takes_bf16_nfast(const float* A, long long a_sm, long long a_sk, const float* B, long long b_sk, long long b_sn, int M, int N, int K,
                                  int a_last_one)
{
    static const bool fp32_mfma = getenv("D3P_GEMM_FP32_MFMA") != nullptr;
    auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const int m_real_h = a_last_one ? M - 1 : M, K4 = (K + 3) & ~3;
    const bool va8 = aligned16(A) && ((a_sk == 1 && a_sm % 4 == 0 && a_sm >= K4) || (a_sm == 1 && a_sk % 4 == 0 && a_sk >= ((m_real_h + 3) & ~3)));
    const bool vb8 = aligned16(B) && b_sn == 1 && b_sk % 4 == 0 && N % 4 == 0;
    return !fp32_mfma && va8 && vb8 && (M > 96 || (M > 32 && K >= 2048));
}

// what a product may ask for beyond C = alpha op(A) op(B) + bias (+ C); every field optional
struct GemmOpts {
    int a_last_one = 0;             // GemmArgs::a_last_one
    float* part = nullptr;          // split-K scratch ([splits][M][N]) and its size: without it the product is not split
    size_t part_floats = 0;
    int epi = 0;                    // GemmArgs::epi and its operands
    float* C2 = nullptr;
    const float* ex_zu = nullptr;
    const float* ex_eps = nullptr;
    int ex_Z = 0;
    float ex_sc = 0.f;
    int* splits_left = nullptr;     // a split-K product is NOT reduced: the partial tiles stay in `part` and *splits_left says how many
                                    // (0: the product went to C as usual); the consumer sums them in fixed order
    const GemmJumps* jumps = nullptr;
    const uint32_t* a_exact16 = nullptr;   // GemmArgs::a_exact16 / a_exact_nonce
    uint32_t a_exact_nonce = 0u;
    GemmGroupPlan* group = nullptr; // (with splits_left) a product that takes the bf16 kernel is appended to the group instead of
    int force_splits = 0;           // being launched (gemm_group_launch), with force_splits K slabs instead of a count of its own
    const float* b_row_scale = nullptr;    // GemmArgs::b_row_scale: bf16 kernel with an n-fast B only (gemm_takes_bf16_nfast)
    const float* ep_x = nullptr;    // epi 4 (with ex_sc): k-fast A, n-fast B on the bf16 kernel, unsplit (gemm_takes_bf16_nfast)
    float *ep_ll = nullptr, *ep_xx = nullptr;
};

static int gemm(hipStream_t s, const float* A, long long a_sm, long long a_sk, const float* B, long long b_sk, long long b_sn,
                float* C, int ldc, int M, int N, int K, const float* bias, float alpha, int accumulate, const GemmOpts& o = GemmOpts())
{
    const int a_last_one = o.a_last_one, epi = o.epi, ex_Z = o.ex_Z, force_splits = o.force_splits;
    float* const part = o.part;
    const size_t part_floats = o.part_floats;
    float* const C2 = o.C2;
    int* const splits_left = o.splits_left;
    const GemmJumps* const jumps = o.jumps;
    const float *const ex_zu = o.ex_zu, *const ex_eps = o.ex_eps, *const b_row_scale = o.b_row_scale;
    const float ex_sc = o.ex_sc;
    const uint32_t* const a_exact16 = o.a_exact16;
    const uint32_t a_exact_nonce = o.a_exact_nonce;
    GemmGroupPlan* const group = o.group;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K;
    g.a_sm = a_sm; g.a_sk = a_sk; g.b_sk = b_sk; g.b_sn = b_sn;
    g.ldc = ldc; g.alpha = alpha; g.accumulate = accumulate;
    g.a_last_one = a_last_one;
    g.epi = epi;
    g.C2 = C2;
    g.a_exact16 = a_exact16;
    g.a_exact_nonce = a_exact_nonce;
    g.b_row_scale = b_row_scale;
    g.ep_x = o.ep_x; g.ep_ll = o.ep_ll; g.ep_xx = o.ep_xx;
    g.ex_zu = ex_zu; g.ex_eps = ex_eps; g.ex_Z = ex_Z; g.ex_sc = ex_sc;
    g.n_seg = jumps ? jumps->n_seg : 0x7fffffff;
    g.k_seg = jumps ? jumps->k_seg : 0x7fffffff;
    g.b_njump = jumps ? jumps->b_njump : 0;
    g.b_kjump = jumps ? jumps->b_kjump : 0;
    g.bias_njump = jumps ? jumps->bias_njump : 0;
    g.c_njump = jumps ? jumps->c_njump : 0;
    // 16-byte fetches along the unit-stride dimension when every such load is aligned: base pointer, the other stride and the
    // K range of a split (k_per is a multiple of 16) -- the kernels guard the M / N edges themselves, K must be a multiple of 4
    auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const bool va = aligned16(A) && K % 4 == 0 && ((a_sk == 1 && a_sm % 4 == 0) || (a_sm == 1 && a_sk % 4 == 0));
    const bool vb = aligned16(B) && K % 4 == 0 && ((b_sn == 1 && b_sk % 4 == 0) || (b_sk == 1 && b_sn % 4 == 0)) &&
                    !(jumps && (jumps->b_njump || jumps->b_kjump));  // displaced B segments: scalar fetch only
    // 128 x 64 tiles (k_gemm_f32_w8): whole float4s only (see its header)
    // (also for a short, very deep product -- the V1 weight gradient, 51 x 400 x 4096: its workgroups run alone on their CUs, where
    // the 64 x 64 kernel's one-slice pipeline leaves every load latency exposed: 21.7 us)
    const int m_real_h = a_last_one ? M - 1 : M, K4 = (K + 3) & ~3;
    // (the eight-wave kernel takes any K: a k-fast operand needs a row stride of at least roundup4(K))
    const bool va8 = aligned16(A) && ((a_sk == 1 && a_sm % 4 == 0 && a_sm >= K4) || (a_sm == 1 && a_sk % 4 == 0 && a_sk >= ((m_real_h + 3) & ~3)));
    const bool vb8 = aligned16(B) && ((b_sn == 1 && b_sk % 4 == 0 && N % 4 == 0) || (b_sk == 1 && b_sn % 4 == 0 && b_sn >= K4)) &&
                     !(jumps && (jumps->b_njump || jumps->b_kjump));
    const bool big = va8 && vb8 && (M > 96 || (M > 32 && K >= 2048));
    static const bool fp32_mfma = getenv("D3P_GEMM_FP32_MFMA") != nullptr;  // developer switch: the fp32-MFMA kernel for the large products
    if (b_row_scale && !(big && !fp32_mfma && b_sn == 1)) return fail(D3P_E_INVALID_ARG, "gemm: b_row_scale on a product that does not take the bf16 kernel");
    if (epi == 4 && !(big && !fp32_mfma && !part && a_sk == 1 && b_sn == 1 && o.ep_x && o.ep_ll && o.ep_xx))
        return fail(D3P_E_INVALID_ARG, "gemm: epilogue 4 on a product that does not take the bf16 kernel unsplit");
    const int tm = big ? D3P_GTM : D3P_GT;
    const unsigned tiles = cdiv(N, D3P_GT) * cdiv(M, tm);
    int splits = 1;
    if (part && tiles < (big ? 160u : 512u) && K >= 8 * D3P_GK) {
        static const bool fp32_big = getenv("D3P_GEMM_FP32_MFMA") != nullptr;
        splits = big ? (fp32_big ? (int)(512 / tiles) : gemm_group_splits(tiles, K))   // (fp32 kernel: two 8-wave workgroups per CU, one round)
                     : (int)((1024 + tiles - 1) / tiles);
        const int max_by_k = K / (4 * D3P_GK);
        if (splits > max_by_k && !(big && !fp32_big)) splits = max_by_k;   // (gemm_group_splits keeps its K ranges at two slices or more)
        if (splits > 16) splits = 16;  // the reduction adds the partial tiles serially
        const size_t max_by_mem = part_floats / ((size_t)M * N);
        if ((size_t)splits > max_by_mem) splits = (int)max_by_mem;
        if (splits < 1) splits = 1;
    }
    if (force_splits > 0 && part && big) {
        splits = force_splits;
        const size_t max_by_mem = part_floats / ((size_t)M * N);
        if ((size_t)splits > max_by_mem) splits = (int)max_by_mem;
        if (splits < 1) splits = 1;
    }
    const int kq = big ? D3P_GKB : D3P_GK;  // K slice of the kernel: a split starts on a slice boundary
    int k_per = (K + splits - 1) / splits;
    k_per = (k_per + kq - 1) / kq * kq;
    splits = (K + k_per - 1) / k_per;
    g.k_per = k_per;
    g.part = splits > 1 ? part : nullptr;
    const dim3 grid(cdiv(N, D3P_GT), cdiv(M, tm), splits);
    if (big && !fp32_mfma && group && splits_left && group->n < D3P_GROUP_MAX &&
        (group->n == 0 || (group->ak == (a_sk == 1) && group->bn == (b_sn == 1)))) {
        const int p = group->n++;
        group->ak = a_sk == 1;
        group->bn = b_sn == 1;
        group->q.g[p] = g;
        group->q.gx[p] = grid.x;
        group->q.gy[p] = grid.y;
        group->q.cnt[p] = grid.x * grid.y * grid.z;
        *splits_left = splits > 1 ? splits : 0;
        return D3P_OK;
    }
    if (big && !fp32_mfma) {
        const bool ak = a_sk == 1, bn = b_sn == 1;
        if (epi == 4) hipLaunchKernelGGL((k_gemm_bf16x3<true, true, true>), grid, dim3(512), 0, s, g);
        else if (ak && bn) hipLaunchKernelGGL((k_gemm_bf16x3<true, true>), grid, dim3(512), 0, s, g);
        else if (ak) hipLaunchKernelGGL((k_gemm_bf16x3<true, false>), grid, dim3(512), 0, s, g);
        else if (bn) hipLaunchKernelGGL((k_gemm_bf16x3<false, true>), grid, dim3(512), 0, s, g);
        else hipLaunchKernelGGL((k_gemm_bf16x3<false, false>), grid, dim3(512), 0, s, g);
    } else if (big) {
        const bool ak = a_sk == 1, bn = b_sn == 1;
        if (ak && bn) hipLaunchKernelGGL((k_gemm_f32_w8<true, true>), grid, dim3(512), 0, s, g);
        else if (ak) hipLaunchKernelGGL((k_gemm_f32_w8<true, false>), grid, dim3(512), 0, s, g);
        else if (bn) hipLaunchKernelGGL((k_gemm_f32_w8<false, true>), grid, dim3(512), 0, s, g);
        else hipLaunchKernelGGL((k_gemm_f32_w8<false, false>), grid, dim3(512), 0, s, g);
    } else if (va && vb) hipLaunchKernelGGL((k_gemm_f32<true, true>), grid, dim3(256), 0, s, g);
    else if (va) hipLaunchKernelGGL((k_gemm_f32<true, false>), grid, dim3(256), 0, s, g);
    else if (vb) hipLaunchKernelGGL((k_gemm_f32<false, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_gemm_f32<false, false>), grid, dim3(256), 0, s, g);
    if (splits_left) *splits_left = splits > 1 ? splits : 0;
    if (splits > 1 && !splits_left) hipLaunchKernelGGL(k_gemm_reduce, dim3(cdiv((uint64_t)M * N, 256)), dim3(256), 0, s, g, splits);
    return check_launch("k_gemm_f32");
}

// ------------------------------------------------------------------------------------------------------------------
// row-wise / element-wise stages
// ------------------------------------------------------------------------------------------------------------------
// DPSVI.evaluate: ONE guide draw for the whole batch -- eps = normal(k_z, (B, Z)) with
// rng_key_eval = split(key)[1], guide_seed = split(.)[1], k_z = split(.)[1]  (numpyro SVI.evaluate / Trace_ELBO / seed handler)
__global__ void k_vae_eval_eps(const uint32_t* __restrict__ jax_key, uint32_t B, int Z, float* __restrict__ eps)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)B * Z;
    if (t >= n) return;
    uint32_t k0 = jax_key[0], k1 = jax_key[1];
#pragma unroll
    for (int lvl = 0; lvl < 3; ++lvl) {
        uint32_t a, b0, b1;
        threefry2x32(k0, k1, 0u, 2u, a, b0);
        threefry2x32(k0, k1, 1u, 3u, a, b1);
        k0 = b0;
        k1 = b1;
    }
    eps[t] = bits_to_normal(tf_iota_word(k0, k1, (uint64_t)n, (uint64_t)t));
}

// Gaussian-mechanism noise for all parameter leaves (10, or 14 with two hidden layers) in one launch: leaf k draws
// normal(site_key_k, leaf shape) (svi.py:487-491), i.e. word w of ChaCha block b of key k is element 16 b + w of that leaf.
#define D3P_VAE_MAX_LEAVES 14 // parameter leaves: 2 (2 nh + 1) + 4
struct SiteNoiseArgs {
    const uint32_t* site_keys;                    // n_leaves x 16
    uint32_t blk_off[D3P_VAE_MAX_LEAVES + 1];     // prefix sums of ceil(leaf size / 16) (unused entries = the total)
    uint32_t elem_off[D3P_VAE_MAX_LEAVES + 1];    // prefix sums of the leaf sizes
    float* noise;
};

__device__ __forceinline__ void site_noise_block(const SiteNoiseArgs& a, uint32_t b)   // b = ChaCha block over all leaves
{
    if (b >= a.blk_off[D3P_VAE_MAX_LEAVES]) return;
    int site = 0;
#pragma unroll
    for (int k = 1; k < D3P_VAE_MAX_LEAVES; ++k) site += (b >= a.blk_off[k]) ? 1 : 0;
    const uint32_t lb = b - a.blk_off[site], n_site = a.elem_off[site + 1] - a.elem_off[site];
    uint32_t key[16], o[16];
    load_key(a.site_keys + 16 * site, key);
    keystream_block(key, lb, o);
    float* dst = a.noise + a.elem_off[site];
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t e = 16u * lb + w;
        if (e < n_site) dst[e] = bits_to_normal(o[w]);
    }
}

__global__ void __launch_bounds__(256) k_vae_site_noise(SiteNoiseArgs a) { site_noise_block(a, blockIdx.x * blockDim.x + threadIdx.x); }

// zl, u (B x Z), eps -> z = zl + exp(u) eps (written over zl), sd = exp(u) (written over u), lat[i] = log q - log p
// (zl / u, like dz / du below, are the two halves of one B x 2 Z array: row stride ld)
// jax_key != nullptr: the guide noise is drawn here -- eps[i][j] = normal word j of example i's sample key (svi.py:289-290;
// single site 'z'; the key of an example is a function of its GLOBAL position pos0 + i in a batch of B_total, so 1 GPU and N
// GPUs draw the same noise) -- and written to eps_out for the backward pass; otherwise eps_in is used (tests, evaluate).
// splits > 0: [zl | u] has not been written yet -- the product h [Wl | Ws] left `splits` split-K partial tiles in part
// ([split][B][2 Z]); they are summed here in the order and with the bias addition of k_gemm_reduce (which this replaces).
// noise_blocks > 0: the first noise_blocks workgroups of the launch draw the Gaussian-mechanism noise of the update instead
// (site_noise_block: needs the step's keys only, so it rides in this short launch instead of one of its own).
struct LatentArgs {
    float *zl, *u;
    const float* eps_in;
    const uint32_t* jax_key;
    uint32_t B_total, pos0;
    float* eps_out;
    uint32_t B;
    int Z, ld;
    float* lat;
    const float* part;
    int splits;
    const float *bias_l, *bias_s;
    unsigned noise_blocks;
    SiteNoiseArgs noise;
};

__global__ void __launch_bounds__(256) k_vae_latent(LatentArgs a)
{
    if (blockIdx.x < a.noise_blocks) {
        site_noise_block(a.noise, blockIdx.x * blockDim.x + threadIdx.x);
        return;
    }
    const uint32_t i = ((blockIdx.x - a.noise_blocks) * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= a.B) return;
    const int Z = a.Z;
    uint32_t k0 = 0, k1 = 0;
    if (a.jax_key) px_sample_key(a.jax_key[0], a.jax_key[1], a.B_total, a.pos0 + i, k0, k1);
    float acc = 0.f;
    for (int j = lane; j < Z; j += 64) {
        const size_t e = (size_t)i * a.ld + j;
        float ee;
        if (a.jax_key) {
            ee = bits_to_normal(tf_iota_word(k0, k1, (uint64_t)Z, (uint64_t)j));
            a.eps_out[(size_t)i * Z + j] = ee;
        } else {
            ee = a.eps_in[(size_t)i * Z + j];
        }
        float loc, uu;
        if (a.splits > 0) {
            const size_t tile = (size_t)a.B * 2 * Z, t = (size_t)i * 2 * Z + j;
            float sl = 0.f, su = 0.f;
            for (int z = 0; z < a.splits; ++z) { sl += a.part[z * tile + t]; su += a.part[z * tile + t + Z]; }  // fixed order
            loc = __fmaf_rn(1.0f, sl, a.bias_l[j]);
            uu = __fmaf_rn(1.0f, su, a.bias_s[j]);
        } else {
            loc = a.zl[e];
            uu = a.u[e];
        }
        const float sd = expf(uu), z = __fmaf_rn(sd, ee, loc);
        a.zl[e] = z;
        a.u[e] = sd;
        acc += (-0.5f * ee * ee - uu) + 0.5f * z * z;  // the log(2 pi) / 2 terms of log q and log p cancel
    }
    acc = wave_sum(acc);
    if (lane == 0) a.lat[i] = acc;
}

// logits a (B x D), x -> da = sc (sigmoid(a) - x) in place; px_loss[i] = sc (lat_i - sum_j (x a - softplus(a))) mask_i
// (also leaves x2[i] = |x_i|^2 for the norm kernel, which then need not read X again)
__global__ void k_vae_out(float* __restrict__ a, const float* __restrict__ X, const uint8_t* __restrict__ mask, uint32_t B, int D,
                          float sc, const float* __restrict__ lat, float* __restrict__ px_loss, float* __restrict__ x2)
{
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= B) return;
    // one exponential per element: e = exp(-|t|) gives softplus(t) = max(t, 0) + log1p(e) and sigmoid(t) = 1 / (1 + e) for
    // t >= 0, e / (1 + e) otherwise
    float xx = 0.f;
    auto elem = [&](float t, float x, float& ll) {
        xx = __fmaf_rn(x, x, xx);
        // (hardware exp2 / log2 / rcp paths: 1 + e is in (1, 2], so log(1 + e) has an absolute error of ~1e-7 against terms >= ln 2 e)
        const float e = __expf(-fabsf(t)), r = __builtin_amdgcn_rcpf(1.0f + e);
        ll += x * t - (fmaxf(t, 0.0f) + __logf(1.0f + e));
        return sc * ((t >= 0.0f ? r : e * r) - x);
    };
    float ll = 0.f;
    float* ar = a + (size_t)i * D;
    const float* xr = X + (size_t)i * D;
    if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(ar) | reinterpret_cast<uintptr_t>(xr)) & 15u) == 0) {
        for (int j = 4 * lane; j < D; j += 256) {
            const float4 t = *reinterpret_cast<const float4*>(ar + j), x = *reinterpret_cast<const float4*>(xr + j);
            float4 o;
            o.x = elem(t.x, x.x, ll); o.y = elem(t.y, x.y, ll); o.z = elem(t.z, x.z, ll); o.w = elem(t.w, x.w, ll);
            *reinterpret_cast<float4*>(ar + j) = o;
        }
    } else {
        for (int j = lane; j < D; j += 64) ar[j] = elem(ar[j], xr[j], ll);
    }
    ll = wave_sum(ll);
    xx = wave_sum(xx);
    if (lane == 0) {
        px_loss[i] = (mask && mask[i] == 0) ? 0.f : sc * (lat[i] - ll);
        x2[i] = xx;
    }
}

// joint L2 norm of every example's gradient by the outer-product identity, clip factor c_i (0 for masked rows); the rows of
// the delta arrays are staged in LDS while their squares are summed and written back scaled by c_i (svi.py:121-122
// folded into the sums), so the deltas are read once and written once (a separate rescaling pass read them a second time).
// One term per dense layer, in the order decoder output layer .. decoder first layer, latent heads, encoder last .. first:
// ||grad||^2 = sum_t (|input_t|^2 + 1) (|delta_t|^2 [+ |delta'_t|^2])  (the + 1 is the bias; the heads share their input)
#define D3P_VAE_MAX_TERMS 6
struct NormTerm {
    const float* in;  // B x in_n, row stride in_ld; nullptr: |x_i|^2 comes from x2 (left by k_vae_out)
    float* d0;        // B x d_n, row stride d_ld: the layer's delta, rescaled in place
    float* d1;        // the second head's delta (same shape), or nullptr
    int in_ld, in_n, d_ld, d_n;
};
struct NormArgs {
    NormTerm t[D3P_VAE_MAX_TERMS];
    int n_terms;
    int stage;  // floats of LDS per wave = sum of the delta widths
    const float* x2;
    const uint8_t* mask;
    uint32_t B;
    float clip;
    float* cf;
    float* norms;  // nullable
    int scale_back;        // 0: the deltas stay as they are -- the weight-gradient products apply cf themselves (GemmArgs::b_row_scale)
    float* px_loss;        // the LAST workgroup of the launch sums px_loss and counts the unmasked examples into loss_n[0..1]
    float* loss_n;         // (fixed order; was a launch of its own)
    // the output layer's epilogue left per-group partial sums instead of px_loss / x2 (GemmArgs::epi 4): every row's wave finishes
    // them here (vae_out_finish_row) -- px_loss is then complete only when this launch ends, so the last workgroup leaves loss_n[0]
    // alone and a later launch sums px_loss (k_vae_finalize / k_vae_tile_sums: vae_block_sum)
    const float* ep_ll;    // nullable: [ep_groups][B]
    const float* ep_xx;
    const float* lat;
    int ep_groups;         // <= 64
    float ep_sc;
};

// (the finishing step of GemmArgs::epi 4 for one example: px_loss[i] = sc (lat_i - sum_g ll[g][i]) mask_i; returns |x_i|^2)
__device__ __forceinline__ float vae_out_finish_row(const float* __restrict__ ep_ll, const float* __restrict__ ep_xx, int groups, uint32_t B,
                                                    uint32_t i, int lane, const float* __restrict__ lat, float sc, bool live,
                                                    float* __restrict__ px_loss)
{
    float l = 0.f, x = 0.f, sl, sx;
    if (lane < groups) {
        l = ep_ll[(size_t)lane * B + i];
        x = ep_xx[(size_t)lane * B + i];
    }
    wave_sum2(l, x, sl, sx);
    if (lane == 0) px_loss[i] = live ? sc * (lat[i] - sl) : 0.f;
    return sx;
}

// sum of v[0 .. n - 1] by one workgroup of 256 threads, fixed order; lds: 256 floats; every thread gets the total
__device__ __forceinline__ float vae_block_sum(const float* __restrict__ v, uint32_t n, float* lds)
{
    float s = 0.f;
    for (uint32_t i = threadIdx.x; i < n; i += 256) s += v[i];
    lds[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) lds[threadIdx.x] += lds[threadIdx.x + off];
        __syncthreads();
    }
    const float tot = lds[0];
    __syncthreads();
    return tot;
}

// this lane's share of a row's sum of squares (16-byte loads: elements 4 l .. 4 l + 3, + 256, ...; rows that cannot take them:
// elements l, l + 64, ...), copying what it reads to `keep` when given -- the wave sums come later, all terms at once
__device__ __forceinline__ bool row_vec4(const float* r, int n) { return (n & 3) == 0 && (reinterpret_cast<uintptr_t>(r) & 15u) == 0; }
__device__ __forceinline__ float row_sumsq_lane(const float* __restrict__ r, int n, int lane, float* __restrict__ keep)
{
    float s = 0.f;
    if (row_vec4(r, n)) {
        for (int j = 4 * lane; j < n; j += 256) {
            const float4 v = *reinterpret_cast<const float4*>(r + j);
            if (keep) *reinterpret_cast<float4*>(keep + j) = v;
            s = __fmaf_rn(v.x, v.x, s); s = __fmaf_rn(v.y, v.y, s); s = __fmaf_rn(v.z, v.z, s); s = __fmaf_rn(v.w, v.w, s);
        }
    } else {
        for (int j = lane; j < n; j += 64) {
            const float v = r[j];
            if (keep) keep[j] = v;
            s = __fmaf_rn(v, v, s);
        }
    }
    return s;
}
__device__ __forceinline__ void row_scale_back(float* __restrict__ r, int n, int lane, const float* __restrict__ keep, float c)
{
    if (row_vec4(r, n)) {
        for (int j = 4 * lane; j < n; j += 256) {
            const float4 v = *reinterpret_cast<const float4*>(keep + j);
            *reinterpret_cast<float4*>(r + j) = make_float4(v.x * c, v.y * c, v.z * c, v.w * c);
        }
    } else {
        for (int j = lane; j < n; j += 64) r[j] = keep[j] * c;
    }
}

__global__ void k_vae_norms(NormArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float norm_stage[];  // 4 waves x a.stage
    if (blockIdx.x == gridDim.x - 1) {  // sums[P] = sum_i px_loss[i], sums[P + 1] = number of unmasked examples
        float* l = norm_stage;
        float* c = norm_stage + 256;
        float s = 0.f, n = 0.f;
        for (uint32_t i = threadIdx.x; i < a.B; i += 256) {
            s += a.px_loss[i];
            n += (a.mask && a.mask[i] == 0) ? 0.f : 1.f;
        }
        l[threadIdx.x] = s;
        c[threadIdx.x] = n;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) { l[threadIdx.x] += l[threadIdx.x + off]; c[threadIdx.x] += c[threadIdx.x + off]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (!a.ep_ll) a.loss_n[0] = l[0];   // (fused output layer: px_loss is being written by this very launch)
            a.loss_n[1] = c[0];
        }
        return;
    }
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= a.B) return;
    float* keep = norm_stage + (size_t)(threadIdx.x >> 6) * a.stage;
    const float x2_row = a.ep_ll ? vae_out_finish_row(a.ep_ll, a.ep_xx, a.ep_groups, a.B, i, lane, a.lat, a.ep_sc, !(a.mask && a.mask[i] == 0), a.px_loss)
                                 : a.x2[i];
    // every term's loads first (per-lane partial sums: nothing waits for a wave sum between two terms), then all wave sums
    float pin[D3P_VAE_MAX_TERMS], pd[D3P_VAE_MAX_TERMS];
    {
        float* k = keep;
#pragma unroll
        for (int t = 0; t < D3P_VAE_MAX_TERMS; ++t) {
            pin[t] = pd[t] = 0.f;
            if (t < a.n_terms) {
                const NormTerm& q = a.t[t];
                if (q.in) pin[t] = row_sumsq_lane(q.in + (size_t)i * q.in_ld, q.in_n, lane, nullptr);
                pd[t] = row_sumsq_lane(q.d0 + (size_t)i * q.d_ld, q.d_n, lane, a.scale_back ? k : nullptr);
                k += (q.d_n + 3) & ~3;
                if (q.d1) {
                    pd[t] += row_sumsq_lane(q.d1 + (size_t)i * q.d_ld, q.d_n, lane, a.scale_back ? k : nullptr);
                    k += (q.d_n + 3) & ~3;
                }
            }
        }
    }
    float n2 = 0.f;
#pragma unroll
    for (int t = 0; t < D3P_VAE_MAX_TERMS; ++t) {
        if (t < a.n_terms) {
            float in2, d;
            wave_sum2(pin[t], pd[t], in2, d);
            if (!a.t[t].in) in2 = x2_row;
            n2 += (in2 + 1.0f) * d;
        }
    }
    const float nrm = sqrtf(n2);
    const bool live = !(a.mask && a.mask[i] == 0);
    const float c = live ? 1.0f / fmaxf(1.0f, nrm / a.clip) : 0.f;  // svi.py:121-122; masked rows contribute nothing
    if (lane == 0) {
        a.cf[i] = c;
        if (a.norms) a.norms[i] = live ? nrm : 0.f;
    }
    if (!a.scale_back) return;
    // every lane reads back exactly the LDS words it wrote: no barrier needed
    float* k = keep;
    for (int t = 0; t < a.n_terms; ++t) {
        const NormTerm& q = a.t[t];
        row_scale_back(q.d0 + (size_t)i * q.d_ld, q.d_n, lane, k, c);
        k += (q.d_n + 3) & ~3;
        if (q.d1) {
            row_scale_back(q.d1 + (size_t)i * q.d_ld, q.d_n, lane, k, c);
            k += (q.d_n + 3) & ~3;
        }
    }
}

// sums[P] = sum_i px_loss[i], sums[P + 1] = number of unmasked examples (one workgroup, fixed order)
__global__ void __launch_bounds__(256) k_vae_loss_n(const float* __restrict__ px_loss, const uint8_t* __restrict__ mask, uint32_t B,
                                                    float* __restrict__ out)
{
    __shared__ float l[256], c[256];
    float s = 0.f, n = 0.f;
    for (uint32_t i = threadIdx.x; i < B; i += 256) {
        s += px_loss[i];
        n += (mask && mask[i] == 0) ? 0.f : 1.f;
    }
    l[threadIdx.x] = s;
    c[threadIdx.x] = n;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { l[threadIdx.x] += l[threadIdx.x + off]; c[threadIdx.x] += c[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = l[0]; out[1] = c[0]; }
}

// the finishing step of GemmArgs::epi 4 as a launch of its own, where no norm kernel follows the forward pass (evaluate)
__global__ void __launch_bounds__(256) k_vae_out_finish(const float* __restrict__ ep_ll, const float* __restrict__ ep_xx, int groups, uint32_t B,
                                                        const float* __restrict__ lat, float sc, const uint8_t* __restrict__ mask,
                                                        float* __restrict__ px_loss, float* __restrict__ x2)
{
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= B) return;
    const float xx = vae_out_finish_row(ep_ll, ep_xx, groups, B, i, lane, lat, sc, !(mask && mask[i] == 0), px_loss);
    if (lane == 0) x2[i] = xx;
}

// mean, Gaussian mechanism, rescale (svi.py:343-346, :365-375), numpyro Adam (svi.py:379-393) over the P parameters
#define D3P_VAE_MAX_BLOCKS 7  // [W | b] blocks of the flat layout: 2 nh + 3 with nh <= 2 hidden layers

// weight-gradient blocks whose split-K partial tiles were left unreduced: block b = columns [w_off[b], w_off[b + 1]) of the flat
// layout, w_splits[b] tiles (0: the product wrote its sums itself)
struct VaeTiles {
    const float* wpart;                                         // nullptr: no block has tiles
    uint32_t w_off[D3P_VAE_MAX_BLOCKS + 1];                     // first column of block b (unused entries = P)
    uint32_t w_base[D3P_VAE_MAX_BLOCKS], w_tile[D3P_VAE_MAX_BLOCKS], w_ld[D3P_VAE_MAX_BLOCKS], w_coff[D3P_VAE_MAX_BLOCKS],
        w_out[D3P_VAE_MAX_BLOCKS];                              // its tiles: wpart + w_base, w_tile apart, element (r, c) of the
    int w_splits[D3P_VAE_MAX_BLOCKS];                           // block at r * w_ld + w_coff + c, c < w_out
};

// column col's sum over its block's tiles, in fixed order (`direct` = what the product wrote when the block has none)
__device__ __forceinline__ float vae_tile_sum(const VaeTiles& a, size_t col, float direct)
{
    if (!a.wpart) return direct;
    int b = 0;
#pragma unroll
    for (int k = 1; k < D3P_VAE_MAX_BLOCKS; ++k) b += (col >= a.w_off[k]) ? 1 : 0;
    if (a.w_splits[b] <= 0) return direct;
    const uint32_t e = (uint32_t)col - a.w_off[b];
    const float* t = a.wpart + a.w_base[b] + (size_t)(e / a.w_out[b]) * a.w_ld[b] + a.w_coff[b] + e % a.w_out[b];
    // (all tiles requested at once -- a loop over a run-time count is one memory round trip per tile: 16.6 -> 13.4 us)
    const int nz = a.w_splits[b];
    const size_t tile = a.w_tile[b];
    float tv[D3P_WPART_SPLITS];
#pragma unroll
    for (int z = 0; z < D3P_WPART_SPLITS; ++z) tv[z] = z < nz ? t[(size_t)z * tile] : 0.f;
    float tot = 0.f;
#pragma unroll
    for (int z = 0; z < D3P_WPART_SPLITS; ++z) tot += z < nz ? tv[z] : 0.f;
    return tot;
}

// sums[col] := the tile sums (stage API and data-parallel local sums: the sums leave the device function as one vector)
// px_loss != nullptr (fused output layer): sums[P] = sum_i px_loss[i] too (workgroup 0; the norm kernel could not: see NormArgs)
// col0 .. col1 - 1: the columns of this launch (a bucket of the data-parallel step's reduce: VaeBuckets)
__global__ void __launch_bounds__(256) k_vae_tile_sums(VaeTiles a, float* __restrict__ sums, size_t P, const float* __restrict__ px_loss, uint32_t B,
                                                       size_t col0, size_t col1)
{
    __shared__ float lds[256];
    if (px_loss && blockIdx.x == 0) {
        const float tot = vae_block_sum(px_loss, B, lds);
        if (threadIdx.x == 0) sums[P] = tot;
    }
    const size_t col = col0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col < col1) sums[col] = vae_tile_sum(a, col, sums[col]);
}

struct VaeFinalArgs {
    const float* sums;  // P + 2
    VaeTiles tiles;     // (single-device update: k_vae_finalize sums the tiles itself)
    const float* noise;
    const float* in_params;  // state before the update (== params / adam_m / adam_v, or the old state's arrays when the update
    const float* in_m;       // is out of place: DPSVI.update returns a NEW state, svi.py:395-434)
    const float* in_v;
    float* params;
    float* adam_m;
    float* adam_v;
    const int32_t* step;
    float* loss_out;
    float* grad_out;
    size_t P;
    uint32_t B;
    d3p_dpsvi_hyper h;
    float obs_scale;
    const float* px_loss;   // nullable (fused output layer, single-device update): the loss sum is taken here, sum_i px_loss[i] over
    uint32_t B_local;       // B_local examples, instead of from sums[P] (NormArgs)
};

// one parameter column of the update: mean + noise (svi.py:343-346, :365-375), rescale, numpyro Adam (svi.py:379-393)
__device__ __forceinline__ void vae_finalize_col(const VaeFinalArgs& a, size_t col, float tot, float n, float factor, float bc0, float bc1)
{
    const float Bf = (float)a.B;
    const float g = (tot / Bf + a.noise[col] * (a.h.dp_scale * (a.h.clip / n))) * a.obs_scale * factor;
    if (a.grad_out) a.grad_out[col] = g;
    float m = a.in_m[col], v = a.in_v[col];
    m = (1.0f - a.h.b1) * g + a.h.b1 * m;
    v = (1.0f - a.h.b2) * g * g + a.h.b2 * v;
    const float mhat = m / bc0;
    const float vhat = v / bc1;
    a.params[col] = a.in_params[col] - a.h.lr * mhat / (sqrtf(vhat) + a.h.adam_eps);
    a.adam_m[col] = m;
    a.adam_v[col] = v;
}

__global__ void __launch_bounds__(256) k_vae_finalize(VaeFinalArgs a)
{
    __shared__ float bc[2];  // Adam's bias corrections 1 - b^(i + 1): two powf per workgroup instead of per column
    __shared__ float lsum[256];
    if (threadIdx.x < 2) bc[threadIdx.x] = 1.0f - powf(threadIdx.x ? a.h.b2 : a.h.b1, (float)(*a.step + 1));
    __syncthreads();
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float n = a.sums[a.P + 1], Bf = (float)a.B;
    const float factor = (n == 0.f) ? 0.f : Bf / n;
    if (blockIdx.x == 0 && a.loss_out) {   // (workgroup-uniform)
        const float ls = a.px_loss ? vae_block_sum(a.px_loss, a.B_local, lsum) : a.sums[a.P];
        if (threadIdx.x == 0) *a.loss_out = (ls / Bf) * a.obs_scale * factor;
    }
    if (col >= a.P) return;
    vae_finalize_col(a, col, vae_tile_sum(a.tiles, col, a.sums[col]), n, factor, bc[0], bc[1]);
}

// The data-parallel step's tile sums, full-mesh all-reduce (d3p_fmesh.hip: reduce-scatter, rank-order owner sums, all-gather as tagged
// 8-byte words through the peers' hipIpc-mapped inboxes) and update in ONE launch: the scatter phase takes a column's value straight
// from the split-K partial tiles (k_vae_tile_sums' work), the gather phase applies noise + Adam to a column the moment its sum over the
// ranks has arrived (k_vae_finalize's work) -- two launches and two passes over the 2.76 MB of sums less per step.  The owner stores
// its chunk's sums into its OWN gather slots too, so that every column (and the global example count every column's update needs) is
// read the same way.  Same arithmetic, column by column, as tile sums -> d3p_fmesh_allreduce -> k_vae_finalize: bit for bit.
__global__ void __launch_bounds__(256) k_vae_fmesh_step(FMeshArgs m, VaeFinalArgs f)
{
    constexpr int U = 4;
    __shared__ float bc[2];
    if (threadIdx.x < 2) bc[threadIdx.x] = 1.0f - powf(threadIdx.x ? f.h.b2 : f.h.b1, (float)(*f.step + 1));
    __syncthreads();
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    char* const mine = m.peer[m.rank];
    const size_t sc_par = (size_t)m.parity * m.world * m.chunk;
    auto local = [&](uint64_t i) { return i < (uint64_t)f.P ? vae_tile_sum(f.tiles, (size_t)i, f.sums[i]) : f.sums[i]; };   // [P]: loss sum, [P + 1]: count
    // ---- reduce-scatter
    for (uint64_t i0 = tid; i0 < m.n; i0 += U * nthreads) {
        float v[U];
        int o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * nthreads;
            o[u] = i < m.n ? (int)(i / m.chunk) : m.rank;
            v[u] = o[u] != m.rank ? local(i) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * nthreads;
            if (o[u] != m.rank) fm_store(m.peer[o[u]], sc_par + (size_t)m.rank * m.chunk + (i - (uint64_t)o[u] * m.chunk), v[u], m.tag);
        }
    }
    // ---- owner sums of my chunk in rank order; the sums go to EVERY rank's gather slot [parity][my rank], mine included
    const uint64_t lo = (uint64_t)m.rank * m.chunk, hi = lo + m.chunk < m.n ? lo + m.chunk : m.n;
    bool ok = true;
    for (uint64_t i = lo + tid; i < hi && ok; i += nthreads) {
        const uint64_t j = i - lo;
        const float own = local(i);
        unsigned long long w[D3P_FMESH_MAX_WORLD];
        bool all = m.world == 1;
        for (uint32_t spins = 0; spins < D3P_FMESH_WAIT_ROUNDS && !all; ++spins) {
#pragma unroll
            for (int r = 0; r < D3P_FMESH_MAX_WORLD; ++r)
                if (r < m.world && r != m.rank)
                    w[r] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(mine) + sc_par + (size_t)r * m.chunk + j, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_SYSTEM);
            all = true;
#pragma unroll
            for (int r = 0; r < D3P_FMESH_MAX_WORLD; ++r)
                if (r < m.world && r != m.rank) all = all && (uint32_t)(w[r] >> 32) == m.tag;
            if (!all && (spins & 255u) == 255u && __hip_atomic_load(m.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        }
        if (!all) { ok = false; break; }
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < D3P_FMESH_MAX_WORLD; ++r)
            if (r < m.world) {
                const float v = r == m.rank ? own : __uint_as_float((uint32_t)w[r]);
                s = r == 0 ? v : s + v;
            }
        for (int p = 0; p < m.world; ++p) fm_store(m.peer[p] + m.gather_off, sc_par + (size_t)m.rank * m.chunk + j, s, m.tag);
    }
    // ---- gather + update: the example count first (every column's update needs it), then U columns' sums requested together
    const unsigned long long* const gin = reinterpret_cast<const unsigned long long*>(mine + m.gather_off) + sc_par;
    auto gword = [&](uint64_t i) { const uint64_t o = i / m.chunk; return (size_t)o * m.chunk + (size_t)(i - o * m.chunk); };
    float n = 0.f;
    {
        const size_t wn = gword((uint64_t)f.P + 1u);
        bool have = false;
        for (uint32_t spins = 0; spins < D3P_FMESH_WAIT_ROUNDS && ok; ++spins) {
            const unsigned long long w = __hip_atomic_load(gin + wn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((uint32_t)(w >> 32) == m.tag) { n = __uint_as_float((uint32_t)w); have = true; break; }
            if ((spins & 255u) == 255u && __hip_atomic_load(m.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        }
        ok = ok && have;
    }
    const float Bf = (float)f.B, factor = (n == 0.f) ? 0.f : Bf / n;
    for (uint64_t i0 = tid; i0 < m.n && ok; i0 += U * nthreads) {
        size_t word[U];
        bool need[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * nthreads;
            need[u] = i <= (uint64_t)f.P;   // the P columns and the loss sum
            word[u] = need[u] ? gword(i) : 0;
        }
        unsigned long long w[U];
        bool all = false;
        for (uint32_t spins = 0; spins < D3P_FMESH_WAIT_ROUNDS && !all; ++spins) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (need[u]) w[u] = __hip_atomic_load(gin + word[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            all = true;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (need[u]) all = all && (uint32_t)(w[u] >> 32) == m.tag;
            if (!all && (spins & 255u) == 255u && __hip_atomic_load(m.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        }
        if (!all) { ok = false; break; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!need[u]) continue;
            const uint64_t i = i0 + (uint64_t)u * nthreads;
            const float s = __uint_as_float((uint32_t)w[u]);
            if (i < (uint64_t)f.P) vae_finalize_col(f, (size_t)i, s, n, factor, bc[0], bc[1]);
            else if (f.loss_out) *f.loss_out = (s / Bf) * f.obs_scale * factor;
        }
    }
    if (!ok) {
        uint32_t expect = 0u;
        (void)__hip_atomic_compare_exchange_strong(m.status, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// rows of the five delta arrays scaled by the clip factors in one launch
// All keys of one update in ONE launch (they were three launches + a step-counter launch + a 64-byte copy, ~5 us each):
// keys[0..47] = [next | gradient | perturbation] = split(state_key, 3) (svi.py:208-211), keys[48 ..] =
// split(perturbation_key, n_sites) (svi.py:491; n_sites = 10 or 14 leaves), keys[D3P_VAE_KEY_JAX ..+1] =
// convert_to_jax_rng_key(gradient_key).  advance: also write the next state key into the other key slot, save the optimiser
// step index in keys[D3P_VAE_KEY_STEP] for k_vae_finalize and advance it.
#define D3P_VAE_KEY_JAX (48 + 16 * D3P_VAE_MAX_LEAVES)
#define D3P_VAE_KEY_STEP (D3P_VAE_KEY_JAX + 2)
#define D3P_VAE_KEY_WORDS (D3P_VAE_KEY_JAX + 4)
// Further workgroups of the launch do work of the step that depends on nothing but its inputs (a launch of their own cost as
// much as this one): workgroups 1 .. x_blocks the exactness pass over the batch (exact16_pass); the ones behind them pack the two
// latent heads, which lie H Z + Z apart in the flat layout with rows of Z floats (no 16-byte fetches: the head products took the
// scalar 64 x 64 kernel), side by side into wcat (HE x 2 Z = [Wl | Ws]) and transposed into wcatT (2 Z x HE = [Wl^T ; Ws^T]).
struct KeysExtra {
    const float* x;
    size_t x_n4;
    uint32_t* x_flag;
    uint32_t x_nonce;
    unsigned x_blocks;
    const float *wl, *wsd;   // the heads' weights (HE x Z each)
    float *wcat, *wcatT;
    int HE, Z;
    unsigned pack_blocks;
    // the run loop's step (d3p_dpvi_vae_run): the workgroups behind the packing ones make the step's batch indices --
    // fold_in(batch_key, batch_i) (minibatch.py:230), its 30 round constants, the Feistel permutation of positions 0 .. n_idx - 1
    // (util.py:248-301), every workgroup deriving the constants for itself (two dependent quad-lane ChaCha blocks) like k_sampler
    const uint32_t* batch_key;   // nullable
    uint32_t batch_i, capacity, n_idx;
    int bits_lower, bits_upper;
    uint32_t* idx;
};

__global__ void __launch_bounds__(256) k_vae_keys(const uint32_t* __restrict__ cur_key, uint32_t* __restrict__ keys,
                                                  uint32_t* __restrict__ next_slot, const int32_t* __restrict__ step,
                                                  int32_t* __restrict__ step_out, int advance, int n_sites, KeysExtra ex)
{
    if (blockIdx.x > ex.x_blocks + ex.pack_blocks) {
        __shared__ uint32_t f_key[16], f_rc[32];
        const int tid = threadIdx.x, quad = tid >> 2, q = tid & 3;
        if (quad == 0) {
            uint32_t ka, kb;
            derive_child_quad(ex.batch_key, 0u, D3P_TAG_FOLD, ex.batch_i, ka, kb);
            f_key[q] = ex.batch_key[q];
            f_key[4 + q] = ka;
            f_key[8 + q] = kb;
            f_key[12 + q] = 0u;
        }
        __syncthreads();
        if (quad < 2) {  // round constants: keystream blocks 0, 1 (util.py:240-246), column 0 forced odd
            uint32_t w[4];
            keystream_block_quad(f_key, (uint32_t)quad, w[0], w[1], w[2], w[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g = 16 * quad + 4 * i + q;
                if (g < 30) f_rc[g] = (g % 3 == 0) ? (w[i] | 1u) : w[i];
            }
        }
        __syncthreads();
        const uint32_t p = (blockIdx.x - ex.x_blocks - ex.pack_blocks - 1) * blockDim.x + (uint32_t)tid;
        if (p < ex.n_idx) ex.idx[p] = feistel_permute_dev(f_rc, ex.capacity, ex.bits_lower, ex.bits_upper, p);
        return;
    }
    if (blockIdx.x > ex.x_blocks) {
        const unsigned t = (blockIdx.x - ex.x_blocks - 1) * blockDim.x + threadIdx.x;
        const unsigned Z2 = 2u * ex.Z;
        if (t < (unsigned)ex.HE * Z2) {
            const unsigned k = t / Z2, n = t % Z2;
            const float v = n < (unsigned)ex.Z ? ex.wl[k * ex.Z + n] : ex.wsd[k * ex.Z + n - ex.Z];
            ex.wcat[t] = v;
            ex.wcatT[(size_t)n * ex.HE + k] = v;
        }
        return;
    }
    if (blockIdx.x > 0) {
        exact16_pass(ex.x, ex.x_n4, ex.x_flag, ex.x_nonce, blockIdx.x - 1, ex.x_blocks);
        return;
    }
    // one quad of lanes per derivation (4-lane ChaCha block, d3p_device.h), on the first wave: the launch is pure latency
    __shared__ uint32_t sk[3][16];
    const bool kt = threadIdx.x < 64;
    const int lane = threadIdx.x & 63, quad = lane >> 2, q = lane & 3;
    auto store_child = [&](uint32_t* dst, const uint32_t* parent, uint32_t a, uint32_t b) {
        dst[q] = parent[q];
        dst[4 + q] = a;
        dst[8 + q] = b;
        dst[12 + q] = 0u;
    };
    if (kt) {
        uint32_t a, b;
        derive_child_quad(cur_key, quad < 3 ? (uint32_t)quad : 0u, D3P_TAG_SPLIT, 0u, a, b);
        if (quad < 3) {
            store_child(keys + 16 * quad, cur_key, a, b);
            store_child(sk[quad], cur_key, a, b);
            if (quad == 0 && advance) store_child(next_slot, cur_key, a, b);
        }
        if (lane == 63 && advance) {
            const int32_t i = *step;
            keys[D3P_VAE_KEY_STEP] = (uint32_t)i;
            *step_out = i + 1;  // (step_out == step, or the new state's counter when the update is out of place)
        }
    }
    __syncthreads();
    if (kt) {
        // quad 0: block 0 of the gradient key's stream -> jax key; quads 1..n_sites (<= 15): split(perturbation_key, n_sites)
        const uint32_t* parent = quad == 0 ? sk[1] : sk[2];
        const bool site = quad >= 1 && quad <= n_sites;
        uint32_t a, b;
        derive_child_quad(parent, site ? (uint32_t)(quad - 1) : 0u, quad == 0 ? 0u : D3P_TAG_SPLIT, 0u, a, b);
        if (quad == 0 && q < 2) keys[D3P_VAE_KEY_JAX + q] = a;
        if (site) store_child(keys + 48 + 16 * (quad - 1), parent, a, b);
    }
}

// The network as lists of dense layers.  Hidden widths hs[0 .. nh - 1]: [H] (the reference, examples/vae.py:80-85) or
// [H, H2] (BASELINE config 5's 784 -> [400, 200] variant, model->H2 > 0):
//   encoder (guide)  x -> hs[0] -> .. -> hs[nh - 1] : dense layers enc[0 .. nh - 1] (softplus), then the heads (Wl, bl), (Ws, bs)
//   decoder (model)  z -> hs[nh - 1] -> .. -> hs[0] -> D : dense layers dec[0 .. nh] (softplus after all but the last)
// Flat layout = tree_flatten order of {'decoder$params', 'encoder$params'}: decoder layers, encoder layers, Wl, bl, Ws, bs; a
// layer is [W (in x out) | b (out)] (stax.Dense).  nh = 1: V1, c1, V2, c2, W1, b1, Wl, bl, Ws, bs -- 10 leaves, 5 blocks.
struct VaeDense { size_t W, b; int in, out; };
struct VaeNet {
    int nh, D, Z, HE;  // HE = hs[nh - 1]: width of the encoder's last hidden layer = input of the heads
    VaeDense dec[3], enc[2];
    size_t Wl, bl, Ws, bs, P;
    int n_leaves() const { return 2 * (2 * nh + 1) + 4; }
    int n_blocks() const { return 2 * nh + 3; }
    // [W | b] block b of the flat layout: decoder layers, encoder layers, Wl, Ws
    const VaeDense* block_layer(int blk) const { return blk <= nh ? &dec[blk] : blk <= 2 * nh ? &enc[blk - nh - 1] : nullptr; }
};

static VaeNet vae_net(const d3p_vae_model* m)
{
    VaeNet n;
    memset(&n, 0, sizeof(n));
    n.nh = m->H2 > 0 ? 2 : 1;
    n.D = m->D;
    n.Z = m->Z;
    const int hs[2] = {m->H, m->H2};
    n.HE = hs[n.nh - 1];
    size_t off = 0;
    auto dense = [&](VaeDense* l, int in, int out) {
        l->in = in; l->out = out;
        l->W = off; off += (size_t)in * out;
        l->b = off; off += (size_t)out;
    };
    for (int l = 0; l <= n.nh; ++l) dense(&n.dec[l], l == 0 ? n.Z : hs[n.nh - l], l == n.nh ? n.D : hs[n.nh - l - 1]);
    for (int l = 0; l < n.nh; ++l) dense(&n.enc[l], l == 0 ? n.D : hs[l - 1], hs[l]);
    n.Wl = off; off += (size_t)n.HE * n.Z;
    n.bl = off; off += (size_t)n.Z;
    n.Ws = off; off += (size_t)n.HE * n.Z;
    n.bs = off; off += (size_t)n.Z;
    n.P = off;
    return n;
}

struct VaeWorkspace {
    float *he[2], *sge[2];   // encoder hidden activations and sigmoid(pre-activation) = softplus'
    float *hd[2], *sgd[2];   // decoder hidden, likewise (hd[l] = output of dec[l])
    float *dd[2], *de[2];    // deltas at the pre-activations of dec[l] / enc[l]
    float *zl, *u, *eps, *a, *dz, *du;
    float *lat, *px_loss, *x2, *cf, *sums, *noise, *part, *wpart;
    size_t part_floats;
    uint32_t* keys;  // 3 x 16 split + up to 14 x 16 site keys + jax key + step index (D3P_VAE_KEY_*)
    uint32_t* x_exact16;  // one word: the batch X is exactly bf16 (GemmArgs::a_exact16), set per forward pass
    float *wcat, *wcatT;  // the latent heads packed for the step (k_vae_keys): [Wl | Ws] (HE x 2 Z) and its transpose
    float *ep_ll, *ep_xx; // per-group partial row sums of the output layer's epilogue (GemmArgs::epi 4): [ceil(D / 32)][B] each
};

// one step of the native run loop (d3p_dpvi_vae_run): the batch is rows of a resident table, chosen by the Feistel sampler
struct VaeRunStep {
    const uint32_t* batch_key;   // batchifier state (device)
    uint32_t batch_i;            // fold_in data: index of the batch (minibatch.py:230)
    const float* table;          // n_rows x D
    uint32_t n_rows;
    uint32_t* idx;               // B indices (written)
};

// what the key launch of an update has already done for the passes that follow it (vae_step_keys)
struct VaeStepPrep {
    bool x_checked = false;     // the exactness pass over X of THIS forward pass has been enqueued
    bool heads_packed = false;  // ws.wcat / ws.wcatT hold the heads of the parameters the pass runs on
};

static size_t vae_carve(const d3p_vae_model* m, uint32_t B, char* base, VaeWorkspace* ws)
{
    const VaeNet N = vae_net(m);
    const size_t D = (size_t)N.D, Z = (size_t)N.Z, P = N.P;
    size_t off = 0;
    auto take = [&](size_t n_floats) { size_t o = off; off += align_up_v(n_floats * sizeof(float), 256); return base ? (float*)(base + o) : nullptr; };
    float* q;
    if (ws) memset(ws, 0, sizeof(*ws));
    for (int l = 0; l < N.nh; ++l) {
        q = take(B * (size_t)N.enc[l].out); if (ws) ws->he[l] = q;
        q = take(B * (size_t)N.enc[l].out); if (ws) ws->sge[l] = q;
    }
    q = take(B * 2 * Z); if (ws) { ws->zl = q; ws->u = q + Z; }    // [z_loc -> z | log z_std -> z_std]: B x 2 Z, row stride 2 Z
    q = take(B * Z); if (ws) ws->eps = q;
    for (int l = 0; l < N.nh; ++l) {
        q = take(B * (size_t)N.dec[l].out); if (ws) ws->hd[l] = q;
        q = take(B * (size_t)N.dec[l].out); if (ws) ws->sgd[l] = q;
    }
    q = take(B * D); if (ws) ws->a = q;
    for (int l = 0; l < N.nh; ++l) { q = take(B * (size_t)N.dec[l].out); if (ws) ws->dd[l] = q; }
    q = take(B * 2 * Z); if (ws) { ws->dz = q; ws->du = q + Z; }  // [dz | du], likewise
    for (int l = 0; l < N.nh; ++l) { q = take(B * (size_t)N.enc[l].out); if (ws) ws->de[l] = q; }
    q = take(B); if (ws) ws->lat = q;
    q = take(B); if (ws) ws->px_loss = q;
    q = take(B); if (ws) ws->x2 = q;
    q = take(B); if (ws) ws->cf = q;
    q = take(P + 2); if (ws) ws->sums = q;
    q = take(P); if (ws) ws->noise = q;
    q = take(D3P_VAE_KEY_WORDS); if (ws) ws->keys = (uint32_t*)q;
    const size_t pf = 16 * (D + 1) * (size_t)m->H;  // split-K partial tiles: up to 16 splits of the largest weight matrix
    q = take(pf); if (ws) { ws->part = q; ws->part_floats = pf; }
    q = take((size_t)D3P_WPART_SPLITS * P); if (ws) ws->wpart = q;  // unreduced weight-gradient tiles
    q = take(64); if (ws) ws->x_exact16 = (uint32_t*)q;
    q = take((size_t)N.HE * 2 * Z); if (ws) ws->wcat = q;
    q = take((size_t)N.HE * 2 * Z); if (ws) ws->wcatT = q;
    q = take((size_t)cdiv(D, 32) * B); if (ws) ws->ep_ll = q;
    q = take((size_t)cdiv(D, 32) * B); if (ws) ws->ep_xx = q;
    return off;
}

static int vae_validate(const d3p_vae_model* m, const char* what)
{
    if (!m) return fail(D3P_E_INVALID_ARG, "%s: null model", what);
    if (!(m->D >= 1 && m->H >= 1 && m->Z >= 1 && m->H2 >= 0 && m->scale > 0.f && m->inv_obs > 0.f))
        return fail(D3P_E_INVALID_ARG, "%s: bad model (D, H, Z >= 1, H2 >= 0, scale > 0, inv_obs > 0)", what);
    return D3P_OK;
}

// [W | b] blocks of the flat layout and their unreduced tiles in ws.wpart: the dense layers, then Wl and Ws -- which come from ONE
// product with N = 2 Z: shared tiles of (HE + 1) x 2 Z in the region of the Wl block
static VaeTiles vae_tiles(const VaeNet& N, const VaeWorkspace& ws, const int* w_splits)
{
    VaeTiles f;
    memset(&f, 0, sizeof(f));
    f.wpart = w_splits ? ws.wpart : nullptr;
    const int nh = N.nh, n_blocks = N.n_blocks();
    for (int b = 0; b < n_blocks; ++b) {
        const VaeDense* lay = N.block_layer(b);
        const size_t first = lay ? lay->W : (b == 2 * nh + 1 ? N.Wl : N.Ws);
        const size_t next = b + 1 < n_blocks ? (N.block_layer(b + 1) ? N.block_layer(b + 1)->W : (b + 1 == 2 * nh + 1 ? N.Wl : N.Ws)) : N.P;
        f.w_off[b] = (uint32_t)first;
        f.w_base[b] = (uint32_t)(D3P_WPART_SPLITS * first);
        f.w_tile[b] = (uint32_t)(next - first);
        f.w_ld[b] = f.w_out[b] = (uint32_t)(lay ? lay->out : N.Z);
        f.w_coff[b] = 0;
        f.w_splits[b] = w_splits ? w_splits[b] : 0;
    }
    const int bl = 2 * nh + 1, bs = 2 * nh + 2;
    f.w_tile[bl] = f.w_tile[bs] = f.w_tile[bl] + f.w_tile[bs];
    f.w_ld[bl] = f.w_ld[bs] = 2u * (uint32_t)N.Z;
    f.w_base[bs] = f.w_base[bl];
    f.w_coff[bs] = (uint32_t)N.Z;
    for (int b = n_blocks; b <= D3P_VAE_MAX_BLOCKS; ++b) f.w_off[b] = (uint32_t)N.P;
    return f;
}

// forward pass: activations, the reparametrised latent, da = sc (sigmoid(a) - x) and px_loss[i] = sc (log q - log p - log lik)
// eps != nullptr: given noise; otherwise drawn from jax_key inside k_vae_latent into ws.eps
// the nonce of the current exactness pass over the batch (per host thread: a forward pass and the weight-gradient products that
// follow it are enqueued by one thread, back to back)
static uint32_t vae_exact_nonce(bool advance)
{
    static thread_local uint32_t nonce = 0x5eed0000u;
    if (advance) ++nonce;
    return nonce;
}

// may the batch take the one-plane path at all (k_exact16_flag reads whole 16-byte words)?
static bool vae_exact_eligible(const float* X, uint32_t B, int D)
{
    static const bool no_exact = getenv("D3P_VAE_NO_EXACT16") != nullptr;   // developer switch (A/B), read once
    return !no_exact && ((size_t)B * D) % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15u) == 0;
}
static unsigned vae_exact_blocks(size_t n4) { return (unsigned)(n4 / 1024 < 1 ? 1 : (n4 / 1024 > 1024 ? 1024 : n4 / 1024)); }

static int vae_enqueue_forward(hipStream_t s, const d3p_vae_model* m, const float* params, const float* X, const uint8_t* mask,
                               uint32_t B, const float* eps, float sc, const VaeWorkspace& ws, const uint32_t* jax_key = nullptr,
                               uint32_t B_total = 0, uint32_t pos0 = 0, VaeStepPrep prep = VaeStepPrep(), const SiteNoiseArgs* noise = nullptr,
                               bool* out_fused = nullptr)
{
    // noise != nullptr: the Gaussian-mechanism noise of the update is drawn beside the latent kernel
    // out_fused != nullptr: the caller's norm kernel finishes a fused output layer (*out_fused says whether there is one: ws.ep_ll /
    // ws.ep_xx then hold per-group partial sums, px_loss / x2 do not exist yet); nullptr: px_loss and x2 are complete on return
    int rc;
    const VaeNet N = vae_net(m);
    const int D = N.D, Z = N.Z, HE = N.HE, nh = N.nh, Bi = (int)B;
    const dim3 rows(cdiv((uint64_t)B * 64, 256));
    // Is the batch exactly bf16 (binarised images, examples/vae.py:157-168)?  One pass over X (13 MB) decides for the two products
    // that take X as their A operand -- the first encoder layer and its weight gradient: they then stage one plane of A instead
    // of splitting it into three and issue three of the six products (GemmArgs::a_exact16).  Any other batch takes the general path.
    const uint32_t* xflag = nullptr;
    if (vae_exact_eligible(X, B, D)) {
        if (!prep.x_checked) {
            const size_t n4 = (size_t)B * D / 4;
            const uint32_t nonce = vae_exact_nonce(true);   // a new nonce per pass: the flag word is never reset
            hipLaunchKernelGGL(k_exact16_flag, dim3(vae_exact_blocks(n4)), dim3(256), 0, s, X, n4, ws.x_exact16, nonce);
        }
        xflag = ws.x_exact16;
    }
    // ---- encoder (guide)
    {
        const float* in = X;
        for (int l = 0; l < nh; ++l) {
            const VaeDense& e = N.enc[l];
            GemmOpts o;   // softplus epilogue (sigmoid = its derivative to sge), split-K where the grid is short, one-plane A for the batch
            o.part = ws.part; o.part_floats = ws.part_floats; o.epi = 1; o.C2 = ws.sge[l];
            o.a_exact16 = l == 0 ? xflag : nullptr; o.a_exact_nonce = vae_exact_nonce(false);
            if ((rc = gemm(s, in, e.in, 1, params + e.W, e.out, 1, ws.he[l], e.out, Bi, e.out, e.in, params + e.b, 1.f, 0, o))) return rc;
            in = ws.he[l];
        }
    }
    // [z_loc | log z_std] = h [Wl | Ws] + [bl | bs] in ONE product of N = 2 Z: Ws lies HE Z behind where Wl's columns Z .. 2 Z - 1
    // would be, and bs likewise behind bl (flat layout: Wl, bl, Ws, bs)
    const int ldz = 2 * Z;
    // (packed heads: B = ws.wcat, a plain HE x 2 Z matrix -- only the bias keeps its displaced second half)
    const GemmJumps enc = {Z, 0x7fffffff, prep.heads_packed ? 0ll : (long long)HE * Z, 0, (long long)HE * Z, 0};
    // (a split-K product leaves its partial tiles for k_vae_latent to sum: no reduction launch)
    int zl_splits = 0;
    {
        GemmOpts o;
        o.part = ws.part; o.part_floats = ws.part_floats; o.splits_left = &zl_splits; o.jumps = &enc;
        if ((rc = gemm(s, ws.he[nh - 1], HE, 1, prep.heads_packed ? ws.wcat : params + N.Wl, prep.heads_packed ? 2 * Z : Z, 1, ws.zl, ldz, Bi, 2 * Z, HE,
                       params + N.bl, 1.f, 0, o)))
            return rc;
    }
    {
        LatentArgs la;
        memset(&la, 0, sizeof(la));
        la.zl = ws.zl; la.u = ws.u;   // zl := z, u := sd
        la.eps_in = eps; la.jax_key = eps ? (const uint32_t*)nullptr : jax_key;
        la.B_total = B_total ? B_total : B; la.pos0 = pos0;
        la.eps_out = ws.eps; la.B = B; la.Z = Z; la.ld = ldz; la.lat = ws.lat;
        la.part = ws.part; la.splits = zl_splits;
        la.bias_l = params + N.bl; la.bias_s = params + N.bs;
        if (noise) {
            la.noise = *noise;
            la.noise_blocks = cdiv(noise->blk_off[D3P_VAE_MAX_LEAVES], 256);
        }
        hipLaunchKernelGGL(k_vae_latent, dim3(rows.x + la.noise_blocks), dim3(256), 0, s, la);
    }
    // ---- decoder (model)
    {
        const float* in = ws.zl;
        int ld_in = ldz;
        for (int l = 0; l < nh; ++l) {
            const VaeDense& d = N.dec[l];
            // (the first product has K = Z: no split)
            GemmOpts o;
            o.part = l ? ws.part : nullptr; o.part_floats = l ? ws.part_floats : 0; o.epi = 1; o.C2 = ws.sgd[l];
            if ((rc = gemm(s, in, ld_in, 1, params + d.W, d.out, 1, ws.hd[l], d.out, Bi, d.out, d.in, params + d.b, 1.f, 0, o))) return rc;
            in = ws.hd[l];
            ld_in = d.out;
        }
        const VaeDense& o = N.dec[nh];
        // the output layer: when the product takes the bf16 kernel its epilogue does k_vae_out's work (GemmArgs::epi 4) instead of a
        // launch behind it
        static const bool no_fuse = getenv("D3P_VAE_NO_OUT_FUSE") != nullptr;   // developer switch (A/B), read once
        const int groups = (int)cdiv(D, 32);
        const bool fuse = !no_fuse && groups <= 64 && gemm_takes_bf16_nfast(in, ld_in, 1, params + o.W, D, 1, Bi, D, o.in, 0);
        if (out_fused) *out_fused = fuse;
        if (fuse) {
            GemmOpts go;
            go.epi = 4; go.ex_sc = sc; go.ep_x = X; go.ep_ll = ws.ep_ll; go.ep_xx = ws.ep_xx;
            if ((rc = gemm(s, in, ld_in, 1, params + o.W, D, 1, ws.a, D, Bi, D, o.in, params + o.b, 1.f, 0, go))) return rc;   // a := da
            if (!out_fused)
                hipLaunchKernelGGL(k_vae_out_finish, rows, dim3(256), 0, s, (const float*)ws.ep_ll, (const float*)ws.ep_xx, groups, B, (const float*)ws.lat,
                                   sc, mask, ws.px_loss, ws.x2);
            return check_launch("d3p_vae forward");
        }
        if ((rc = gemm(s, in, ld_in, 1, params + o.W, D, 1, ws.a, D, Bi, D, o.in, params + o.b, 1.f, 0))) return rc;
    }
    hipLaunchKernelGGL(k_vae_out, rows, dim3(256), 0, s, ws.a, X, mask, B, D, sc, (const float*)ws.lat, ws.px_loss, ws.x2);  // a := da
    return check_launch("d3p_vae forward");
}

// forward + backward + norms + clipped sums into ws.sums[P + 2]; eps_dev given or drawn from jax_key
// Data-parallel step (d3p_dpvi_vae_run_dist): the clipped sums leave the rank in TWO buckets, so that the reduce of the first
// travels while the products of the second still run.  (Every clipped sum needs the clip factors, i.e. the whole backward pass:
// what can overlap a reduce is the tail of the step -- the weight-gradient products, 63 of its 240 us at 4096 examples per rank.)
// Bucket 0 = the decoder's leaves (columns 0 .. split - 1 of the flat layout), bucket 1 = the encoder's and the latent heads' +
// [loss sum, count]; the weight-gradient products go out as one grouped launch per bucket, each followed by the tile sums of its
// columns and `ready(bucket)` (the caller enqueues the reduce behind an event).
// developer switch (A/B), read once: the weight-gradient products launched one by one.  The two-bucket hand-over exists only in the
// grouped form, so every caller that decides "two buckets" asks this predicate too (else no reduce would be enqueued at all).
static bool vae_no_group()
{
    static const bool v = getenv("D3P_VAE_NO_GROUP") != nullptr;
    return v;
}
struct VaeBuckets {
    size_t split;
    int (*ready)(void* ctx, int bucket);
    void* ctx;
};

static int vae_enqueue_sums(hipStream_t s, const d3p_vae_model* m, const float* params, const float* X, const uint8_t* mask,
                            uint32_t B, const float* eps_ext, const uint32_t* jax_key, float clip, const VaeWorkspace& ws,
                            float* norms_out, uint32_t B_total = 0, uint32_t pos0 = 0, int* w_splits = nullptr, VaeStepPrep prep = VaeStepPrep(),
                            const SiteNoiseArgs* noise = nullptr, bool* loss_pending = nullptr, const VaeBuckets* bk = nullptr)
{
    // The split-K partial tiles of the weight-gradient products stay in ws.wpart.  w_splits != nullptr (single-device update):
    // w_splits[0 .. n_blocks - 1] says how many each, and k_vae_finalize sums them; otherwise ONE launch (k_vae_tile_sums) sums
    // them into ws.sums here.
    if (B_total == 0) B_total = B;
    int rc;
    const VaeNet N = vae_net(m);
    const int D = N.D, Z = N.Z, HE = N.HE, nh = N.nh, Bi = (int)B;
    const float sc = m->inv_obs * m->scale;
    const float* eps = eps_ext ? eps_ext : ws.eps;  // (drawn inside k_vae_latent when not given)
    const dim3 rows(cdiv((uint64_t)B * 64, 256));
    // loss_pending (with w_splits, the single-device update): *loss_pending = the output layer was fused and ws.sums[P] is NOT the loss
    // sum -- the caller's k_vae_finalize takes it from ws.px_loss; without w_splits the tile-sum launch below does
    bool out_fused = false;
    if ((rc = vae_enqueue_forward(s, m, params, X, mask, B, eps_ext, sc, ws, jax_key, B_total, pos0, prep, noise, &out_fused))) return rc;
    // ---- backward (data): delta_in = (delta_out W^T) . softplus'(pre) down the decoder
    {
        const float* delta = ws.a;
        for (int l = nh - 1; l >= 0; --l) {
            const VaeDense& d = N.dec[l + 1];
            GemmOpts o;   // epilogue 2: times softplus'(pre)
            o.part = ws.part; o.part_floats = ws.part_floats; o.epi = 2; o.C2 = ws.sgd[l];
            if ((rc = gemm(s, delta, d.out, 1, params + d.W, 1, d.out, ws.dd[l], d.in, Bi, d.in, d.out, nullptr, 1.f, 0, o))) return rc;
            delta = ws.dd[l];
        }
    }
    const int ldz = 2 * Z;
    // dz = dpre V1^T + sc z and du = dz sd eps - sc in the product's epilogue (epi 3; was the k_vae_dlatent launch)
    {
        GemmOpts o;
        o.part = ws.part; o.part_floats = ws.part_floats; o.epi = 3; o.ex_zu = ws.zl; o.ex_eps = eps; o.ex_Z = Z; o.ex_sc = sc;
        if ((rc = gemm(s, ws.dd[0], N.dec[0].out, 1, params + N.dec[0].W, 1, N.dec[0].out, ws.dz, ldz, Bi, Z, N.dec[0].out, nullptr, 1.f, 0, o))) return rc;
    }
    // dpre = ([dz | du] [Wl^T ; Ws^T]) . softplus'(pre): ONE product of K = 2 Z (rows Z .. 2 Z - 1 of the stacked B are Ws^T,
    // HE Z behind where Wl^T's would be)
    // (packed heads: B = ws.wcatT, a plain 2 Z x HE matrix with 16-byte rows -- the product takes the bf16 kernel)
    const GemmJumps dec = {0x7fffffff, Z, 0, (long long)HE * Z, 0, 0};
    {
        GemmOpts o;
        o.epi = 2; o.C2 = ws.sge[nh - 1];
        if (prep.heads_packed) {
            rc = gemm(s, ws.dz, ldz, 1, ws.wcatT, HE, 1, ws.de[nh - 1], HE, Bi, HE, 2 * Z, nullptr, 1.f, 0, o);
        } else {
            o.jumps = &dec;
            rc = gemm(s, ws.dz, ldz, 1, params + N.Wl, 1, Z, ws.de[nh - 1], HE, Bi, HE, 2 * Z, nullptr, 1.f, 0, o);
        }
        if (rc) return rc;
    }
    for (int l = nh - 2; l >= 0; --l) {
        const VaeDense& e = N.enc[l + 1];
        GemmOpts o;
        o.part = ws.part; o.part_floats = ws.part_floats; o.epi = 2; o.C2 = ws.sge[l];
        if ((rc = gemm(s, ws.de[l + 1], e.out, 1, params + e.W, 1, e.out, ws.de[l], e.in, Bi, e.in, e.out, nullptr, 1.f, 0, o))) return rc;
    }
    // ---- clipped sums: weights  A^T (diag(c) Delta)  (GEMMs over the batch), biases = column sums
    // [W | b] of every layer is contiguous in the flat layout, so the bias gradient is row `in` of a GEMM whose A carries a
    // virtual row of ones
    float* S = ws.sums;
    // one product per dense layer (decoder, then encoder), then ONE for [Wl | Ws] (B operand [dz | du], N = 2 Z; the Ws block of
    // the sums lies HE Z behind where columns Z .. 2 Z - 1 of a Z-wide C would be)
    struct WG { const float* A; long long a_sk; int in; const float* Bm; int ldb; int out; int ldc; size_t off; const GemmJumps* j; int blk; };
    const GemmJumps wls = {Z, 0x7fffffff, 0, 0, 0, (long long)HE * Z};
    WG wg[6];
    int n_wg = 0;
    for (int l = 0; l <= nh; ++l) {
        const VaeDense& d = N.dec[l];
        wg[n_wg++] = {l == 0 ? ws.zl : ws.hd[l - 1], l == 0 ? ldz : d.in, d.in, l == nh ? ws.a : ws.dd[l], d.out, d.out, d.out, d.W, nullptr, l};
    }
    for (int l = 0; l < nh; ++l) {
        const VaeDense& e = N.enc[l];
        wg[n_wg++] = {l == 0 ? X : ws.he[l - 1], e.in, e.in, ws.de[l], e.out, e.out, e.out, e.W, nullptr, nh + 1 + l};
    }
    wg[n_wg++] = {ws.he[nh - 1], HE, HE, ws.dz, ldz, 2 * Z, Z, N.Wl, &wls, 2 * nh + 1};
    // Where do the clip factors meet the deltas?  When every product takes the bf16 kernel it multiplies the rows of its B operand by
    // c_i while staging them (GemmArgs::b_row_scale) and the norm kernel only reads; otherwise the norm kernel writes the rows back scaled.
    static const bool scale_back_env = getenv("D3P_VAE_SCALE_BACK") != nullptr;   // developer switch (A/B), read once
    bool scale_in_gemm = !scale_back_env;
    for (int b = 0; b < n_wg; ++b)
        scale_in_gemm = scale_in_gemm && gemm_takes_bf16_nfast(wg[b].A, 1, wg[b].a_sk, wg[b].Bm, wg[b].ldb, 1, wg[b].in + 1, wg[b].out, Bi, 1);
    // ---- per-example norms and clip factors (and, unless scale_in_gemm, the rows of every delta come back scaled by c_i)
    NormArgs na;
    memset(&na, 0, sizeof(na));
    na.scale_back = scale_in_gemm ? 0 : 1;
    {
        int t = 0;
        auto term = [&](const float* in, int in_ld, int in_n, float* d0, float* d1, int d_ld, int d_n) {
            NormTerm& q = na.t[t++];
            q.in = in; q.in_ld = in_ld; q.in_n = in_n; q.d0 = d0; q.d1 = d1; q.d_ld = d_ld; q.d_n = d_n;
            na.stage += ((d_n + 3) & ~3) * (d1 ? 2 : 1);   // (every row of the stage starts on a 16-byte boundary)
        };
        term(ws.hd[nh - 1], N.dec[nh].in, N.dec[nh].in, ws.a, nullptr, D, D);
        for (int l = nh - 1; l >= 1; --l) term(ws.hd[l - 1], N.dec[l].in, N.dec[l].in, ws.dd[l], nullptr, N.dec[l].out, N.dec[l].out);
        term(ws.zl, ldz, Z, ws.dd[0], nullptr, N.dec[0].out, N.dec[0].out);
        term(ws.he[nh - 1], HE, HE, ws.dz, ws.du, ldz, Z);
        for (int l = nh - 1; l >= 1; --l) term(ws.he[l - 1], N.enc[l].in, N.enc[l].in, ws.de[l], nullptr, N.enc[l].out, N.enc[l].out);
        term(nullptr, 0, 0, ws.de[0], nullptr, N.enc[0].out, N.enc[0].out);
        na.n_terms = t;
    }
    na.x2 = ws.x2;
    na.mask = mask; na.B = B; na.clip = clip; na.cf = ws.cf; na.norms = norms_out;
    na.px_loss = ws.px_loss; na.loss_n = ws.sums + N.P;
    if (out_fused) { na.ep_ll = ws.ep_ll; na.ep_xx = ws.ep_xx; na.lat = ws.lat; na.ep_groups = (int)cdiv(D, 32); na.ep_sc = sc; }
    {
        size_t stage_floats = na.scale_back ? 4 * (size_t)na.stage : 0;
        if (stage_floats < 512) stage_floats = 512;  // the last workgroup's loss / count reduction uses 2 x 256 floats
        hipLaunchKernelGGL(k_vae_norms, dim3(rows.x + 1), dim3(256), stage_floats * sizeof(float), s, na);
    }
    // single-device update: the products go out as ONE grouped launch with a common K range per workgroup
    // (bk: one grouped launch per bucket)
    const bool no_group = vae_no_group();
    int splits_here[D3P_VAE_MAX_BLOCKS] = {0};
    int* const tile_splits = w_splits ? w_splits : splits_here;
    bool loss_summed = false;   // (fused output layer: a group launch sums px_loss into S[P] beside its products)
    const int n_buckets = (bk && !no_group) ? 2 : 1;
    for (int bucket = 0; bucket < n_buckets; ++bucket) {
        auto in_bucket = [&](int b) { return n_buckets == 1 || (bucket == 0) == (wg[b].off < bk->split); };
        GemmGroupPlan plan;
        GemmGroupPlan* group = !no_group ? &plan : nullptr;
        int group_splits = 0;
        if (group) {
            unsigned tiles = 0;
            for (int b = 0; b < n_wg; ++b)
                if (in_bucket(b)) tiles += cdiv(wg[b].out, D3P_GT) * cdiv(wg[b].in + 1, D3P_GTM);
            group_splits = gemm_group_splits(tiles, Bi);
        }
        for (int b = 0; b < n_wg; ++b) {
            if (!in_bucket(b)) continue;
            float* part = ws.wpart + (size_t)D3P_WPART_SPLITS * wg[b].off;
            const size_t part_floats = (size_t)D3P_WPART_SPLITS * (wg[b].in + 1) * wg[b].out;
            int left = 0;
            // (A = X^T: the flag of the forward pass holds -- same batch; the virtual row of ones is exact too)
            const uint32_t* xflag = (wg[b].A == X && vae_exact_eligible(X, B, D)) ? ws.x_exact16 : nullptr;
            GemmOpts o;   // A carries the virtual row of ones (bias gradient); tiles left unreduced; grouped; clip factors on B's rows
            o.a_last_one = 1; o.part = part; o.part_floats = part_floats; o.splits_left = &left; o.jumps = wg[b].j;
            o.a_exact16 = xflag; o.a_exact_nonce = vae_exact_nonce(false);
            o.group = group; o.force_splits = group_splits; o.b_row_scale = scale_in_gemm ? ws.cf : nullptr;
            if ((rc = gemm(s, wg[b].A, 1, wg[b].a_sk, wg[b].Bm, wg[b].ldb, 1, S + wg[b].off, wg[b].ldc, wg[b].in + 1, wg[b].out, Bi, nullptr, 1.f, 0, o)))
                return rc;
            tile_splits[wg[b].blk] = left;
            if (b == n_wg - 1) tile_splits[wg[b].blk + 1] = left;
        }
        const bool last = bucket == n_buckets - 1;
        if (group && out_fused && plan.n > 0 && last) {
            plan.sum_in = ws.px_loss; plan.sum_out = S + N.P; plan.sum_n = B;
            loss_summed = true;
        }
        if (group && (rc = gemm_group_launch(s, plan))) return rc;
        if (!w_splits) {
            const size_t c0 = (n_buckets == 2 && bucket == 1) ? bk->split : 0, c1 = (n_buckets == 2 && bucket == 0) ? bk->split : (size_t)N.P;
            hipLaunchKernelGGL(k_vae_tile_sums, dim3(cdiv(c1 - c0, 256)), dim3(256), 0, s, vae_tiles(N, ws, tile_splits), S, N.P,
                               (last && out_fused && !loss_summed) ? (const float*)ws.px_loss : (const float*)nullptr, B, c0, c1);
            if (n_buckets == 2 && (rc = bk->ready(bk->ctx, bucket))) return rc;
        }
    }
    if (loss_pending) *loss_pending = out_fused && !loss_summed;
    return check_launch("d3p_vae sums");
}

}  // namespace d3p

using namespace d3p;

extern "C" {

// C = alpha op(A) op(B) (+ bias) (+ C) on the matrix cores; exported so the GEMM can be tested on its own.
int d3p_gemm_f32(void* stream, const float* A_dev, int64_t a_sm, int64_t a_sk, const float* B_dev, int64_t b_sk, int64_t b_sn,
                 float* C_dev, int32_t ldc, int32_t M, int32_t N, int32_t K, const float* bias_dev, float alpha, int32_t accumulate)
{
    D3P_REQUIRE(A_dev && B_dev && C_dev, "d3p_gemm_f32: null pointer");
    D3P_REQUIRE(M >= 1 && N >= 1 && K >= 1 && ldc >= N, "d3p_gemm_f32: bad shape");
    return gemm((hipStream_t)stream, A_dev, a_sm, a_sk, B_dev, b_sk, b_sn, C_dev, ldc, M, N, K, bias_dev, alpha, accumulate);
}

int64_t d3p_vae_num_params(const d3p_vae_model* model)
{
    if (!model || model->D < 1 || model->H < 1 || model->Z < 1 || model->H2 < 0) return 0;
    return (int64_t)vae_net(model).P;
}

size_t d3p_dpvi_vae_workspace(const d3p_vae_model* model, uint32_t B)
{
    if (!model || model->D < 1 || model->H < 1 || model->Z < 1 || model->H2 < 0) return 0;
    return vae_carve(model, B, nullptr, nullptr);
}

int d3p_vae_step_sums(void* stream, const d3p_vae_model* model, const float* params_dev, const float* X_dev, const uint8_t* mask_dev,
                      uint32_t B, const float* eps_dev, const uint32_t* jax_key_dev, float clip, float* sums_dev, float* norms_dev,
                      float* px_loss_dev, void* workspace_dev, size_t workspace_bytes)
{
    if (int rc = vae_validate(model, "d3p_vae_step_sums")) return rc;
    D3P_REQUIRE(params_dev && X_dev && sums_dev && workspace_dev && (eps_dev || jax_key_dev), "d3p_vae_step_sums: null pointer");
    D3P_REQUIRE(B >= 1 && clip > 0.f, "d3p_vae_step_sums: B >= 1 and clip > 0 required");
    if (workspace_bytes < d3p_dpvi_vae_workspace(model, B)) return fail(D3P_E_WORKSPACE, "d3p_vae_step_sums: workspace too small");
    VaeWorkspace ws;
    vae_carve(model, B, (char*)workspace_dev, &ws);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = vae_enqueue_sums(s, model, params_dev, X_dev, mask_dev, B, eps_dev, jax_key_dev, clip, ws, norms_dev)) return rc;
    const size_t P = vae_net(model).P;
    D3P_HIP_TRY(hipMemcpyAsync(sums_dev, ws.sums, (P + 2) * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (px_loss_dev) D3P_HIP_TRY(hipMemcpyAsync(px_loss_dev, ws.px_loss, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, s));
    return D3P_OK;
}

int d3p_vae_evaluate(void* stream, const d3p_vae_model* model, const float* params_dev, const float* X_dev, uint32_t B,
                     const uint32_t* jax_key_dev, const float* eps_dev, float* loss_dev, void* workspace_dev, size_t workspace_bytes)
{
    if (int rc = vae_validate(model, "d3p_vae_evaluate")) return rc;
    D3P_REQUIRE(params_dev && X_dev && loss_dev && workspace_dev && (jax_key_dev || eps_dev), "d3p_vae_evaluate: null pointer");
    D3P_REQUIRE(B >= 1, "d3p_vae_evaluate: B must be >= 1");
    if (workspace_bytes < d3p_dpvi_vae_workspace(model, B)) return fail(D3P_E_WORKSPACE, "d3p_vae_evaluate: workspace too small");
    VaeWorkspace ws;
    vae_carve(model, B, (char*)workspace_dev, &ws);
    hipStream_t s = (hipStream_t)stream;
    const float* eps = eps_dev;
    if (!eps) {
        hipLaunchKernelGGL(k_vae_eval_eps, dim3(cdiv((uint64_t)B * model->Z, 256)), dim3(256), 0, s, jax_key_dev, B, model->Z, ws.eps);
        eps = ws.eps;
    }
    if (int rc = vae_enqueue_forward(s, model, params_dev, X_dev, nullptr, B, eps, model->inv_obs * model->scale, ws)) return rc;
    const size_t P = vae_net(model).P;
    hipLaunchKernelGGL(k_vae_loss_n, dim3(1), dim3(256), 0, s, (const float*)ws.px_loss, (const uint8_t*)nullptr, B, ws.sums + P);
    D3P_HIP_TRY(hipMemcpyAsync(loss_dev, ws.sums + P, sizeof(float), hipMemcpyDeviceToDevice, s));
    return check_launch("d3p_vae_evaluate");
}

static int vae_update_checks(const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                             const void* workspace_dev, const char* what)
{
    if (int rc = vae_validate(model, what)) return rc;
    if (!(hyper && state && state->rng_key && state->params && state->adam_m && state->adam_v && state->step && workspace_dev))
        return fail(D3P_E_INVALID_ARG, "%s: null pointer", what);
    if (!(hyper->clip > 0.f) || !std::isfinite(hyper->clip))
        return fail(D3P_E_INVALID_ARG, "%s: the clipping threshold must be finite and greater than 0", what);
    return D3P_OK;
}

// keys of one update, all functions of the state key: [next | gradient | perturbation] = split(key, 3) (svi.py:208-211),
// split(perturbation_key, 10) (svi.py:491), convert_to_jax_rng_key(gradient_key)
// from != nullptr: the update reads the key and the step counter of `from` and writes the next key / counter into `state`
// (slot 1 of its key buffer; state->key_slot is taken as 0)
// X != nullptr (with prep): the exactness pass over the batch the forward pass is about to take, and the packing of the latent
// heads of `params` (the parameters that pass runs on), ride in this launch; *prep says which of them did
static int vae_step_keys(hipStream_t s, const d3p_vae_model* model, const d3p_dpsvi_state* state, const VaeWorkspace& ws, bool advance,
                         const d3p_dpsvi_state* from = nullptr, const float* X = nullptr, uint32_t B = 0, const float* params = nullptr,
                         VaeStepPrep* prep = nullptr, const VaeRunStep* rs = nullptr)
{
    // rs != nullptr (the run loop): X is the batch buffer the step's rows are ABOUT to be gathered into -- this launch makes the
    // indices, the caller gathers (and checks exactness) behind it
    const VaeNet N = vae_net(model);
    const int slot = state->key_slot & 1, n_sites = N.n_leaves();
    KeysExtra ex;
    memset(&ex, 0, sizeof(ex));
    unsigned pack_blocks = 0, feistel_blocks = 0;
    if (rs) {
        int bits = 0;   // bit_length(n_rows - 1), util.py:230
        for (uint32_t v = rs->n_rows - 1; v; v >>= 1) ++bits;
        ex.batch_key = rs->batch_key; ex.batch_i = rs->batch_i; ex.capacity = rs->n_rows; ex.n_idx = B;
        ex.bits_lower = bits >> 1; ex.bits_upper = bits - (bits >> 1);
        ex.idx = rs->idx;
        feistel_blocks = cdiv(B, 256);
    } else if (X && prep && vae_exact_eligible(X, B, model->D)) {
        ex.x = X;
        ex.x_n4 = (size_t)B * model->D / 4;
        ex.x_flag = ws.x_exact16;
        ex.x_nonce = vae_exact_nonce(true);
        ex.x_blocks = vae_exact_blocks(ex.x_n4);
        prep->x_checked = true;
    }
    static const bool no_pack = getenv("D3P_VAE_NO_HEAD_PACK") != nullptr;   // developer switch (A/B), read once
    if (params && prep && !no_pack && (2 * N.Z) % 4 == 0 && N.HE % 4 == 0) {
        ex.wl = params + N.Wl; ex.wsd = params + N.Ws;
        ex.wcat = ws.wcat; ex.wcatT = ws.wcatT;
        ex.HE = N.HE; ex.Z = N.Z;
        pack_blocks = cdiv((uint64_t)N.HE * 2 * N.Z, 256);
        prep->heads_packed = true;
    }
    ex.pack_blocks = pack_blocks;
    const dim3 grid(1 + ex.x_blocks + pack_blocks + feistel_blocks);
    if (from)
        hipLaunchKernelGGL(k_vae_keys, grid, dim3(256), 0, s, (const uint32_t*)(from->rng_key + 16 * (from->key_slot & 1)), ws.keys,
                           state->rng_key + 16, (const int32_t*)from->step, state->step, advance ? 1 : 0, n_sites, ex);
    else
        hipLaunchKernelGGL(k_vae_keys, grid, dim3(256), 0, s, (const uint32_t*)(state->rng_key + 16 * slot), ws.keys,
                           state->rng_key + 16 * (slot ^ 1), (const int32_t*)state->step, state->step, advance ? 1 : 0, n_sites, ex);
    return check_launch("k_vae_keys");
}

// the Gaussian-mechanism noise of an update: one key per leaf, normal(site_key, leaf shape) (svi.py:487)
static SiteNoiseArgs vae_site_noise_args(const VaeNet& N, const VaeWorkspace& ws)
{
    size_t leaf_off[D3P_VAE_MAX_LEAVES + 1];
    int k = 0;
    for (int l = 0; l <= N.nh; ++l) { leaf_off[k++] = N.dec[l].W; leaf_off[k++] = N.dec[l].b; }
    for (int l = 0; l < N.nh; ++l) { leaf_off[k++] = N.enc[l].W; leaf_off[k++] = N.enc[l].b; }
    leaf_off[k++] = N.Wl; leaf_off[k++] = N.bl; leaf_off[k++] = N.Ws; leaf_off[k++] = N.bs;
    for (; k <= D3P_VAE_MAX_LEAVES; ++k) leaf_off[k] = N.P;
    SiteNoiseArgs na;
    na.site_keys = ws.keys + 48;
    na.noise = ws.noise;
    na.blk_off[0] = 0;
    for (k = 0; k < D3P_VAE_MAX_LEAVES; ++k) {
        na.elem_off[k] = (uint32_t)leaf_off[k];
        na.blk_off[k + 1] = na.blk_off[k] + (uint32_t)((leaf_off[k + 1] - leaf_off[k] + 15) / 16);
    }
    na.elem_off[D3P_VAE_MAX_LEAVES] = (uint32_t)leaf_off[D3P_VAE_MAX_LEAVES];
    return na;
}

static int vae_apply_impl(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                          const float* sums_dev, uint32_t B_total, uint32_t B_local, float* loss_dev, float* grad_out_dev,
                          void* workspace_dev, size_t workspace_bytes, bool derive_keys, const int* w_splits = nullptr,
                          const d3p_dpsvi_state* from = nullptr, bool loss_pending = false, bool noise_drawn = false);

// advance = true (single-device update): the key kernel also writes the next state key and advances the step counter, so that
// vae_apply_impl(derive_keys = false) has nothing left to launch for them
static int vae_local_sums_impl(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                               const float* X_dev, const uint8_t* mask_dev, uint32_t B_local, uint32_t B_total, uint32_t pos0,
                               const float* eps_dev, float* sums_dev, void* workspace_dev, size_t workspace_bytes, bool advance,
                               int* w_splits = nullptr, const d3p_dpsvi_state* from = nullptr, const VaeRunStep* rs = nullptr,
                               bool* loss_pending = nullptr, bool draw_noise = false, const VaeBuckets* bk = nullptr)
{
    // draw_noise (with advance; the native data-parallel loop): the Gaussian-mechanism noise of the update is drawn beside the latent
    // kernel although the sums leave the device function for a reduce -- it depends on the step's keys only
    // rs != nullptr: X_dev is the batch BUFFER; the step's rows are gathered into it here (indices from the key launch)
    // loss_pending (with w_splits): vae_enqueue_sums' -- the caller hands it on to vae_apply_impl
    if (int rc = vae_update_checks(model, hyper, state, workspace_dev, "d3p_dpvi_vae_local_sums")) return rc;
    D3P_REQUIRE(X_dev && sums_dev, "d3p_dpvi_vae_local_sums: null pointer");
    D3P_REQUIRE(B_local >= 1 && (uint64_t)pos0 + B_local <= B_total, "d3p_dpvi_vae_local_sums: need 1 <= B_local and pos0 + B_local <= B_total");
    if (workspace_bytes < d3p_dpvi_vae_workspace(model, B_local)) return fail(D3P_E_WORKSPACE, "d3p_dpvi_vae_local_sums: workspace too small");
    VaeWorkspace ws;
    vae_carve(model, B_local, (char*)workspace_dev, &ws);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    VaeStepPrep prep;
    if ((rc = vae_step_keys(s, model, state, ws, advance, from, X_dev, B_local, from ? from->params : state->params, &prep, rs))) return rc;
    if (rs) {   // gather + exactness pass in one sweep (d3p_take_rows and k_exact16_flag's work)
        const bool chk = vae_exact_eligible(X_dev, B_local, model->D);
        const uint32_t d4 = (uint32_t)model->D / 4;
        const size_t n = (size_t)B_local * d4;
        hipLaunchKernelGGL(k_vae_gather_check, dim3((unsigned)(n / 512 < 1 ? 1 : (n / 512 > 2048 ? 2048 : n / 512))), dim3(256), 0, s, rs->table,
                           (const uint32_t*)rs->idx, B_local, d4, const_cast<float*>(X_dev), chk ? ws.x_exact16 : (uint32_t*)nullptr,
                           chk ? vae_exact_nonce(true) : 0u);
        prep.x_checked = chk;
    }
    // w_splits != nullptr = the single-device update: vae_apply_impl follows on the same workspace with these keys, so its noise is
    // drawn here, beside the latent kernel
    const SiteNoiseArgs noise = vae_site_noise_args(vae_net(model), ws);
    if ((rc = vae_enqueue_sums(s, model, from ? from->params : state->params, X_dev, mask_dev, B_local, eps_dev, ws.keys + D3P_VAE_KEY_JAX, hyper->clip, ws, nullptr,
                               B_total, pos0, w_splits, prep, (w_splits || draw_noise) ? &noise : nullptr, w_splits ? loss_pending : nullptr, bk)))
        return rc;
    if (sums_dev != ws.sums)
        D3P_HIP_TRY(hipMemcpyAsync(sums_dev, ws.sums, (vae_net(model).P + 2) * sizeof(float), hipMemcpyDeviceToDevice, s));
    return check_launch("d3p_dpvi_vae_local_sums");
}

int d3p_dpvi_vae_local_sums(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                            const float* X_dev, const uint8_t* mask_dev, uint32_t B_local, uint32_t B_total, uint32_t pos0,
                            const float* eps_dev, float* sums_dev, void* workspace_dev, size_t workspace_bytes)
{
    return vae_local_sums_impl(stream, model, hyper, state, X_dev, mask_dev, B_local, B_total, pos0, eps_dev, sums_dev, workspace_dev,
                               workspace_bytes, false);
}

int d3p_dpvi_vae_apply(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                       const float* sums_dev, uint32_t B_total, uint32_t B_local, float* loss_dev, float* grad_out_dev,
                       void* workspace_dev, size_t workspace_bytes)
{
    return vae_apply_impl(stream, model, hyper, state, sums_dev, B_total, B_local, loss_dev, grad_out_dev, workspace_dev,
                          workspace_bytes, true);
}

// the arguments of the update of one step (k_vae_finalize, k_vae_fmesh_step)
static VaeFinalArgs vae_final_args(const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state, const VaeWorkspace& ws,
                                   const float* sums_dev, uint32_t B_total, uint32_t B_local, float* loss_dev, float* grad_out_dev, const int* w_splits,
                                   const d3p_dpsvi_state* from, bool loss_pending)
{
    const VaeNet N = vae_net(model);
    VaeFinalArgs f;
    memset(&f, 0, sizeof(f));
    f.sums = sums_dev;
    f.tiles = vae_tiles(N, ws, w_splits);
    f.noise = ws.noise;
    f.in_params = from ? from->params : state->params;
    f.in_m = from ? from->adam_m : state->adam_m;
    f.in_v = from ? from->adam_v : state->adam_v;
    f.params = state->params;
    f.adam_m = state->adam_m;
    f.adam_v = state->adam_v;
    f.step = reinterpret_cast<const int32_t*>(ws.keys + D3P_VAE_KEY_STEP);  // the step index k_vae_keys saved before advancing it
    f.loss_out = loss_dev;
    f.grad_out = grad_out_dev;
    f.P = N.P;
    f.B = B_total;
    f.h = *hyper;
    f.obs_scale = 1.0f / model->inv_obs;
    f.px_loss = loss_pending ? ws.px_loss : nullptr;
    f.B_local = B_local;
    return f;
}

// derive_keys = false: the keys of this update are already in the workspace (left there by d3p_dpvi_vae_local_sums on the same
// workspace and state, as in the single-device update)
static int vae_apply_impl(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                          const float* sums_dev, uint32_t B_total, uint32_t B_local, float* loss_dev, float* grad_out_dev,
                          void* workspace_dev, size_t workspace_bytes, bool derive_keys, const int* w_splits, const d3p_dpsvi_state* from,
                          bool loss_pending, bool noise_drawn)
{
    // loss_pending (single-device update): sums_dev[P] does not hold the loss sum -- k_vae_finalize takes it from ws.px_loss
    if (int rc = vae_update_checks(model, hyper, state, workspace_dev, "d3p_dpvi_vae_apply")) return rc;
    D3P_REQUIRE(sums_dev && B_total >= 1 && B_local >= 1, "d3p_dpvi_vae_apply: null pointer or empty batch");
    if (workspace_bytes < d3p_dpvi_vae_workspace(model, B_local)) return fail(D3P_E_WORKSPACE, "d3p_dpvi_vae_apply: workspace too small");
    VaeWorkspace ws;
    vae_carve(model, B_local, (char*)workspace_dev, &ws);
    hipStream_t s = (hipStream_t)stream;
    const VaeNet N = vae_net(model);
    int rc;
    if (derive_keys && (rc = vae_step_keys(s, model, state, ws, true))) return rc;
    if (derive_keys || (!w_splits && !noise_drawn)) {  // (the single-device update and the native data-parallel loop drew the noise beside their latent kernel: vae_local_sums_impl)
        const SiteNoiseArgs na = vae_site_noise_args(N, ws);
        hipLaunchKernelGGL(k_vae_site_noise, dim3(cdiv(na.blk_off[D3P_VAE_MAX_LEAVES], 256)), dim3(256), 0, s, na);
    }
    const VaeFinalArgs f = vae_final_args(model, hyper, state, ws, sums_dev, B_total, B_local, loss_dev, grad_out_dev, w_splits, from, loss_pending);
    hipLaunchKernelGGL(k_vae_finalize, dim3(cdiv(N.P, 256)), dim3(256), 0, s, f);
    return check_launch("d3p_dpvi_vae_apply");
}

// d3p_dpvi_vae_update as a function of an immutable state: reads `from` (key slot from->key_slot, parameters, Adam moments, step
// counter), writes the new state into `state` (its arrays may be uninitialised; the next key goes to slot 1 of state->rng_key)
// -- no copy of the 3 x 689 k floats of state per update.
int d3p_dpvi_vae_update_from(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                             const d3p_dpsvi_state* from, const float* X_dev, const uint8_t* mask_dev, uint32_t B, const float* eps_dev,
                             float* loss_dev, float* grad_out_dev, void* workspace_dev, size_t workspace_bytes)
{
    if (int rc = vae_update_checks(model, hyper, state, workspace_dev, "d3p_dpvi_vae_update_from")) return rc;
    D3P_REQUIRE(from && from->rng_key && from->params && from->adam_m && from->adam_v && from->step, "d3p_dpvi_vae_update_from: null source state");
    D3P_REQUIRE(X_dev, "d3p_dpvi_vae_update_from: null pointer");
    D3P_REQUIRE(B >= 1, "d3p_dpvi_vae_update_from: B must be >= 1");
    D3P_REQUIRE(state->key_slot == 0, "d3p_dpvi_vae_update_from: state->key_slot must be 0");
    if (workspace_bytes < d3p_dpvi_vae_workspace(model, B)) return fail(D3P_E_WORKSPACE, "d3p_dpvi_vae_update_from: workspace too small");
    VaeWorkspace ws;
    vae_carve(model, B, (char*)workspace_dev, &ws);
    int w_splits[D3P_VAE_MAX_BLOCKS];
    bool loss_pending = false;
    if (int rc = vae_local_sums_impl(stream, model, hyper, state, X_dev, mask_dev, B, B, 0, eps_dev, ws.sums, workspace_dev,
                                     workspace_bytes, true, w_splits, from, nullptr, &loss_pending))
        return rc;
    return vae_apply_impl(stream, model, hyper, state, ws.sums, B, B, loss_dev, grad_out_dev, workspace_dev, workspace_bytes, false,
                          w_splits, from, loss_pending);
}

static int vae_update_impl(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                           const float* X_dev, const uint8_t* mask_dev, uint32_t B, const float* eps_dev, float* loss_dev,
                           float* grad_out_dev, void* workspace_dev, size_t workspace_bytes, const VaeRunStep* rs)
{
    // the single-device update IS the data-parallel one with one rank: local sums, (no reduce), apply
    if (int rc = vae_update_checks(model, hyper, state, workspace_dev, "d3p_dpvi_vae_update")) return rc;
    D3P_REQUIRE(X_dev, "d3p_dpvi_vae_update: null pointer");
    D3P_REQUIRE(B >= 1, "d3p_dpvi_vae_update: B must be >= 1");
    if (workspace_bytes < d3p_dpvi_vae_workspace(model, B)) return fail(D3P_E_WORKSPACE, "d3p_dpvi_vae_update: workspace too small");
    VaeWorkspace ws;
    vae_carve(model, B, (char*)workspace_dev, &ws);
    int w_splits[D3P_VAE_MAX_BLOCKS];  // split-K partial tiles of the weight gradients are summed by k_vae_finalize, not by reduction launches
    bool loss_pending = false;
    if (int rc = vae_local_sums_impl(stream, model, hyper, state, X_dev, mask_dev, B, B, 0, eps_dev, ws.sums, workspace_dev,
                                     workspace_bytes, true, w_splits, nullptr, rs, &loss_pending))
        return rc;
    return vae_apply_impl(stream, model, hyper, state, ws.sums, B, B, loss_dev, grad_out_dev, workspace_dev, workspace_bytes, false,
                          w_splits, nullptr, loss_pending);
}

int d3p_dpvi_vae_update(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                        const float* X_dev, const uint8_t* mask_dev, uint32_t B, const float* eps_dev, float* loss_dev,
                        float* grad_out_dev, void* workspace_dev, size_t workspace_bytes)
{
    return vae_update_impl(stream, model, hyper, state, X_dev, mask_dev, B, eps_dev, loss_dev, grad_out_dev, workspace_dev, workspace_bytes,
                           nullptr);
}

// num_steps x (get_batch(first_batch + t, batch_key) of subsample_batchify_data -> update) on the resident data set X_dev
// (n_rows x D): the body of the example's jit(fori_loop(...)) epoch (examples/vae.py:227-246) -- per step fold_in, the Feistel
// indices, the row gather into xb_dev (B x D) and the update, all enqueued back to back: no host work depends on a result.
// `state` is advanced in place (its key_slot field says which slot of state->rng_key holds the key BEFORE the run; afterwards
// the key is in slot (key_slot + num_steps) & 1).  idx_dev: B uint32 + 16 (the step's batch key).
int d3p_dpvi_vae_run(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                     const uint32_t* batch_key_dev, uint32_t first_batch, const float* X_dev, uint32_t n_rows, uint32_t B,
                     uint32_t num_steps, float* losses_dev, float* xb_dev, uint32_t* idx_dev, void* workspace_dev, size_t workspace_bytes)
{
    if (int rc = vae_update_checks(model, hyper, state, workspace_dev, "d3p_dpvi_vae_run")) return rc;
    D3P_REQUIRE(batch_key_dev && X_dev && xb_dev && idx_dev, "d3p_dpvi_vae_run: null pointer");
    D3P_REQUIRE(B >= 1 && B <= n_rows, "d3p_dpvi_vae_run: need 1 <= B <= n_rows");
    if (workspace_bytes < d3p_dpvi_vae_workspace(model, B)) return fail(D3P_E_WORKSPACE, "d3p_dpvi_vae_run: workspace too small");
    d3p_dpsvi_state st = *state;
    uint32_t* step_key = idx_dev + B;
    // 16-byte rows: the step's indices come out of its key launch and ONE sweep gathers the rows and checks their exactness (two
    // launches where fold_in, the sampler, the gather and the key launch were four: 264 -> 254 us per step at B = 4096)
    static const bool no_fuse = getenv("D3P_VAE_RUN_UNFUSED") != nullptr;   // developer switch (A/B), read once
    const bool fused = !no_fuse && model->D % 4 == 0 && ((reinterpret_cast<uintptr_t>(X_dev) | reinterpret_cast<uintptr_t>(xb_dev)) & 15u) == 0;
    for (uint32_t t = 0; t < num_steps; ++t) {
        int rc;
        if (fused) {
            const VaeRunStep rs = {batch_key_dev, first_batch + t, X_dev, n_rows, idx_dev};
            if ((rc = vae_update_impl(stream, model, hyper, &st, xb_dev, nullptr, B, nullptr, losses_dev ? losses_dev + t : nullptr, nullptr,
                                      workspace_dev, workspace_bytes, &rs)))
                return rc;
        } else {
            if ((rc = d3p_rng_fold_in(stream, batch_key_dev, first_batch + t, step_key))) return rc;          // minibatch.py:230
            if ((rc = d3p_feistel_sample(stream, step_key, n_rows, B, idx_dev))) return rc;                   // minibatch.py:231
            if ((rc = d3p_take_rows(stream, X_dev, n_rows, (uint32_t)(model->D * sizeof(float)), idx_dev, B, nullptr, xb_dev))) return rc;
            if ((rc = d3p_dpvi_vae_update(stream, model, hyper, &st, xb_dev, nullptr, B, nullptr, losses_dev ? losses_dev + t : nullptr,
                                          nullptr, workspace_dev, workspace_bytes)))
                return rc;
        }
        st.key_slot ^= 1;
    }
    return D3P_OK;
}

// ---- the data-parallel epoch body as ONE call (examples/vae.py:227-246 with the batch sharded by position)
namespace {
struct VaeDistSide {   // per (device, caller's stream): the stream the reduces travel on and the events that order them against the step's stream
    hipStream_t cs = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr}, done = nullptr;
};
static int vae_dist_side(hipStream_t caller, VaeDistSide** out)
{
    // keyed by the caller's stream too: in-process ranks (one stream each) and host threads must not share a side stream or re-record
    // each other's events -- two ranks of one communicator on ONE side stream would wait for each other's reduce forever
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, VaeDistSide> sides;
    int dev = 0;
    D3P_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    VaeDistSide& v = sides[std::make_pair(dev, caller)];
    if (!v.cs) {
        D3P_HIP_TRY(hipStreamCreateWithFlags(&v.cs, hipStreamNonBlocking));
        for (hipEvent_t* e : {&v.ready[0], &v.ready[1], &v.done}) D3P_HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    *out = &v;
    return D3P_OK;
}
struct VaeDistStep {
    hipStream_t s;
    VaeDistSide* side;
    void* comm;
    float* sums;
    size_t split, total;   // bucket 0 = [0, split), bucket 1 = [split, total)
};
static int vae_dist_bucket_ready(void* ctx, int bucket)
{
    // bucket 0: its reduce starts on the side stream as soon as its tile sums are complete -- the second bucket's products run beside it;
    // bucket 1: behind the first on the same stream (a communicator runs one collective at a time anyway)
    VaeDistStep* d = (VaeDistStep*)ctx;
    D3P_HIP_TRY(hipEventRecord(d->side->ready[bucket], d->s));
    D3P_HIP_TRY(hipStreamWaitEvent(d->side->cs, d->side->ready[bucket], 0));
    const size_t lo = bucket == 0 ? 0 : d->split, hi = bucket == 0 ? d->split : d->total;
    if (int rc = rccl_allreduce_f32(d->comm, d->sums + lo, hi - lo, d->side->cs)) return rc;
    if (bucket == 1) {
        D3P_HIP_TRY(hipEventRecord(d->side->done, d->side->cs));
        D3P_HIP_TRY(hipStreamWaitEvent(d->s, d->side->done, 0));
    }
    return D3P_OK;
}
}  // namespace

// num_steps x [local sums of the rank's B_local examples (positions pos0 .. of the global batch of B_total) -> sum-all-reduce of the
// P + 2 fp32 sums over `comm` -> noise once + Adam, identical on every rank] on the RESIDENT shard X_local_dev, enqueued back to
// back: no host work between steps, no per-step state or key copies (the state advances in place, its key ping-pongs between the
// two slots of state->rng_key like d3p_dpvi_vae_run), the step's keys are derived once, the Gaussian-mechanism noise is drawn
// beside the latent kernel (it depends on the keys only), and apply is ONE launch behind the reduce.
// buckets = 2: the sums travel in two buckets on a second stream -- the decoder's leaves while the encoder's weight-gradient
// products still run (VaeBuckets); 1: one all-reduce on `stream` itself; 0: the library's choice (1: see below).
// comm = NULL: no collective (one rank; B_local == B_total).
int d3p_dpvi_vae_run_dist(void* stream, void* comm, void* fmesh, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                          const float* X_local_dev, const uint8_t* mask_dev, uint32_t B_local, uint32_t B_total, uint32_t pos0,
                          uint32_t num_steps, float* losses_dev, int32_t buckets, void* workspace_dev, size_t workspace_bytes)
{
    // fmesh != NULL: the step's collective is the full-mesh reduce-scatter + all-gather of d3p_fmesh.hip (one launch in the stream)
    // instead of RCCL's all-reduce
    D3P_REQUIRE(!(comm && fmesh), "d3p_dpvi_vae_run_dist: one collective, RCCL (comm) or the full mesh (fmesh), not both");
    if (int rc = vae_update_checks(model, hyper, state, workspace_dev, "d3p_dpvi_vae_run_dist")) return rc;
    D3P_REQUIRE(X_local_dev, "d3p_dpvi_vae_run_dist: null pointer");
    D3P_REQUIRE(B_local >= 1 && (uint64_t)pos0 + B_local <= B_total, "d3p_dpvi_vae_run_dist: need 1 <= B_local and pos0 + B_local <= B_total");
    D3P_REQUIRE(comm || fmesh || B_local == B_total, "d3p_dpvi_vae_run_dist: a shard of the batch needs a communicator");
    D3P_REQUIRE(buckets >= 0 && buckets <= 2, "d3p_dpvi_vae_run_dist: buckets must be 0, 1 or 2");
    if (workspace_bytes < d3p_dpvi_vae_workspace(model, B_local)) return fail(D3P_E_WORKSPACE, "d3p_dpvi_vae_run_dist: workspace too small");
    VaeWorkspace ws;
    vae_carve(model, B_local, (char*)workspace_dev, &ws);
    const VaeNet N = vae_net(model);
    hipStream_t s = (hipStream_t)stream;
    static const int env_buckets = [] { const char* e = getenv("D3P_VAE_DP_BUCKETS"); return e ? atoi(e) : 0; }();   // developer switch (A/B), read once
    // (the library's choice is ONE bucket: on one GPU the two-bucket form costs 40 us per step more -- two event hand-overs between the
    // streams, one more grouped launch and tile-sum launch -- than the 30 us of products it can put beside a reduce;
    // profiles/r05_vae_dp_loop_rank_local.jsonl.  bench.py --gpus N times both over real links.)
    if (buckets == 0 && !fmesh) buckets = (env_buckets == 1 || env_buckets == 2) ? env_buckets : 1;
    VaeDistStep step = {s, nullptr, comm, ws.sums, N.enc[0].W, (size_t)N.P + 2};
    const bool two = comm && buckets == 2 && !vae_no_group() && step.split > 0 && step.split < (size_t)N.P;
    if (two)
        if (int rc = vae_dist_side(s, &step.side)) return rc;
    const VaeBuckets bk = {step.split, vae_dist_bucket_ready, &step};
    d3p_dpsvi_state st = *state;
    // full mesh: the tile sums, the collective and the update as ONE launch (k_vae_fmesh_step); D3P_VAE_FMESH_UNFUSED=1 keeps them apart
    static const bool fm_unfused = getenv("D3P_VAE_FMESH_UNFUSED") != nullptr;   // developer switch (A/B), read once
    for (uint32_t t = 0; t < num_steps; ++t) {
        int rc;
        if (fmesh && !fm_unfused && buckets != 1) {   // (buckets = 1 with a mesh: tile sums, collective and update as three launches)
            FMesh* x = (FMesh*)fmesh;
            D3P_REQUIRE(x->n == (uint64_t)N.P + 2, "d3p_dpvi_vae_run_dist: the mesh was created for another vector length (P + 2)");
            for (int p = 0; p < x->world; ++p) D3P_REQUIRE(x->peer[p], "d3p_dpvi_vae_run_dist: the peers' inboxes are not mapped (d3p_fmesh_connect)");
            int w_splits[D3P_VAE_MAX_BLOCKS];
            bool loss_pending = false;
            // (the single-device update's local phase: the split-K partial tiles stay where they are, the noise is drawn beside the latent kernel)
            if ((rc = vae_local_sums_impl(stream, model, hyper, &st, X_local_dev, mask_dev, B_local, B_total, pos0, nullptr, ws.sums, workspace_dev,
                                          workspace_bytes, true, w_splits, nullptr, nullptr, &loss_pending)))
                return rc;
            if (loss_pending)   // (the loss sum is still per example -- the output layer's epilogue without the grouped launch: one more launch sums it)
                hipLaunchKernelGGL(k_vae_loss_n, dim3(1), dim3(256), 0, s, (const float*)ws.px_loss, mask_dev, B_local, ws.sums + N.P);
            const VaeFinalArgs f = vae_final_args(model, hyper, &st, ws, ws.sums, B_total, B_local, losses_dev ? losses_dev + t : nullptr, nullptr, w_splits,
                                                  nullptr, false);
            FMeshArgs ma;
            fmesh_next_args(x, ws.sums, &ma);
            hipLaunchKernelGGL(k_vae_fmesh_step, dim3((unsigned)x->wgs), dim3(256), 0, s, ma, f);
            if ((rc = check_launch("k_vae_fmesh_step"))) return rc;
            st.key_slot ^= 1;
            continue;
        }
        if ((rc = vae_local_sums_impl(stream, model, hyper, &st, X_local_dev, mask_dev, B_local, B_total, pos0, nullptr, ws.sums, workspace_dev,
                                      workspace_bytes, true, nullptr, nullptr, nullptr, nullptr, true, two ? &bk : nullptr)))
            return rc;
        if (comm && !two && (rc = rccl_allreduce_f32(comm, ws.sums, (size_t)N.P + 2, s))) return rc;
        if (fmesh && (rc = fmesh_enqueue_allreduce(s, fmesh, ws.sums, (uint64_t)N.P + 2))) return rc;
        if ((rc = vae_apply_impl(stream, model, hyper, &st, ws.sums, B_total, B_local, losses_dev ? losses_dev + t : nullptr, nullptr, workspace_dev,
                                 workspace_bytes, false, nullptr, nullptr, false, true)))
            return rc;
        st.key_slot ^= 1;
    }
    return D3P_OK;
}

}  // extern "C"
