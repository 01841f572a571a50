// Device-side primitives shared by the gfx950 kernels: ChaCha20 block, threefry2x32, the
// bits->uniform->normal transforms and wave64 reductions.  gfx950 only: wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define D3P_WAVE 64

#define D3P_TAG_SPLIT 0x00000001u
#define D3P_TAG_FOLD 0x00000002u

#define D3P_NORMAL_LO (-0.99999994f)  // np.nextafter(float32(-1), 0): d3p/random/__init__.py:78
#define D3P_SQRT2 1.41421354f         // float32(sqrt(2)):            d3p/random/__init__.py:81
#define D3P_HALF_LOG_2PI 0.918938533204672742f

namespace d3p {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r)
{
    // v_alignbit_b32: ({x,x} >> (32-r)) = rotate left by r
    return __builtin_amdgcn_alignbit(x, x, 32 - r);
}

#define D3P_QR(a, b, c, d)                \
    a += b; d ^= a; d = d3p::rotl32(d, 16); \
    c += d; b ^= c; b = d3p::rotl32(b, 12); \
    a += b; d ^= a; d = d3p::rotl32(d, 8);  \
    c += d; b ^= c; b = d3p::rotl32(b, 7);

// RFC 8439 2.3 block function; `in` and `out` are 16 registers each.
__device__ __forceinline__ void chacha20_block(const uint32_t (&in)[16], uint32_t (&out)[16])
{
    uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6],
             x7 = in[7], x8 = in[8], x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12],
             x13 = in[13], x14 = in[14], x15 = in[15];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        D3P_QR(x0, x4, x8, x12) D3P_QR(x1, x5, x9, x13) D3P_QR(x2, x6, x10, x14) D3P_QR(x3, x7, x11, x15)
        D3P_QR(x0, x5, x10, x15) D3P_QR(x1, x6, x11, x12) D3P_QR(x2, x7, x8, x13) D3P_QR(x3, x4, x9, x14)
    }
    out[0] = x0 + in[0]; out[1] = x1 + in[1]; out[2] = x2 + in[2]; out[3] = x3 + in[3];
    out[4] = x4 + in[4]; out[5] = x5 + in[5]; out[6] = x6 + in[6]; out[7] = x7 + in[7];
    out[8] = x8 + in[8]; out[9] = x9 + in[9]; out[10] = x10 + in[10]; out[11] = x11 + in[11];
    out[12] = x12 + in[12]; out[13] = x13 + in[13]; out[14] = x14 + in[14]; out[15] = x15 + in[15];
}

__device__ __forceinline__ void load_key(const uint32_t* __restrict__ key, uint32_t (&s)[16])
{
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = key[i];
}

// Block `blk` of the keystream of `key` (random_bits domain: nonce untouched).
__device__ __forceinline__ void keystream_block(const uint32_t (&key)[16], uint32_t blk, uint32_t (&out)[16])
{
    uint32_t in[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) in[i] = key[i];
    in[12] += blk;
    chacha20_block(in, out);
}

// ============================ KEY LAYOUT (this build's own -- PARITY UNPINNED, DESIGN.md sections 2-3) ============================
// jax-chacha-prng defines how split / fold_in derive a child key; it is not vendored and no reference test pins it.  What
// THIS build does is stated in exactly two functions -- layout_child_tweak (which words of the parent state are changed
// before the block function) and the "child = constants | block words 0..7 | zero counter and nonce" rule in derive_child /
// the quad forms below -- plus keystream_block above (random_bits: counter + block index, nonce untouched).  oracle/d3p_oracle.c
// has the same three places (derive_child, d3po_key_from_bytes, d3po_random_words).  Once tests/golden/capture_from_reference.py
// has been run against the real package, adopting its layout means editing these and nothing else.
__device__ __forceinline__ void layout_child_tweak(uint32_t& w12, uint32_t& w13, uint32_t& w15, uint32_t ctr_add, uint32_t data,
                                                   uint32_t tag)
{
    w12 += ctr_add;  // counter:  split child index
    w13 ^= data;     // nonce[0]: fold_in data
    w15 ^= tag;      // nonce[2]: D3P_TAG_SPLIT / D3P_TAG_FOLD -- keeps bits, split and fold_in in disjoint domains
}

// Child key derivation shared by split (tag 1, ctr_add = i) and fold_in (tag 2, data): the child
// key is words 0..7 of the parent's block at (counter + ctr_add, nonce ^ (data, 0, tag)); child
// counter/nonce are zero.
__device__ __forceinline__ void derive_child(const uint32_t (&parent)[16], uint32_t ctr_add, uint32_t data,
                                             uint32_t tag, uint32_t (&child)[16])
{
    uint32_t in[16], blk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) in[i] = parent[i];
    layout_child_tweak(in[12], in[13], in[15], ctr_add, data, tag);
    chacha20_block(in, blk);
#pragma unroll
    for (int i = 0; i < 4; ++i) child[i] = parent[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) child[4 + i] = blk[i];
    child[12] = child[13] = child[14] = child[15] = 0u;
}

// ---- 4-lane ChaCha20: lane q of a quad (lane & 3) holds column q of the 4x4 state as (a, b, c, d) =
// (x[q], x[4+q], x[8+q], x[12+q]).  Column rounds are lane-local; for the diagonal rounds b, c, d are
// rotated by 1, 2, 3 lanes inside the quad with DPP quad_perm and rotated back afterwards.  ~310
// instructions per block instead of ~970 for the one-lane form: used where a ChaCha block sits on a
// serial critical path (the key chain).  All four lanes of the quad must be active.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}

__device__ __forceinline__ void chacha20_block_quad(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d)
{
    const uint32_t a0 = a, b0 = b, c0 = c, d0 = d;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        D3P_QR(a, b, c, d)
        b = dpp_mov_u32<0x39>(b);  // quad_perm [1,2,3,0]: lane q takes b of lane q+1
        c = dpp_mov_u32<0x4E>(c);  // quad_perm [2,3,0,1]
        d = dpp_mov_u32<0x93>(d);  // quad_perm [3,0,1,2]
        D3P_QR(a, b, c, d)
        b = dpp_mov_u32<0x93>(b);
        c = dpp_mov_u32<0x4E>(c);
        d = dpp_mov_u32<0x39>(d);
    }
    a += a0;
    b += b0;
    c += c0;
    d += d0;
}

// split(parent, .)[child] in quad form: on return (a, b) of lane q are words q and 4+q of the child's 256-bit
// key, i.e. the child state's column q is (const[q], a, b, 0).
__device__ __forceinline__ void derive_child_quad(const uint32_t* __restrict__ parent, uint32_t child, uint32_t tag,
                                                  uint32_t data, uint32_t& a, uint32_t& b)
{
    const int q = threadIdx.x & 3;
    a = parent[q];
    b = parent[4 + q];
    uint32_t c = parent[8 + q], d = parent[12 + q];
    {  // lane q holds word 12 + q of the state: apply the layout rule to the word this lane owns
        uint32_t w12 = d, w13 = d, w15 = d;
        layout_child_tweak(w12, w13, w15, child, data, tag);
        d = q == 0 ? w12 : q == 1 ? w13 : q == 3 ? w15 : d;
    }
    chacha20_block_quad(a, b, c, d);
}

// Block `blk` of the keystream of `key` in quad form (keystream_block above): on return lane q of the quad holds words
// q, 4 + q, 8 + q, 12 + q of the block in (a, b, c, d).  `key` may live in global memory or LDS.
__device__ __forceinline__ void keystream_block_quad(const uint32_t* __restrict__ key, uint32_t blk, uint32_t& a, uint32_t& b, uint32_t& c,
                                                     uint32_t& d)
{
    const int q = threadIdx.x & 3;
    a = key[q];
    b = key[4 + q];
    c = key[8 + q];
    d = key[12 + q] + (q == 0 ? blk : 0u);
    chacha20_block_quad(a, b, c, d);
}

// the same with the lane's four parent words (rows q, 4+q, 8+q, 12+q of the parent state) already in registers
__device__ __forceinline__ void derive_child_quad_regs(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t child,
                                                       uint32_t tag, uint32_t data, uint32_t& a, uint32_t& b)
{
    const int q = threadIdx.x & 3;
    a = p0;
    b = p1;
    uint32_t c = p2, d = p3;
    {
        uint32_t w12 = d, w13 = d, w15 = d;
        layout_child_tweak(w12, w13, w15, child, data, tag);
        d = q == 0 ? w12 : q == 1 ? w13 : q == 3 ? w15 : d;
    }
    chacha20_block_quad(a, b, c, d);
}

// threefry2x32-20 (Random123), as used by jax.random.
// D3P_TF_PIN: x0 only ever grows by additions (x0 += x1 in every round, + a key word every fourth), and LLVM's
// reassociation pass likes to keep it as "sum of the x1's" + "sum of the key words and counters" and to rebuild x0 with two
// v_add3_u32 in EVERY round for the xor -- 5 VALU instructions per round instead of 3 (measured on gfx950, ROCm 7.2: 109
// instead of 75 integer instructions per call; integer VALU instructions issue at 4 cycles per wave64).  An empty asm that
// "modifies" x0 after each key injection makes it one opaque value again; it emits no instruction.
#define D3P_TF_PIN(x) asm volatile("" : "+v"(x), "+v"(x1))
__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0,
                                             uint32_t& o1)
{
    const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
    uint32_t x0 = c0 + k0, x1 = c1 + k1;
    D3P_TF_PIN(x0);
#define D3P_TF_R(r) x0 += x1; x1 = d3p::rotl32(x1, r); x1 ^= x0;
    D3P_TF_R(13) D3P_TF_R(15) D3P_TF_R(26) D3P_TF_R(6)
    x0 += k1; x1 += k2 + 1u;
    D3P_TF_PIN(x0);
    D3P_TF_R(17) D3P_TF_R(29) D3P_TF_R(16) D3P_TF_R(24)
    x0 += k2; x1 += k0 + 2u;
    D3P_TF_PIN(x0);
    D3P_TF_R(13) D3P_TF_R(15) D3P_TF_R(26) D3P_TF_R(6)
    x0 += k0; x1 += k1 + 3u;
    D3P_TF_PIN(x0);
    D3P_TF_R(17) D3P_TF_R(29) D3P_TF_R(16) D3P_TF_R(24)
    x0 += k1; x1 += k2 + 4u;
    D3P_TF_PIN(x0);
    D3P_TF_R(13) D3P_TF_R(15) D3P_TF_R(26) D3P_TF_R(6)
    x0 += k2; x1 += k0 + 5u;
#undef D3P_TF_R
    o0 = x0;
    o1 = x1;
}

// Word j of jax's threefry_2x32(key, iota(n)): counts are split in two halves (odd n zero-padded).
__device__ __forceinline__ uint32_t tf_iota_word(uint32_t k0, uint32_t k1, uint64_t n, uint64_t j)
{
    const uint64_t half = (n + 1) >> 1;
    uint32_t a, b;
    if (j < half) {
        const uint64_t c1 = j + half;
        threefry2x32(k0, k1, (uint32_t)j, c1 < n ? (uint32_t)c1 : 0u, a, b);
        return a;
    }
    threefry2x32(k0, k1, (uint32_t)(j - half), (uint32_t)j, a, b);
    return b;
}

// jax.random.uniform's float32 construction: (bits >> 9) | 0x3f800000, minus 1, affine, clamp at lo.
__device__ __forceinline__ float bits_to_uniform(uint32_t bits, float lo, float hi)
{
    const float f = __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, bits, 9)) - 1.0f;
    const float scale = hi - lo;
    const float r = __fmaf_rn(f, scale, lo);
    return fmaxf(lo, r);
}

// float32 erf_inv (Giles' single-precision polynomial, the form XLA lowers lax.erf_inv to).
__device__ __forceinline__ float erfinv_f32(float x)
{
    float w = -__logf(__fmaf_rn(-x, x, 1.0f));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = __fmaf_rn(p, w, 3.43273939e-07f);
        p = __fmaf_rn(p, w, -3.5233877e-06f);
        p = __fmaf_rn(p, w, -4.39150654e-06f);
        p = __fmaf_rn(p, w, 0.00021858087f);
        p = __fmaf_rn(p, w, -0.00125372503f);
        p = __fmaf_rn(p, w, -0.00417768164f);
        p = __fmaf_rn(p, w, 0.246640727f);
        p = __fmaf_rn(p, w, 1.50140941f);
    } else {
        w = __fsqrt_rn(w) - 3.0f;
        p = -0.000200214257f;
        p = __fmaf_rn(p, w, 0.000100950558f);
        p = __fmaf_rn(p, w, 0.00134934322f);
        p = __fmaf_rn(p, w, -0.00367342844f);
        p = __fmaf_rn(p, w, 0.00573950773f);
        p = __fmaf_rn(p, w, -0.0076224613f);
        p = __fmaf_rn(p, w, 0.00943887047f);
        p = __fmaf_rn(p, w, 1.00167406f);
        p = __fmaf_rn(p, w, 2.83297682f);
    }
    return p * x;
}

// Same function with the rare tail polynomial (|x| > 0.9966, 0.34 % of draws) behind a wave-uniform
// branch: identical arithmetic per lane, but a wave only pays for the tail when one of its lanes
// needs it.  All 64 lanes must call it together.
__device__ __forceinline__ float erfinv_f32_wu(float x)
{
    // 1 - x^2 >= 2^-23 for every x this is called with, so the raw v_log_f32 (no denormal pre-scaling)
    // is exact enough: w = -ln2 * log2(1 - x^2)
    const float w = -0.693147182f * __builtin_amdgcn_logf(__fmaf_rn(-x, x, 1.0f));
    const float wc = w - 2.5f;
    float p = 2.81022636e-08f;
    p = __fmaf_rn(p, wc, 3.43273939e-07f);
    p = __fmaf_rn(p, wc, -3.5233877e-06f);
    p = __fmaf_rn(p, wc, -4.39150654e-06f);
    p = __fmaf_rn(p, wc, 0.00021858087f);
    p = __fmaf_rn(p, wc, -0.00125372503f);
    p = __fmaf_rn(p, wc, -0.00417768164f);
    p = __fmaf_rn(p, wc, 0.246640727f);
    p = __fmaf_rn(p, wc, 1.50140941f);
    const bool tail = !(w < 5.0f);
    if (__builtin_amdgcn_ballot_w64(tail) != 0ull) {
        const float wt = __fsqrt_rn(w) - 3.0f;
        float q = -0.000200214257f;
        q = __fmaf_rn(q, wt, 0.000100950558f);
        q = __fmaf_rn(q, wt, 0.00134934322f);
        q = __fmaf_rn(q, wt, -0.00367342844f);
        q = __fmaf_rn(q, wt, 0.00573950773f);
        q = __fmaf_rn(q, wt, -0.0076224613f);
        q = __fmaf_rn(q, wt, 0.00943887047f);
        q = __fmaf_rn(q, wt, 1.00167406f);
        q = __fmaf_rn(q, wt, 2.83297682f);
        p = tail ? q : p;
    }
    return p * x;
}

__device__ __forceinline__ float bits_to_normal_wu(uint32_t bits)
{
    return D3P_SQRT2 * erfinv_f32_wu(bits_to_uniform(bits, D3P_NORMAL_LO, 1.0f));
}

// d3p.random.normal / jax.random.normal transform of one 32-bit word.
__device__ __forceinline__ float bits_to_normal(uint32_t bits)
{
    return D3P_SQRT2 * erfinv_f32(bits_to_uniform(bits, D3P_NORMAL_LO, 1.0f));
}

// ---- wave64 all-reduce (sum): 4 DPP steps inside each row of 16 lanes, then the 4 row sums are
// read through SGPRs.  Fixed order => bitwise reproducible.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror: every lane of a row of 16 holds the row's sum r0 .. r3
    // across the rows with the two broadcast steps of the GFX9 DPP encoding: row 1 += lane 15 (r0), row 3 += lane 47 (r2), then
    // rows 2, 3 += lane 31 (r0 + r1): lane 63 holds (r0 + r1) + (r2 + r3) -- the same value, bit for bit, as reading the four
    // row sums through SGPRs and adding them (round 2: 4 v_readlane + 3 v_add), in 2 + 1 instructions.  (A masked-off row keeps
    // its value, which `old` of __builtin_amdgcn_update_dpp cannot express for an add; the s_nop are the VALU -> DPP hazard.)
    // Hazards the compiler's recognizer cannot see inside the asm: (i) VALU write of a VGPR -> DPP read of it (2 wait states): the
    // s_nop 1 in front of each v_add_f32_dpp; (ii) VALU write of EXEC -> DPP (5 wait states): the asm is always preceded by the four
    // compiler-emitted v_add_f32_dpp above plus the first s_nop 1 -- six issue slots -- and the compiler itself keeps an EXEC write
    // five wait states away from THOSE, so no EXEC write can sit closer than that to the asm block.
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The wave sums of TWO values at once: v_permlane32_swap puts the low halves of both side by side (a' = [a_lo | b_lo],
// b' = [a_hi | b_hi]), so a' + b' holds a's pair sums in lanes 0 - 31 and b's in lanes 32 - 63; four DPP steps sum each row of
// 16, one row_bcast:15 adds row 0 into row 1 and row 2 into row 3: lane 31 holds sum(a), lane 63 sum(b).  Nine instructions
// instead of 2 x 8.  (Another fixed summation order than wave_sum's: used where no other kernel form has to match bit for bit.)
__device__ __forceinline__ void wave_sum2(float a_, float b_, float& sa, float& sb)
{
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_), __float_as_uint(b_), false, false);
    float c = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    c += dpp_mov<0xB1>(c);
    c += dpp_mov<0x4E>(c);
    c += dpp_mov<0x141>(c);
    c += dpp_mov<0x140>(c);
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1" : "+v"(c));
    sa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 31));
    sb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
}

// softplus(t) = max(t, 0) + log(1 + exp(-|t|)); exp(-|t|) is in (0, 1], so log(1 + e) via v_log_f32 has an
// absolute error of ~1e-7 (relative to values >= ln 2 * e): inside the stated float32 tolerance.
// (1 + exp(-|t|) lies in (1, 2]: a normal number, so the raw v_log_f32 (log2) needs none of the denormal scaling and fix-up
// code __logf carries -- ten instructions, on the critical path of the step kernel's update prologue and of every example)
__device__ __forceinline__ float softplus_f(float t)
{
    return fmaxf(t, 0.0f) + 0.693147180559945309f * __builtin_amdgcn_logf(1.0f + __expf(-fabsf(t)));
}
__device__ __forceinline__ float sigmoid_f(float t) { return __builtin_amdgcn_rcpf(1.0f + __expf(-t)); }  // v_rcp_f32: 1 ulp

// Row layout of a fixed-point accumulator replica: P gradient columns (scale 2^40 / C) | loss, fine part | example count |
// loss, coarse part | spare.  The loss partial s of a workgroup (any magnitude a float can hold up to 2^78) is split exactly
// as s = hi 2^27 + lo with hi = rint(s 2^-27) and the remainder lo kept at a resolution of 2^-24: both parts are integers that
// sum exactly and order-independently like the gradient columns, and no model-dependent bound has to be guessed.
#define D3P_ACC_COLS(P) ((P) + 4)
#define D3P_LOSS_HI_UNIT 134217728.0   // 2^27
#define D3P_LOSS_LO_SCALE 16777216.0   // 2^24

// round(v * sg) as a 64-bit integer for |v * sg| < 2^51: ONE fused multiply-add in double onto 1.5 * 2^52 leaves the rounded
// (to nearest, ties to even) integer in the low mantissa bits, a 64-bit subtraction of the constant's bit pattern extracts it --
// 1 conversion + 1 v_fma_f64 + 2 integer instructions where __double2ll_rn expands to ~15.  (One rounding instead of the two of
// "multiply, then round": may differ from it in the last fixed-point digit, 2^-40 C; every form of the sum stays exact.)
__device__ __forceinline__ long long fixed_point_rn(float v, double sg, bool& in_range)
{
    const double t = fma((double)v, sg, 6755399441055744.0);
    in_range = fabs((double)v * sg) < 2251799813685248.0;  // 2^51 (NaN / Inf fail the comparison)
    return (long long)(__double_as_longlong(t) - 0x4338000000000000ll);
}

// (double)v for every int64 v, in three instructions instead of the ~12 of the generic conversion: v = hi 2^32 + lo with hi the
// signed upper and lo the unsigned lower word, both exact in double; the fused multiply-add rounds their exact sum once.
__device__ __forceinline__ double i64_to_f64(long long v)
{
    return fma((double)(int)(v >> 32), 4294967296.0, (double)(unsigned int)v);
}

// the two integer parts of a workgroup's loss partial; false when it is not finite or beyond 2^78
__device__ __forceinline__ bool loss_split(float s, long long& hi, long long& lo)
{
    // (the same two integers as rint / llrint give, through the 1.5 * 2^52 trick of fixed_point_rn: |s 2^-27| < 2^51 and the
    // remainder times 2^24 is below 2^51 in magnitude, so both roundings are exact integer extractions)
    const double sd = (double)s;
    const double th = fma(sd, 1.0 / D3P_LOSS_HI_UNIT, 6755399441055744.0);
    const double h = th - 6755399441055744.0;   // rint(sd 2^-27), ties to even
    hi = (long long)(__double_as_longlong(th) - 0x4338000000000000ll);
    const double tl = fma(fma(-h, D3P_LOSS_HI_UNIT, sd), D3P_LOSS_LO_SCALE, 6755399441055744.0);
    lo = (long long)(__double_as_longlong(tl) - 0x4338000000000000ll);
    return fabs(sd) < 3.0e23;
}

__device__ __forceinline__ double loss_join(long long hi, long long lo)
{
    return (double)hi * D3P_LOSS_HI_UNIT + (double)lo * (1.0 / D3P_LOSS_LO_SCALE);
}

// The loss of a batch with NO valid example (a suppressed Poisson batch, minibatch.py:119-122).  The reference multiplies every
// example's loss by its mask (svi.py:271-281): 0 for finite parameters, but NaN * 0 = NaN once any parameter is not finite --
// which is the state the first empty batch leaves (svi.py:365: C / 0 = inf).  The step kernels evaluate no masked
// example, so their reporter asks the parameters the step ran with.  Cold path: one thread, only when n = 0.
// x(c): parameter c as the step saw it.
template <class X>
__device__ inline float empty_batch_loss(int P, X x)
{
    bool bad = false;
    for (int c = 0; c < P; ++c) {
        const float v = x(c);
        bad |= !(fabsf(v) <= 3.402823466e38f);
    }
    return bad ? __builtin_nanf("") : 0.0f;
}

// Position -> row of the Feistel permutation sampler (util.py:248-301): ten rounds over the (upper, lower) halves of the position,
// cycle-walking until the value falls below the capacity.  rc: the 30 round constants (column 0 already forced odd).
__device__ __forceinline__ uint32_t feistel_permute_dev(const uint32_t* rc, uint32_t capacity, int bits_lower,
                                                        int bits_upper, uint32_t position)
{
    const uint32_t mask_lower = (1u << bits_lower) - 1u, mask_upper = (1u << bits_upper) - 1u;
    uint32_t x = position;
    do {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const uint32_t k0 = rc[3 * j], k1 = rc[3 * j + 1], k2 = rc[3 * j + 2];
            const uint32_t xu = x >> bits_lower, xl = x & mask_lower;
            const uint32_t yu = xl ^ ((((xu * k1) >> bits_upper) ^ k2) & mask_lower);
            const uint32_t yl = (xu * k0) & mask_upper;
            x = (yu << bits_upper) | yl;
        }
    } while (x >= capacity);
    return x;
}

}  // namespace d3p
