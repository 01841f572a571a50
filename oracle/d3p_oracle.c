/*
 * d3p_oracle.c -- CPU restatement of the d3p DP-VI hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP path in d3p_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * package (d3p_amd/) never imports, links or calls anything in oracle/.
 *
 * Every function cites the reference lines (relative to /root/reference) it restates.
 *
 * PARITY STATUS
 *   pinned   : ChaCha20 block function (RFC 8439 2.3.2 / A.1 vectors, OpenSSL cross-check),
 *              threefry2x32 + jax.random split/random_bits/normal layout (Random123 KATs and
 *              values published in the JAX documentation / test-suite),
 *              Feistel sampler arithmetic (d3p/util.py:229-301, fully specified there),
 *              clip / mean / perturbation-scale arithmetic (known-answer tests of
 *              tests/test_dpsvi.py, tests/test_gradient_manipulators.py).
 *   UNPINNED : the key/nonce/counter layout of chacha.random (PyPI jax-chacha-prng >=1,<2 is
 *              not vendored in the reference and not installed here; no reference test holds
 *              a known-answer vector for split / fold_in / random_bits / uniform).  The layout
 *              below is THIS BUILD'S OWN, documented in DESIGN.md ("parity unpinned").
 *              Also unpinned: numpyro's AutoDiagonalNormal parametrisation and Trace_ELBO /
 *              seed-handler key plumbing (numpyro is absent); restated from its published
 *              behaviour and flagged where used.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define D3P_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * ChaCha20 block function, RFC 8439 section 2.3 (20 rounds, 32-bit counter, 96-bit nonce).
 * State layout (16 x u32): [0..3] "expand 32-byte k", [4..11] key, [12] counter, [13..15] nonce.
 * Replaces: chacha.random / chacha.cipher of jax-chacha-prng (d3p/random/__init__.py:25-32).
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

#define QR(a, b, c, d)                 \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7);

D3P_API void d3po_chacha20_block(const uint32_t in[16], uint32_t out[16])
{
    uint32_t x[16];
    memcpy(x, in, sizeof(x));
    for (int i = 0; i < 10; ++i) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13])
        QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12])
        QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
}

/* Build's own key layout (PARITY UNPINNED, see header).
 * PRNGKey(seed) (d3p/random/__init__.py:35-47): 32 seed bytes -> key words little-endian as in
 * RFC 8439 2.3; counter = 0; nonce = 0.  (int seeds are reduced mod 2^256 and serialised
 * big-endian to 32 bytes, shorter byte strings are right-padded with zeros: host side.) */
D3P_API void d3po_key_from_bytes(const uint8_t seed[32], uint32_t state[16])
{
    state[0] = 0x61707865u; state[1] = 0x3320646eu; state[2] = 0x79622d32u; state[3] = 0x6b206574u;
    for (int i = 0; i < 8; ++i)
        state[4 + i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) |
                       ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    state[12] = state[13] = state[14] = state[15] = 0;
}

#define D3P_TAG_SPLIT 0x00000001u
#define D3P_TAG_FOLD 0x00000002u

/* child key = first 8 words of the block of (parent, counter += i, nonce ^= (data, 0, tag));
 * child counter and nonce are reset to zero, so random_bits (nonce tag 0), split (tag 1) and
 * fold_in (tag 2) draw from disjoint (nonce, counter) domains of the parent key. */
static void derive_child(const uint32_t parent[16], uint32_t ctr_add, uint32_t data, uint32_t tag,
                         uint32_t child[16])
{
    uint32_t in[16], blk[16];
    memcpy(in, parent, sizeof(in));
    in[12] += ctr_add;
    in[13] ^= data;
    in[15] ^= tag;
    d3po_chacha20_block(in, blk);
    memcpy(child, parent, 4 * sizeof(uint32_t));
    memcpy(child + 4, blk, 8 * sizeof(uint32_t));
    child[12] = child[13] = child[14] = child[15] = 0;
}

/* rng_suite.split (d3p/random/__init__.py:29; call sites svi.py:210, :491) */
D3P_API void d3po_split(const uint32_t key[16], int num, uint32_t* out /* num x 16 */)
{
    for (int i = 0; i < num; ++i) derive_child(key, (uint32_t)i, 0u, D3P_TAG_SPLIT, out + 16 * i);
}

/* rng_suite.fold_in (d3p/random/__init__.py:30; call sites minibatch.py:115, :207, :230) */
D3P_API void d3po_fold_in(const uint32_t key[16], uint32_t data, uint32_t out[16])
{
    derive_child(key, 0u, data, D3P_TAG_FOLD, out);
}

/* rng_suite.random_bits, 32-bit words (d3p/random/__init__.py:31): keystream word j is word
 * (j mod 16) of the block with counter key[12] + j/16. */
D3P_API void d3po_random_words(const uint32_t key[16], uint64_t first_word, uint64_t n_words,
                               uint32_t* out)
{
    uint32_t in[16], blk[16];
    memcpy(in, key, sizeof(in));
    uint64_t cur_blk = (uint64_t)-1;
    for (uint64_t j = 0; j < n_words; ++j) {
        uint64_t w = first_word + j, b = w >> 4;
        if (b != cur_blk) {
            in[12] = key[12] + (uint32_t)b;
            d3po_chacha20_block(in, blk);
            cur_blk = b;
        }
        out[j] = blk[w & 15];
    }
}

/* random_bits for bit_width 8/16/32/64: element e of the flat output is taken little-endian
 * from the keystream bytes (width 8/16: low bits of a word first; width 64: lo word first). */
D3P_API int d3po_random_bits(const uint32_t key[16], int bit_width, uint64_t count, void* out)
{
    if (bit_width != 8 && bit_width != 16 && bit_width != 32 && bit_width != 64) return -1;
    uint64_t n_words = (count * (uint64_t)bit_width + 31) / 32;
    uint32_t* w = (uint32_t*)malloc((n_words ? n_words : 1) * sizeof(uint32_t));
    if (!w) return -2;
    d3po_random_words(key, 0, n_words, w);
    for (uint64_t e = 0; e < count; ++e) {
        switch (bit_width) {
        case 8: ((uint8_t*)out)[e] = (uint8_t)(w[e >> 2] >> (8 * (e & 3))); break;
        case 16: ((uint16_t*)out)[e] = (uint16_t)(w[e >> 1] >> (16 * (e & 1))); break;
        case 32: ((uint32_t*)out)[e] = w[e]; break;
        default: ((uint64_t*)out)[e] = (uint64_t)w[2 * e] | ((uint64_t)w[2 * e + 1] << 32);
        }
    }
    free(w);
    return 0;
}

/* bits -> float32 in [lo, hi), the jax.random.uniform construction that chacha.random.uniform
 * mirrors (d3p/random/__init__.py:32, :80): mantissa bits | 1.0f, minus 1, affine, clamp at lo. */
static inline float bits_to_uniform(uint32_t bits, float lo, float hi)
{
    union { uint32_t u; float f; } c;
    c.u = (bits >> 9) | 0x3f800000u;
    float f = c.f - 1.0f;
    float scale = hi - lo;
    float r = fmaf(f, scale, lo); /* XLA permits mul+add fusion (AllowFPOpFusion = Fast): one rounding */
    return r < lo ? lo : r;
}

D3P_API void d3po_uniform(const uint32_t key[16], uint64_t n, float lo, float hi, float* out)
{
    uint32_t in[16], blk[16];
    memcpy(in, key, sizeof(in));
    for (uint64_t j = 0; j < n; ++j) {
        if ((j & 15) == 0) {
            in[12] = key[12] + (uint32_t)(j >> 4);
            d3po_chacha20_block(in, blk);
        }
        out[j] = bits_to_uniform(blk[j & 15], lo, hi);
    }
}

/* float32 erf_inv as lowered by XLA for jax.lax.erf_inv (d3p/random/__init__.py:81): M. Giles,
 * "Approximating the erfinv function", single-precision polynomial; +-inf at |x| == 1. */
D3P_API float d3po_erfinv_f32(float x)
{
    /* XLA writes w = -log1p(-x*x); the rounding of x*x costs up to ~1e-5 relative accuracy for
     * |result| > 3.  The single-rounding form below is within 2 ulp of the exact value; parity with
     * the reference's XLA lowering is therefore stated at 1e-5 relative. */
    float w = -logf(fmaf(-x, x, 1.0f));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = fmaf(p, w, 3.43273939e-07f);
        p = fmaf(p, w, -3.5233877e-06f);
        p = fmaf(p, w, -4.39150654e-06f);
        p = fmaf(p, w, 0.00021858087f);
        p = fmaf(p, w, -0.00125372503f);
        p = fmaf(p, w, -0.00417768164f);
        p = fmaf(p, w, 0.246640727f);
        p = fmaf(p, w, 1.50140941f);
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = fmaf(p, w, 0.000100950558f);
        p = fmaf(p, w, 0.00134934322f);
        p = fmaf(p, w, -0.00367342844f);
        p = fmaf(p, w, 0.00573950773f);
        p = fmaf(p, w, -0.0076224613f);
        p = fmaf(p, w, 0.00943887047f);
        p = fmaf(p, w, 1.00167406f);
        p = fmaf(p, w, 2.83297682f);
    }
    if (fabsf(x) == 1.0f) return x * INFINITY;
    return p * x;
}

#define D3P_NORMAL_LO (-0.99999994f) /* np.nextafter(-1f, 0f): d3p/random/__init__.py:78 */
#define D3P_SQRT2 1.41421354f        /* np.float32(np.sqrt(2)): d3p/random/__init__.py:81 */

static inline float bits_to_normal(uint32_t bits)
{
    return D3P_SQRT2 * d3po_erfinv_f32(bits_to_uniform(bits, D3P_NORMAL_LO, 1.0f));
}

/* d3p.random.normal / _normal (d3p/random/__init__.py:50-81) */
D3P_API void d3po_normal(const uint32_t key[16], uint64_t n, float* out)
{
    uint32_t in[16], blk[16];
    memcpy(in, key, sizeof(in));
    for (uint64_t j = 0; j < n; ++j) {
        if ((j & 15) == 0) {
            in[12] = key[12] + (uint32_t)(j >> 4);
            d3po_chacha20_block(in, blk);
        }
        out[j] = bits_to_normal(blk[j & 15]);
    }
}

/* d3p.random._randint for 32-bit dtypes (d3p/random/__init__.py:108-146).  delta, the float32
 * log2 at :125 and the masked rejection loop (:130-143) are restated literally; since a lane
 * that was accepted never changes again the all-lanes while_loop equals per-lane iteration. */
D3P_API int d3po_randint32(const uint32_t key_in[16], uint64_t n, int32_t minval, int32_t maxval,
                           int32_t* out)
{
    uint32_t delta = (uint32_t)(maxval - 1 - minval);
    float l2 = log2f((float)delta) + 1.0f;          /* jnp.log2(jnp.float32(delta)) + 1 */
    uint32_t lg;
    if (!(l2 > 0.0f)) lg = 0;                        /* udtype(-inf) / udtype(nan) -> 0 on XLA:CPU */
    else if (l2 >= 32.0f) lg = 32;
    else lg = (uint32_t)l2;
    if (lg > 32) lg = 32;
    uint32_t bitmask = (lg >= 32) ? 0xffffffffu : ((1u << lg) - 1u);
    uint32_t* u = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t* nu = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!u || !nu) { free(u); free(nu); return -2; }
    uint32_t key[16], ks[32];
    memcpy(key, key_in, sizeof(key));
    d3po_split(key, 2, ks);
    memcpy(key, ks, sizeof(key));
    d3po_random_words(ks + 16, 0, n, u);
    for (uint64_t j = 0; j < n; ++j) u[j] &= bitmask;
    for (int round = 0; round < 4096; ++round) {
        int any = 0;
        for (uint64_t j = 0; j < n; ++j) any |= (u[j] > delta);
        if (!any) break;
        d3po_split(key, 2, ks);
        memcpy(key, ks, sizeof(key));
        d3po_random_words(ks + 16, 0, n, nu);
        for (uint64_t j = 0; j < n; ++j)
            if (u[j] > delta) u[j] = nu[j] & bitmask;
    }
    for (uint64_t j = 0; j < n; ++j) out[j] = (int32_t)u[j] + minval;
    free(u); free(nu);
    return 0;
}

/* The same for every integer dtype of d3p/random/__init__.py:115-123 (8, 16, 32, 64 bits): delta and the mask live in the
 * unsigned dtype of that width (wrapping), element j of random_bits(round_key, nbits, shape) is the little-endian view of
 * the keystream bytes, the result wraps in the signed dtype.  `out` holds n elements of nbits bits. */
D3P_API int d3po_randint_bits(const uint32_t key_in[16], uint64_t n, int nbits, int64_t minval, int64_t maxval, void* out)
{
    if (nbits != 8 && nbits != 16 && nbits != 32 && nbits != 64) return -1;
    const uint64_t full = nbits == 64 ? ~0ull : ((1ull << nbits) - 1ull);
    const uint64_t delta = (uint64_t)(maxval - 1 - minval) & full;      /* udtype(maxval - 1 - minval) */
    float l2 = log2f((float)delta) + 1.0f;                              /* jnp.log2(jnp.float32(delta)) + 1 */
    int lg;
    if (!(l2 > 0.0f)) lg = 0;
    else if (l2 >= (float)nbits) lg = nbits;
    else lg = (int)l2;
    const uint64_t bitmask = lg >= nbits ? full : ((1ull << lg) - 1ull);
    const int nb = nbits / 8;
    const uint64_t nwords = (n * (uint64_t)nb + 3) / 4 + 1;
    uint32_t* w = (uint32_t*)malloc(nwords * sizeof(uint32_t));
    uint64_t* u = (uint64_t*)malloc((n ? n : 1) * sizeof(uint64_t));
    uint8_t* done = (uint8_t*)calloc(n ? n : 1, 1);
    if (!w || !u || !done) { free(w); free(u); free(done); return -2; }
    uint32_t key[16], ks[32];
    memcpy(key, key_in, sizeof(key));
    for (int round = 0; round < 4096; ++round) {
        int any = 0;
        if (round > 0) {
            for (uint64_t j = 0; j < n; ++j) any |= !done[j];
            if (!any) break;
        }
        d3po_split(key, 2, ks);
        memcpy(key, ks, sizeof(key));
        d3po_random_words(ks + 16, 0, nwords, w);
        const uint8_t* bytes = (const uint8_t*)w;   /* little-endian host, as the reference's */
        for (uint64_t j = 0; j < n; ++j) {
            if (done[j]) continue;
            uint64_t raw = 0;
            memcpy(&raw, bytes + j * (uint64_t)nb, (size_t)nb);
            u[j] = raw & bitmask;
            done[j] = u[j] <= delta;
        }
    }
    for (uint64_t j = 0; j < n; ++j) {
        const uint64_t v = (u[j] + (uint64_t)minval) & full;            /* vdtype(uvals) + minval, wrapping */
        memcpy((uint8_t*)out + j * (uint64_t)nb, &v, (size_t)nb);
    }
    free(w); free(u); free(done);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * threefry2x32-20 and the jax.random (<= 0.4.10, non-partitionable) array layouts.
 * Used for (i) d3p.random.debug (d3p/random/debug.py:34-80), (ii) the per-example guide noise
 * eps_i, which the reference draws from JAX's default PRNG (d3p/svi.py:259, :289-290;
 * README.md:49-50).  Pinned by Random123 KATs and JAX-published values (tests/test_oracle_pins).
 * ---------------------------------------------------------------------------------------- */
D3P_API void d3po_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t out[2])
{
    static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
    for (int i = 0; i < 5; ++i) {
        for (int r = 0; r < 4; ++r) {
            x0 += x1;
            x1 = rotl32(x1, R[i & 1][r]);
            x1 ^= x0;
        }
        x0 += ks[(i + 1) % 3];
        x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
    }
    out[0] = x0;
    out[1] = x1;
}

/* jax threefry_2x32(key, iota(n))[j]: the count array is split in halves (padded to even). */
static inline uint32_t tf_iota_word(uint32_t k0, uint32_t k1, uint64_t n, uint64_t j)
{
    uint64_t half = (n + 1) / 2;
    uint32_t o[2];
    if (j < half) {
        uint64_t c1 = j + half;
        d3po_threefry2x32(k0, k1, (uint32_t)j, (c1 < n) ? (uint32_t)c1 : 0u, o);
        return o[0];
    }
    d3po_threefry2x32(k0, k1, (uint32_t)(j - half), (uint32_t)j, o);
    return o[1];
}

D3P_API void d3po_tf_random_words(const uint32_t key[2], uint64_t n, uint32_t* out)
{
    for (uint64_t j = 0; j < n; ++j) out[j] = tf_iota_word(key[0], key[1], n, j);
}

D3P_API void d3po_tf_split(const uint32_t key[2], int num, uint32_t* out /* num x 2 */)
{
    d3po_tf_random_words(key, 2 * (uint64_t)num, out);
}

D3P_API void d3po_tf_fold_in(const uint32_t key[2], uint32_t data, uint32_t out[2])
{
    d3po_threefry2x32(key[0], key[1], 0u, data, out);
}

D3P_API void d3po_tf_uniform(const uint32_t key[2], uint64_t n, float lo, float hi, float* out)
{
    for (uint64_t j = 0; j < n; ++j)
        out[j] = bits_to_uniform(tf_iota_word(key[0], key[1], n, j), lo, hi);
}

D3P_API void d3po_tf_normal(const uint32_t key[2], uint64_t n, float* out)
{
    for (uint64_t j = 0; j < n; ++j) out[j] = bits_to_normal(tf_iota_word(key[0], key[1], n, j));
}

/* Key of the guide's `_auto_latent` sample for the example at batch position p
 * (d3p/svi.py:289-290 px_rng_keys = jax.random.split(jax_rng_key, B); numpyro Trace_ELBO:
 * model_seed, guide_seed = split(px_key); numpyro.handlers.seed: key, sample_key = split(key)).
 * UNPINNED against numpyro (absent); the threefry arithmetic itself is pinned. */
D3P_API void d3po_px_sample_key(const uint32_t jax_key[2], uint32_t B, uint32_t p, uint32_t out[2])
{
    uint32_t px[2], s[4];
    px[0] = tf_iota_word(jax_key[0], jax_key[1], 2ull * B, 2ull * p);
    px[1] = tf_iota_word(jax_key[0], jax_key[1], 2ull * B, 2ull * p + 1);
    d3po_tf_split(px, 2, s); /* guide_seed = s[2..3] */
    uint32_t g[2] = {s[2], s[3]};
    d3po_tf_split(g, 2, s);  /* sample key = s[2..3] */
    out[0] = s[2];
    out[1] = s[3];
}

/* Multi-site guides (examples/logistic_regression.py:67-86: `sample('w', ...)` then `sample('intercept', ...)`): numpyro's `seed`
 * handler advances its key at every sample statement, in program order -- rng, site_key = split(rng) -- starting from the guide
 * seed of the example (as in d3po_px_sample_key, which is the n_sites = 1 case, and d3po_gmm_site_keys).  UNPINNED: numpyro's
 * key plumbing is restated from its published behaviour, the reference holds no vector for it (header, "PARITY UNPINNED"). */
D3P_API void d3po_px_site_keys(const uint32_t jax_key[2], uint32_t B, uint32_t p, int n_sites, uint32_t* out /* 2 n_sites */)
{
    uint32_t px[2], s[4], r[2];
    px[0] = tf_iota_word(jax_key[0], jax_key[1], 2ull * B, 2ull * p);
    px[1] = tf_iota_word(jax_key[0], jax_key[1], 2ull * B, 2ull * p + 1);
    d3po_tf_split(px, 2, s); /* guide_seed = s[2..3] */
    r[0] = s[2]; r[1] = s[3];
    for (int site = 0; site < n_sites; ++site) {
        d3po_tf_split(r, 2, s);
        out[2 * site] = s[2]; out[2 * site + 1] = s[3];
        r[0] = s[0]; r[1] = s[1];
    }
}

/* eps[i] = [normal(site_key_0, (size_0,)) | normal(site_key_1, (size_1,)) | ...] for the examples at positions pos0 .. pos0 + B_local - 1
 * of a batch of B_total (svi.py:289-290: one key per example from split(jax_key, B_total)); a scalar site (shape ()) has size 1. */
D3P_API void d3po_px_eps_sites(const uint32_t jax_key[2], uint32_t B_total, uint32_t pos0, uint32_t B_local, const int32_t* site_sizes,
                               int n_sites, float* eps)
{
    int D = 0;
    for (int s = 0; s < n_sites; ++s) D += site_sizes[s];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)B_local; ++i) {
        uint32_t sk[2 * 16];
        d3po_px_site_keys(jax_key, B_total, pos0 + (uint32_t)i, n_sites, sk);
        float* e = eps + (size_t)i * D;
        for (int s = 0; s < n_sites; ++s) {
            d3po_tf_normal(sk + 2 * s, (uint64_t)site_sizes[s], e);
            e += site_sizes[s];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Feistel sampler: d3p/util.py:216-301 (verbatim uint32 arithmetic).
 * ---------------------------------------------------------------------------------------- */
static inline int bit_length_u32(uint32_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

D3P_API void d3po_feistel_constants(const uint32_t key[16], uint32_t rc[30])
{
    d3po_random_words(key, 0, 30, rc);            /* util.py:240-242 */
    for (int j = 0; j < 10; ++j) rc[3 * j] |= 1u; /* util.py:245-246 */
}

D3P_API uint32_t d3po_feistel_permute(const uint32_t rc[30], uint32_t capacity, uint32_t position)
{
    int bits = bit_length_u32(capacity - 1); /* util.py:230 */
    int bits_lower = bits >> 1, bits_upper = bits - bits_lower;
    uint32_t mask_lower = (1u << bits_lower) - 1u, mask_upper = (1u << bits_upper) - 1u;
    uint32_t x = position;
    do {
        for (int j = 0; j < 10; ++j) { /* util.py:273-285 */
            const uint32_t* k = rc + 3 * j;
            uint32_t xu = x >> bits_lower, xl = x & mask_lower;
            uint32_t yu = xl ^ ((((xu * k[1]) >> bits_upper) ^ k[2]) & mask_lower);
            uint32_t yl = (xu * k[0]) & mask_upper;
            x = (yu << bits_upper) | yl;
        }
    } while (x >= capacity); /* util.py:291-296 */
    return x;
}

D3P_API void d3po_feistel_sample(const uint32_t key[16], uint32_t capacity, uint32_t n, uint32_t* out)
{
    uint32_t rc[30];
    d3po_feistel_constants(key, rc);
    for (uint32_t p = 0; p < n; ++p) out[p] = d3po_feistel_permute(rc, capacity, p);
}

/* ------------------------------------------------------------------------------------------
 * Poisson selection: d3p/minibatch.py:29-39 (+ :119-124).  argsort of the boolean selectors is
 * stable, so [::-1] yields the selected indices in DESCENDING order followed by the unselected
 * ones in descending order; the first `cutoff` of them are returned.  counts[0] = raw number
 * selected (:35), counts[1] = after truncate / suppress (:119-122).
 * ---------------------------------------------------------------------------------------- */
D3P_API void d3po_poisson_select(const uint32_t key[16], float q, uint32_t N, uint32_t cutoff,
                                 int suppress, uint32_t* idx_out, uint32_t counts[2])
{
    float* u = (float*)malloc((N ? N : 1) * sizeof(float));
    d3po_uniform(key, N, 0.0f, 1.0f, u);
    uint32_t nsel = 0, w = 0;
    for (uint32_t j = 0; j < N; ++j) nsel += (u[j] <= q);
    for (int64_t j = (int64_t)N - 1; j >= 0 && w < cutoff; --j)
        if (u[j] <= q) idx_out[w++] = (uint32_t)j;
    for (int64_t j = (int64_t)N - 1; j >= 0 && w < cutoff; --j)
        if (!(u[j] <= q)) idx_out[w++] = (uint32_t)j;
    counts[0] = nsel;
    counts[1] = suppress ? (nsel <= cutoff ? nsel : 0u) : (nsel < cutoff ? nsel : cutoff);
    free(u);
}

/* ------------------------------------------------------------------------------------------
 * DP-VI update for Bayesian logistic regression + AutoDiagonalNormal guide.
 *
 * Model (README.md:89-97, examples/logistic_regression.py:49-66): w ~ N(0, prior_w^2 I_d),
 * optional intercept ~ N(0, prior_b^2); y_i ~ Bernoulli(logits = x_i.w + b) inside
 * plate(N, subsample_size=batch).  Guide: numpyro AutoDiagonalNormal over the D = d (+1) latents,
 * unconstrained params auto_loc (D) and auto_scale_unc (D) with scale = softplus(unc)
 * (UNPINNED numpyro >= 0.8 behaviour).  Per-example loss as wrapped at d3p/svi.py:271-281 for a
 * batch of one example (plate scale = N):
 *     L_i = inv_obs * ( logq(z_i) - logp(z_i) - lik_scale * loglik(y_i | x_i, z_i) ) * mask_i
 * with z_i = loc + scale * eps_i.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t d;          /* feature columns of X */
    int32_t intercept;  /* 0/1: latent D = d + intercept */
    float prior_w;      /* prior std of w */
    float prior_b;      /* prior std of the intercept */
    float lik_scale;    /* plate scale N (num_obs_total) */
    float inv_obs;      /* 1 / observation_scale (svi.py:278) */
    int32_t family;     /* 0: Bernoulli-logit regression; 1: obs ~ Normal(mu, lik_sigma).to_event(1), mu ~ N(0, prior_w)
                         * (examples/simple_gaussian_posterior.py:51-65; labels unused) */
    int32_t guide_exp;  /* 0: scale = softplus(u) (AutoDiagonalNormal); 1: scale = exp(u)
                         * (examples/simple_gaussian_posterior.py:77-81, params mu_loc / mu_std_log) */
    float lik_sigma;    /* observation std of family 1 */
} d3po_logreg_spec;

/* scale of the guide and its derivative wrt the unconstrained parameter */
static inline void guide_scale(const d3po_logreg_spec* sp, float u, float* s, float* ds);

static inline float softplus_f(float t) { return fmaxf(t, 0.0f) + log1pf(expf(-fabsf(t))); }
static inline float sigmoid_f(float t) { return 1.0f / (1.0f + expf(-t)); }

#define HALF_LOG_2PI 0.918938533204672742f

static inline void guide_scale(const d3po_logreg_spec* sp, float u, float* s, float* ds)
{
    if (sp->guide_exp) { *s = *ds = expf(u); }
    else { *s = softplus_f(u); *ds = sigmoid_f(u); }
}

/* One example: loss and gradient wrt (auto_loc, auto_scale_unc); grad has 2*D entries
 * [d/dloc (D) | d/dunc (D)] = tree_flatten order of {auto_loc, auto_scale} (svi.py:490). */
D3P_API float d3po_logreg_px_loss_grad(const d3po_logreg_spec* sp, const float* loc, const float* unc,
                                       const float* x, float y, const float* eps, float mask,
                                       float* grad /* 2*D or NULL */)
{
    const int d = sp->d, D = sp->d + (sp->intercept ? 1 : 0);
    const int gauss = sp->family == 1;
    const double inv_var = gauss ? 1.0 / ((double)sp->lik_sigma * (double)sp->lik_sigma) : 0.0;
    double t = 0.0, lq = 0.0, lp = 0.0; /* accumulate in double: the oracle is the accurate side */
    for (int j = 0; j < D; ++j) {
        float s, sg;
        guide_scale(sp, unc[j], &s, &sg);
        float z = fmaf(s, eps[j], loc[j]);
        float ps = (j < d) ? sp->prior_w : sp->prior_b;
        float xv = (j < d) ? x[j] : 1.0f;
        if (gauss) { double r = (double)xv - (double)z; t += r * r; } /* squared residual norm */
        else t += (double)xv * (double)z;                             /* logit */
        lq += -0.5 * (double)eps[j] * (double)eps[j] - log((double)s) - (double)HALF_LOG_2PI;
        lp += -0.5 * ((double)z / ps) * ((double)z / ps) - log((double)ps) - (double)HALF_LOG_2PI;
    }
    float tf = (float)t;
    double loglik = gauss ? -0.5 * t * inv_var - D * (log((double)sp->lik_sigma) + (double)HALF_LOG_2PI)
                          : (double)y * t - (double)softplus_f(tf);
    double L = (double)sp->inv_obs * ((lq - lp) - (double)sp->lik_scale * loglik) * mask;
    if (grad) {
        /* dL/dz_j = inv_obs * z_j / ps^2 + A * xa_j, with (A, xa) = (inv_obs N (sigmoid(t) - y), x) for the
         * Bernoulli family and (-inv_obs N / sigma^2, x - z) for the Gaussian one */
        float A = gauss ? (float)(-(double)sp->inv_obs * sp->lik_scale * inv_var)
                        : sp->inv_obs * sp->lik_scale * (sigmoid_f(tf) - y);
        for (int j = 0; j < D; ++j) {
            float s, sg;
            guide_scale(sp, unc[j], &s, &sg);
            float z = fmaf(s, eps[j], loc[j]);
            float ps = (j < d) ? sp->prior_w : sp->prior_b;
            float xv = (j < d) ? x[j] : 1.0f;
            float xa = gauss ? xv - z : xv;
            float g = sp->inv_obs * z / (ps * ps) + A * xa;
            float h = (g * eps[j] - sp->inv_obs / s) * sg;
            grad[j] = g * mask;
            grad[D + j] = h * mask;
        }
    }
    return (float)L;
}

/* svi.py:238-308 -- materialise px_losses (B) and px_grads (B x 2D) like jax.vmap does.
 * Xb: B x d gathered rows, yb: B, eps: B x D, mask: B floats (0/1) or NULL (= all valid).
 * Returns num_elements; *factor = B/n (0 if n == 0) (svi.py:305); losses are rescaled by
 * obs_scale * factor (svi.py:306). */
D3P_API int d3po_logreg_px_grads(const d3po_logreg_spec* sp, const float* loc, const float* unc,
                                 const float* Xb, const float* yb, const float* eps,
                                 const float* mask, int B, float* px_loss, float* px_grads,
                                 float* factor)
{
    const int D = sp->d + (sp->intercept ? 1 : 0);
    int n = 0;
    for (int i = 0; i < B; ++i) n += (mask ? (mask[i] != 0.0f) : 1);
    float f = (n == 0) ? 0.0f : (float)B / (float)n;
    float obs = 1.0f / sp->inv_obs;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i) {
        float m = mask ? mask[i] : 1.0f;
        float L = d3po_logreg_px_loss_grad(sp, loc, unc, Xb + (size_t)i * sp->d, yb ? yb[i] : 0.0f,
                                           eps + (size_t)i * D, m, px_grads + (size_t)i * 2 * D);
        px_loss[i] = L * obs * f;
    }
    *factor = f;
    return n;
}

/* svi.py:68-87 full_norm over a flat row; svi.py:106-124 clip_gradient; svi.py:310-325 vmapped.
 * Returns -1 (the reference raises ValueError) when c == 0. */
D3P_API int d3po_clip_rows(float* px_grads, int B, int P, float c)
{
    if (c == 0.0f) return -1;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i) {
        float* g = px_grads + (size_t)i * P;
        double ss = 0.0;
        for (int j = 0; j < P; ++j) ss += (double)g[j] * (double)g[j];
        float norm = (float)sqrt(ss);
        float ratio = norm / c;
        float scale = 1.0f / (ratio != ratio ? ratio : fmaxf(1.0f, ratio)); /* jnp.maximum propagates NaN; fmaxf would not */
        for (int j = 0; j < P; ++j) g[j] *= scale;
    }
    return 0;
}

D3P_API float d3po_full_norm(const float* v, uint64_t n)
{
    double ss = 0.0;
    for (uint64_t j = 0; j < n; ++j) ss += (double)v[j] * (double)v[j];
    return (float)sqrt(ss);
}

/* svi.py:327-348: mean over the (padded) batch axis of every column, and of the losses. */
D3P_API float d3po_combine(const float* px_grads, const float* px_loss, int B, int P, float* avg)
{
#pragma omp parallel for schedule(static)
    for (int j = 0; j < P; ++j) {
        double s = 0.0;
        for (int i = 0; i < B; ++i) s += (double)px_grads[(size_t)i * P + j];
        avg[j] = (float)(s / B);
    }
    double l = 0.0;
    for (int i = 0; i < B; ++i) l += (double)px_loss[i];
    return (float)(l / B);
}

/* svi.py:350-377 + perturbation_function svi.py:470-498.  `site_sizes` lists the sites in
 * tree_flatten order; site k uses split(key, n_sites)[k] (svi.py:491) and normal(site_key, shape)
 * (svi.py:487).  scale = dp_scale * C / n (svi.py:365-366); the result is multiplied by
 * obs_scale * factor (svi.py:375).  n == 0 reproduces the reference's inf/NaN. */
D3P_API void d3po_perturb(const uint32_t key[16], const float* avg, const int32_t* site_sizes,
                          int n_sites, float dp_scale, float c, float num_elements, float obs_scale,
                          float factor, float* out)
{
    uint32_t* ks = (uint32_t*)malloc((size_t)n_sites * 16 * sizeof(uint32_t));
    d3po_split(key, n_sites, ks);
    float scale = dp_scale * (c / num_elements);
    size_t off = 0;
    for (int k = 0; k < n_sites; ++k) {
        float* z = (float*)malloc(((size_t)site_sizes[k] + 1) * sizeof(float));
        d3po_normal(ks + 16 * k, (uint64_t)site_sizes[k], z);
        for (int j = 0; j < site_sizes[k]; ++j)
            out[off + j] = (avg[off + j] + z[j] * scale) * obs_scale * factor;
        off += (size_t)site_sizes[k];
        free(z);
    }
    free(ks);
}

/* d3p/optimizers.py:29-112 -- ADADP (Koskela & Honkela, arXiv:1809.03832) on the flat parameter vector.
 * State: x, lr, x_stepped, x_prev; `i` is the step count before the update.
 *   every step : new_x = x - 0.5 lr g                                             (optimizers.py:100)
 *   even i     : x_prev = x; x_stepped = x - lr g; x = new_x                      (:62-69)
 *   odd i      : err = sqrt(sum(((x_stepped - new_x) / max(1, x_stepped))^2))     (:75-87; max(1, x), not |x|)
 *                lr *= min(max(sqrt(tol / err), 0.9), 1.1)                        (:89-91; the bounds are literals
 *                                                                                  there, alpha_min/max are unused)
 *                x = (stability_check && err > tol) ? x_prev : new_x              (:93-98)
 * Pinned by the known answers of tests/test_adadp_optimizer.py:66-131. */
D3P_API void d3po_adadp(float* x, float* lr, float* x_stepped, float* x_prev, const float* g, int P, int i, float tol,
                        int stability_check)
{
    const float l = *lr;
    if ((i & 1) == 0) {
        for (int j = 0; j < P; ++j) {
            x_prev[j] = x[j];
            x_stepped[j] = x[j] - l * g[j];
            x[j] = x[j] - (0.5f * l) * g[j];
        }
        return;
    }
    double ss = 0.0;
    for (int j = 0; j < P; ++j) {
        float nx = x[j] - (0.5f * l) * g[j];
        float e = (x_stepped[j] - nx) / fmaxf(1.0f, x_stepped[j]);
        ss += (double)e * (double)e;
    }
    const float err = (float)sqrt(ss);
    *lr = l * fminf(fmaxf(sqrtf(tol / err), 0.9f), 1.1f);
    const int reject = stability_check && err > tol;
    for (int j = 0; j < P; ++j) x[j] = reject ? x_prev[j] : x[j] - (0.5f * l) * g[j];
}

/* numpyro.optim.Adam == jax.example_libraries.optimizers.adam(step, 0.9, 0.999, 1e-8)
 * (svi.py:379-393; examples/logistic_regression.py:141).  `i` is the step count before the update. */
D3P_API void d3po_adam(float* x, float* m, float* v, const float* g, int P, int i, float lr, float b1,
                       float b2, float eps)
{
    float c1 = 1.0f - powf(b1, (float)(i + 1)), c2 = 1.0f - powf(b2, (float)(i + 1));
    for (int j = 0; j < P; ++j) {
        m[j] = (1.0f - b1) * g[j] + b1 * m[j];
        v[j] = (1.0f - b2) * g[j] * g[j] + b2 * v[j];
        float mhat = m[j] / c1, vhat = v[j] / c2;
        x[j] = x[j] - lr * mhat / (sqrtf(vhat) + eps);
    }
}

/* ------------------------------------------------------------------------------------------
 * One full DPSVI.update (svi.py:395-434) on an explicit batch, following the reference dataflow
 * stage by stage (materialised B x P like jax.vmap).  Used by the parity tests and as bench.py's
 * cpu_baseline ("port").  params = [loc (D) | unc (D)], adam m/v same layout.
 * Keys: state_key -> split(.,3) = next, gradient, perturbation (svi.py:208-211, :413-414);
 * jax key = random_bits(gradient_key, 32, (2,)) (random/__init__.py:155).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    float clip;      /* clipping_threshold */
    float dp_scale;  /* sigma */
    float lr, b1, b2, adam_eps;
} d3po_dpsvi_hyper;

D3P_API float d3po_logreg_update(const d3po_logreg_spec* sp, const d3po_dpsvi_hyper* hy,
                                 uint32_t state_key[16], float* params, float* adam_m, float* adam_v,
                                 int32_t* adam_step, const float* Xb, const float* yb,
                                 const float* mask, int B, const float* eps_ext /* B x D or NULL */,
                                 float* scratch /* B*(2D) + B + B*D + 4D floats */,
                                 float* grad_out /* 2D or NULL */)
{
    const int D = sp->d + (sp->intercept ? 1 : 0), P = 2 * D;
    uint32_t ks[48], jaxkey[2];
    d3po_split(state_key, 3, ks);
    d3po_random_words(ks + 16, 0, 2, jaxkey);
    float* px_grads = scratch;
    float* px_loss = px_grads + (size_t)B * P;
    float* eps = px_loss + B;
    float* avg = eps + (size_t)B * D;
    float* pert = avg + P;
    if (eps_ext) {
        memcpy(eps, eps_ext, (size_t)B * D * sizeof(float));
    } else {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < B; ++i) {
            uint32_t sk[2];
            d3po_px_sample_key(jaxkey, (uint32_t)B, (uint32_t)i, sk);
            d3po_tf_normal(sk, (uint64_t)D, eps + (size_t)i * D);
        }
    }
    float factor;
    int n = d3po_logreg_px_grads(sp, params, params + D, Xb, yb, eps, mask, B, px_loss, px_grads,
                                 &factor);
    d3po_clip_rows(px_grads, B, P, hy->clip);
    float loss = d3po_combine(px_grads, px_loss, B, P, avg);
    int32_t sites[2] = {D, D};
    d3po_perturb(ks + 32, avg, sites, 2, hy->dp_scale, hy->clip, (float)n, 1.0f / sp->inv_obs, factor,
                 pert);
    if (grad_out) memcpy(grad_out, pert, (size_t)P * sizeof(float));
    d3po_adam(params, adam_m, adam_v, pert, P, *adam_step, hy->lr, hy->b1, hy->b2, hy->adam_eps);
    *adam_step += 1;
    memcpy(state_key, ks, 16 * sizeof(uint32_t));
    return loss;
}

/* ------------------------------------------------------------------------------------------
 * bench.py's cpu_baseline ("port"): `steps` consecutive DPSVI.update calls (svi.py:395-434) on Feistel minibatches
 * (minibatch.py:217-237: fold_in, sample_from_array, take) drawn from a RESIDENT table, entirely in C -- no numpy gather,
 * no Python between steps -- on `threads` OpenMP threads (1 = the single-thread figure SURVEY 8(d) asks for).
 * The dataflow is the reference's, stage by stage: gather the batch (`jnp.take`: B x d copied), per-example noise from
 * the threefry keys (:289-290), per-example gradients MATERIALISED as a B x P tensor (what jax.vmap yields, :299), a
 * clipping pass over its rows (:321), the mean over the batch axis (:343), perturbation (:470-498), Adam (:379-393).
 * Arithmetic is float32 like the reference's; the per-column guide terms softplus(u), sigmoid(u), log s are computed once
 * per step (XLA hoists them out of the vmapped body), unlike d3po_logreg_px_loss_grad above, which is the accurate
 * float64 checker and recomputes them per example.  Logistic regression without intercept, softplus guide.
 * tests/test_oracle_pins.py checks this loop against the step-by-step composition of the checker functions.
 * scratch: B*d + B + B*2D + B + B*D + 8*D floats, idx: B words.  Returns the loss of the last step. */
D3P_API float d3po_logreg_run_feistel(const d3po_logreg_spec* sp, const d3po_dpsvi_hyper* hy, uint32_t state_key[16],
                                      float* params, float* adam_m, float* adam_v, int32_t* adam_step, const float* X,
                                      const float* y, uint32_t n_rows, const uint32_t batch_key[16], uint32_t first_batch,
                                      int B, int steps, int threads, float* scratch, uint32_t* idx)
{
    const int d = sp->d, D = d, P = 2 * D;
    if (threads < 1) threads = 1;
    float* Xb = scratch;
    float* yb = Xb + (size_t)B * d;
    float* px_grads = yb + B;
    float* px_loss = px_grads + (size_t)B * P;
    float* eps = px_loss + B;
    float* col = eps + (size_t)B * D; /* s | sg | q | lc  (4D) */
    float* avg = col + 4 * (size_t)D;
    float* pert = avg + P;
    const float inv_obs = sp->inv_obs, obs = 1.0f / sp->inv_obs;
    const float c1 = inv_obs / (sp->prior_w * sp->prior_w), hz = 0.5f / (sp->prior_w * sp->prior_w);
    const float A_scale = inv_obs * sp->lik_scale;
    float loss = 0.0f;
    for (int t = 0; t < steps; ++t) {
        uint32_t ks[48], jaxkey[2], bk[16], rc[30];
        d3po_split(state_key, 3, ks);
        d3po_random_words(ks + 16, 0, 2, jaxkey);
        d3po_fold_in(batch_key, first_batch + (uint32_t)t, bk);
        d3po_feistel_constants(bk, rc);
        const float* loc = params;
        for (int j = 0; j < D; ++j) {
            float s, sg;
            guide_scale(sp, params[D + j], &s, &sg);
            col[j] = s;
            col[D + j] = sg;
            col[2 * D + j] = inv_obs * sg / s;
            col[3 * D + j] = logf(sp->prior_w) - logf(s);
        }
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int i = 0; i < B; ++i) {
            /* minibatch.py:231-233: Feistel index, take */
            const uint32_t r = d3po_feistel_permute(rc, n_rows, (uint32_t)i);
            idx[i] = r;
            memcpy(Xb + (size_t)i * d, X + (size_t)r * d, (size_t)d * sizeof(float));
            yb[i] = y[r];
        }
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int i = 0; i < B; ++i) {
            uint32_t sk[2];
            d3po_px_sample_key(jaxkey, (uint32_t)B, (uint32_t)i, sk);
            float* e = eps + (size_t)i * D;
            d3po_tf_normal(sk, (uint64_t)D, e);
            const float* x = Xb + (size_t)i * d;
            float* g = px_grads + (size_t)i * P;
            float tl = 0.0f, lp = 0.0f;
            for (int j = 0; j < D; ++j) {
                const float z = fmaf(col[j], e[j], loc[j]);
                g[j] = z; /* z kept in the gradient row until the logit is known */
                tl = fmaf(x[j], z, tl);
                lp += fmaf(hz * z, z, fmaf(-0.5f * e[j], e[j], col[3 * D + j]));
            }
            const float A = A_scale * (sigmoid_f(tl) - yb[i]);
            for (int j = 0; j < D; ++j) {
                const float gz = fmaf(c1, g[j], A * x[j]);
                g[j] = gz;
                g[D + j] = fmaf(gz * e[j], col[D + j], -col[2 * D + j]);
            }
            const float loglik = yb[i] * tl - softplus_f(tl);
            px_loss[i] = inv_obs * (lp - sp->lik_scale * loglik) * obs; /* factor = 1: no padding (svi.py:305-306) */
        }
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int i = 0; i < B; ++i) { /* svi.py:321: clip_gradient per row */
            float* g = px_grads + (size_t)i * P;
            float ss = 0.0f;
            for (int j = 0; j < P; ++j) ss = fmaf(g[j], g[j], ss);
            const float ratio = sqrtf(ss) / hy->clip;
            const float scale = 1.0f / (ratio != ratio ? ratio : fmaxf(1.0f, ratio));
            for (int j = 0; j < P; ++j) g[j] *= scale;
        }
        /* svi.py:343-346: mean over the batch axis (threads own column blocks, rows streamed) */
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int jb = 0; jb < P; jb += 64) {
            float acc[64];
            const int w = (P - jb) < 64 ? (P - jb) : 64;
            for (int j = 0; j < w; ++j) acc[j] = 0.0f;
            for (int i = 0; i < B; ++i) {
                const float* g = px_grads + (size_t)i * P + jb;
                for (int j = 0; j < w; ++j) acc[j] += g[j];
            }
            for (int j = 0; j < w; ++j) avg[jb + j] = acc[j] / (float)B;
        }
        double l = 0.0;
        for (int i = 0; i < B; ++i) l += (double)px_loss[i];
        loss = (float)(l / B);
        int32_t sites[2] = {D, D};
        d3po_perturb(ks + 32, avg, sites, 2, hy->dp_scale, hy->clip, (float)B, obs, 1.0f, pert);
        d3po_adam(params, adam_m, adam_v, pert, P, *adam_step, hy->lr, hy->b1, hy->b2, hy->adam_eps);
        *adam_step += 1;
        memcpy(state_key, ks, 16 * sizeof(uint32_t));
    }
    return loss;
}

/* DPSVI.evaluate (svi.py:436-449) -> numpyro SVI.evaluate: `_, key = split(rng_key)`; -Trace_ELBO on the whole
 * batch with ONE guide draw: model_seed, guide_seed = split(key); sample key = split(guide_seed)[1]
 * (numpyro.handlers.seed); plate(N, B) scales the likelihood by N / B.  (numpyro plumbing UNPINNED.) */
D3P_API float d3po_logreg_evaluate_sites(const d3po_logreg_spec* sp, const float* loc, const float* unc, const float* Xb,
                                         const float* yb, int B, const uint32_t jax_key[2], const int32_t* site_sizes, int n_sites);

D3P_API float d3po_logreg_evaluate(const d3po_logreg_spec* sp, const float* loc, const float* unc, const float* Xb,
                                   const float* yb, int B, const uint32_t jax_key[2])
{
    const int32_t D = sp->d + (sp->intercept ? 1 : 0);
    return d3po_logreg_evaluate_sites(sp, loc, unc, Xb, yb, B, jax_key, &D, 1);   /* ONE site: `_auto_latent` / the one-site guides */
}

/* The same with the guide's sample sites in program order (examples/logistic_regression.py:67-86: 'w' (d), 'intercept' (1)): the seed
 * handler advances rng, site_key = split(rng) per sample statement; loc / unc in the kernels' order = site order.  UNPINNED plumbing. */
D3P_API float d3po_logreg_evaluate_sites(const d3po_logreg_spec* sp, const float* loc, const float* unc, const float* Xb,
                                         const float* yb, int B, const uint32_t jax_key[2], const int32_t* site_sizes, int n_sites)
{
    const int d = sp->d, D = sp->d + (sp->intercept ? 1 : 0);
    uint32_t s[4], k[2];
    d3po_tf_split(jax_key, 2, s);
    k[0] = s[2]; k[1] = s[3];                 /* rng_key_eval */
    d3po_tf_split(k, 2, s);
    k[0] = s[2]; k[1] = s[3];                 /* guide_seed */
    float* eps = (float*)malloc((size_t)D * sizeof(float));
    float* z = (float*)malloc((size_t)D * sizeof(float));
    for (int site = 0, off = 0; site < n_sites; ++site) {
        d3po_tf_split(k, 2, s);               /* rng, key of this sample site */
        uint32_t sk[2] = {s[2], s[3]};
        k[0] = s[0]; k[1] = s[1];
        d3po_tf_normal(sk, (uint64_t)site_sizes[site], eps + off);
        off += site_sizes[site];
    }
    double lq = 0.0, lp = 0.0;
    for (int j = 0; j < D; ++j) {
        float sc, dsc;
        guide_scale(sp, unc[j], &sc, &dsc);
        float ps = (j < d) ? sp->prior_w : sp->prior_b;
        z[j] = fmaf(sc, eps[j], loc[j]);
        lq += -0.5 * (double)eps[j] * eps[j] - log((double)sc) - (double)HALF_LOG_2PI;
        lp += -0.5 * ((double)z[j] / ps) * ((double)z[j] / ps) - log((double)ps) - (double)HALF_LOG_2PI;
    }
    double ll = 0.0;
    for (int i = 0; i < B; ++i) {
        if (sp->family == 1) {
            double t = 0.0;
            for (int j = 0; j < d; ++j) { double r = (double)Xb[(size_t)i * d + j] - (double)z[j]; t += r * r; }
            ll += -0.5 * t / ((double)sp->lik_sigma * sp->lik_sigma) - d * (log((double)sp->lik_sigma) + (double)HALF_LOG_2PI);
            continue;
        }
        double t = sp->intercept ? (double)z[d] : 0.0;
        for (int j = 0; j < d; ++j) t += (double)Xb[(size_t)i * d + j] * (double)z[j];
        ll += (double)yb[i] * t - (double)softplus_f((float)t);
    }
    free(eps); free(z);
    return (float)(-(lp + ((double)sp->lik_scale / B) * ll - lq));
}

/* jax.random.randint for 32-bit integers (d3p/random/debug.py:39 `randint = jrng.randint`), jax <= 0.4.10:
 * k1, k2 = split(key); offset = ((bits(k1) % span) * (2^32 % span) + bits(k2) % span) % span, with
 * 2^32 % span computed as ((2^16 % span)^2) % span in uint32.  (UNPINNED: restated from memory of jax/_src/random.py.) */
D3P_API void d3po_tf_randint32(const uint32_t key[2], uint64_t n, int32_t minval, int32_t maxval, int32_t* out)
{
    uint32_t ks[4];
    d3po_tf_split(key, 2, ks);
    uint32_t span = (maxval <= minval) ? 1u : (uint32_t)maxval - (uint32_t)minval;
    uint32_t mult = 65536u % span;
    mult = (mult * mult) % span;
    for (uint64_t j = 0; j < n; ++j) {
        uint32_t hi = tf_iota_word(ks[0], ks[1], n, j), lo = tf_iota_word(ks[2], ks[3], n, j);
        uint32_t off = ((hi % span) * mult + (lo % span)) % span;
        out[j] = (int32_t)((uint32_t)minval + off);
    }
}

/* GaussianMixture.log_prob (d3p/gmm.py:71-86): logsumexp_k( log pi_k + sum_d log N(x_d; mu_kd, sigma_kd) ) per row. */
D3P_API void d3po_gmm_log_prob(const float* x, int B, int d, const float* locs, const float* scales, const float* pis,
                               int K, float* out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i) {
        double best = -INFINITY;
        double* comp = (double*)malloc((size_t)K * sizeof(double));
        for (int k = 0; k < K; ++k) {
            double s = log((double)pis[k]);
            for (int j = 0; j < d; ++j) {
                double sc = scales[(size_t)k * d + j], z = ((double)x[(size_t)i * d + j] - locs[(size_t)k * d + j]) / sc;
                s += -0.5 * z * z - log(sc) - (double)HALF_LOG_2PI;
            }
            comp[k] = s;
            if (s > best) best = s;
        }
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += exp(comp[k] - best);
        out[i] = (float)(best + log(acc));
        free(comp);
    }
}

/* ------------------------------------------------------------------------------------------
 * DP-VI per-example gradient for the Gaussian-mixture MODEL of BASELINE config 3
 * (examples/gaussian_mixture_model.py:51-85):
 *   model: pis ~ Dirichlet(1_K); mus ~ Normal(0, 10)^(K x d); sigs ~ InverseGamma(1, 1)^(K x d);
 *          obs_i ~ GaussianMixture(mus, sigs, pis)                      (d3p/gmm.py:71-86)
 *   guide: pis ~ Dirichlet(exp(alpha_log)); mus ~ Normal(mus_loc, 1); sigs ~ InverseGamma(1, 1)
 *   params (tree_flatten order): alpha_log (K), mus_loc (K x d);  P = K + K d.
 * Per example (svi.py:271-281, one guide draw each): L = inv_obs * ((log q - log p)(latents) - N loglik(x | latents)).
 * The InverseGamma terms of log q and log p are the same number and cancel.  Gradients are pathwise: mus = loc + eps,
 * pis = g / sum(g) with g_k ~ Gamma(alpha_k) and dg_k/dalpha_k the implicit-reparametrisation derivative at fixed CDF
 * value (what jax.random.gamma's JVP rule computes).
 *
 * PARITY UNPINNED for the sampling layout: the seed handler's key chain over the guide's three sample sites and the
 * normal layout of `mus` follow numpyro / jax (k_pis = split(r0)[1], r1 = split(r0)[0], k_mus = split(r1)[1], ...), but
 * jax.random.gamma's rejection sampler (per-element key splitting inside a while_loop) is NOT reproduced bit for bit:
 * this build draws  g_k  by Marsaglia-Tsang with (normal, uniform) = threefry2x32(k_pis, (k, attempt)) and
 * sigs = 1 / Exponential(1) with the uniforms of threefry(k_sigs, iota(K d)).  Same distributions, different streams.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t K, d;
    float prior_mu_scale; /* 10 in the example */
    float lik_scale;      /* plate scale N */
    float inv_obs;        /* 1 / observation_scale */
} d3po_gmm_spec;

/* digamma for x > 0: recurrence up to x >= 10, then the asymptotic series */
D3P_API double d3po_digamma(double x)
{
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x
           - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760))))));
}

/* dx/dalpha of x = F^{-1}(u; alpha) for the Gamma(alpha, 1) CDF F at fixed u:  -(dF/dalpha) / f.
 * With P(alpha, x) = x^alpha e^-x S / Gamma(alpha + 1), S = sum_n t_n, t_n = x^n / ((alpha+1)...(alpha+n)):
 *   dP/dalpha = P (log x - psi(alpha + 1) + S'/S),  S' = -sum_n t_n sum_{j<=n} 1/(alpha + j),  P / f = x S / alpha
 * so  dx/dalpha = -(x / alpha) (S (log x - psi(alpha + 1)) + S').  The series is used for all x (terms up to n ~ x + 60);
 * cancellation grows like e^x, fine for the x <~ 40 a Gamma with alpha = O(1..10) produces. */
D3P_API double d3po_gamma_grad(double alpha, double x)
{
    if (!(x > 0.0)) return 0.0;
    double t = 1.0, h = 0.0, S = 1.0, Sp = 0.0;
    for (int n = 1; n < 2000; ++n) {
        t *= x / (alpha + n);
        h += 1.0 / (alpha + n);
        S += t;
        Sp -= t * h;
        if (t < 1e-18 * S && n > x) break;
    }
    return -(x / alpha) * (S * (log(x) - d3po_digamma(alpha + 1.0)) + Sp);
}

static inline double bits_to_open_unit_d(uint32_t b) { return ((double)b + 0.5) * (1.0 / 4294967296.0); }

/* Gamma(alpha, 1) draw for mixture component `comp`: Marsaglia & Tsang (2000); alpha < 1 boosted by U^(1/alpha). */
D3P_API double d3po_gamma_sample(const uint32_t key[2], uint32_t comp, double alpha)
{
    /* NaN / +inf concentration (a diverged state): every acceptance test is false; jax.random.gamma's while_loop condition is false
     * on NaN as well and returns NaN at once.  The loop is bounded in any case (acceptance >= 0.95 per attempt). */
    if (!(alpha < 1.7976931348623157e308)) return alpha;
    const double a = alpha < 1.0 ? alpha + 1.0 : alpha;
    const double dd = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
    double g = 0.0;
    for (uint32_t attempt = 0; attempt < 1024u; ++attempt) {
        uint32_t b[2];
        d3po_threefry2x32(key[0], key[1], comp, attempt, b);
        const double x = (double)bits_to_normal(b[0]);
        const double U = bits_to_open_unit_d(b[1]);
        const double v1 = 1.0 + c * x;
        if (v1 <= 0.0) continue;
        const double v = v1 * v1 * v1;
        if (log(U) < 0.5 * x * x + dd - dd * v + dd * log(v)) { g = dd * v; break; }
    }
    if (alpha < 1.0) {
        uint32_t b[2];
        d3po_threefry2x32(key[0], key[1], comp, 0x80000000u, b);
        g *= pow(bits_to_open_unit_d(b[0]), 1.0 / alpha);
    }
    return g;
}

/* Keys of the guide's three sample sites for the example at batch position p (numpyro.handlers.seed: every sample
 * statement does key, sample_key = split(key)):  r0 = guide_seed = split(px_key)[1];  k_pis = split(r0)[1];
 * r1 = split(r0)[0]; k_mus = split(r1)[1]; r2 = split(r1)[0]; k_sigs = split(r2)[1]. */
D3P_API void d3po_gmm_site_keys(const uint32_t jax_key[2], uint32_t B, uint32_t p, uint32_t out[6])
{
    uint32_t px[2], s[4], r[2];
    px[0] = tf_iota_word(jax_key[0], jax_key[1], 2ull * B, 2ull * p);
    px[1] = tf_iota_word(jax_key[0], jax_key[1], 2ull * B, 2ull * p + 1);
    d3po_tf_split(px, 2, s);
    r[0] = s[2]; r[1] = s[3];
    for (int site = 0; site < 3; ++site) {
        d3po_tf_split(r, 2, s);
        out[2 * site] = s[2]; out[2 * site + 1] = s[3];
        r[0] = s[0]; r[1] = s[1];
    }
}

/* Latent draws of one example: g[K] (Gamma(alpha_k)), eps[K d] (standard normals of mus), sigs[K d]. */
D3P_API void d3po_gmm_px_latents(const d3po_gmm_spec* sp, const float* alpha_log, const uint32_t jax_key[2], uint32_t B,
                                 uint32_t p, double* g, float* eps, float* sigs)
{
    const int K = sp->K, n = sp->K * sp->d;
    uint32_t sk[6];
    d3po_gmm_site_keys(jax_key, B, p, sk);
    for (int k = 0; k < K; ++k) g[k] = d3po_gamma_sample(sk, (uint32_t)k, exp((double)alpha_log[k]));
    d3po_tf_normal(sk + 2, (uint64_t)n, eps);
    for (int j = 0; j < n; ++j) {
        /* Exponential(1) = Gamma(1, 1) by inversion in float32: u = (m + 1/2) 2^-23 in [2^-24, 1 - 2^-24] is exact,
         * so e = -log(u) lies in [6e-8, 16.7] and sigs = 1 / e is finite */
        const uint32_t b = tf_iota_word(sk[4], sk[5], (uint64_t)n, (uint64_t)j);
        const float u = ((float)(b >> 9) + 0.5f) * 1.1920928955078125e-07f;
        sigs[j] = 1.0f / -logf(u);
    }
}

/* Loss and gradient of one example given its latent draws; grad = [d/dalpha_log (K) | d/dmus_loc (K d)] or NULL. */
D3P_API float d3po_gmm_px_loss_grad_given(const d3po_gmm_spec* sp, const float* alpha_log, const float* mus_loc,
                                          const float* x, const double* g, const float* eps, const float* sigs,
                                          float mask, float* grad)
{
    const int K = sp->K, d = sp->d;
    const double ps = sp->prior_mu_scale, N = sp->lik_scale, io = sp->inv_obs;
    double* alpha = (double*)malloc(sizeof(double) * 4 * (size_t)K);
    double *pis = alpha + K, *a = pis + K, *r = a + K;
    double S = 0.0, A0 = 0.0;
    for (int k = 0; k < K; ++k) { alpha[k] = exp((double)alpha_log[k]); A0 += alpha[k]; S += g[k]; }
    double lq = lgamma(A0) - lgamma((double)K), lmu = 0.0;
    for (int k = 0; k < K; ++k) {
        pis[k] = g[k] / S;
        lq += -lgamma(alpha[k]) + (alpha[k] - 1.0) * log(pis[k]);
    }
    double best = -INFINITY;
    for (int k = 0; k < K; ++k) {
        double ll = 0.0;
        for (int j = 0; j < d; ++j) {
            const size_t e = (size_t)k * d + j;
            const float mu = mus_loc[e] + eps[e];
            const double z = ((double)x[j] - mu) / sigs[e];
            ll += -0.5 * z * z - log((double)sigs[e]) - (double)HALF_LOG_2PI;
            lmu += -0.5 * (double)eps[e] * eps[e] - (-0.5 * ((double)mu / ps) * ((double)mu / ps) - log(ps));
        }
        a[k] = log(pis[k]) + ll;
        if (a[k] > best) best = a[k];
    }
    double se = 0.0;
    for (int k = 0; k < K; ++k) { r[k] = exp(a[k] - best); se += r[k]; }
    const double loglik = best + log(se);
    for (int k = 0; k < K; ++k) r[k] /= se;
    const double L = io * ((lq + lmu) - N * loglik) * mask;
    if (grad) {
        const double psi0 = d3po_digamma(A0);
        for (int k = 0; k < K; ++k) {
            const double gp = d3po_gamma_grad(alpha[k], g[k]) / S; /* (dg_k/dalpha_k) / S */
            const double dq = psi0 - d3po_digamma(alpha[k]) + log(pis[k]) + gp * ((alpha[k] - 1.0) / pis[k] - (A0 - K));
            const double dl = gp * (r[k] / pis[k] - 1.0);
            grad[k] = (float)(alpha[k] * io * (dq - N * dl) * mask);
            for (int j = 0; j < d; ++j) {
                const size_t e = (size_t)k * d + j;
                const float mu = mus_loc[e] + eps[e];
                const double w = ((double)x[j] - mu) / ((double)sigs[e] * sigs[e]);
                grad[K + e] = (float)(io * ((double)mu / (ps * ps) - N * r[k] * w) * mask);
            }
        }
    }
    free(alpha);
    return (float)L;
}

/* svi.py:238-308 for the mixture model: px_losses (B) and px_grads (B x P), latents drawn per example from jax_key. */
D3P_API int d3po_gmm_px_grads(const d3po_gmm_spec* sp, const float* params, const float* Xb, const float* mask, int B,
                              const uint32_t jax_key[2], float* px_loss, float* px_grads, float* factor)
{
    const int K = sp->K, d = sp->d, P = K + K * d;
    int n = 0;
    for (int i = 0; i < B; ++i) n += (mask ? (mask[i] != 0.0f) : 1);
    const float f = (n == 0) ? 0.0f : (float)B / (float)n;
    const float obs = 1.0f / sp->inv_obs;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i) {
        double* g = (double*)malloc(sizeof(double) * (size_t)K);
        float* eps = (float*)malloc(sizeof(float) * 2 * (size_t)K * d);
        float* sigs = eps + (size_t)K * d;
        d3po_gmm_px_latents(sp, params, jax_key, (uint32_t)B, (uint32_t)i, g, eps, sigs);
        const float m = mask ? mask[i] : 1.0f;
        const float L = d3po_gmm_px_loss_grad_given(sp, params, params + K, Xb + (size_t)i * d, g, eps, sigs, m,
                                                    px_grads + (size_t)i * P);
        px_loss[i] = L * obs * f;
        free(g); free(eps);
    }
    *factor = f;
    return n;
}

/* DPSVI.evaluate (svi.py:436-449) for the mixture model: -ELBO of the batch with ONE guide draw.
 * rng_key_eval = split(jax_key)[1]; guide_seed = split(rng_key_eval)[1]; site keys as in d3po_gmm_site_keys;
 * plate(N, B) scales the likelihood by N / B. */
D3P_API float d3po_gmm_evaluate(const d3po_gmm_spec* sp, const float* params, const float* Xb, int B,
                                const uint32_t jax_key[2])
{
    const int K = sp->K, d = sp->d, n = K * d;
    uint32_t s[4], r[2], sk[6];
    d3po_tf_split(jax_key, 2, s);
    r[0] = s[2]; r[1] = s[3];
    d3po_tf_split(r, 2, s);
    r[0] = s[2]; r[1] = s[3];
    for (int site = 0; site < 3; ++site) {
        d3po_tf_split(r, 2, s);
        sk[2 * site] = s[2]; sk[2 * site + 1] = s[3];
        r[0] = s[0]; r[1] = s[1];
    }
    double* g = (double*)malloc(sizeof(double) * (size_t)K);
    float* eps = (float*)malloc(sizeof(float) * 4 * (size_t)n + sizeof(float) * (size_t)K + sizeof(float) * (size_t)B);
    float *sigs = eps + n, *mus = sigs + n, *pis = mus + n, *ll = pis + K;
    double S = 0.0, A0 = 0.0, lat = 0.0;
    for (int k = 0; k < K; ++k) {
        const double alpha = exp((double)params[k]);
        g[k] = d3po_gamma_sample(sk, (uint32_t)k, alpha);
        S += g[k];
        A0 += alpha;
    }
    lat = lgamma(A0) - lgamma((double)K);
    for (int k = 0; k < K; ++k) {
        const double alpha = exp((double)params[k]);
        pis[k] = (float)(g[k] / S);
        lat += -lgamma(alpha) + (alpha - 1.0) * log(g[k] / S);
    }
    d3po_tf_normal(sk + 2, (uint64_t)n, eps);
    for (int j = 0; j < n; ++j) {
        const uint32_t b = tf_iota_word(sk[4], sk[5], (uint64_t)n, (uint64_t)j);
        const float u = ((float)(b >> 9) + 0.5f) * 1.1920928955078125e-07f;
        sigs[j] = 1.0f / -logf(u);
        mus[j] = params[K + j] + eps[j];
        const double ps = sp->prior_mu_scale;
        lat += -0.5 * (double)eps[j] * eps[j] - (-0.5 * ((double)mus[j] / ps) * ((double)mus[j] / ps) - log(ps));
    }
    d3po_gmm_log_prob(Xb, B, d, mus, sigs, pis, K, ll);
    double tot = 0.0;
    for (int i = 0; i < B; ++i) tot += (double)ll[i];
    free(g); free(eps);
    return (float)(lat - ((double)sp->lik_scale / B) * tot);
}

/* ------------------------------------------------------------------------------------------
 * DP-VI step sums for the variational auto-encoder of BASELINE config 5 (examples/vae.py:65-153):
 *   encoder (guide):  h1 = softplus(x W1 + b1);  z_loc = h1 Wl + bl;  z_std = exp(h1 Ws + bs);  z ~ Normal(z_loc, z_std)
 *   decoder (model):  z ~ Normal(0, 1);  h2 = softplus(z V1 + c1);  obs ~ Bernoulli(sigmoid(h2 V2 + c2))
 * Parameters in tree_flatten order of {'decoder$params', 'encoder$params'} (stax.Dense: W (in, out), b (out)):
 *   V1 (Z x H), c1 (H), V2 (H x D), c2 (D), W1 (D x H), b1 (H), Wl (H x Z), bl (Z), Ws (H x Z), bs (Z);
 *   D = 784, H = 400, Z = 50 -> P = 688 884.
 * Per example (z is local to the plate, so every term carries the same site scale `scale` = plate scale x the
 * example's handlers.scale(1 / N), vae.py:194-195):
 *   L_i = inv_obs * scale * ( log q(z_i | x_i) - log p(z_i) - log p(x_i | z_i) ),   z_i = z_loc + z_std * eps_i
 * with the Bernoulli log-likelihood evaluated on logits a: x a - softplus(a) (numpyro clamps the probabilities at
 * float32 eps instead; the two agree to that eps).  This function materialises every per-example gradient (P floats)
 * explicitly -- it is the check for the device path, which never does (norms by the outer-product identity
 * ||a d^T||_F = ||a|| ||d||, clipped sums as GEMMs).
 * sums[P + 2] = [sum_i c_i g_i | sum_i L_i mask_i | n];  norms[B] (optional) = ||g_i||_2 before clipping.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t D, H, Z;
    float scale;    /* site scale (1 in the example) */
    float inv_obs;  /* 1 / observation_scale */
    int32_t H2;     /* > 0: a second hidden layer on each side (BASELINE config 5's 784 -> [400, 200] -> 50 variant; the
                     * reference has one): encoder x -> H -> H2 -> heads, decoder z -> H2 -> H -> D; leaves
                     * V1 (Z x H2), c1, V2 (H2 x H), c2, V3 (H x D), c3, W1 (D x H), b1, W2 (H x H2), b2, Wl (H2 x Z), bl, Ws, bs */
} d3po_vae_spec;

/* the dense layers of the network in tree_flatten order: decoder dec[0 .. nh], encoder enc[0 .. nh - 1], then the heads */
typedef struct { int64_t W, b; int in, out; } d3po_dense;
typedef struct { int nh, HE; d3po_dense dec[3], enc[2]; int64_t Wl, bl, Ws, bs, P; } d3po_vae_net;

static d3po_vae_net d3po_vae_layers(const d3po_vae_spec* sp)
{
    d3po_vae_net n;
    memset(&n, 0, sizeof(n));
    n.nh = sp->H2 > 0 ? 2 : 1;
    const int hs[2] = {sp->H, sp->H2};
    n.HE = hs[n.nh - 1];
    int64_t off = 0;
    for (int l = 0; l <= n.nh; ++l) {
        d3po_dense* d = &n.dec[l];
        d->in = l == 0 ? sp->Z : hs[n.nh - l];
        d->out = l == n.nh ? sp->D : hs[n.nh - l - 1];
        d->W = off; off += (int64_t)d->in * d->out;
        d->b = off; off += d->out;
    }
    for (int l = 0; l < n.nh; ++l) {
        d3po_dense* e = &n.enc[l];
        e->in = l == 0 ? sp->D : hs[l - 1];
        e->out = hs[l];
        e->W = off; off += (int64_t)e->in * e->out;
        e->b = off; off += e->out;
    }
    n.Wl = off; off += (int64_t)n.HE * sp->Z;
    n.bl = off; off += sp->Z;
    n.Ws = off; off += (int64_t)n.HE * sp->Z;
    n.bs = off; off += sp->Z;
    n.P = off;
    return n;
}

D3P_API int64_t d3po_vae_num_params(const d3po_vae_spec* sp)
{
    return d3po_vae_layers(sp).P;
}

static double d3po_softplus(double t) { return fmax(t, 0.0) + log1p(exp(-fabs(t))); }
static double d3po_sigmoid(double t) { return 1.0 / (1.0 + exp(-t)); }

D3P_API void d3po_vae_step_sums(const d3po_vae_spec* sp, const float* params, const float* X, const float* mask, int B,
                                const float* eps /* B x Z */, float clip, float* sums, float* norms, float* px_loss)
{
    const int D = sp->D, Z = sp->Z;
    const d3po_vae_net N = d3po_vae_layers(sp);
    const int nh = N.nh, HE = N.HE;
    const int64_t P = N.P;
    int wmax = D > Z ? D : Z;
    if (sp->H > wmax) wmax = sp->H;
    if (sp->H2 > wmax) wmax = sp->H2;
    double* acc = (double*)calloc((size_t)P + 2, sizeof(double));
    const double sc = (double)sp->inv_obs * sp->scale;
#pragma omp parallel
    {
        double* g = (double*)malloc(sizeof(double) * (size_t)P);
        /* per layer: pre-activation, activation, delta (encoder and decoder), plus the latent vectors and the logits */
        double* buf = (double*)malloc(sizeof(double) * (size_t)(12 * wmax + 6 * Z + 2 * D));
        double *pre_e[2], *h_e[2], *d_e[2], *pre_d[2], *h_d[2], *d_d[2];
        {
            double* q = buf;
            for (int l = 0; l < 2; ++l) {
                pre_e[l] = q; q += wmax; h_e[l] = q; q += wmax; d_e[l] = q; q += wmax;
                pre_d[l] = q; q += wmax; h_d[l] = q; q += wmax; d_d[l] = q; q += wmax;
            }
        }
        double *zl = buf + 12 * wmax, *u = zl + Z, *z = u + Z, *dz = z + Z, *du = dz + Z, *sd = du + Z, *a = sd + Z, *da = a + D;
        double* xin = (double*)malloc(sizeof(double) * (size_t)D);
#pragma omp for schedule(dynamic)
        for (int i = 0; i < B; ++i) {
            const float m = mask ? mask[i] : 1.0f;
            const float* x = X + (size_t)i * D;
            const float* e = eps + (size_t)i * Z;
            for (int k = 0; k < D; ++k) xin[k] = (double)x[k];
            /* ---- encoder */
            {
                const double* in = xin;
                for (int l = 0; l < nh; ++l) {
                    const d3po_dense* L = &N.enc[l];
                    const float *W = params + L->W, *bv = params + L->b;
                    for (int j = 0; j < L->out; ++j) {
                        double t = bv[j];
                        for (int k = 0; k < L->in; ++k) t += in[k] * W[(size_t)k * L->out + j];
                        pre_e[l][j] = t;
                        h_e[l][j] = d3po_softplus(t);
                    }
                    in = h_e[l];
                }
            }
            const double* hlast = h_e[nh - 1];
            const float *Wl = params + N.Wl, *bl = params + N.bl, *Ws = params + N.Ws, *bs = params + N.bs;
            double lq = 0.0, lp = 0.0;
            for (int j = 0; j < Z; ++j) {
                double tl = bl[j], tu = bs[j];
                for (int k = 0; k < HE; ++k) { tl += hlast[k] * Wl[(size_t)k * Z + j]; tu += hlast[k] * Ws[(size_t)k * Z + j]; }
                zl[j] = tl; u[j] = tu; sd[j] = exp(tu);
                z[j] = tl + sd[j] * (double)e[j];
                lq += -0.5 * (double)e[j] * e[j] - tu - (double)HALF_LOG_2PI;
                lp += -0.5 * z[j] * z[j] - (double)HALF_LOG_2PI;
            }
            /* ---- decoder */
            {
                const double* in = z;
                for (int l = 0; l < nh; ++l) {
                    const d3po_dense* L = &N.dec[l];
                    const float *W = params + L->W, *bv = params + L->b;
                    for (int j = 0; j < L->out; ++j) {
                        double t = bv[j];
                        for (int k = 0; k < L->in; ++k) t += in[k] * W[(size_t)k * L->out + j];
                        pre_d[l][j] = t;
                        h_d[l][j] = d3po_softplus(t);
                    }
                    in = h_d[l];
                }
            }
            double ll = 0.0;
            {
                const d3po_dense* L = &N.dec[nh];
                const float *W = params + L->W, *bv = params + L->b;
                const double* in = h_d[nh - 1];
                for (int j = 0; j < D; ++j) {
                    double t = bv[j];
                    for (int k = 0; k < L->in; ++k) t += in[k] * W[(size_t)k * D + j];
                    a[j] = t;
                    ll += (double)x[j] * t - d3po_softplus(t);
                    da[j] = sc * (d3po_sigmoid(t) - (double)x[j]);   /* d(-ll)/da */
                }
            }
            const double L_i = sc * (lq - lp - ll);
            /* ---- backward, explicit per-example gradient: for a dense layer with input v and output delta d,
             * grad W = v d^T, grad b = d, delta of the input's pre-activation = (W d) . softplus' */
            {
                const double* delta = da;
                for (int l = nh; l >= 0; --l) {
                    const d3po_dense* L = &N.dec[l];
                    const float* W = params + L->W;
                    const double* in = l == 0 ? z : h_d[l - 1];
                    for (int k = 0; k < L->in; ++k) for (int j = 0; j < L->out; ++j) g[L->W + (int64_t)k * L->out + j] = in[k] * delta[j];
                    for (int j = 0; j < L->out; ++j) g[L->b + j] = delta[j];
                    if (l > 0) {
                        for (int k = 0; k < L->in; ++k) {
                            double t = 0.0;
                            for (int j = 0; j < L->out; ++j) t += delta[j] * W[(size_t)k * L->out + j];
                            d_d[l - 1][k] = t * d3po_sigmoid(pre_d[l - 1][k]);     /* softplus' = sigmoid */
                        }
                        delta = d_d[l - 1];
                    } else {
                        for (int k = 0; k < Z; ++k) {
                            double t = sc * z[k];                    /* from -log p(z) = z^2 / 2 */
                            for (int j = 0; j < L->out; ++j) t += delta[j] * W[(size_t)k * L->out + j];
                            dz[k] = t;
                            du[k] = t * sd[k] * (double)e[k] - sc;   /* z = zl + exp(u) eps;  log q contributes -u */
                        }
                    }
                }
            }
            for (int k = 0; k < HE; ++k) for (int j = 0; j < Z; ++j) {
                g[N.Wl + (int64_t)k * Z + j] = hlast[k] * dz[j];
                g[N.Ws + (int64_t)k * Z + j] = hlast[k] * du[j];
            }
            for (int j = 0; j < Z; ++j) { g[N.bl + j] = dz[j]; g[N.bs + j] = du[j]; }
            for (int k = 0; k < HE; ++k) {
                double t = 0.0;
                for (int j = 0; j < Z; ++j) t += dz[j] * Wl[(size_t)k * Z + j] + du[j] * Ws[(size_t)k * Z + j];
                d_e[nh - 1][k] = t * d3po_sigmoid(pre_e[nh - 1][k]);
            }
            for (int l = nh - 1; l >= 0; --l) {
                const d3po_dense* L = &N.enc[l];
                const float* W = params + L->W;
                const double* in = l == 0 ? xin : h_e[l - 1];
                const double* delta = d_e[l];
                for (int k = 0; k < L->in; ++k) for (int j = 0; j < L->out; ++j) g[L->W + (int64_t)k * L->out + j] = in[k] * delta[j];
                for (int j = 0; j < L->out; ++j) g[L->b + j] = delta[j];
                if (l > 0)
                    for (int k = 0; k < L->in; ++k) {
                        double t = 0.0;
                        for (int j = 0; j < L->out; ++j) t += delta[j] * W[(size_t)k * L->out + j];
                        d_e[l - 1][k] = t * d3po_sigmoid(pre_e[l - 1][k]);
                    }
            }
            double ss = 0.0;
            for (int64_t j = 0; j < P; ++j) ss += g[j] * g[j];
            const double nrm = sqrt(ss);
            if (norms) norms[i] = (float)(nrm * m);
            if (px_loss) px_loss[i] = (float)(L_i * m);
            if (m != 0.0f) {
                const double cf = 1.0 / fmax(1.0, nrm / clip);
#pragma omp critical
                {
                    for (int64_t j = 0; j < P; ++j) acc[j] += cf * g[j];
                    acc[P] += L_i;
                    acc[P + 1] += 1.0;
                }
            }
        }
        free(g); free(buf); free(xin);
    }
    for (int64_t j = 0; j < P + 2; ++j) sums[j] = (float)acc[j];
    free(acc);
}

/* DPSVI.evaluate for the VAE: -ELBO of the batch with one guide draw; eps (B x Z) = tf_normal(k_z, B Z) with
 * k_z = split(split(split(jax_key)[1])[1])[1]; every site is scaled by sp->scale (= handlers.scale x N / B). */
D3P_API float d3po_vae_evaluate(const d3po_vae_spec* sp, const float* params, const float* X, int B, const uint32_t jax_key[2])
{
    uint32_t s[4], k[2] = {jax_key[0], jax_key[1]};
    for (int lvl = 0; lvl < 3; ++lvl) { d3po_tf_split(k, 2, s); k[0] = s[2]; k[1] = s[3]; }
    float* eps = (float*)malloc(sizeof(float) * (size_t)B * sp->Z);
    d3po_tf_normal(k, (uint64_t)B * sp->Z, eps);
    const int64_t P = d3po_vae_num_params(sp);
    float* sums = (float*)malloc(sizeof(float) * ((size_t)P + 2));
    d3po_vae_step_sums(sp, params, X, NULL, B, eps, 1e30f, sums, NULL, NULL);
    const float loss = sums[P];
    free(eps); free(sums);
    return loss;
}

/* Synthetic logistic-regression table, element (r, c) a pure function of (seed, r, c) so that any
 * shard can be regenerated (mirrors examples/logistic_regression.py:88-104 in
 * distribution): X[r][c] = normal from threefry2x32((seed, 0x58), (r, c))[0];
 * w_true[c], b_true from key (seed, 0x57); y[r] = 1 if u(r) < sigmoid(x_r.w + b) with
 * u(r) = uniform from threefry2x32((seed, 0x59), (r, 0))[0]. */
D3P_API void d3po_synth_wtrue(uint32_t seed, int d, float* w_true /* d + 1 */)
{
    for (int c = 0; c <= d; ++c) {
        uint32_t o[2];
        d3po_threefry2x32(seed, 0x57u, (uint32_t)c, 0u, o);
        w_true[c] = bits_to_normal(o[0]);
    }
}

D3P_API void d3po_synth_logreg(uint32_t seed, uint64_t row0, uint64_t nrows, int d, float* X, float* y)
{
    float* w = (float*)malloc(((size_t)d + 1) * sizeof(float));
    d3po_synth_wtrue(seed, d, w);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)nrows; ++i) {
        uint64_t r = row0 + (uint64_t)i;
        double t = (double)w[d];
        for (int c = 0; c < d; ++c) {
            uint32_t o[2];
            d3po_threefry2x32(seed, 0x58u, (uint32_t)r, (uint32_t)c, o);
            float xv = bits_to_normal(o[0]);
            X[(size_t)i * d + c] = xv;
            t += (double)xv * (double)w[c];
        }
        uint32_t o[2];
        d3po_threefry2x32(seed, 0x59u, (uint32_t)r, 0u, o);
        float u = bits_to_uniform(o[0], 0.0f, 1.0f);
        y[i] = (u < sigmoid_f((float)t)) ? 1.0f : 0.0f;
    }
    free(w);
}
