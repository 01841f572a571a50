// rng_suite kernels (ChaCha20 CSPRNG and the threefry debug suite) and the minibatch samplers
// (Feistel permutation, Poisson selection, row gather) for gfx950, plus their C-ABI entry points.
#include "d3p_device.h"
#include "d3p_host.h"

namespace d3p {

char* last_error_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}

// ------------------------------------------------------------------------------------------
// ChaCha20 suite
// ------------------------------------------------------------------------------------------
__global__ void k_rng_derive(const uint32_t* __restrict__ key, int num, uint32_t data, uint32_t tag,
                             uint32_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num) return;
    uint32_t parent[16], child[16];
    load_key(key, parent);
    derive_child(parent, tag == D3P_TAG_SPLIT ? (uint32_t)i : 0u, data, tag, child);
#pragma unroll
    for (int w = 0; w < 16; ++w) out[16 * (size_t)i + w] = child[w];
}

// One thread = one 64-byte ChaCha block.  MODE 0: raw words, 1: uniform(lo,hi), 2: normal.
template <int MODE>
__global__ void k_rng_stream(const uint32_t* __restrict__ key, uint64_t n_blocks, uint64_t n_elems, float lo,
                             float hi, void* __restrict__ out)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    uint32_t k[16], o[16];
    load_key(key, k);
    keystream_block(k, (uint32_t)b, o);
    if (MODE == 0) {
        uint4* dst = reinterpret_cast<uint4*>(out) + 4 * b;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    } else {
        float* dst = reinterpret_cast<float*>(out);
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint64_t e = 16 * b + w;
            if (e < n_elems) dst[e] = (MODE == 1) ? bits_to_uniform(o[w], lo, hi) : bits_to_normal(o[w]);
        }
    }
}

// d3p.random._randint (d3p/random/__init__.py:108-146), one thread per element.  The reference's
// while_loop re-draws all lanes each round but only rejected lanes take the new value, so every
// element evolves independently: round r uses word j of round_key_r where
// (key_{r+1}, round_key_r) = split(key_r, 2).
__global__ void k_rng_randint(const uint32_t* __restrict__ key, uint64_t n, uint32_t delta, uint32_t bitmask,
                              int32_t minval, int32_t* __restrict__ out)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t cur[16], nxt[16], rk[16], o[16];
    load_key(key, cur);
    uint32_t u = 0;
    for (int round = 0; round < 4096; ++round) {
        derive_child(cur, 0u, 0u, D3P_TAG_SPLIT, nxt);
        derive_child(cur, 1u, 0u, D3P_TAG_SPLIT, rk);
        keystream_block(rk, (uint32_t)(j >> 4), o);
        uint32_t w = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) w = ((j & 15) == (uint64_t)t) ? o[t] : w;
        u = w & bitmask;
        if (u <= delta) break;
#pragma unroll
        for (int t = 0; t < 16; ++t) cur[t] = nxt[t];
    }
    out[j] = (int32_t)u + minval;
}

// The same for the 8-, 16-, 32- and 64-bit integer dtypes (d3p/random/__init__.py:115-123): element j of
// random_bits(round_key, nbits, shape) is the little-endian nbits-wide view of the keystream at byte j * nbits / 8.
template <typename U, typename V>
__global__ void k_rng_randint_bits(const uint32_t* __restrict__ key, uint64_t n, U delta, U bitmask, V minval, V* __restrict__ out)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    constexpr int NB = (int)sizeof(U);
    const uint64_t byte0 = j * NB;
    const uint32_t blk = (uint32_t)(byte0 >> 6), w0 = (uint32_t)(byte0 & 63) >> 2, sh = (uint32_t)(byte0 & 3) * 8u;
    uint32_t cur[16], nxt[16], rk[16], o[16];
    load_key(key, cur);
    U u = 0;
    for (int round = 0; round < 4096; ++round) {
        derive_child(cur, 0u, 0u, D3P_TAG_SPLIT, nxt);
        derive_child(cur, 1u, 0u, D3P_TAG_SPLIT, rk);
        keystream_block(rk, blk, o);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            lo = (w0 == (uint32_t)t) ? o[t] : lo;
            hi = (w0 + 1u == (uint32_t)t) ? o[t] : hi;
        }
        const unsigned long long raw = NB == 8 ? (((unsigned long long)hi << 32) | lo) : (unsigned long long)(lo >> sh);
        u = (U)raw & bitmask;
        if (u <= delta) break;
#pragma unroll
        for (int t = 0; t < 16; ++t) cur[t] = nxt[t];
    }
    out[j] = (V)((V)u + minval);  // vdtype(uvals) + minval, wrapping in the value dtype
}

// ------------------------------------------------------------------------------------------
// threefry (jax.random layouts)
// ------------------------------------------------------------------------------------------
// MODE 0 raw, 1 uniform, 2 normal.  Thread j < half produces words j and j + half from one call.
template <int MODE>
__global__ void k_tf_stream(const uint32_t* __restrict__ key, uint64_t n, float lo, float hi, void* __restrict__ out)
{
    const uint64_t half = (n + 1) >> 1;
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    const uint64_t j2 = j + half;
    uint32_t a, b;
    threefry2x32(key[0], key[1], (uint32_t)j, j2 < n ? (uint32_t)j2 : 0u, a, b);
    if (MODE == 0) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(out);
        dst[j] = a;
        if (j2 < n) dst[j2] = b;
    } else {
        float* dst = reinterpret_cast<float*>(out);
        dst[j] = (MODE == 1) ? bits_to_uniform(a, lo, hi) : bits_to_normal(a);
        if (j2 < n) dst[j2] = (MODE == 1) ? bits_to_uniform(b, lo, hi) : bits_to_normal(b);
    }
}

// Per-example, per-SITE guide noise of a multi-site mean-field guide (examples/logistic_regression.py:67-86; oracle d3po_px_eps_sites):
// example p's key = split(jax_key, B_total)[p] (svi.py:289-290), guide seed = split(.)[1], then numpyro's seed handler advances
// rng, site_key = split(rng) at every sample statement; eps row = [normal(site_key_0, (size_0,)) | normal(site_key_1, (size_1,)) | ...].
// One workgroup per example: every thread derives the site keys (2 + 2 per site threefry calls: cheaper than a barrier and a
// broadcast), then the threads share the sites' word pairs (words j and j + ceil(size / 2) of a site come from ONE threefry call).
#define D3P_MAX_GUIDE_SITES 8
struct EpsSitesArgs {
    const uint32_t* jax_key;
    uint32_t B_total, pos0, B_local;
    int n_sites;
    int32_t size[D3P_MAX_GUIDE_SITES];
    int32_t row;     // sum of the sizes
    float* eps;
};

__global__ void __launch_bounds__(256) k_px_eps_sites(EpsSitesArgs a)
{
    const uint32_t i = blockIdx.x;
    if (i >= a.B_local) return;
    const uint32_t p = a.pos0 + i, j0 = a.jax_key[0], j1 = a.jax_key[1];
    const uint32_t px0 = tf_iota_word(j0, j1, 2ull * a.B_total, 2ull * p), px1 = tf_iota_word(j0, j1, 2ull * a.B_total, 2ull * p + 1);
    uint32_t t, r0, r1;
    threefry2x32(px0, px1, 0u, 2u, t, r0);   // split(key, 2): counts [0, 1 | 2, 3]; key 1 = (y1(0, 2), y1(1, 3)), key 0 = (y0(0, 2), y0(1, 3))
    threefry2x32(px0, px1, 1u, 3u, t, r1);   // (r0, r1) = the guide seed
    float* row = a.eps + (size_t)i * a.row;
    int off = 0;
    for (int s = 0; s < a.n_sites; ++s) {
        uint32_t c0, c1, k0, k1;
        threefry2x32(r0, r1, 0u, 2u, c0, k0);
        threefry2x32(r0, r1, 1u, 3u, c1, k1);
        r0 = c0; r1 = c1;                    // the handler's key after this site
        const int n = a.size[s], half = (n + 1) >> 1;
        for (int j = threadIdx.x; j < half; j += blockDim.x) {
            const int j2 = j + half;
            uint32_t wa, wb;
            threefry2x32(k0, k1, (uint32_t)j, j2 < n ? (uint32_t)j2 : 0u, wa, wb);
            row[off + j] = bits_to_normal(wa);
            if (j2 < n) row[off + j2] = bits_to_normal(wb);
        }
        off += n;
    }
}

__global__ void k_tf_fold_in(const uint32_t* __restrict__ key, uint32_t data, uint32_t* __restrict__ out)
{
    uint32_t a, b;
    threefry2x32(key[0], key[1], 0u, data, a, b);
    out[0] = a;
    out[1] = b;
}

// ------------------------------------------------------------------------------------------
// Feistel sampler (d3p/util.py:216-301)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void feistel_constants_to_lds(const uint32_t* __restrict__ key, uint32_t* rc_lds)
{
    // random_bits(key, 32, (10, 3)) = first 30 words of blocks 0 and 1 (util.py:240-242);
    // column 0 forced odd (util.py:245-246).
    if (threadIdx.x < 2) {
        uint32_t k[16], o[16];
        load_key(key, k);
        keystream_block(k, threadIdx.x, o);
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int g = 16 * (int)threadIdx.x + w;
            if (g < 30) rc_lds[g] = (g % 3 == 0) ? (o[w] | 1u) : o[w];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t feistel_permute(const uint32_t* rc, uint32_t capacity, int bits_lower,
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

__global__ void k_feistel_sample(const uint32_t* __restrict__ key, uint32_t capacity, int bits_lower,
                                 int bits_upper, uint32_t n, uint32_t* __restrict__ out)
{
    __shared__ uint32_t rc[32];
    feistel_constants_to_lds(key, rc);
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = feistel_permute(rc, capacity, bits_lower, bits_upper, p);
}

// Same permutation from explicit round constants (any rng_suite's random_bits(key, 32, (10, 3)),
// util.py:240-242); the |1 of util.py:245-246 is applied here.
__global__ void k_feistel_from_rc(const uint32_t* __restrict__ rc_in, uint32_t capacity, int bits_lower,
                                  int bits_upper, uint32_t n, uint32_t* __restrict__ out)
{
    __shared__ uint32_t rc[32];
    if (threadIdx.x < 30) rc[threadIdx.x] = (threadIdx.x % 3 == 0) ? (rc_in[threadIdx.x] | 1u) : rc_in[threadIdx.x];
    __syncthreads();
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = feistel_permute(rc, capacity, bits_lower, bits_upper, p);
}

// ------------------------------------------------------------------------------------------
// Poisson selection (d3p/minibatch.py:29-39, :119-124)
// Pass 1: thread t owns keystream block t = elements 16t..16t+15; writes a 16-bit selection mask
//         and the block's selected count.
// Pass 2: one workgroup scans the per-workgroup counts from the TOP of the table downwards
//         (descending order of the reversed stable argsort) and writes the two counts.
// Pass 3: every element computes its output slot; slots < cutoff are written.
// ------------------------------------------------------------------------------------------
#define D3P_PS_THREADS 256

template <int RNG>  // 0: ChaCha20 keystream (d3p.random), 1: threefry iota stream (d3p.random.debug)
__global__ void __launch_bounds__(D3P_PS_THREADS)
k_poisson_flags(const uint32_t* __restrict__ key, size_t key_stride_words, float q, uint32_t N,
                uint16_t* __restrict__ flags, uint32_t* __restrict__ wg_counts, size_t ws_stride_bytes,
                uint32_t chunk0, uint32_t n_chunks, uint32_t elo, uint32_t ehi)
{
    // chunk0, n_chunks, [elo, ehi): the SHARD of the mask this launch makes (the whole mask: 0, ceil(N / 16), [0, N)).  Element e's
    // draw is keystream word e whoever generates it (block e / 16, word e % 16), so a rank of a row-sharded data-parallel run makes
    // the blocks of its own rows only -- 1 / world of the work -- and obtains the same bits as a single GPU.
    // blockIdx.y selects the step of a batch: every per-step array is `stride` apart
    key += (size_t)blockIdx.y * key_stride_words;
    flags = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(flags) + (size_t)blockIdx.y * ws_stride_bytes);
    wg_counts = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(wg_counts) + (size_t)blockIdx.y * ws_stride_bytes);
    __shared__ uint32_t red[D3P_PS_THREADS / 64];
    const uint32_t t = blockIdx.x * D3P_PS_THREADS + threadIdx.x;
    const uint32_t ch = chunk0 + t;   // the keystream block / group of 16 elements of this thread
    uint32_t m = 0;
    if (t < n_chunks) {
        uint32_t o[16];
        if (RNG == 0) {
            uint32_t k[16];
            load_key(key, k);
            keystream_block(k, ch, o);
        } else {
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const uint32_t e = 16u * ch + w;
                o[w] = (e < N) ? tf_iota_word(key[0], key[1], N, e) : 0u;
            }
        }
        // uniform(key, (N,))[e] <= q (minibatch.py:34-35) on the raw word: the uniform of a word is EXACTLY (word >> 9) 2^-23
        // ((1.m - 1) (1 - 0) + 0, clamped at 0), and q 2^23 is an exact float product, so the float comparison is the integer
        // comparison (word >> 9) <= floor(q 2^23) -- two integer instructions per element and no conversion, select or branch
        // (bit for bit the same mask; the float form stays for the last, partial chunk's bounds)
        const uint32_t thr = q >= 1.0f ? 0xffffffffu : (uint32_t)floorf(q * 8388608.0f);
        if (16u * ch >= elo && 16u * ch + 15u < ehi) {
#pragma unroll
            for (int w = 0; w < 16; ++w) m |= (((o[w] >> 9) - thr - 1u) >> 31) << w;   // (x <= thr <=> x - thr - 1 wraps negative; x, thr < 2^31)
            if (q >= 1.0f) m = 0xffffu;
        } else {   // a chunk that straddles the end of the table or of the shard
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const uint32_t e = 16u * ch + w;
                const bool sel = (e >= elo) && (e < ehi) && (bits_to_uniform(o[w], 0.0f, 1.0f) <= q);
                m |= (sel ? 1u : 0u) << w;
            }
        }
        flags[t] = (uint16_t)m;
    }
    uint32_t c = __popc(m);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < D3P_PS_THREADS / 64; ++w) s += red[w];
        wg_counts[blockIdx.x] = s;
    }
}

// wg_above[g] = number of selected elements in workgroups with a higher index than g.
__global__ void __launch_bounds__(1024)
k_poisson_scan(const uint32_t* __restrict__ wg_counts, uint32_t n_wg, uint32_t cutoff, int suppress,
               uint32_t* __restrict__ wg_above, uint32_t* __restrict__ counts, size_t ws_stride_bytes,
               size_t counts_stride_words, uint32_t* __restrict__ shard_counts)
{
    // shard_counts != nullptr: the launch scans a SHARD of the mask; the shard's selected count of step blockIdx.y goes there and
    // `counts` is left to whoever knows the other shards' counts (d3p_xchg_poisson_counts)
    wg_counts = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(wg_counts) + (size_t)blockIdx.y * ws_stride_bytes);
    wg_above = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(wg_above) + (size_t)blockIdx.y * ws_stride_bytes);
    counts += (size_t)blockIdx.y * counts_stride_words;
    __shared__ uint32_t part[1024];
    const uint32_t per = (n_wg + 1023u) / 1024u;
    // thread 0 owns the TOP `per` workgroups, thread 1 the next, ...
    const int64_t hi = (int64_t)n_wg - 1 - (int64_t)threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t i = 0; i < per; ++i) {
        const int64_t g = hi - i;
        if (g >= 0) s += wg_counts[g];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    // exclusive prefix over threads (1024 values, simple Hillis-Steele in LDS)
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (threadIdx.x >= (unsigned)off) ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t i = 0; i < per; ++i) {
        const int64_t g = hi - i;
        if (g >= 0) {
            wg_above[g] = run;
            run += wg_counts[g];
        }
    }
    if (threadIdx.x == 1023) {
        const uint32_t nsel = part[1023];
        if (shard_counts) {
            shard_counts[blockIdx.y] = nsel;
        } else {
            counts[0] = nsel;
            counts[1] = suppress ? (nsel <= cutoff ? nsel : 0u) : (nsel < cutoff ? nsel : cutoff);
        }
    }
}

__global__ void __launch_bounds__(D3P_PS_THREADS)
k_poisson_write(const uint16_t* __restrict__ flags, const uint32_t* __restrict__ wg_above,
                const uint32_t* __restrict__ counts, uint32_t N, uint32_t cutoff, uint32_t* __restrict__ out_idx,
                size_t ws_stride_bytes, size_t counts_stride_words, size_t idx_stride_words,
                uint32_t chunk0, uint32_t n_chunks, const uint32_t* __restrict__ shard_above, uint32_t* __restrict__ plist)
{
    // shard_above != nullptr: the flags are those of a SHARD (chunks chunk0 ...); shard_above[step] = selected elements in the
    // shards ABOVE this one (higher rows).  Only the shard's selected elements are written, at their GLOBAL positions -- the
    // selected elements come first, in descending row order (minibatch.py:36-37), so the shard's positions are the contiguous range
    // shard_above .. + its count -- and only the valid ones (position < counts[1]: truncation keeps the globally highest rows,
    // minibatch.py:119-124); plist[j] = shard_above + j is the shard's dense list of owned positions.
    flags = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(flags) + (size_t)blockIdx.y * ws_stride_bytes);
    wg_above = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(wg_above) + (size_t)blockIdx.y * ws_stride_bytes);
    counts += (size_t)blockIdx.y * counts_stride_words;
    out_idx += (size_t)blockIdx.y * idx_stride_words;
    __shared__ uint32_t wave_cnt[D3P_PS_THREADS / 64];
    if (plist) plist += (size_t)blockIdx.y * idx_stride_words;
    const uint32_t t = blockIdx.x * D3P_PS_THREADS + threadIdx.x;
    const uint32_t m = (t < n_chunks) ? flags[t] : 0u;
    const uint32_t c = __popc(m);
    // selected elements in higher threads of this wave (suffix sum over lanes)
    const int lane = threadIdx.x & 63;
    uint32_t suf = c;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_down(suf, off);
        if (lane + off < 64) suf += v;
    }
    if (lane == 0) wave_cnt[threadIdx.x >> 6] = suf;
    __syncthreads();
    uint32_t above = wg_above[blockIdx.x] + (suf - c);
    for (int w = (threadIdx.x >> 6) + 1; w < D3P_PS_THREADS / 64; ++w) above += wave_cnt[w];
    if (t >= n_chunks) return;
    if (shard_above) {
        const uint32_t first = shard_above[blockIdx.y], n_valid = counts[1];
        above += first;
        for (int w = 15; w >= 0; --w) {
            if (!((m >> w) & 1u)) continue;
            if (above < n_valid) {
                out_idx[above] = 16u * (chunk0 + t) + (uint32_t)w;
                if (plist) plist[above - first] = above;
            }
            ++above;
        }
        return;
    }
    const uint32_t nsel = counts[0];
    // walk this thread's 16 elements from the top
    for (int w = 15; w >= 0; --w) {
        const uint32_t e = 16u * t + w;
        if (e >= N) continue;
        const bool sel = (m >> w) & 1u;
        const uint32_t pos = sel ? above : nsel + ((N - 1u - e) - above);
        if (pos < cutoff) out_idx[pos] = e;
        above += sel ? 1u : 0u;
    }
}

// ------------------------------------------------------------------------------------------
// jnp.take(a, idx, axis=0): one wave per output row.
// ------------------------------------------------------------------------------------------
template <typename VEC>
__global__ void k_take_rows(const char* __restrict__ table, uint64_t n_rows, uint32_t row_bytes, const uint32_t* __restrict__ idx,
                            uint32_t n, const uint32_t* __restrict__ valid_count, char* __restrict__ out)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    const bool valid = valid_count ? (wave < *valid_count) : true;
    // (an index past the table is clamped to its last row -- jnp.take's "clip" -- instead of being dereferenced: the samplers never
    //  produce one, a caller's own index array might)
    uint64_t row = idx[wave];
    if (row >= n_rows) row = n_rows - 1;
    const VEC* src = reinterpret_cast<const VEC*>(table + (size_t)row * row_bytes);
    VEC* dst = reinterpret_cast<VEC*>(out + (size_t)wave * row_bytes);
    const uint32_t nv = row_bytes / sizeof(VEC);
    VEC zero;
    memset(&zero, 0, sizeof(VEC));
    for (uint32_t v = lane; v < nv; v += 64) dst[v] = valid ? src[v] : zero;
}

}  // namespace d3p

using namespace d3p;

// delta, log2 in float32 and the power-of-two mask exactly as d3p/random/__init__.py:124-128, in the unsigned dtype
template <typename U>
static void randint_mask(int64_t minval, int64_t maxval, U* delta, U* bitmask)
{
    constexpr int nbits = 8 * (int)sizeof(U);
    *delta = (U)(unsigned long long)(maxval - 1 - minval);
    const float l2 = log2f((float)*delta) + 1.0f;
    int lg;
    if (!(l2 > 0.0f)) lg = 0;                  // udtype(-inf): 0
    else if (l2 >= (float)nbits) lg = nbits;   // jnp.minimum(..., nbits)
    else lg = (int)l2;
    *bitmask = lg >= nbits ? (U)~(U)0 : (U)(((U)1 << lg) - (U)1);
}

extern "C" {

int d3p_abi_version(void) { return D3P_ABI_VERSION; }
const char* d3p_last_error(void) { return last_error_buf(); }

int d3p_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int d3p_rng_split(void* stream, const uint32_t* key_dev, int num, uint32_t* out_keys_dev)
{
    D3P_REQUIRE(key_dev && out_keys_dev, "d3p_rng_split: null pointer");
    D3P_REQUIRE(num >= 0, "d3p_rng_split: num must be >= 0");
    if (num == 0) return D3P_OK;
    hipLaunchKernelGGL(k_rng_derive, dim3(cdiv(num, 64)), dim3(64), 0, (hipStream_t)stream, key_dev, num, 0u,
                       D3P_TAG_SPLIT, out_keys_dev);
    return check_launch("d3p_rng_split");
}

int d3p_rng_fold_in(void* stream, const uint32_t* key_dev, uint32_t data, uint32_t* out_key_dev)
{
    D3P_REQUIRE(key_dev && out_key_dev, "d3p_rng_fold_in: null pointer");
    hipLaunchKernelGGL(k_rng_derive, dim3(1), dim3(64), 0, (hipStream_t)stream, key_dev, 1, data, D3P_TAG_FOLD,
                       out_key_dev);
    return check_launch("d3p_rng_fold_in");
}

int d3p_rng_random_bits(void* stream, const uint32_t* key_dev, int bit_width, uint64_t count, void* out_dev)
{
    D3P_REQUIRE(key_dev && out_dev, "d3p_rng_random_bits: null pointer");
    D3P_REQUIRE(bit_width == 8 || bit_width == 16 || bit_width == 32 || bit_width == 64,
                "d3p_rng_random_bits: bit_width must be 8, 16, 32 or 64");
    const uint64_t n_blocks = (count * (uint64_t)bit_width + 511) / 512;
    D3P_REQUIRE(n_blocks <= 0xFFFFFFFFull, "d3p_rng_random_bits: more than 2^32 ChaCha blocks requested");
    if (n_blocks == 0) return D3P_OK;
    hipLaunchKernelGGL(k_rng_stream<0>, dim3(cdiv(n_blocks, 256)), dim3(256), 0, (hipStream_t)stream, key_dev,
                       n_blocks, count, 0.f, 0.f, out_dev);
    return check_launch("d3p_rng_random_bits");
}

int d3p_rng_uniform(void* stream, const uint32_t* key_dev, uint64_t n, float minval, float maxval, float* out_dev)
{
    D3P_REQUIRE(key_dev && out_dev, "d3p_rng_uniform: null pointer");
    const uint64_t n_blocks = (n + 15) / 16;
    D3P_REQUIRE(n_blocks <= 0xFFFFFFFFull, "d3p_rng_uniform: too many elements");
    if (n == 0) return D3P_OK;
    hipLaunchKernelGGL(k_rng_stream<1>, dim3(cdiv(n_blocks, 256)), dim3(256), 0, (hipStream_t)stream, key_dev,
                       n_blocks, n, minval, maxval, (void*)out_dev);
    return check_launch("d3p_rng_uniform");
}

int d3p_rng_normal(void* stream, const uint32_t* key_dev, uint64_t n, float* out_dev)
{
    D3P_REQUIRE(key_dev && out_dev, "d3p_rng_normal: null pointer");
    const uint64_t n_blocks = (n + 15) / 16;
    D3P_REQUIRE(n_blocks <= 0xFFFFFFFFull, "d3p_rng_normal: too many elements");
    if (n == 0) return D3P_OK;
    hipLaunchKernelGGL(k_rng_stream<2>, dim3(cdiv(n_blocks, 256)), dim3(256), 0, (hipStream_t)stream, key_dev,
                       n_blocks, n, 0.f, 0.f, (void*)out_dev);
    return check_launch("d3p_rng_normal");
}

int d3p_rng_randint(void* stream, const uint32_t* key_dev, uint64_t n, int32_t minval, int32_t maxval,
                    int32_t* out_dev)
{
    D3P_REQUIRE(key_dev && out_dev, "d3p_rng_randint: null pointer");
    if (n == 0) return D3P_OK;
    // delta / bitmask exactly as d3p/random/__init__.py:124-128 (float32 log2).
    const uint32_t delta = (uint32_t)(maxval - 1 - minval);
    const float l2 = log2f((float)delta) + 1.0f;
    uint32_t lg;
    if (!(l2 > 0.0f)) lg = 0;
    else if (l2 >= 32.0f) lg = 32;
    else lg = (uint32_t)l2;
    const uint32_t bitmask = (lg >= 32) ? 0xffffffffu : ((1u << lg) - 1u);
    hipLaunchKernelGGL(k_rng_randint, dim3(cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, key_dev, n, delta,
                       bitmask, minval, out_dev);
    return check_launch("d3p_rng_randint");
}

int d3p_rng_randint_bits(void* stream, const uint32_t* key_dev, uint64_t n, int bit_width, int64_t minval, int64_t maxval,
                         void* out_dev)
{
    D3P_REQUIRE(key_dev && out_dev, "d3p_rng_randint_bits: null pointer");
    D3P_REQUIRE(bit_width == 8 || bit_width == 16 || bit_width == 32 || bit_width == 64, "d3p_rng_randint_bits: bit_width must be 8, 16, 32 or 64");
    D3P_REQUIRE((n * (uint64_t)(bit_width / 8) + 63) / 64 <= 0xFFFFFFFFull, "d3p_rng_randint_bits: too many elements");
    if (n == 0) return D3P_OK;
    const dim3 grid(cdiv(n, 128)), block(128);
    hipStream_t s = (hipStream_t)stream;
    if (bit_width == 8) {
        uint8_t d, m;
        randint_mask<uint8_t>(minval, maxval, &d, &m);
        hipLaunchKernelGGL((k_rng_randint_bits<uint8_t, int8_t>), grid, block, 0, s, key_dev, n, d, m, (int8_t)minval, (int8_t*)out_dev);
    } else if (bit_width == 16) {
        uint16_t d, m;
        randint_mask<uint16_t>(minval, maxval, &d, &m);
        hipLaunchKernelGGL((k_rng_randint_bits<uint16_t, int16_t>), grid, block, 0, s, key_dev, n, d, m, (int16_t)minval, (int16_t*)out_dev);
    } else if (bit_width == 32) {
        uint32_t d, m;
        randint_mask<uint32_t>(minval, maxval, &d, &m);
        hipLaunchKernelGGL((k_rng_randint_bits<uint32_t, int32_t>), grid, block, 0, s, key_dev, n, d, m, (int32_t)minval, (int32_t*)out_dev);
    } else {
        unsigned long long d, m;
        randint_mask<unsigned long long>(minval, maxval, &d, &m);
        hipLaunchKernelGGL((k_rng_randint_bits<unsigned long long, long long>), grid, block, 0, s, key_dev, n, d, m, (long long)minval,
                           (long long*)out_dev);
    }
    return check_launch("d3p_rng_randint_bits");
}

int d3p_tf_split(void* stream, const uint32_t* key_dev, int num, uint32_t* out_keys_dev)
{
    D3P_REQUIRE(key_dev && out_keys_dev, "d3p_tf_split: null pointer");
    D3P_REQUIRE(num >= 0, "d3p_tf_split: num must be >= 0");
    if (num == 0) return D3P_OK;
    return d3p_tf_random_bits(stream, key_dev, 2ull * (uint64_t)num, out_keys_dev);
}

int d3p_px_eps_sites(void* stream, const uint32_t* jax_key_dev, uint32_t B_total, uint32_t pos0, uint32_t B_local, const int32_t* site_sizes_host,
                     int32_t n_sites, float* eps_dev)
{
    D3P_REQUIRE(jax_key_dev && site_sizes_host && eps_dev, "d3p_px_eps_sites: null pointer");
    D3P_REQUIRE(n_sites >= 1 && n_sites <= D3P_MAX_GUIDE_SITES, "d3p_px_eps_sites: 1 <= n_sites <= 8");
    D3P_REQUIRE((uint64_t)pos0 + B_local <= B_total, "d3p_px_eps_sites: pos0 + B_local must not exceed B_total");
    if (B_local == 0) return D3P_OK;
    EpsSitesArgs a;
    a.jax_key = jax_key_dev; a.B_total = B_total; a.pos0 = pos0; a.B_local = B_local; a.n_sites = n_sites; a.eps = eps_dev;
    int64_t row = 0;
    for (int s = 0; s < n_sites; ++s) {
        D3P_REQUIRE(site_sizes_host[s] >= 1, "d3p_px_eps_sites: a site has at least one element (a scalar site has size 1)");
        a.size[s] = site_sizes_host[s];
        row += site_sizes_host[s];
    }
    D3P_REQUIRE(row <= 0x7fffffff, "d3p_px_eps_sites: row too long");
    a.row = (int32_t)row;
    hipLaunchKernelGGL(k_px_eps_sites, dim3(B_local), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("d3p_px_eps_sites");
}

int d3p_tf_fold_in(void* stream, const uint32_t* key_dev, uint32_t data, uint32_t* out_key_dev)
{
    D3P_REQUIRE(key_dev && out_key_dev, "d3p_tf_fold_in: null pointer");
    hipLaunchKernelGGL(k_tf_fold_in, dim3(1), dim3(1), 0, (hipStream_t)stream, key_dev, data, out_key_dev);
    return check_launch("d3p_tf_fold_in");
}

int d3p_tf_random_bits(void* stream, const uint32_t* key_dev, uint64_t n_words, uint32_t* out_dev)
{
    D3P_REQUIRE(key_dev && out_dev, "d3p_tf_random_bits: null pointer");
    D3P_REQUIRE(n_words <= 0xFFFFFFFFull, "d3p_tf_random_bits: too many words");
    if (n_words == 0) return D3P_OK;
    hipLaunchKernelGGL(k_tf_stream<0>, dim3(cdiv((n_words + 1) / 2, 256)), dim3(256), 0, (hipStream_t)stream,
                       key_dev, n_words, 0.f, 0.f, (void*)out_dev);
    return check_launch("d3p_tf_random_bits");
}

int d3p_tf_uniform(void* stream, const uint32_t* key_dev, uint64_t n, float minval, float maxval, float* out_dev)
{
    D3P_REQUIRE(key_dev && out_dev, "d3p_tf_uniform: null pointer");
    D3P_REQUIRE(n <= 0xFFFFFFFFull, "d3p_tf_uniform: too many elements");
    if (n == 0) return D3P_OK;
    hipLaunchKernelGGL(k_tf_stream<1>, dim3(cdiv((n + 1) / 2, 256)), dim3(256), 0, (hipStream_t)stream, key_dev, n,
                       minval, maxval, (void*)out_dev);
    return check_launch("d3p_tf_uniform");
}

int d3p_tf_normal(void* stream, const uint32_t* key_dev, uint64_t n, float* out_dev)
{
    D3P_REQUIRE(key_dev && out_dev, "d3p_tf_normal: null pointer");
    D3P_REQUIRE(n <= 0xFFFFFFFFull, "d3p_tf_normal: too many elements");
    if (n == 0) return D3P_OK;
    hipLaunchKernelGGL(k_tf_stream<2>, dim3(cdiv((n + 1) / 2, 256)), dim3(256), 0, (hipStream_t)stream, key_dev, n,
                       0.f, 0.f, (void*)out_dev);
    return check_launch("d3p_tf_normal");
}

static inline int bit_length_u32(uint32_t v)
{
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

int d3p_feistel_sample(void* stream, const uint32_t* key_dev, uint32_t capacity, uint32_t n, uint32_t* out_idx_dev)
{
    D3P_REQUIRE(key_dev && (out_idx_dev || n == 0), "d3p_feistel_sample: null pointer");
    D3P_REQUIRE(capacity >= 1, "d3p_feistel_sample: capacity must be >= 1");
    D3P_REQUIRE(n <= capacity, "d3p_feistel_sample: cannot sample more than capacity without replacement");
    if (n == 0) return D3P_OK;
    const int bits = bit_length_u32(capacity - 1);  // util.py:230
    const int bits_lower = bits >> 1, bits_upper = bits - bits_lower;
    hipLaunchKernelGGL(k_feistel_sample, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, key_dev, capacity,
                       bits_lower, bits_upper, n, out_idx_dev);
    return check_launch("d3p_feistel_sample");
}

int d3p_feistel_from_constants(void* stream, const uint32_t* rc_dev, uint32_t capacity, uint32_t n,
                               uint32_t* out_idx_dev)
{
    D3P_REQUIRE(rc_dev && (out_idx_dev || n == 0), "d3p_feistel_from_constants: null pointer");
    D3P_REQUIRE(capacity >= 1, "d3p_feistel_from_constants: capacity must be >= 1");
    D3P_REQUIRE(n <= capacity, "d3p_feistel_from_constants: cannot sample more than capacity without replacement");
    if (n == 0) return D3P_OK;
    const int bits = bit_length_u32(capacity - 1);
    const int bits_lower = bits >> 1, bits_upper = bits - bits_lower;
    hipLaunchKernelGGL(k_feistel_from_rc, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rc_dev, capacity,
                       bits_lower, bits_upper, n, out_idx_dev);
    return check_launch("d3p_feistel_from_constants");
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t d3p_poisson_select_workspace(uint32_t N)
{
    const size_t n_chunks = ((size_t)N + 15) / 16;
    const size_t n_wg = (n_chunks + D3P_PS_THREADS - 1) / D3P_PS_THREADS;
    return align_up(n_chunks * sizeof(uint16_t), 256) + 2 * align_up((n_wg + 1) * sizeof(uint32_t), 256);
}

int d3p_poisson_select(void* stream, const uint32_t* key_dev, float q, uint32_t N, uint32_t cutoff, int suppress,
                       uint32_t* out_idx_dev, uint32_t* out_counts_dev, void* workspace_dev, size_t workspace_bytes)
{
    return d3p_poisson_select_rng(stream, 0, key_dev, q, N, cutoff, suppress, out_idx_dev, out_counts_dev,
                                  workspace_dev, workspace_bytes);
}

int d3p_poisson_select_rng(void* stream, int rng_kind, const uint32_t* key_dev, float q, uint32_t N, uint32_t cutoff,
                           int suppress, uint32_t* out_idx_dev, uint32_t* out_counts_dev, void* workspace_dev,
                           size_t workspace_bytes)
{
    return d3p_poisson_select_batch(stream, rng_kind, key_dev, 0, q, N, cutoff, suppress, out_idx_dev, 0, out_counts_dev, 0,
                                    1, workspace_dev, workspace_bytes);
}

int d3p_poisson_select_batch(void* stream, int rng_kind, const uint32_t* keys_dev, size_t key_stride_words, float q,
                             uint32_t N, uint32_t cutoff, int suppress, uint32_t* out_idx_dev, size_t idx_stride_words,
                             uint32_t* out_counts_dev, size_t counts_stride_words, uint32_t num_steps,
                             void* workspace_dev, size_t workspace_bytes)
{
    D3P_REQUIRE(rng_kind == 0 || rng_kind == 1, "d3p_poisson_select: rng_kind must be 0 (chacha) or 1 (threefry)");
    D3P_REQUIRE(keys_dev && out_counts_dev && workspace_dev, "d3p_poisson_select: null pointer");
    D3P_REQUIRE(out_idx_dev || cutoff == 0, "d3p_poisson_select: null index buffer");
    D3P_REQUIRE(N >= 1, "d3p_poisson_select: N must be >= 1");
    D3P_REQUIRE(cutoff <= N, "d3p_poisson_select: cutoff must be <= N");
    D3P_REQUIRE(num_steps >= 1 && num_steps <= 65535, "d3p_poisson_select: 1 <= num_steps <= 65535");
    const size_t per_step = d3p_poisson_select_workspace(N);
    if (workspace_bytes < per_step * num_steps)
        return fail(D3P_E_WORKSPACE, "d3p_poisson_select: workspace too small (%zu < %zu)", workspace_bytes,
                    per_step * num_steps);
    const size_t n_chunks = ((size_t)N + 15) / 16;
    const uint32_t n_wg = (uint32_t)((n_chunks + D3P_PS_THREADS - 1) / D3P_PS_THREADS);
    char* ws = (char*)workspace_dev;
    uint16_t* flags = (uint16_t*)ws;
    uint32_t* wg_counts = (uint32_t*)(ws + align_up(n_chunks * sizeof(uint16_t), 256));
    uint32_t* wg_above = (uint32_t*)((char*)wg_counts + align_up((n_wg + 1) * sizeof(uint32_t), 256));
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(n_wg, num_steps);
    if (rng_kind == 0)
        hipLaunchKernelGGL(k_poisson_flags<0>, grid, dim3(D3P_PS_THREADS), 0, s, keys_dev, key_stride_words, q, N, flags,
                           wg_counts, per_step, 0u, (uint32_t)n_chunks, 0u, N);
    else
        hipLaunchKernelGGL(k_poisson_flags<1>, grid, dim3(D3P_PS_THREADS), 0, s, keys_dev, key_stride_words, q, N, flags,
                           wg_counts, per_step, 0u, (uint32_t)n_chunks, 0u, N);
    hipLaunchKernelGGL(k_poisson_scan, dim3(1, num_steps), dim3(1024), 0, s, (const uint32_t*)wg_counts, n_wg, cutoff, suppress,
                       wg_above, out_counts_dev, per_step, counts_stride_words, (uint32_t*)nullptr);
    if (cutoff > 0)
        hipLaunchKernelGGL(k_poisson_write, grid, dim3(D3P_PS_THREADS), 0, s, (const uint16_t*)flags,
                           (const uint32_t*)wg_above, (const uint32_t*)out_counts_dev, N, cutoff, out_idx_dev, per_step,
                           counts_stride_words, idx_stride_words, 0u, (uint32_t)n_chunks, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    return check_launch("d3p_poisson_select");
}

// The same selection made by the ranks of a row-sharded data-parallel run, each for the rows [row_lo, row_hi) it holds (SURVEY
// 8(e)): two calls with the exchange of the shards' counts between them (d3p_xchg_poisson_counts in the run loops; any other
// all-gather of num_steps words per rank will do).
//   d3p_poisson_shard_flags: the shard's part of the Bernoulli mask (ceil((row_hi - row_lo) / 16) + <= 1 ChaCha20 blocks per step
//     instead of N / 16) and its selected count per step -> shard_counts_dev[t];
//   d3p_poisson_shard_write: given, per step, counts {selected in the whole table, valid = after truncate / suppress} (stride
//     counts_stride_words) and above_dev[t] = selected elements in the shards with HIGHER rows: the shard's valid selected rows at
//     their global batch positions in out_idx (the other entries of out_idx are not touched), its dense list of owned positions
//     (plist_dev, nullable; same stride as out_idx) -- the union over the shards is bit for bit the single-GPU selection.
int d3p_poisson_shard_flags(void* stream, int rng_kind, const uint32_t* keys_dev, size_t key_stride_words, float q, uint32_t N,
                            uint32_t row_lo, uint32_t row_hi, uint32_t num_steps, uint32_t* shard_counts_dev, void* workspace_dev,
                            size_t workspace_bytes)
{
    D3P_REQUIRE(rng_kind == 0 || rng_kind == 1, "d3p_poisson_shard_flags: rng_kind must be 0 (chacha) or 1 (threefry)");
    D3P_REQUIRE(keys_dev && shard_counts_dev && workspace_dev, "d3p_poisson_shard_flags: null pointer");
    D3P_REQUIRE(N >= 1 && row_lo <= row_hi && row_hi <= N, "d3p_poisson_shard_flags: need 0 <= row_lo <= row_hi <= N, N >= 1");
    D3P_REQUIRE(num_steps >= 1 && num_steps <= 65535, "d3p_poisson_shard_flags: 1 <= num_steps <= 65535");
    const uint32_t chunk0 = row_lo / 16u;
    const size_t n_chunks = row_hi > row_lo ? ((size_t)row_hi + 15) / 16 - chunk0 : 0;
    const size_t per_step = d3p_poisson_select_workspace(n_chunks ? (uint32_t)(16 * n_chunks) : 1u);
    if (workspace_bytes < per_step * num_steps)
        return fail(D3P_E_WORKSPACE, "d3p_poisson_shard_flags: workspace too small (%zu < %zu)", workspace_bytes, per_step * num_steps);
    hipStream_t s = (hipStream_t)stream;
    if (n_chunks == 0) {  // an empty shard selects nothing
        D3P_HIP_TRY(hipMemsetAsync(shard_counts_dev, 0, (size_t)num_steps * sizeof(uint32_t), s));
        return D3P_OK;
    }
    const uint32_t n_wg = (uint32_t)((n_chunks + D3P_PS_THREADS - 1) / D3P_PS_THREADS);
    char* ws = (char*)workspace_dev;
    uint16_t* flags = (uint16_t*)ws;
    uint32_t* wg_counts = (uint32_t*)(ws + align_up(n_chunks * sizeof(uint16_t), 256));
    uint32_t* wg_above = (uint32_t*)((char*)wg_counts + align_up((n_wg + 1) * sizeof(uint32_t), 256));
    const dim3 grid(n_wg, num_steps);
    if (rng_kind == 0)
        hipLaunchKernelGGL(k_poisson_flags<0>, grid, dim3(D3P_PS_THREADS), 0, s, keys_dev, key_stride_words, q, N, flags, wg_counts, per_step,
                           chunk0, (uint32_t)n_chunks, row_lo, row_hi);
    else
        hipLaunchKernelGGL(k_poisson_flags<1>, grid, dim3(D3P_PS_THREADS), 0, s, keys_dev, key_stride_words, q, N, flags, wg_counts, per_step,
                           chunk0, (uint32_t)n_chunks, row_lo, row_hi);
    hipLaunchKernelGGL(k_poisson_scan, dim3(1, num_steps), dim3(1024), 0, s, (const uint32_t*)wg_counts, n_wg, 0u, 0, wg_above,
                       (uint32_t*)nullptr, per_step, (size_t)0, shard_counts_dev);
    return check_launch("d3p_poisson_shard_flags");
}

int d3p_poisson_shard_write(void* stream, uint32_t N, uint32_t row_lo, uint32_t row_hi, uint32_t cutoff, const uint32_t* counts_dev,
                            size_t counts_stride_words, const uint32_t* above_dev, uint32_t* out_idx_dev, uint32_t* plist_dev,
                            size_t idx_stride_words, uint32_t num_steps, void* workspace_dev, size_t workspace_bytes)
{
    D3P_REQUIRE(counts_dev && above_dev && workspace_dev, "d3p_poisson_shard_write: null pointer");
    D3P_REQUIRE(out_idx_dev || cutoff == 0, "d3p_poisson_shard_write: null index buffer");
    D3P_REQUIRE(N >= 1 && row_lo <= row_hi && row_hi <= N, "d3p_poisson_shard_write: need 0 <= row_lo <= row_hi <= N, N >= 1");
    D3P_REQUIRE(num_steps >= 1 && num_steps <= 65535, "d3p_poisson_shard_write: 1 <= num_steps <= 65535");
    const uint32_t chunk0 = row_lo / 16u;
    const size_t n_chunks = row_hi > row_lo ? ((size_t)row_hi + 15) / 16 - chunk0 : 0;
    if (n_chunks == 0 || cutoff == 0) return D3P_OK;
    const size_t per_step = d3p_poisson_select_workspace((uint32_t)(16 * n_chunks));
    if (workspace_bytes < per_step * num_steps)
        return fail(D3P_E_WORKSPACE, "d3p_poisson_shard_write: workspace too small (%zu < %zu)", workspace_bytes, per_step * num_steps);
    const uint32_t n_wg = (uint32_t)((n_chunks + D3P_PS_THREADS - 1) / D3P_PS_THREADS);
    char* ws = (char*)workspace_dev;
    const uint16_t* flags = (const uint16_t*)ws;
    const uint32_t* wg_counts = (const uint32_t*)(ws + align_up(n_chunks * sizeof(uint16_t), 256));
    const uint32_t* wg_above = (const uint32_t*)((const char*)wg_counts + align_up((n_wg + 1) * sizeof(uint32_t), 256));
    hipLaunchKernelGGL(k_poisson_write, dim3(n_wg, num_steps), dim3(D3P_PS_THREADS), 0, (hipStream_t)stream, flags, wg_above, counts_dev, N, cutoff,
                       out_idx_dev, per_step, counts_stride_words, idx_stride_words, chunk0, (uint32_t)n_chunks, above_dev, plist_dev);
    return check_launch("d3p_poisson_shard_write");
}

int d3p_take_rows(void* stream, const void* table_dev, uint64_t n_rows, uint32_t row_bytes, const uint32_t* idx_dev,
                  uint32_t n, const uint32_t* valid_count_dev, void* out_dev)
{
    D3P_REQUIRE(table_dev && idx_dev && (out_dev || n == 0), "d3p_take_rows: null pointer");
    D3P_REQUIRE(n_rows >= 1 || n == 0, "d3p_take_rows: rows requested from an empty table");
    D3P_REQUIRE(row_bytes > 0 && row_bytes % 4 == 0, "d3p_take_rows: row_bytes must be a positive multiple of 4");
    if (n == 0) return D3P_OK;
    const dim3 grid(cdiv((uint64_t)n * 64, 256)), block(256);
    const bool vec16 = (row_bytes % 16 == 0) && (((uintptr_t)table_dev | (uintptr_t)out_dev) % 16 == 0);
    if (vec16)
        hipLaunchKernelGGL(k_take_rows<uint4>, grid, block, 0, (hipStream_t)stream, (const char*)table_dev, n_rows, row_bytes,
                           idx_dev, n, valid_count_dev, (char*)out_dev);
    else
        hipLaunchKernelGGL(k_take_rows<uint32_t>, grid, block, 0, (hipStream_t)stream, (const char*)table_dev, n_rows,
                           row_bytes, idx_dev, n, valid_count_dev, (char*)out_dev);
    return check_launch("d3p_take_rows");
}

}  // extern "C"
