/*
 * d3p_hip.h -- C-ABI of libd3p_hip.so: the MI355X (gfx950) hot path of d3p's DP-VI update step.
 *
 * The reference (DPBayes/d3p 0.2.0) is pure Python; it has no FFI boundary of its own
 * (SURVEY.md F5).  Each entry point below replaces one Python-level call on the hot path and
 * cites it (file:line relative to the reference root).  INTEGRATION.md shows the ctypes stub a
 * d3p maintainer would add to route the reference's modules through this library.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*, NULL = the default stream).
 *   - Every `*_dev` / device pointer is a raw device address (e.g. torch.Tensor.data_ptr()).
 *     The caller owns all memory; the library allocates nothing and keeps no references.
 *   - All calls are asynchronous on `stream`; nothing synchronises the device.
 *   - Return value: 0 = D3P_OK, negative = error; d3p_last_error() returns a thread-local
 *     message for the last failing call on this thread.
 *   - ChaCha20 keys ("PRNGState", d3p/random/__init__.py:28) are 16 x uint32 RFC 8439 states:
 *     [0..3] constants, [4..11] key, [12] counter, [13..15] nonce (layout: DESIGN.md section 3).
 *   - Threefry keys (d3p/random/debug.py, and the per-example guide noise) are 2 x uint32.
 */
#ifndef D3P_HIP_H
#define D3P_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3P_OK 0
#define D3P_E_INVALID_ARG (-1)
#define D3P_E_HIP (-2)
#define D3P_E_UNSUPPORTED (-3)
#define D3P_E_WORKSPACE (-4)

#define D3P_ABI_VERSION 9
#define D3P_IPC_HANDLE_BYTES 80   /* what d3p_xchg_create / d3p_fmesh_create hand out for the peers (ABI 9) */

int d3p_abi_version(void);
const char* d3p_last_error(void);
/* Number of visible HIP devices (<= 0: none; the Python layer refuses to run without one). */
int d3p_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * rng_suite: ChaCha20 CSPRNG -- replaces chacha.random as aliased by d3p/random/__init__.py:28-32
 * and the functions d3p adds on top (:50-155).
 * ------------------------------------------------------------------------------------------- */
/* split(key, num) -> num keys   (d3p/random/__init__.py:29; svi.py:210, :491) */
int d3p_rng_split(void* stream, const uint32_t* key_dev, int num, uint32_t* out_keys_dev);
/* fold_in(key, data)            (d3p/random/__init__.py:30; minibatch.py:115, :207, :230) */
int d3p_rng_fold_in(void* stream, const uint32_t* key_dev, uint32_t data, uint32_t* out_key_dev);
/* random_bits(key, bit_width in {8,16,32,64}, shape) -> `count` elements
 * (d3p/random/__init__.py:31; util.py:240-242).  out_dev must hold
 * ceil(count*bit_width/512) * 64 bytes (whole ChaCha blocks are written). */
int d3p_rng_random_bits(void* stream, const uint32_t* key_dev, int bit_width, uint64_t count,
                        void* out_dev);
/* uniform(key, shape, float32, minval, maxval)  (d3p/random/__init__.py:32, :80; minibatch.py:34) */
int d3p_rng_uniform(void* stream, const uint32_t* key_dev, uint64_t n, float minval, float maxval,
                    float* out_dev);
/* normal(key, shape, float32) = sqrt(2) * erf_inv(uniform(nextafter(-1,0), 1))
 * (d3p/random/__init__.py:50-81; svi.py:487) */
int d3p_rng_normal(void* stream, const uint32_t* key_dev, uint64_t n, float* out_dev);
/* randint(key, shape, minval, maxval, int32): masked rejection sampling
 * (d3p/random/__init__.py:84-146; minibatch.py:208) */
int d3p_rng_randint(void* stream, const uint32_t* key_dev, uint64_t n, int32_t minval,
                    int32_t maxval, int32_t* out_dev);

/* ---------------------------------------------------------------------------------------------
 * debug rng_suite: threefry2x32 in jax.random's array layout -- replaces d3p/random/debug.py:34-80.
 * The same generator produces the per-example guide noise (svi.py:259, :289-290).
 * ------------------------------------------------------------------------------------------- */
/* ABI 3: randint for the 8-, 16-, 32- and 64-bit integer dtypes (d3p/random/__init__.py:108-146, the dtype table at
 * :115-123; tests/test_random.py:74-135 draw int8 over the full range and int16 up to 2^15): delta and the power-of-two
 * mask are formed in the unsigned dtype of that width (float32 log2, :124-128), element j of random_bits(round_key, bit_width,
 * shape) is the little-endian bit_width-wide view of the keystream, the result wraps in the signed dtype.  out_dev holds n
 * elements of bit_width bits. */
int d3p_rng_randint_bits(void* stream, const uint32_t* key_dev, uint64_t n, int bit_width, int64_t minval, int64_t maxval,
                         void* out_dev);
int d3p_tf_split(void* stream, const uint32_t* key_dev, int num, uint32_t* out_keys_dev);
int d3p_tf_fold_in(void* stream, const uint32_t* key_dev, uint32_t data, uint32_t* out_key_dev);
int d3p_tf_random_bits(void* stream, const uint32_t* key_dev, uint64_t n_words, uint32_t* out_dev);
int d3p_tf_uniform(void* stream, const uint32_t* key_dev, uint64_t n, float minval, float maxval,
                   float* out_dev);
int d3p_tf_normal(void* stream, const uint32_t* key_dev, uint64_t n, float* out_dev);
/* jax.random.randint(key, shape, minval, maxval, int32) (d3p/random/debug.py:39) */
int d3p_tf_randint(void* stream, const uint32_t* key_dev, uint64_t n, int32_t minval, int32_t maxval,
                   int32_t* out_dev);

/* ---------------------------------------------------------------------------------------------
 * Minibatch samplers
 * ------------------------------------------------------------------------------------------- */
/* sample_from_array(key, arange(capacity), n, 0): 10-round Feistel permutation with cycle walking,
 * evaluated at positions 0..n-1 (d3p/util.py:216-301; minibatch.py:231, :289). */
int d3p_feistel_sample(void* stream, const uint32_t* key_dev, uint32_t capacity, uint32_t n,
                       uint32_t* out_idx_dev);

/* Same permutation from explicit round constants rc_dev[30] = rng_suite.random_bits(key, 32, (10, 3))
 * (util.py:240-242), so that any rng_suite (e.g. d3p.random.debug) can drive the sampler. */
int d3p_feistel_from_constants(void* stream, const uint32_t* rc_dev, uint32_t capacity, uint32_t n,
                               uint32_t* out_idx_dev);

/* poisson_sample_idxs + truncate/suppress bookkeeping (d3p/minibatch.py:29-39, :119-124).
 * Element j is selected iff uniform word j <= q.  out_idx_dev[cutoff]: selected indices in
 * descending order, then unselected indices in descending order (stable argsort reversed).
 * out_counts_dev[0] = number selected, [1] = valid count after truncate (suppress=0) or
 * suppress (suppress=1).  Rank r of `world` generates only keystream words of rows
 * [row_lo, row_hi) when those are passed (0, N for a single GPU). */
size_t d3p_poisson_select_workspace(uint32_t N);
int d3p_poisson_select(void* stream, const uint32_t* key_dev, float q, uint32_t N, uint32_t cutoff,
                       int suppress, uint32_t* out_idx_dev, uint32_t* out_counts_dev,
                       void* workspace_dev, size_t workspace_bytes);
/* rng_kind 0: key_dev is a ChaCha state (d3p.random); 1: a threefry key (d3p.random.debug). */
int d3p_poisson_select_rng(void* stream, int rng_kind, const uint32_t* key_dev, float q, uint32_t N,
                           uint32_t cutoff, int suppress, uint32_t* out_idx_dev,
                           uint32_t* out_counts_dev, void* workspace_dev, size_t workspace_bytes);

/* The same for `num_steps` independent draws in one set of launches (the fused run loop prepares 128 steps at
 * once): step t uses the key at keys_dev + t * key_stride_words and writes out_idx_dev + t * idx_stride_words,
 * out_counts_dev + t * counts_stride_words; the workspace is num_steps x d3p_poisson_select_workspace(N). */
int d3p_poisson_select_batch(void* stream, int rng_kind, const uint32_t* keys_dev, size_t key_stride_words,
                             float q, uint32_t N, uint32_t cutoff, int suppress, uint32_t* out_idx_dev,
                             size_t idx_stride_words, uint32_t* out_counts_dev, size_t counts_stride_words,
                             uint32_t num_steps, void* workspace_dev, size_t workspace_bytes);

/* The same selection made by the ranks of a row-sharded data-parallel run, each for the rows [row_lo, row_hi) it holds (SURVEY 8(e);
 * d3p/minibatch.py:29-39: element e's draw is keystream word e whoever generates it, so a rank makes the ChaCha20 blocks of its own
 * rows only -- 1 / world of the work -- and obtains the single-GPU mask bit for bit).  Two calls with an all-gather of the shards'
 * counts between them (d3p_xchg_poisson_counts below; the run loops with the one-shot exchange do all three per prepared batch):
 *   d3p_poisson_shard_flags: the shard's part of the mask and its selected count per step -> shard_counts_dev[t];
 *   d3p_poisson_shard_write: given, per step, counts {selected in the whole table, valid after truncate / suppress
 *     (minibatch.py:119-124)} at counts_dev + t * counts_stride_words and above_dev[t] = selected elements in the shards with HIGHER
 *     rows: the shard's valid selected rows at their GLOBAL batch positions in out_idx (descending row order, :36-37; other entries
 *     are not touched) and its dense list of owned positions (plist_dev, nullable; stride idx_stride_words like out_idx).
 * Same workspace for both (num_steps x d3p_poisson_select_workspace(16 * chunks of the shard) <= that of the whole table).  ABI 7. */
int d3p_poisson_shard_flags(void* stream, int rng_kind, const uint32_t* keys_dev, size_t key_stride_words, float q, uint32_t N,
                            uint32_t row_lo, uint32_t row_hi, uint32_t num_steps, uint32_t* shard_counts_dev, void* workspace_dev,
                            size_t workspace_bytes);
int d3p_poisson_shard_write(void* stream, uint32_t N, uint32_t row_lo, uint32_t row_hi, uint32_t cutoff, const uint32_t* counts_dev,
                            size_t counts_stride_words, const uint32_t* above_dev, uint32_t* out_idx_dev, uint32_t* plist_dev,
                            size_t idx_stride_words, uint32_t num_steps, void* workspace_dev, size_t workspace_bytes);

/* jnp.take(a, idx, axis=0) for a row-major table (minibatch.py:126-129, :210, :233, :306).
 * If valid_count_dev != NULL, output rows >= *valid_count_dev are zero-filled (the mask multiply
 * of minibatch.py:127-129).  row_bytes must be a multiple of 4.  An index >= n_rows is clamped to the last row
 * (jnp.take's "clip"; the samplers never produce one), never dereferenced. */
int d3p_take_rows(void* stream, const void* table_dev, uint64_t n_rows, uint32_t row_bytes,
                  const uint32_t* idx_dev, uint32_t n, const uint32_t* valid_count_dev,
                  void* out_dev);

/* ---------------------------------------------------------------------------------------------
 * DP-VI stages on materialised (B x P) per-example gradients.  The reference's tests call these
 * five stages directly (tests/test_dpsvi.py:128-129, :158-159, :171, :185, :199-202).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t d;         /* feature columns of X */
    int32_t intercept; /* 0/1; latent dimension D = d + intercept; P = 2*D */
    float prior_w;     /* prior std of w   (README.md:93; examples/logistic_regression.py:61) */
    float prior_b;     /* prior std of the intercept (examples/logistic_regression.py:62) */
    float lik_scale;   /* plate scale = num_obs_total (examples/logistic_regression.py:65) */
    float inv_obs;     /* 1 / observation_scale (svi.py:278) */
    /* ABI 2: likelihood family and guide transform.  Zero-initialised tails give the ABI-1 meaning. */
    int32_t family;          /* D3P_FAMILY_LOGREG: ys ~ Bernoulli(logits = xs.w + b)   (README.md:89-99)
                              * D3P_FAMILY_GAUSS_MEAN: obs ~ Normal(mu, lik_sigma).to_event(1), mu ~ Normal(0, prior_w)
                              *   (examples/simple_gaussian_posterior.py:51-65; y_dev unused, intercept must be 0) */
    int32_t guide_transform; /* D3P_GUIDE_SOFTPLUS: scale = softplus(u) (AutoDiagonalNormal);
                              * D3P_GUIDE_EXP: scale = exp(u) (the hand-written guides of the examples,
                              *   examples/simple_gaussian_posterior.py:77-81: mu_loc, mu_std_log) */
    float lik_sigma;         /* observation std of D3P_FAMILY_GAUSS_MEAN */
} d3p_logreg_model;

#define D3P_FAMILY_LOGREG 0
#define D3P_FAMILY_GAUSS_MEAN 1
#define D3P_GUIDE_SOFTPLUS 0
#define D3P_GUIDE_EXP 1

/* _compute_per_example_gradients for the logistic-regression + AutoDiagonalNormal workload
 * (svi.py:238-308).  params_dev = [auto_loc (D) | auto_scale unconstrained (D)].
 * Noise source: eps_dev (B x D, "parity mode") or, if NULL, generated on chip from the threefry
 * key jax_key_dev exactly as svi.py:289-290 + numpyro's handlers would.
 * mask_dev: B bytes (0/1) or NULL.  Outputs: px_loss_dev[B], px_grads_dev[B x P],
 * meta_dev[0] = num_elements, meta_dev[1] = batch_mask_scaling_factor (svi.py:305). */
size_t d3p_logreg_px_grads_workspace(const d3p_logreg_model* model, uint32_t B);
int d3p_logreg_px_grads(void* stream, const d3p_logreg_model* model, const float* params_dev,
                        const float* X_dev, const float* y_dev, const uint8_t* mask_dev, uint32_t B,
                        const float* eps_dev, const uint32_t* jax_key_dev, float* px_loss_dev,
                        float* px_grads_dev, float* meta_dev, void* workspace_dev,
                        size_t workspace_bytes);

/* DPSVI.evaluate (svi.py:436-449): -ELBO of the batch (B rows) at the current parameters with one guide
 * draw; jax_key_dev = convert_to_jax_rng_key(split(state.rng_key, 1)[0]); model->lik_scale = num_obs_total. */
size_t d3p_logreg_evaluate_workspace(const d3p_logreg_model* model, uint32_t B);
int d3p_logreg_evaluate(void* stream, const d3p_logreg_model* model, const float* params_dev,
                        const float* X_dev, const float* y_dev, uint32_t B, const uint32_t* jax_key_dev,
                        float* loss_dev, void* workspace_dev, size_t workspace_bytes);

/* GaussianMixture(locs, scales, pis).log_prob(x) for B rows of x (d3p/gmm.py:71-86; BASELINE config 3's
 * density).  x: B x d, locs/scales: K x d, pis: K (a simplex; validated by the host layer), K <= 64. */
int d3p_gmm_log_prob(void* stream, const float* x_dev, uint32_t B, int32_t d, const float* locs_dev,
                     const float* scales_dev, const float* pis_dev, int32_t K, float* out_dev);

/* _compute_per_example_gradients for the Gaussian-mixture MODEL of BASELINE config 3
 * (examples/gaussian_mixture_model.py:51-85: pis ~ Dirichlet(1), mus ~ Normal(0, prior_mu_scale), sigs ~
 * InverseGamma(1, 1), obs ~ GaussianMixture(mus, sigs, pis); guide pis ~ Dirichlet(exp(alpha_log)),
 * mus ~ Normal(mus_loc, 1), sigs ~ InverseGamma(1, 1); svi.py:238-308).
 * params_dev = [alpha_log (K) | mus_loc (K x d)], P = K + K d; one guide draw per example from jax_key_dev
 * (stream layout: oracle/d3p_oracle.c, d3po_gmm_*).  Outputs as d3p_logreg_px_grads; latents_out_dev (optional,
 * B x (K + 2 K d)) receives every example's (gamma draws, eps of mus, sigs).  K <= 16 with d <= 256 or
 * K <= 32 with d <= 128. */
typedef struct {
    int32_t K, d;
    float prior_mu_scale; /* 10 (examples/gaussian_mixture_model.py:65) */
    float lik_scale;      /* plate scale = num_obs_total */
    float inv_obs;        /* 1 / observation_scale (svi.py:278) */
} d3p_gmm_model;

size_t d3p_gmm_px_grads_workspace(int32_t K, uint32_t B);
int d3p_gmm_px_grads(void* stream, const d3p_gmm_model* model, const float* params_dev, const float* X_dev,
                     const uint8_t* mask_dev, uint32_t B, const uint32_t* jax_key_dev, float* px_loss_dev,
                     float* px_grads_dev, float* meta_dev, float* latents_out_dev, void* workspace_dev,
                     size_t workspace_bytes);

/* _clip_gradients: every row scaled by 1/max(1, ||row||_2 / c) in place (svi.py:68-124, :310-325).
 * c == 0 -> D3P_E_INVALID_ARG (the reference raises ValueError, svi.py:119-120). */
int d3p_clip_rows(void* stream, float* px_grads_dev, uint32_t B, uint32_t P, float c);

/* full_norm of a flat vector (svi.py:68-87) -> out_dev[0]. n == 0 gives 0. */
int d3p_full_norm(void* stream, const float* v_dev, uint64_t n, float* out_dev, void* workspace_dev,
                  size_t workspace_bytes);
/* ABI 5: numpy.linalg.norm(v, ord) of the same vector for any order (`ord` of full_norm / normalize_gradient,
 * d3p/svi.py:68-103): 0 = number of non-zero entries, 1, 2 (= d3p_full_norm), +-inf = largest / smallest magnitude, any
 * other p: (sum |x|^p)^(1/p).  n >= 1. */
int d3p_full_norm_ord(void* stream, const float* v_dev, uint64_t n, double ord, float* out_dev);

/* _combine_gradients: column means over the padded batch and mean loss (svi.py:327-348). */
int d3p_combine(void* stream, const float* px_grads_dev, const float* px_loss_dev, uint32_t B,
                uint32_t P, float* avg_dev, float* loss_dev);

/* _perturb_and_reassemble_gradients + perturbation_function (svi.py:350-377, :470-498):
 * out = (avg + normal(site_key) * dp_scale * c / n) * obs_scale * factor with one key per site
 * from split(key, n_sites); meta_dev = {n, factor} as produced above. */
int d3p_perturb(void* stream, const uint32_t* key_dev, const float* avg_dev,
                const int32_t* site_sizes_host, int n_sites, float dp_scale, float c,
                const float* meta_dev, float obs_scale, float* out_dev,
                uint32_t* site_keys_dev /* scratch: n_sites x 16 words */);

/* The same with the standard-normal draws supplied (noise_dev[n], from any rng_suite.normal):
 * out = (avg + noise * dp_scale * c / meta[0]) * obs_scale * meta[1]  (svi.py:365-375, :487-488). */
int d3p_perturb_apply(void* stream, const float* avg_dev, const float* noise_dev, uint64_t n,
                      float dp_scale, float c, const float* meta_dev, float obs_scale, float* out_dev);

/* numpyro.optim.Adam step (svi.py:379-393; examples/logistic_regression.py:141).
 * step_dev: int32 device counter (incremented). */
int d3p_adam_step(void* stream, float* params_dev, float* m_dev, float* v_dev, int32_t* step_dev,
                  const float* grads_dev, uint32_t P, float lr, float b1, float b2, float eps);

/* numpyro.optim.SGD step (tests/test_dpsvi.py:57): params -= lr * grads; step_dev incremented. */
int d3p_sgd_step(void* stream, float* params_dev, int32_t* step_dev, const float* grads_dev, uint32_t P,
                 float lr);

/* d3p.optimizers.ADADP step (optimizers.py:29-112; Koskela & Honkela's adaptive learning rate): even steps
 * store x_prev / x_stepped and take half a step, odd steps take the second half step, estimate the local error
 * err = ||(x_stepped - x) / max(1, x_stepped)||_2, rescale lr by min(max(sqrt(tol / err), 0.9), 1.1) and, with
 * stability_check, fall back to x_prev when err > tol.  lr_dev: one float; step_dev incremented.
 * Known answers: tests/test_adadp_optimizer.py:66-131. */
size_t d3p_adadp_workspace(void);
int d3p_adadp_step(void* stream, float* params_dev, float* lr_dev, float* x_stepped_dev, float* x_prev_dev,
                   int32_t* step_dev, const float* grads_dev, uint32_t P, float tol, int stability_check,
                   void* workspace_dev, size_t workspace_bytes);

/* ---------------------------------------------------------------------------------------------
 * Fused DPSVI.update (svi.py:395-434) -- the north-star path.
 * per-example gradient -> joint L2 clip -> sum, X read once, B x P never materialised.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float clip;      /* clipping_threshold C (svi.py:182) */
    float dp_scale;  /* sigma (svi.py:183) */
    float lr, b1, b2, adam_eps;
} d3p_dpsvi_hyper;

/* Device-resident training state (DPSVIState, svi.py:37-40, plus Adam moments).  All pointers
 * are device addresses owned by the caller. */
typedef struct {
    uint32_t* rng_key;   /* 2 x 16 words: ping-pong ChaCha state keys */
    int32_t key_slot;    /* host value: which of the two slots holds the current key at entry; a
                            call that performs k updates leaves it in slot (key_slot + k) & 1 */
    float* params;       /* P = 2D: [auto_loc | auto_scale unconstrained] */
    float* adam_m;       /* P */
    float* adam_v;       /* P */
    int32_t* step;       /* optimiser step counter i (numpyro _NumPyroOptim state) */
} d3p_dpsvi_state;

/* Where a step's batch comes from. */
#define D3P_BATCH_EXPLICIT 0 /* X/y ARE the batch (B rows), optional mask: DPSVI.update(state, X, y, mask=) */
#define D3P_BATCH_FEISTEL 1  /* subsample_batchify_data get_batch(i, key) fused in (minibatch.py:217-237) */
#define D3P_BATCH_POISSON 2  /* poisson_batchify_data get_batch(i, key) fused in (minibatch.py:103-131) */

typedef struct {
    int32_t kind;            /* D3P_BATCH_* */
    uint32_t B;              /* batch size (FEISTEL), max_batch_size (POISSON), rows (EXPLICIT) */
    float q;                 /* POISSON sampling rate */
    int32_t suppress;        /* POISSON: handle_oversized_batch == "suppress" */
    const uint32_t* batch_key; /* batchifier state key (16 words, device); NULL for EXPLICIT */
    uint32_t* batch_index;   /* device counter i, fold_in(key, i); incremented per step */
    const uint8_t* mask;     /* EXPLICIT only: B bytes or NULL */
    uint64_t n_rows;         /* N: rows of the full (global) table */
    uint64_t row_lo, row_hi; /* rows held by this rank: X_dev/y_dev point at row_lo (0, N on 1 GPU) */
} d3p_batch_source;

size_t d3p_dpvi_logreg_workspace(const d3p_logreg_model* model, const d3p_batch_source* src);

/* Phase 1 (per rank): key schedule, minibatch indices, fused per-example gradient / clip / sum.
 * sums_dev[P + 2] = [sum_i c_i g_i (P) | sum_i masked loss_i | number of valid examples] over
 * the examples whose rows this rank holds.  eps_dev: optional B x D parity-mode noise. */
int d3p_dpvi_logreg_local_sums(void* stream, const d3p_logreg_model* model,
                               const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                               const d3p_batch_source* src, const float* X_dev, const float* y_dev,
                               const float* eps_dev, float* sums_dev, void* workspace_dev,
                               size_t workspace_bytes);

/* Phase 2 (replicated): mean, Gaussian mechanism (noise added ONCE, after any all-reduce of
 * sums_dev), rescale, Adam, advance keys/counters.  loss_dev[0] receives the batch loss
 * (svi.py:342, :306); grad_out_dev (P, optional) the perturbed gradient. */
int d3p_dpvi_logreg_finalize(void* stream, const d3p_logreg_model* model,
                             const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                             const d3p_batch_source* src, const float* sums_dev, float* loss_dev,
                             float* grad_out_dev, void* workspace_dev, size_t workspace_bytes);

/* Stepwise form of the two phases for the data-parallel loop (d3p_amd/dist.py): the key schedule and
 * the sampler are evaluated for up to 128 steps at once (`prepare`), so that each step only launches
 * the fused kernel + partial reduction (`step_sums`), and -- after the caller's sum-all-reduce of
 * sums_dev -- `step_finalize`.  `t` is the step index inside the prepared batch.
 *   begin -> { prepare(K) -> K x [ step_sums(t) -> all-reduce -> step_finalize(t) ] }* -> end(total steps)
 * `end` stores the key after `steps_done` updates in slot (key_slot + steps_done) & 1. */
int d3p_dpvi_logreg_begin(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                          const d3p_dpsvi_state* state, const d3p_batch_source* src,
                          void* workspace_dev, size_t workspace_bytes);
int d3p_dpvi_logreg_prepare(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                            const d3p_dpsvi_state* state, const d3p_batch_source* src,
                            uint32_t num_steps, void* workspace_dev, size_t workspace_bytes);
int d3p_dpvi_logreg_step_sums(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                              const d3p_dpsvi_state* state, const d3p_batch_source* src, uint32_t t,
                              const float* X_dev, const float* y_dev, const float* eps_dev,
                              float* sums_dev, void* workspace_dev, size_t workspace_bytes);
int d3p_dpvi_logreg_step_finalize(void* stream, const d3p_logreg_model* model,
                                  const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                                  const d3p_batch_source* src, uint32_t t, const float* sums_dev,
                                  float* loss_dev, float* grad_out_dev, void* workspace_dev,
                                  size_t workspace_bytes);
int d3p_dpvi_logreg_end(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                        const d3p_dpsvi_state* state, const d3p_batch_source* src,
                        uint32_t steps_done, void* workspace_dev, size_t workspace_bytes);

/* One-launch-per-step form (what d3p_dpvi_logreg_run uses internally), exposed for the data-parallel loop:
 * the cross-workgroup sums of a step live in a 64-bit FIXED-POINT accumulator (integer addition is
 * associative: exact and bitwise reproducible, on one GPU and under any all-reduce order); the update of
 * step g is applied in the prologue of launch g+1.  Accumulator g % 3 of the workspace (layout from
 * d3p_dpvi_logreg_acc_layout: byte offset, int64 words per buffer; three consecutive buffers) is what the
 * caller sum-all-reduces (int64) between launch g and launch g+1.
 *   begin -> acc_reset -> { prepare_buf(K, buf) -> K x [ fused_step(g, t, buf, ...) -> all-reduce(acc[g % 3]) ] }*
 *         -> fused_step(flush_only = 1) -> end(total steps) */
int d3p_dpvi_logreg_prepare_buf(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                                const d3p_dpsvi_state* state, const d3p_batch_source* src,
                                uint32_t num_steps, int buf, void* workspace_dev, size_t workspace_bytes);
int d3p_dpvi_logreg_acc_layout(const d3p_logreg_model* model, const d3p_batch_source* src,
                               size_t* offset_bytes, size_t* words_per_buffer);
/* 1 when the model's rows fit the one-launch step (d3p_dpvi_logreg_fused_step), 0 when they run as two-kernel steps (rows too wide for
 * the register-tiled kernel: D > 2048, or D > 1024 unless d % 8 == 0 without intercept): fused_step then returns D3P_E_UNSUPPORTED.  ABI 9. */
int d3p_dpvi_logreg_fused_step_supported(const d3p_logreg_model* model, const d3p_batch_source* src);
int d3p_dpvi_logreg_acc_reset(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                              const d3p_dpsvi_state* state, const d3p_batch_source* src,
                              void* workspace_dev, size_t workspace_bytes);
int d3p_dpvi_logreg_fused_step(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                               const d3p_dpsvi_state* state, const d3p_batch_source* src, uint32_t g,
                               uint32_t t, int buf, int have_prev, uint32_t prev_t, int prev_buf,
                               const float* X_dev, const float* y_dev, float* prev_loss_dev,
                               int flush_only, void* workspace_dev, size_t workspace_bytes);

/* Single-GPU convenience: `num_steps` x (phase 1 + phase 2) enqueued back to back, i.e. the body of
 * the reference's jit(fori_loop(update)) epoch (examples/logistic_regression.py:149-160).
 * losses_dev: num_steps floats (or NULL). */
int d3p_dpvi_logreg_run(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                        const d3p_dpsvi_state* state, const d3p_batch_source* src,
                        const float* X_dev, const float* y_dev, uint32_t num_steps,
                        float* losses_dev, void* workspace_dev, size_t workspace_bytes);

/* The same run as a function of an immutable state (DPSVI.update returns a NEW state, svi.py:395-434; the epoch body of
 * examples/logistic_regression.py:149-160 threads it through fori_loop): it starts from `from` (rng key in slot
 * from->key_slot, optimiser state and step counter; read only) and from batch index `first_batch` (by value;
 * src->batch_index may be NULL) and leaves the result in `state` (state->key_slot must be 0; the final key is in slot
 * num_steps & 1 of state->rng_key).  The state copy and the batch-index word are written by the run's first kernel, so a
 * caller with functional semantics needs no launches of its own around the call.  ABI 4. */
int d3p_dpvi_logreg_run_from(void* stream, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                             const d3p_dpsvi_state* state, const d3p_dpsvi_state* from, const d3p_batch_source* src,
                             uint32_t first_batch, const float* X_dev, const float* y_dev, uint32_t num_steps,
                             float* losses_dev, void* workspace_dev, size_t workspace_bytes);
/* ... and the data-parallel runs d3p_dpvi_logreg_run_dist / d3p_dpvi_logreg_run_xchg (declared below) in the same form:
 * comm (d3p_comm_*: RCCL) or xchg (d3p_xchg_*: one-shot exchange) or neither; the rank's shard is src->row_lo .. row_hi. */
int d3p_dpvi_logreg_run_dist_from(void* stream, void* comm, void* xchg, const d3p_logreg_model* model,
                                  const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state, const d3p_dpsvi_state* from,
                                  const d3p_batch_source* src, uint32_t first_batch, const float* X_dev, const float* y_dev,
                                  uint32_t num_steps, float* losses_dev, void* workspace_dev, size_t workspace_bytes);

/* Times only the dominant kernel (fused gradient/clip/sum) of one step over `reps` launches on
 * `stream` (host out-pointers):
 *   *avg_us       device-side duration: every workgroup stamps the 100 MHz wall clock at entry and
 *                 exit, duration = last exit - first entry (what rocprofv3's kernel trace reports);
 *   *avg_event_us HIP start/stop events recorded around each launch (hipExtLaunchKernel); includes
 *                 the ~4 us dispatch floor an empty kernel also shows (optional, may be NULL).
 * Synchronises the stream; measurement tooling for bench.py, not part of the training path. */
int d3p_dpvi_logreg_time_main_kernel(void* stream, const d3p_logreg_model* model,
                                     const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                                     const d3p_batch_source* src, const float* X_dev,
                                     const float* y_dev, void* workspace_dev, size_t workspace_bytes,
                                     int reps, float* avg_us, float* avg_event_us);

/* DPSVI.evaluate (svi.py:436-449) for the mixture model: -ELBO of the batch with one guide draw;
 * jax_key_dev = convert_to_jax_rng_key(split(state.rng_key, 1)[0]); model->lik_scale = num_obs_total. */
size_t d3p_gmm_evaluate_workspace(const d3p_gmm_model* model, uint32_t B);
int d3p_gmm_evaluate(void* stream, const d3p_gmm_model* model, const float* params_dev, const float* X_dev, uint32_t B,
                     const uint32_t* jax_key_dev, float* loss_dev, void* workspace_dev, size_t workspace_bytes);

/* DPSVI.update (svi.py:395-434) for the mixture model in one call: key split, per-example gradients clipped and
 * summed without materialising B x P, per-site Gaussian noise, numpyro Adam, all enqueued on `stream`.  state as for
 * the logistic-regression path (params = [alpha_log | mus_loc], P = K + K d; the new state key lands in the other
 * key slot).  loss_dev[0]: batch loss; grad_out_dev (P, optional): the perturbed gradient. */
size_t d3p_dpvi_gmm_workspace(const d3p_gmm_model* model, uint32_t B);
int d3p_dpvi_gmm_update(void* stream, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper,
                        const d3p_dpsvi_state* state, const float* X_dev, const uint8_t* mask_dev, uint32_t B,
                        float* loss_dev, float* grad_out_dev, void* workspace_dev, size_t workspace_bytes);

/* Data-parallel form of d3p_dpvi_gmm_update, as for the VAE below: a rank holds the B_local examples at positions
 * pos0 .. pos0 + B_local - 1 of the global batch of B_total (the per-example site keys are functions of the GLOBAL position).
 * d3p_dpvi_gmm_local_sums leaves sums_dev[P + 2] = [sum_i c_i g_i | sum_i loss_i | n] of the rank's examples and does not
 * touch the state; the caller sum-all-reduces sums_dev over the ranks; d3p_dpvi_gmm_apply adds the per-site noise once,
 * applies Adam and advances the state identically on every rank.  The workspace is sized for B_local. */
int d3p_dpvi_gmm_local_sums(void* stream, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper,
                            const d3p_dpsvi_state* state, const float* X_dev, const uint8_t* mask_dev, uint32_t B_local,
                            uint32_t B_total, uint32_t pos0, float* sums_dev, void* workspace_dev, size_t workspace_bytes);
int d3p_dpvi_gmm_apply(void* stream, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                       float* sums_dev, uint32_t B_total, uint32_t B_local, float* loss_dev, float* grad_out_dev,
                       void* workspace_dev, size_t workspace_bytes);

/* The single-GPU run loop executes the steps of a prepared batch (<= 128) as ONE launch whose workgroups wait on
 * arrival counters for the previous step (bounded waits).  This reads back, after synchronising `stream`, whether any
 * wait of the last run hit its bound (aborted_out != 0: the results of that run are invalid). */
int d3p_dpvi_logreg_chain_status(void* stream, const d3p_logreg_model* model, const d3p_batch_source* src,
                                 void* workspace_dev, size_t workspace_bytes, int32_t* aborted_out);

/* ABI 3: both sticky status words of the last run, after synchronising `stream`.
 *   aborted_out   != 0: a bounded wait of the chained launch ran out; from then on no workgroup applied, published or
 *                       arrived anywhere, i.e. the run stopped advancing -- state and losses of that run are invalid
 *                       (d3p_amd.svi.DPSVI.run_steps raises D3PError).  The value is the code of the FIRST wait that ran
 *                       out: kind | step of its launch << 8 | detail << 20, kinds 1 a wait not told apart, 2 exchange
 *                       workgroup waiting for the step's arrivals, 3 exchange workgroup waiting for the row of rank
 *                       `detail`, 4 compute workgroup waiting for the previous step's release, 5 key-chain link, 6 k_xchg
 *                       waiting for the row of rank `detail` (D3P_ABORT_* in csrc/d3p_logreg_kernel.h;
 *                       d3p_amd._lib.describe_abort turns it into text).  All waits of a launch start with it and run out
 *                       together, so the step names a waiter, not necessarily the place the run stood at; D3P_DBG=64 prints
 *                       the earliest step per kind.
 *   nonfinite_out != 0: a workgroup partial was NaN / Inf or left the fixed-point range of the accumulator; the update that
 *                       followed turned the parameters and the loss into NaN, which is what the reference's float sums
 *                       (svi.py:342-346) give for a diverged model -- never finite garbage. */
int d3p_dpvi_logreg_run_status(void* stream, const d3p_logreg_model* model, const d3p_batch_source* src,
                               void* workspace_dev, size_t workspace_bytes, int32_t* aborted_out, int32_t* nonfinite_out);

/* ABI 8: launch geometry of the chained launch for this model and batch source, without running anything: workgroups per step
 * (0: the shape takes the generic one-launch-per-step kernels) and waves per workgroup, for a single-rank run (data_parallel = 0)
 * or a rank of a data-parallel run with the in-launch exchange (1: src->row_lo / row_hi give its share of the batch).  What the
 * tests use to make sure they cover grids that are NOT a multiple of the 8 XCDs (arrival groups, updaters: DESIGN.md section 6u). */
int d3p_dpvi_logreg_chain_grid(const d3p_logreg_model* model, const d3p_batch_source* src, int data_parallel,
                               uint32_t* workgroups_per_step_out, int32_t* waves_out);

/* Form of the run loops' launches: 0 (default) -- the chained launch where the shape has one: the steps of a prepared batch in
 * ONE launch whose workgroups hand over through arrival counters, which needs the launch's workgroups to make progress
 * together (a GPU shared with other work can starve one: the bounded waits then stop the run, d3p_dpvi_logreg_run_status);
 * 1 -- one launch per step: no cross-workgroup waits at all, ~2 x slower.  DPSVI.run_steps re-runs a stopped run in form 1
 * (the reference's jit(fori_loop) cannot stall; a drop-in must not either).  The switch belongs to the CALLING THREAD and a run
 * reads it once, when it is enqueued: setting it around one run cannot change the form of a run another thread is enqueueing.  ABI 6. */
int d3p_dpvi_logreg_set_run_form(int form);

/* Measurement hook for the run loops (d3p_dpvi_logreg_run, d3p_dpvi_logreg_run_dist): while enabled, every launch of the
 * step kernel is bracketed by HIP start/stop events on the launch stream (hipExtLaunchKernel).
 * d3p_dpvi_logreg_kernel_timing_read synchronises the recorded events, returns the summed kernel time (microseconds), the
 * number of launches and the number of DP-VI steps they covered since the last read, and clears the record.  Process-wide
 * switch, not thread-safe; used by bench.py for the roofline figure of the kernel that runs in the timed region. */
int d3p_dpvi_logreg_kernel_timing_enable(int enable);
int d3p_dpvi_logreg_kernel_timing_read(double* total_us_out, uint32_t* launches_out, uint32_t* steps_out);

/* ---------------------------------------------------------------------------------------------
 * Data-parallel run over the GPUs of one node (the reference is single-device).  One process per GPU;
 * every rank calls the same sequence.  The communicator is RCCL's, created from an id that rank 0 obtains and the
 * host layer distributes (torch.distributed broadcast); librccl.so is resolved at run time.
 * d3p_dpvi_logreg_run_dist = d3p_dpvi_logreg_run on this rank's row shard (src->row_lo / row_hi) with ONE collective
 * per step: the in-place sum-all-reduce of the rank's fixed-point accumulator (4 x (P + 2) int64 words: exact integer
 * sums, so every rank applies bitwise the same update under any reduction order); the Gaussian noise is added once,
 * after the reduce, from the same key on every rank.  comm == NULL runs without the collective.
 * ------------------------------------------------------------------------------------------- */
/* One-shot full-mesh exchange over xGMI (ABI 3; SURVEY 5 / 8e: for a message of 8 KB one hop beats a ring's 2 (n - 1)).
 * Every rank owns an inbox (uncached device memory, double-buffered slots of self-validating words: 32 data bits + the
 * 32-bit tag of the exchange's epoch per 8-byte word, so that a row is its own arrival signal) that its peers map with hipIpc:
 *   d3p_xchg_create   allocates the inbox for messages of `words` int64 words and returns its 80-byte handle (ABI 9: D3P_IPC_HANDLE_BYTES --
 *                     the inbox is a range of ONE hipIpc arena per process, exported once and mapped once by every peer: the handle is
 *                     the arena's 64-byte hipIpc handle + the range's offset + a marker; handle_bytes / handle_stride >= 80);
 *   d3p_xchg_connect  takes the handles of ALL ranks (world x handle_stride bytes, the host layer gathers them, e.g. with
 *                     torch.distributed.all_gather_object) and maps the peers' inboxes;
 *   d3p_xchg_allreduce  enqueues ONE kernel: fold acc_dev[replicas][words] -> write the folded row into slot [rank] of every
 *                     inbox -> poll the n rows of the own inbox until every word carries the epoch's tag -> acc_dev row 0 =
 *                     the sum of the n rows (int64: exact, identical on every rank), rows 1.. = 0.  Every rank must call it
 *                     the same number of times.  Bounded waits (a run stopped by one reports which: D3P_ABORT_* in
 *                     d3p_logreg_kernel.h, handed out as `aborted` by d3p_dpvi_logreg_run_status).
 *   d3p_dpvi_logreg_run_xchg = d3p_dpvi_logreg_run_dist with this exchange as the step's one collective (`words` must be
 *                     the accumulator row of the model: 2 D + 4). */
int d3p_xchg_create(int32_t world, int32_t rank, uint32_t words, void** xchg_out, uint8_t* handle_out, size_t handle_bytes);
int d3p_xchg_connect(void* xchg, const uint8_t* handles, size_t handle_stride);
 /* All-gather of the shards' selected counts of the <= 128 steps of a prepared batch through the exchange's count box (tagged words like
 * the rows; its own epoch), for d3p_poisson_shard_write: per step t counts_dev[t * counts_stride_words + {0, 1}] = {selected in the
 * whole table, valid after truncate (cutoff) / suppress}, above_dev[t] = selected in the shards of the HIGHER ranks (the table is
 * sharded contiguously in rank order), n_owned_dev[t * n_owned_stride_words] = this rank's valid selected rows.  Every rank of the
 * exchange must call it the same number of times.  ABI 7. */
int d3p_xchg_poisson_counts(void* stream, void* xchg, const uint32_t* shard_counts_dev, uint32_t num_steps, uint32_t cutoff, int suppress,
                            uint32_t* counts_dev, size_t counts_stride_words, uint32_t* above_dev, uint32_t* n_owned_dev,
                            size_t n_owned_stride_words);

/* Test / rehearsal helper: the other world - 1 ranks of an exchange played by ONE workgroup on this GPU, enqueued on `stream` (not
 * the stream of the run): for each of the next `num_exchanges` exchanges it waits until this rank's row has arrived in its next
 * peer's inbox and then delivers all-zero rows of every other rank to this rank's inbox.  A run of the 8-rank code paths on one GPU
 * (tests/test_dist.py); the result equals the rank's run with an exchange of its own.  Needs mapped peer inboxes.  ABI 7. */
int d3p_xchg_simulate_peers(void* stream, void* xchg, uint32_t num_exchanges);

/* d3p_xchg_connect_local: the same for ranks that live in ONE process (one stream each): `peers` = the `world` exchange
  * objects of the group, in rank order; their inboxes are wired directly. */
int d3p_xchg_connect_local(void* xchg, void* const* peers, int32_t world);
int d3p_xchg_disconnect(void* xchg);   /* ABI 9: unmap the peers' inboxes; teardown = every rank disconnects -> barrier -> every rank destroys */
int d3p_xchg_destroy(void* xchg);
int d3p_xchg_allreduce(void* stream, void* xchg, long long* acc_dev, int32_t replicas);
int d3p_dpvi_logreg_run_xchg(void* stream, void* xchg, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                             const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev, const float* y_dev,
                             uint32_t num_steps, float* losses_dev, void* workspace_dev, size_t workspace_bytes);

int d3p_comm_unique_id(uint8_t* id_out, size_t id_bytes); /* id_bytes >= 128 */
int d3p_comm_init(const uint8_t* id, size_t id_bytes, int32_t nranks, int32_t rank, void** comm_out);
int d3p_comm_destroy(void* comm);
int d3p_dpvi_logreg_run_dist(void* stream, void* comm, const d3p_logreg_model* model, const d3p_dpsvi_hyper* hyper,
                             const d3p_dpsvi_state* state, const d3p_batch_source* src, const float* X_dev,
                             const float* y_dev, uint32_t num_steps, float* losses_dev, void* workspace_dev,
                             size_t workspace_bytes);

/* num_steps x (get_batch(first_batch + t, batch_key) of subsample_batchify_data -> update) on the resident table
 * X_dev (n_rows x d): the body of the example's jit(fori_loop(...)) epoch (examples/gaussian_mixture_model.py:219-230
 * with the subsampling batchifier).  losses_dev[num_steps] optional. */
int d3p_dpvi_gmm_run(void* stream, const d3p_gmm_model* model, const d3p_dpsvi_hyper* hyper,
                     const d3p_dpsvi_state* state, const uint32_t* batch_key_dev, uint32_t first_batch,
                     const float* X_dev, uint32_t n_rows, uint32_t B, uint32_t num_steps, float* losses_dev,
                     void* workspace_dev, size_t workspace_bytes);

/* ---------------------------------------------------------------------------------------------
 * Variational auto-encoder of BASELINE config 5 (examples/vae.py:65-153) -- the GEMM-shaped model on the path.
 * encoder: h1 = softplus(x W1 + b1), z_loc = h1 Wl + bl, z_std = exp(h1 Ws + bs);  decoder: h2 = softplus(z V1 + c1),
 * obs ~ Bernoulli(sigmoid(h2 V2 + c2)).  Parameter vector = leaves of {'decoder$params', 'encoder$params'} in
 * tree_flatten order: V1 (Z x H), c1 (H), V2 (H x D), c2 (D), W1 (D x H), b1 (H), Wl (H x Z), bl (Z), Ws (H x Z), bs (Z).
 * H2 > 0 (ABI 5): BASELINE config 5's literal 784 -> [400, 200] -> 50 variant, one more dense softplus layer on each side
 * (the reference itself has one hidden layer): encoder x -> H -> H2 -> heads, decoder z -> H2 -> H -> D; leaves
 * V1 (Z x H2), c1, V2 (H2 x H), c2, V3 (H x D), c3, W1 (D x H), b1, W2 (H x H2), b2, Wl (H2 x Z), bl, Ws (H2 x Z), bs
 * (P = 819 284 at 784 / [400, 200] / 50).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t D, H, Z;  /* observation, hidden and latent dimension (784, 400, 50) */
    float scale;      /* scale of every site: plate scale x handlers.scale (vae.py:194-195 -> 1) */
    float inv_obs;    /* 1 / observation_scale (svi.py:278) */
    int32_t H2;       /* width of the second hidden layer; 0 = the reference's one-hidden-layer network */
} d3p_vae_model;

int64_t d3p_vae_num_params(const d3p_vae_model* model);
size_t d3p_dpvi_vae_workspace(const d3p_vae_model* model, uint32_t B);

/* fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32): C[M x N] = alpha * op(A) op(B) (+ bias[n]) (+ C) with element
 * strides A(m, k) = A[m a_sm + k a_sk], B(k, n) = B[k b_sk + n b_sn]; C row-major, leading dimension ldc. */
int d3p_gemm_f32(void* stream, const float* A_dev, int64_t a_sm, int64_t a_sk, const float* B_dev, int64_t b_sk,
                 int64_t b_sn, float* C_dev, int32_t ldc, int32_t M, int32_t N, int32_t K, const float* bias_dev,
                 float alpha, int32_t accumulate);

/* Stages 1-3 of DPSVI.update fused (svi.py:238-348): sums_dev[P + 2] = [sum_i c_i g_i | sum_i loss_i | n] without ever
 * forming a per-example gradient: norms by ||a d^T||_F = ||a|| ||d||, clipped sums as A^T (diag(c) Delta) GEMMs.
 * eps_dev (B x Z, optional parity-mode noise) or jax_key_dev (per-example threefry keys, svi.py:289-290).
 * norms_dev (B, optional): per-example gradient norms before clipping; px_loss_dev (B, optional). */
int d3p_vae_step_sums(void* stream, const d3p_vae_model* model, const float* params_dev, const float* X_dev,
                      const uint8_t* mask_dev, uint32_t B, const float* eps_dev, const uint32_t* jax_key_dev, float clip,
                      float* sums_dev, float* norms_dev, float* px_loss_dev, void* workspace_dev,
                      size_t workspace_bytes);

/* DPSVI.evaluate (svi.py:436-449) for the VAE: -ELBO of the batch with one guide draw (eps for all B rows from one key,
 * jax_key_dev = convert_to_jax_rng_key(split(state.rng_key, 1)[0]); eps_dev optionally overrides it);
 * model->scale = handlers.scale x num_obs_total / B (the plate scale of the batch), model->inv_obs = 1. */
int d3p_vae_evaluate(void* stream, const d3p_vae_model* model, const float* params_dev, const float* X_dev, uint32_t B,
                     const uint32_t* jax_key_dev, const float* eps_dev, float* loss_dev, void* workspace_dev,
                     size_t workspace_bytes);

/* One DPSVI.update (svi.py:395-434) for the VAE: key split, the fused sums above, one Gaussian-noise key per
 * parameter leaf (svi.py:487-491; 10 leaves, 14 with model->H2 > 0), numpyro Adam; state as for the other models (P = d3p_vae_num_params). */
int d3p_dpvi_vae_update(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper,
                        const d3p_dpsvi_state* state, const float* X_dev, const uint8_t* mask_dev, uint32_t B,
                        const float* eps_dev, float* loss_dev, float* grad_out_dev, void* workspace_dev,
                        size_t workspace_bytes);
/* The same as a function of an immutable state (as d3p_dpvi_logreg_run_from): reads `from`, writes the new state into
 * `state` (key_slot 0; the next key goes to slot 1 of state->rng_key) -- no 3 x P-float state copy per update.  ABI 4. */
int d3p_dpvi_vae_update_from(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper,
                             const d3p_dpsvi_state* state, const d3p_dpsvi_state* from, const float* X_dev,
                             const uint8_t* mask_dev, uint32_t B, const float* eps_dev, float* loss_dev, float* grad_out_dev,
                             void* workspace_dev, size_t workspace_bytes);

/* num_steps x (get_batch(first_batch + t, batch_key) of subsample_batchify_data -> update) on the resident data set
 * X_dev (n_rows x D): the body of the example's jit(lax.fori_loop(...)) epoch (examples/vae.py:227-246).  Per step
 * fold_in, the Feistel indices, the row gather and the update are enqueued back to back; nothing waits for the host.
 * state is advanced in place: its key is read from slot state->key_slot of state->rng_key (2 x 16 words) and ends up
 * in slot (key_slot + num_steps) & 1.  xb_dev: B x D floats (the gathered batch); idx_dev: B + 16 uint32.  ABI 6. */
int d3p_dpvi_vae_run(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                     const uint32_t* batch_key_dev, uint32_t first_batch, const float* X_dev, uint32_t n_rows, uint32_t B,
                     uint32_t num_steps, float* losses_dev, float* xb_dev, uint32_t* idx_dev, void* workspace_dev,
                     size_t workspace_bytes);

/* Data-parallel form of d3p_dpvi_vae_update (BASELINE config 5, "1 vs 8 GPU"): a rank holds the B_local
 * examples at positions pos0 .. pos0 + B_local - 1 of the global batch of B_total (per-example noise keys are functions of
 * the GLOBAL position, svi.py:289-290).  d3p_dpvi_vae_local_sums leaves sums_dev[P + 2] = [sum_i c_i g_i | sum_i loss_i | n]
 * of the rank's examples; the caller sum-all-reduces sums_dev over the ranks (the ONE collective of the step, 2.76 MB at
 * 784/400/50) and every rank calls d3p_dpvi_vae_apply with the reduced sums: noise once per parameter leaf from the same
 * perturbation key on every rank (svi.py:487-491), identical Adam step, new state key in the other key slot -- replicas stay
 * identical without a broadcast.  state is read by both calls and advanced by apply only.  The workspace is sized for
 * B_local.  d3p_dpvi_vae_update is exactly local_sums + apply with one rank. */
int d3p_dpvi_vae_local_sums(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper,
                            const d3p_dpsvi_state* state, const float* X_dev, const uint8_t* mask_dev, uint32_t B_local,
                            uint32_t B_total, uint32_t pos0, const float* eps_dev, float* sums_dev, void* workspace_dev,
                            size_t workspace_bytes);
int d3p_dpvi_vae_apply(void* stream, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper, const d3p_dpsvi_state* state,
                       const float* sums_dev, uint32_t B_total, uint32_t B_local, float* loss_dev, float* grad_out_dev,
                       void* workspace_dev, size_t workspace_bytes);

/* ABI 8: the data-parallel epoch body as ONE call (examples/vae.py:227-246 with the batch sharded by position): num_steps x
 * [d3p_dpvi_vae_local_sums on the RESIDENT shard X_local_dev -> sum-all-reduce of the P + 2 fp32 sums over `comm` (d3p_comm_*:
 * RCCL; NULL = one rank) -> d3p_dpvi_vae_apply], enqueued back to back on `stream`: nothing waits for the host, the state advances
 * in place (its key ends in slot (key_slot + num_steps) & 1 of state->rng_key, as d3p_dpvi_vae_run leaves it), the step's keys
 * are derived once, the Gaussian-mechanism noise is drawn beside the latent kernel and apply is one launch behind the reduce.
 * buckets = 2: the sums travel in two buckets on a second stream, the decoder's leaves while the encoder's weight-gradient products
 * still run (every clipped sum needs the whole backward pass first: only the tail of the step can overlap a reduce); 1: one
 * all-reduce on `stream`; 0: the library's choice.  fmesh (d3p_fmesh_*, below) instead of comm: the full-mesh collective -- by
 * default FUSED with the tile sums in front of it and the update behind it into one launch (k_vae_fmesh_step: the scatter phase reads
 * the split-K partial tiles, the gather phase applies noise + Adam to a column the moment its sum has arrived); buckets = 1 keeps the
 * three launches apart (same results bit for bit).  losses_dev: num_steps floats or NULL. */
int d3p_dpvi_vae_run_dist(void* stream, void* comm, void* fmesh, const d3p_vae_model* model, const d3p_dpsvi_hyper* hyper,
                          const d3p_dpsvi_state* state, const float* X_local_dev, const uint8_t* mask_dev, uint32_t B_local,
                          uint32_t B_total, uint32_t pos0, uint32_t num_steps, float* losses_dev, int32_t buckets,
                          void* workspace_dev, size_t workspace_bytes);

/* ABI 8: full-mesh sum-all-reduce of a float vector over the GPUs of one node -- the collective of the data-parallel VAE step as
 * the survey asks for it (for MB-sized messages on point-to-point xGMI a full-mesh reduce-scatter + all-gather: two
 * hops with every link carrying 1 / world of the vector, where a ring takes 2 (world - 1) dependent hops).  The reference has no
 * collective (single device).  Rank r owns chunk r of the vector: every rank stores its partials of chunk o into rank o's inbox,
 * the owner adds the world's partials in RANK ORDER (every rank then receives bit for bit the same sums) and stores them into every
 * peer's gather inbox.  A float travels as one 8-byte word {fp32 bits | epoch tag}: the data is its own arrival signal, no fence,
 * no flag.  Inboxes are uncached device memory mapped into the peers with hipIpc handles (d3p_fmesh_create -> exchange the 80-byte
 * handles (D3P_IPC_HANDLE_BYTES, as d3p_xchg_create) -> d3p_fmesh_connect; d3p_fmesh_connect_local wires meshes that live in one process).  d3p_fmesh_allreduce: one launch on
 * `stream`, in place, bounded waits; d3p_fmesh_status (after synchronising `stream`): non-zero when a wait ran out -- the vector of
 * that and every later call is then undefined.  d3p_dpvi_vae_run_dist(..., fmesh) uses it instead of RCCL. */
int d3p_fmesh_create(int32_t world, int32_t rank, uint64_t n_floats, void** fmesh_out, uint8_t* handle_out, size_t handle_bytes);
int d3p_fmesh_connect(void* fmesh, const uint8_t* handles, size_t handle_stride);
int d3p_fmesh_connect_local(void* fmesh, void* const* peers, int32_t world);
int d3p_fmesh_set_grid(void* fmesh, int32_t workgroups);   /* workgroups per launch (default 512 = two per CU, ALL resident together; D3P_FMESH_WGS overrides; ranks that share a GPU: fewer) */
int d3p_fmesh_allreduce(void* stream, void* fmesh, float* buf_dev, uint64_t n_floats);
int d3p_fmesh_status(void* stream, void* fmesh, int32_t* stopped_out);
int d3p_fmesh_disconnect(void* fmesh);   /* ABI 9: as d3p_xchg_disconnect */
int d3p_fmesh_destroy(void* fmesh);

/* Self-test of the wave-level sums the step kernels are built on (wave_sum / wave_sum2: DPP adds from inline assembly), taken
 * directly behind divergent branches: in_dev holds 64 floats per wave, out_dev[3 w + {0, 1, 2}] = the sum of wave w's inputs by
 * wave_sum, and the sums of x and 2 x by wave_sum2.  No reference counterpart (jnp.sum); tests/test_gpu_rng.py.  ABI 7. */
int d3p_selftest_wave_sums(void* stream, const float* in_dev, uint32_t n_waves, float* out_dev);

/* Synthetic workload of SURVEY 8(d) / examples/logistic_regression.py:88-104, generated on device:
 * X[r][c] and y[r] are pure functions of (seed, global row, column). */
int d3p_synth_logreg(void* stream, uint32_t seed, uint64_t row0, uint64_t n_rows, int32_t d,
                     float* X_dev, float* y_dev);

/* ABI 9 -- per-example, per-SITE guide noise of a multi-site mean-field guide (the logistic-regression example's own guide,
 * examples/logistic_regression.py:67-86: `sample('w', Normal(w_loc, exp(w_std_log)))` then `sample('intercept', ...)`), replacing what
 * jax.vmap + numpyro's `seed` handler draw inside svi.py:262-290: eps_dev[i] = [normal(site_key_0, (size_0,)) | normal(site_key_1, ...) | ...]
 * for the examples at positions pos0 .. pos0 + B_local - 1 of a batch of B_total; example p's key = split(jax_key, B_total)[p], its guide
 * seed = split(.)[1], the handler advances `rng, site_key = split(rng)` per sample statement (UNPINNED like the rest of numpyro's key
 * plumbing; n_sites = 1 is the stream the fused kernels draw on chip).  site_sizes_host: n_sites <= 8 sizes (a scalar site: 1). */
int d3p_px_eps_sites(void* stream, const uint32_t* jax_key_dev, uint32_t B_total, uint32_t pos0, uint32_t B_local,
                     const int32_t* site_sizes_host, int32_t n_sites, float* eps_dev);

/* ABI 9 -- DPSVI.update (d3p/svi.py:395-434) for a parameter dict with SEVERAL leaves around the fused clipped sums, e.g. the example's own
 * guide (examples/logistic_regression.py:67-86: intercept_loc, intercept_std_log, w_loc, w_std_log), in nine launches:
 *   d3p_dpvi_leaves_begin -> d3p_px_eps_sites -> d3p_dpvi_logreg_local_sums(eps) -> d3p_dpvi_leaves_finalize.
 * The state's parameters / Adam moments are flat in TREE order (the leaves of the dict sorted by name, as jax.tree_util flattens it);
 * col_of_dev[j] = the fused kernels' column ([loc (D) | unconstrained scale (D)], latent elements in site order) of tree element j.
 * begin: (next_key, gradient key, perturbation key) = split(state key, 3) (svi.py:413-415); jax_key = the gradient key's two threefry words
 * (random/__init__.py:149-155); leaf_keys[k] = split(perturbation key, n_leaves)[k] (svi.py:491); params_kernel[col_of[j]] = params_tree[j]. */
int d3p_dpvi_leaves_begin(void* stream, const uint32_t* state_key_dev, int32_t n_leaves, const float* params_tree_dev,
                          const int32_t* col_of_dev, uint32_t P, uint32_t* next_key_dev, uint32_t* jax_key_dev, uint32_t* leaf_keys_dev,
                          float* params_kernel_dev);
/* finalize: sums_dev = [P clipped sums in kernel column order | loss sum | number of valid examples] as d3p_dpvi_logreg_local_sums (or an
 * all-reduce of it) leaves them.  Per tree element j of leaf k: g = (sums[col_of[j]] / B + normal(leaf_keys[k])[e] * dp_scale * clip / n)
 * * obs_scale * (B / n) (svi.py:343-346, :365-375, :487-488), then numpyro's Adam (svi.py:379-393) from (params, m, v, step)_in into the
 * _out arrays (the update is functional, the inputs are not written); loss = (loss sum / B) * obs_scale * B / n (svi.py:342, :306; n = 0: 0, or
 * NaN when a parameter is not finite, as the masked sum gives there).  leaf_sizes_host: n_leaves <= 16 sizes in tree order; grad_out_dev
 * (nullable): the perturbed gradient in tree order. */
int d3p_dpvi_leaves_finalize(void* stream, const d3p_dpsvi_hyper* hyper, const float* sums_dev, const int32_t* col_of_dev,
                             const uint32_t* leaf_keys_dev, const int32_t* leaf_sizes_host, int32_t n_leaves, uint32_t B, float obs_scale,
                             const float* params_in_dev, const float* m_in_dev, const float* v_in_dev, const int32_t* step_in_dev,
                             float* params_out_dev, float* m_out_dev, float* v_out_dev, int32_t* step_out_dev, float* loss_dev,
                             float* grad_out_dev);

/* ABI 9 -- d3p_logreg_evaluate for a guide with several sample sites (the example's own guide: 'w' (d) then 'intercept' (1)): the one
 * guide draw of SVI.evaluate takes every site's eps from its own key, `rng, site_key = split(rng)` per sample statement; params_dev in the
 * kernels' order [loc (D) | unconstrained scale (D)], latent elements in site order.  site_sizes_host = NULL, n_sites = 1: d3p_logreg_evaluate. */
int d3p_logreg_evaluate_sites(void* stream, const d3p_logreg_model* model, const float* params_dev, const float* X_dev,
                              const float* y_dev, uint32_t B, const uint32_t* jax_key_dev, const int32_t* site_sizes_host,
                              int32_t n_sites, float* loss_dev, void* workspace_dev, size_t workspace_bytes);

/* ABI 9 -- measurement aid, not part of the reference's path: a streaming device-to-device copy of `bytes` (multiple of 16, both
 * buffers 16-byte aligned) with `bytes_per_lane` = 16 (float4: the figure quoted as the achievable HBM rate, SURVEY 8(d) "also measure
 * a device-to-device copy peak on the box"), 8 or 4 (to calibrate the FETCH_SIZE / WRITE_SIZE counters per access width). */
int d3p_hbm_copy(void* stream, void* dst_dev, const void* src_dev, uint64_t bytes, int32_t bytes_per_lane);

#ifdef __cplusplus
}
#endif
#endif /* D3P_HIP_H */
