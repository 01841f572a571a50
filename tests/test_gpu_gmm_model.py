"""DP-VI step for the Gaussian-mixture MODEL of BASELINE config 3 (examples/gaussian_mixture_model.py:51-85):
HIP per-example gradients vs the CPU oracle (latent draws, losses, gradients), the stage-wise DPSVI.update vs the
oracle's stage composition, and learning on the example's three-cluster toy data.

Tolerances: latent draws -- normals exact to 1e-6 abs (same words, same transform), sigs / gamma draws 2e-6 relative
(float log implementations differ in the last ulp); per-example gradients rtol 1e-4 + 1e-5 * max|g| (float32 sums over
K d terms with magnitudes up to N / sig^2 on the device, float64 in the oracle)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def np_(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def rng(gpu):
    import d3p_amd.random as r
    return r


def make_svi(K, d, N, C=1.0, sigma=1.0, lr=1e-2, optim=None):
    from d3p_amd.models import Adam, GaussianMixtureGuide, GaussianMixtureModel, Trace_ELBO
    from d3p_amd.svi import DPSVI
    model = GaussianMixtureModel()
    return DPSVI(model, GaussianMixtureGuide(model), optim or Adam(lr), Trace_ELBO(), C, sigma, k=K, d=d, num_obs_total=N)


def state_with(svi, key, params, N):
    from d3p_amd.svi import DPSVIState
    return DPSVIState(svi.optim.init(torch.tensor(params).cuda()), key, float(N))


def problem(B, K, d, seed):
    r = np.random.default_rng(seed)
    X = (r.normal(size=(B, d)) * 3).astype(np.float32)
    params = np.concatenate([r.normal(size=K) * 0.4, r.normal(size=K * d) * 2]).astype(np.float32)
    return X, params


@pytest.mark.parametrize("B,K,d", [(9, 4, 3), (33, 16, 64), (7, 3, 2), (12, 6, 100), (5, 32, 128), (6, 16, 256), (4, 5, 70)])
def test_px_grads_and_latents_vs_oracle(rng, O, B, K, d):
    N = 2000
    X, params = problem(B, K, d, 10 * B + K)
    mask = np.random.default_rng(1).random(B) < 0.8
    svi = make_svi(K, d, N)
    key = rng.PRNGKey(K * d)
    st = state_with(svi, key, params, N)
    lat = torch.empty((B, K + 2 * K * d), device="cuda")
    _, px_loss, grads, n, f = svi._compute_per_example_gradients(st, key, torch.tensor(X).cuda(),
                                                                 mask=torch.tensor(mask).cuda(), _latents_out=lat)
    assert tuple(grads["alpha_log"].shape) == (B, K) and tuple(grads["mus_loc"].shape) == (B, K, d)
    spec = O.gmm_spec(K, d, 10.0, lik_scale=N, obs_scale=N)
    jax_key = O.convert_to_jax_rng_key(O.PRNGKey(K * d))
    lat = np_(lat)
    for p in range(B):
        g, eps, sigs = O.gmm_px_latents(spec, params[:K], jax_key, B, p)
        np.testing.assert_allclose(lat[p, :K], g, rtol=2e-6)
        np.testing.assert_allclose(lat[p, K:K + K * d], eps.ravel(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(lat[p, K + K * d:], sigs.ravel(), rtol=2e-6)
    eL, eG, en, ef = O.gmm_px_grads(spec, params, X, jax_key, mask.astype(np.float32))
    assert float(n) == en and abs(float(f) - ef) < 1e-6
    G = np.concatenate([np_(grads["alpha_log"]), np_(grads["mus_loc"]).reshape(B, -1)], axis=1)
    for p in range(B):
        np.testing.assert_allclose(G[p], eG[p], rtol=1e-4, atol=1e-5 * np.abs(eG[p]).max())
    np.testing.assert_allclose(np_(px_loss), eL, rtol=2e-5, atol=1e-6 * np.abs(eL).max())
    assert np.all(G[~mask] == 0) and np.all(np_(px_loss)[~mask] == 0)


def test_staged_update_vs_oracle(rng, O):
    """DPSVI.update for the mixture model = the reference's five stages; sites in tree_flatten order
    (alpha_log, mus_loc) get their own perturbation keys (svi.py:491)."""
    B, K, d, N = 40, 16, 64, 5000
    X, params = problem(B, K, d, 3)
    svi = make_svi(K, d, N, C=20.0, sigma=0.7, lr=1e-2)
    st = state_with(svi, rng.PRNGKey(8), params, N)
    new_st, loss = svi.update(st, torch.tensor(X).cuda())
    spec = O.gmm_spec(K, d, 10.0, lik_scale=N, obs_scale=N)
    ks = O.split(O.PRNGKey(8), 3)
    L, G, n, f = O.gmm_px_grads(spec, params, X, O.convert_to_jax_rng_key(ks[1]))
    eloss, avg = O.combine(O.clip_rows(G, 20.0), L)
    g = O.perturb(ks[2], avg, [K, K * d], 0.7, 20.0, n, N, f)
    x, m, v = O.adam(params, np.zeros_like(params), np.zeros_like(params), g, 0, lr=1e-2)
    assert abs(float(loss) - eloss) <= 5e-5 * abs(eloss)
    assert np.array_equal(np_(new_st.rng_key), ks[0])
    np.testing.assert_allclose(np_(new_st.optim_state[1]), x, rtol=1e-4, atol=1e-5)
    p = svi.get_params(new_st)
    assert tuple(p["alpha_log"].shape) == (K,) and tuple(p["mus_loc"].shape) == (K, d)


def test_learns_the_examples_toy_clusters(rng):
    """examples/gaussian_mixture_model.py:87-110: three clusters at -10, 10 and -2 (the last twice as frequent), d = 2,
    k = 3, C = 20; after training, every true centre has a learned component mean nearby."""
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, K, B = 4000, 2, 3, 200
    g = torch.Generator().manual_seed(0)
    comp = torch.multinomial(torch.tensor([0.25, 0.25, 0.5]), N, replacement=True, generator=g)
    centres = torch.tensor([-10.0, 10.0, -2.0])
    scales = torch.tensor([0.1, 1.0, 0.1])
    X = (centres[comp, None] + scales[comp, None] * torch.randn(N, d, generator=g)).cuda()
    svi = make_svi(K, d, N, C=20.0, sigma=0.1, lr=5e-2)
    key, k_init, k_batch = rng.split(rng.PRNGKey(0), 3)
    init, get_batch = subsample_batchify_data((X,), B)
    nb, bstate = init(k_batch)
    st = svi.init(k_init, *get_batch(0, bstate))
    p0 = svi.get_params(st)
    assert torch.all(p0["alpha_log"] == 0) and torch.all(p0["mus_loc"] == 0)
    losses = []
    for i in range(600):
        st, l = svi.update(st, *get_batch(i % nb, bstate))
        losses.append(float(l))
    mus = svi.get_params(st)["mus_loc"].cpu()
    assert np.isfinite(losses).all()
    assert np.mean(losses[-50:]) < np.mean(losses[:50])
    for c in centres.tolist():
        assert float((mus - c).abs().max(dim=1).values.min()) < 1.5, (c, mus)


@pytest.mark.parametrize("B,K,d,masked", [(40, 16, 64, False), (70, 3, 2, True), (3000, 16, 64, True), (25, 5, 70, False)])
def test_fused_update_matches_the_stage_composition(rng, B, K, d, masked):
    """d3p_dpvi_gmm_update (clip + sum inside the gradient kernel, no B x P tensor) against the five stages run one by
    one on the device (which the oracle tests pin); B = 3000 makes several examples share a wavefront."""
    N = 10**5
    X, params = problem(B, K, d, 5)
    mask = torch.tensor(np.random.default_rng(2).random(B) < 0.7).cuda() if masked else True
    svi = make_svi(K, d, N, C=20.0, sigma=0.7, lr=1e-2)
    st = state_with(svi, rng.PRNGKey(8), params, N)
    Xt = torch.tensor(X).cuda()
    gout = torch.empty(K + K * d, device="cuda")
    s_f, l_f = svi._update_gmm_fused(st, Xt, mask=mask, _grad_out=gout)
    s_s, l_s = svi._update_staged(st, Xt, mask=mask)
    assert abs(float(l_f) - float(l_s)) <= 2e-5 * abs(float(l_s))
    assert torch.equal(s_f.rng_key, s_s.rng_key) and int(s_f.optim_state[0]) == 1
    np.testing.assert_allclose(np_(s_f.optim_state[2]), np_(s_s.optim_state[2]), rtol=2e-4, atol=1e-6 * float(s_s.optim_state[2].abs().max()))
    np.testing.assert_allclose(np_(s_f.optim_state[1]), np_(s_s.optim_state[1]), rtol=1e-5, atol=1e-6)
    # bitwise reproducible
    s_f2, l_f2 = svi._update_gmm_fused(st, Xt, mask=mask)
    assert torch.equal(s_f2.optim_state[1], s_f.optim_state[1]) and float(l_f2) == float(l_f)


@pytest.mark.parametrize("seed", [1, 2])
def test_fused_update_random_shapes_match_the_stage_composition(rng, seed):
    """Twenty-five random shapes per seed -- 1 .. 32 components, 1 .. 256 dimensions, batches of 1 .. 1000, masked or not: the
    shape picks the per-example kernel's instantiation (component pairs per wave x 64-dimension slots, paired or single threefry
    calls, the full-tile form) -- fused update against the five stages run one by one (which the oracle tests pin)."""
    rs = np.random.default_rng(200 + seed)
    N = 10**5
    for _ in range(25):
        K = int(rs.choice([1, 2, 3, 5, 8, 16, 17, 32]))
        d = int(rs.choice([1, 2, 10, 63, 64, 65, 128, 200, 256]))
        if K > 16 and d > 128:
            d = 128   # (the fused step's register tiles: K <= 16 with d <= 256, K <= 32 with d <= 128)
        B = int(rs.choice([1, 33, 257, 1000]))
        masked = bool(rs.integers(2))
        X, params = problem(B, K, d, 7 + K + d)
        mask = torch.tensor(np.random.default_rng(3).random(B) < 0.7).cuda() if masked else True
        svi = make_svi(K, d, N, C=20.0, sigma=0.7, lr=1e-2)
        st = state_with(svi, rng.PRNGKey(9), params, N)
        Xt = torch.tensor(X).cuda()
        s_f, l_f = svi._update_gmm_fused(st, Xt, mask=mask)
        s_s, l_s = svi._update_staged(st, Xt, mask=mask)
        what = f"K={K} d={d} B={B} masked={masked}"
        if not np.isfinite(float(l_s)):   # (every row masked: both give NaN, svi.py:305)
            assert not np.isfinite(float(l_f)), what
            continue
        assert abs(float(l_f) - float(l_s)) <= 2e-5 * abs(float(l_s)), what
        assert torch.equal(s_f.rng_key, s_s.rng_key), what
        # (m = 0.1 g: sums of up to 1000 clipped float32 gradients in two different orders -- fixed-point partial sums per wave against
        # the stage-wise float sum: a column with cancellation differs by a few 1e-4 of its value)
        np.testing.assert_allclose(np_(s_f.optim_state[2]), np_(s_s.optim_state[2]), rtol=5e-4, atol=4e-6 * float(s_s.optim_state[2].abs().max()),
                                   err_msg=what)
        np.testing.assert_allclose(np_(s_f.optim_state[1]), np_(s_s.optim_state[1]), rtol=1e-5, atol=2e-6, err_msg=what)


def test_run_steps_walks_the_update_trajectory(rng):
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, K, B, steps = 5000, 64, 16, 128, 5
    X = (torch.randn(N, d, generator=torch.Generator().manual_seed(1)) * 3).cuda()
    svi = make_svi(K, d, N, C=20.0, sigma=0.5, lr=1e-2)
    init, get_batch = subsample_batchify_data((X,), B)
    nb, bstate = init(rng.PRNGKey(21))
    st = svi.init(rng.PRNGKey(20), *get_batch(0, bstate))
    new_st, losses = svi.run_steps(st, get_batch, bstate, 3, steps)
    ref = st
    for t in range(steps):
        ref, l = svi.update(ref, *get_batch(3 + t, bstate))
        assert abs(float(l) - float(losses[t])) <= 1e-6 * abs(float(l))
    assert torch.equal(ref.rng_key, new_st.rng_key) and int(new_st.optim_state[0]) == steps
    assert torch.equal(ref.optim_state[1], new_st.optim_state[1])


def test_run_steps_across_prepared_batches(rng):
    """A run longer than one prepared batch of steps (64: keys, Feistel rows, site keys and noise are prepared per batch and
    the slot / noise arrays alternate between batches) ends where the step-by-step loop ends, bit for bit, also when the run
    starts from an odd step count."""
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, K, B, steps = 3000, 2, 3, 64, 131
    X = (torch.randn(N, d, generator=torch.Generator().manual_seed(3)) * 3).cuda()
    svi = make_svi(K, d, N, C=20.0, sigma=0.5, lr=1e-2)
    init, get_batch = subsample_batchify_data((X,), B)
    nb, bstate = init(rng.PRNGKey(31))
    st = svi.init(rng.PRNGKey(30), *get_batch(0, bstate))
    st, _ = svi.update(st, *get_batch(0, bstate))          # odd starting step
    new_st, losses = svi.run_steps(st, get_batch, bstate, 1, steps)
    ref = st
    for t in range(steps):
        ref, l = svi.update(ref, *get_batch(1 + t, bstate))
        assert float(l) == float(losses[t]), t
    assert torch.equal(ref.rng_key, new_st.rng_key) and int(new_st.optim_state[0]) == steps + 1
    for a, b in zip(ref.optim_state[1:], new_st.optim_state[1:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("steps", [1, 64, 65, 129, 200])
def test_run_steps_key_chain_vs_oracle_split(rng, O, steps):
    """The state key after `steps` steps of the native GMM loop is `steps` successive split(key, 3)[0] (svi.py:208-211), bit for
    bit against the ORACLE's split -- also across the prepared batches of 64 steps, whose serial key chains are made link by link by
    the extra workgroup of the `k_gmm_head` launches of the batch before them (not only against this build's own update())."""
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, K, B = 3000, 2, 3, 64
    X = (torch.randn(N, d, generator=torch.Generator().manual_seed(5)) * 3).cuda()
    svi = make_svi(K, d, N, C=20.0, sigma=0.5, lr=1e-2)
    init, get_batch = subsample_batchify_data((X,), B)
    nb, bstate = init(rng.PRNGKey(41))
    st = svi.init(rng.PRNGKey(40), *get_batch(0, bstate))
    new_st, losses = svi.run_steps(st, get_batch, bstate, 0, steps)
    assert bool(torch.isfinite(losses).all()) and int(new_st.optim_state[0]) == steps
    key = O.PRNGKey(40)
    assert np.array_equal(np_(st.rng_key).ravel(), np.asarray(key).ravel())     # init retains the key (svi.py:213-236)
    for _ in range(steps):
        key = O.split(key, 3)[0]
    assert np.array_equal(np_(new_st.rng_key).ravel(), np.asarray(key).ravel())


@pytest.mark.parametrize("B,K,d", [(50, 16, 64), (17, 3, 2), (300, 5, 70)])
def test_evaluate_vs_oracle(rng, O, B, K, d):
    N = 10**4
    X, params = problem(B, K, d, 9)
    svi = make_svi(K, d, N)
    st = state_with(svi, rng.PRNGKey(99), params, N)
    got = float(svi.evaluate(st, torch.tensor(X).cuda()))
    spec = O.gmm_spec(K, d, 10.0, lik_scale=N, obs_scale=1.0)
    jax_key = O.convert_to_jax_rng_key(O.split(O.PRNGKey(99), 1)[0])
    exp = O.gmm_evaluate(spec, params, X, jax_key)
    assert abs(got - exp) <= 2e-5 * abs(exp)
