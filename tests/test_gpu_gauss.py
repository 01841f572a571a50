"""Second model family on the same kernels: Gaussian observations with a latent mean and the hand-written
exp-transformed guide of examples/simple_gaussian_posterior.py (BASELINE configs[0]).  HIP path vs the CPU
oracle, and convergence to the conjugate posterior (the reference example's `analytical_solution` check).
Tolerances as in test_gpu_dpsvi.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PX_RTOL, PX_ATOL = 2e-5, 2e-6


def np_(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def rng(gpu):
    import d3p_amd.random as r
    return r


def make_svi(d, N, auto_guide=False, C=1.0, sigma=1.0, prior=1.0, lik_sigma=0.1, lr=1e-3, unscale=True):
    from d3p_amd.models import Adam, AutoDiagonalNormal, DiagonalNormalGuide, GaussianMean, Trace_ELBO
    from d3p_amd.svi import DPSVI
    model = GaussianMean(d, prior_scale=prior, obs_scale=lik_sigma)
    guide = AutoDiagonalNormal(model) if auto_guide else DiagonalNormalGuide(model)
    return DPSVI(model, guide, Adam(lr), Trace_ELBO(), C, sigma, clip_unscaled_observations=unscale,
                 d=d, num_obs_total=N)


def state_with(svi, key, loc, unc, N):
    from d3p_amd.svi import DPSVIState
    params = torch.tensor(np.concatenate([loc, unc]), device="cuda")
    return DPSVIState(svi.optim.init(params), key, float(N) if svi._clip_unscaled_observations else 1.0)


def problem(B, d, seed):
    r = np.random.default_rng(seed)
    X = (1.0 + 0.5 * r.normal(size=(B, d))).astype(np.float32)
    loc = (0.8 + r.normal(size=d) * 0.3).astype(np.float32)
    unc = (r.normal(size=d) * 0.3 - 1.0).astype(np.float32)
    return X, loc, unc


@pytest.mark.parametrize("B,d", [(9, 4), (16, 100), (40, 512), (5, 1000), (6, 1500), (5, 2600)])   # (the last two: column-chunked kernel)
@pytest.mark.parametrize("auto_guide", [False, True])
@pytest.mark.parametrize("onchip", [False, True])
def test_px_grads_vs_oracle(rng, O, B, d, auto_guide, onchip):
    N = 1000
    X, loc, unc = problem(B, d, 100 * B + d)
    mask = np.random.default_rng(5).random(B) < 0.8
    svi = make_svi(d, N, auto_guide, prior=1.5, lik_sigma=0.7)
    key = rng.PRNGKey(B + d)
    st = state_with(svi, key, loc, unc, N)
    jax_key = O.convert_to_jax_rng_key(O.PRNGKey(B + d))
    eps = O.px_eps(jax_key, B, d) if onchip else np.random.default_rng(1).normal(size=(B, d)).astype(np.float32)
    kw = {} if onchip else {"_eps": torch.tensor(eps).cuda()}
    _, px_loss, px_grads, n, f = svi._compute_per_example_gradients(st, key, torch.tensor(X).cuda(),
                                                                    mask=torch.tensor(mask).cuda(), **kw)
    names = ("auto_loc", "auto_scale") if auto_guide else ("mu_loc", "mu_std_log")
    assert tuple(sorted(px_grads)) == names
    spec = O.gauss_mean_spec(d, prior=1.5, lik_sigma=0.7, lik_scale=N, obs_scale=N, guide_exp=not auto_guide)
    eL, eG, en, ef = O.logreg_px_grads(spec, loc, unc, X, None, eps, mask.astype(np.float32))
    assert float(n) == en and abs(float(f) - ef) < 1e-6
    G = np.concatenate([np_(px_grads[names[0]]), np_(px_grads[names[1]])], axis=1)
    np.testing.assert_allclose(G, eG, rtol=PX_RTOL, atol=PX_ATOL * np.abs(eG).max())
    np.testing.assert_allclose(np_(px_loss), eL, rtol=PX_RTOL, atol=PX_ATOL * np.abs(eL).max())
    assert np.all(G[~mask] == 0) and not np.allclose(G[mask], 0)


@pytest.mark.parametrize("B,d,masked", [(16, 4, False), (50, 96, True), (64, 512, False), (12, 1500, True), (9, 2600, False)])
@pytest.mark.parametrize("onchip", [False, True])
def test_fused_update_vs_oracle(rng, O, B, d, masked, onchip):
    N = 5000
    X, loc, unc = problem(B, d, 7 * B + d)
    mask = (np.random.default_rng(6).random(B) < 0.7) if masked else None
    svi = make_svi(d, N, C=0.7, sigma=1.3, prior=1.5, lik_sigma=0.6, lr=1e-2)
    st = state_with(svi, rng.PRNGKey(4242), loc, unc, N)
    eps = None if onchip else np.random.default_rng(2).normal(size=(B, d)).astype(np.float32)
    gout = torch.empty(2 * d, device="cuda")
    new_st, loss = svi._update_fused(st, torch.tensor(X).cuda(), mask=torch.tensor(mask).cuda() if masked else True,
                                     _eps=None if onchip else torch.tensor(eps).cuda(), _grad_out=gout)
    spec = O.gauss_mean_spec(d, prior=1.5, lik_sigma=0.6, lik_scale=N, obs_scale=N)
    hy = O.Hyper(0.7, 1.3, 1e-2, 0.9, 0.999, 1e-8)
    ost = O.LogregState(O.PRNGKey(4242), d, loc, unc)
    eloss, egrad = O.logreg_update(spec, hy, ost, X, None, None if not masked else mask.astype(np.float32), eps)
    np.testing.assert_allclose(np_(gout), egrad, rtol=1e-4, atol=1e-6 * np.abs(egrad).max())
    assert abs(float(loss) - eloss) <= 2e-5 * abs(eloss) + 1e-6
    assert np.array_equal(np_(new_st.rng_key).ravel(), ost.key)
    np.testing.assert_allclose(np_(new_st.optim_state[1]), ost.params, rtol=1e-5, atol=1e-6)


def test_run_steps_and_update_walk_the_oracle_trajectory(rng, O):
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, B, steps = 1000, 4, 10, 8           # the example's defaults: q = 0.01
    r = np.random.default_rng(0)
    X = (1.0 + 0.1 * r.normal(size=(N, d))).astype(np.float32)
    Xt = torch.tensor(X).cuda()
    svi = make_svi(d, N, C=2.0, sigma=0.5, lr=1e-2)
    st = svi.init(rng.PRNGKey(100), Xt[:B])
    assert float(st.observation_scale) == N
    p0 = svi.get_params(st)
    assert torch.all(p0["mu_loc"] == 0) and torch.all(p0["mu_std_log"] == 0)   # guide starts from the prior
    init, get_batch = subsample_batchify_data((Xt,), B)
    nb, bstate = init(rng.PRNGKey(200))
    new_st, losses = svi.run_steps(st, get_batch, bstate, 0, steps)

    spec = O.gauss_mean_spec(d, lik_scale=N, obs_scale=N)
    hy = O.Hyper(2.0, 0.5, 1e-2, 0.9, 0.999, 1e-8)
    ost = O.LogregState(O.PRNGKey(100), d)
    el = [O.logreg_update(spec, hy, ost, X[O.feistel_sample(O.fold_in(O.PRNGKey(200), t), N, B)], None)[0]
          for t in range(steps)]
    np.testing.assert_allclose(np_(losses), el, rtol=5e-5)
    assert np.array_equal(np_(new_st.rng_key).ravel(), ost.key)
    np.testing.assert_allclose(np_(new_st.optim_state[1]), ost.params, rtol=2e-4, atol=2e-5)
    st2 = st
    for t in range(steps):
        (bx,) = get_batch(t, bstate)
        st2, l2 = svi.update(st2, bx)
        assert abs(float(l2) - el[t]) <= 5e-5 * abs(el[t])
    assert torch.equal(st2.rng_key, new_st.rng_key)


def test_evaluate_vs_oracle(rng, O):
    N, B, d = 10**4, 60, 33
    X, loc, unc = problem(B, d, 77)
    svi = make_svi(d, N, prior=1.5, lik_sigma=0.7)
    st = state_with(svi, rng.PRNGKey(99), loc, unc, N)
    got = float(svi.evaluate(st, torch.tensor(X).cuda()))
    spec = O.gauss_mean_spec(d, prior=1.5, lik_sigma=0.7, lik_scale=N, obs_scale=1.0)
    jax_key = O.convert_to_jax_rng_key(O.split(O.PRNGKey(99), 1)[0])
    exp = O.logreg_evaluate(spec, loc, unc, X, None, jax_key)
    assert abs(got - exp) <= 2e-5 * abs(exp)


def test_config1_converges_to_the_analytical_posterior(rng):
    """examples/simple_gaussian_posterior.py end to end on the device: N = 1000, d = 4, mu_true = 1; compares
    with the conjugate posterior like the example's final printout (its `analytical_solution`)."""
    from d3p_amd.minibatch import subsample_batchify_data
    from d3p_amd.models import GaussianMean
    N, d, B = 1000, 4, 100
    X = (1.0 + 0.1 * torch.randn(N, d, generator=torch.Generator().manual_seed(1234))).cuda()
    svi = make_svi(d, N, C=20.0, sigma=0.01, lr=2e-2)
    key, k_init, k_batch = rng.split(rng.PRNGKey(0), 3)
    init, get_batch = subsample_batchify_data((X,), B)
    nb, bstate = init(k_batch)
    st = svi.init(k_init, *get_batch(0, bstate))
    st, losses = svi.run_steps(st, get_batch, bstate, 0, 1500)
    p = svi.get_params(st)
    loc_exact, std_exact = GaussianMean.analytical_solution(X, 1.0, 0.1)
    assert float((p["mu_loc"] - loc_exact).abs().max()) < 0.02
    assert float(torch.exp(p["mu_std_log"]).max()) < 0.05
    assert float(losses[-50:].mean()) < float(losses[:50].mean())


def test_family_validation(gpu):
    import ctypes as C
    import d3p_amd._lib as L
    lib = L.load()
    x = torch.zeros(8, device="cuda")
    for model, msg in ((L.LogregModel(4, 1, 1.0, 1.0, 1.0, 1.0, 1, 0, 0.1), b"no intercept"),
                       (L.LogregModel(4, 0, 1.0, 1.0, 1.0, 1.0, 1, 0, 0.0), b"lik_sigma"),
                       (L.LogregModel(4, 0, 1.0, 1.0, 1.0, 1.0, 7, 0, 0.1), b"unknown likelihood family"),
                       (L.LogregModel(4, 0, 1.0, 1.0, 1.0, 1.0, 0, 5, 0.0), b"unknown guide transform"),
                       (L.LogregModel(4, 0, 1.0, 1.0, 1.0, 1.0, 0, 0, 0.0), b"null label pointer")):
        rc = lib.d3p_logreg_evaluate(None, C.byref(model), L.ptr(x), L.ptr(x), None, 2, L.ptr(x), L.ptr(x), L.ptr(x), 1 << 20)
        assert rc == -1 and msg in lib.d3p_last_error()
