"""DP-VI update step: HIP kernels vs the CPU oracle, and the reference's tests/test_dpsvi.py and
tests/test_gradient_manipulators.py re-expressed against the d3p_amd surface.

Floating-point tolerance (stated once): per-example quantities agree with the oracle to
rtol 2e-5 (float32 dot products of length d accumulate in a different order; the oracle
accumulates in float64); batch-level gradients after the sum over B examples to
rtol 1e-4 / atol 1e-6 * max|g|.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PX_RTOL, PX_ATOL = 2e-5, 2e-6
G_RTOL = 1e-4


def np_(t):
    return t.detach().cpu().numpy()


def close(a, b, rtol=G_RTOL, atol_rel=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_rel * max(1e-30, np.abs(b).max()))


@pytest.fixture(scope="module")
def rng(gpu):
    import d3p_amd.random as r
    return r


def make_problem(B, d, icpt, seed, N=1000, mask_frac=None):
    r = np.random.default_rng(seed)
    D = d + int(icpt)
    X = r.normal(size=(B, d)).astype(np.float32)
    y = (r.random(B) < 0.5).astype(np.float32)
    loc = (r.normal(size=D) * 0.3).astype(np.float32)
    unc = (r.normal(size=D) * 0.5 - 1.0).astype(np.float32)
    mask = None
    if mask_frac is not None:
        mask = (r.random(B) < mask_frac)
    return X, y, loc, unc, mask


def make_svi(d, icpt, N, C=1.0, sigma=1.0, prior=1.0, lr=1e-3, rng_suite=None, unscale=True, optim=None):
    from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.svi import DPSVI
    import d3p_amd.random as strong
    model = LogisticRegression(d, prior_scale=prior, intercept=icpt, intercept_prior_scale=2.0 * prior)
    guide = AutoDiagonalNormal(model)
    return DPSVI(model, guide, optim or Adam(lr), Trace_ELBO(), C, sigma, rng_suite=rng_suite or strong,
                 clip_unscaled_observations=unscale, N=N)


def state_with(svi, key, loc, unc):
    from d3p_amd.svi import DPSVIState
    params = torch.tensor(np.concatenate([loc, unc]), device="cuda")
    return DPSVIState(svi.optim.init(params), key, float(svi.static_kwargs["N"]) if svi._clip_unscaled_observations else 1.0)


# --------------------------------------------------------------------------- stage 1 vs oracle
@pytest.mark.parametrize("B,d,icpt", [(7, 8, False), (13, 5, True), (64, 512, False), (33, 520, False),
                                      (9, 130, True), (5, 1024, False), (3, 1, False), (4, 1, True), (6, 2048, False),
                                      # rows whose second half is not 16-byte aligned (odd half / intercept column):
                                      # the scalar-load form with up to 8 column pairs per lane; (20, 512, True) is
                                      # examples/logistic_regression.py's shape
                                      (20, 512, True), (11, 128, True), (10, 300, False), (9, 132, False),
                                      # rows too wide for the register-tiled kernel: the column-chunked kernel's
                                      # materialising form
                                      (6, 1024, True), (5, 3000, True), (4, 4096, False)])
@pytest.mark.parametrize("onchip", [False, True])
def test_px_grads_vs_oracle(rng, O, B, d, icpt, onchip):
    N = 1000
    X, y, loc, unc, mask = make_problem(B, d, icpt, B * 1000 + d, N, mask_frac=0.8)
    D = d + int(icpt)
    svi = make_svi(d, icpt, N, prior=1.5)
    key = rng.PRNGKey(B + d)
    st = state_with(svi, key, loc, unc)
    Xt, yt, mt = torch.tensor(X).cuda(), torch.tensor(y).cuda(), torch.tensor(mask).cuda()
    jax_key = O.convert_to_jax_rng_key(O.PRNGKey(B + d))
    eps = O.px_eps(jax_key, B, D) if onchip else np.random.default_rng(1).normal(size=(B, D)).astype(np.float32)
    kw = {} if onchip else {"_eps": torch.tensor(eps).cuda()}
    _, px_loss, px_grads, n, f = svi._compute_per_example_gradients(st, key, Xt, yt, mask=mt, **kw)
    spec = O.logreg_spec(d, icpt, 1.5, 3.0, lik_scale=N, obs_scale=N)
    eL, eG, en, ef = O.logreg_px_grads(spec, loc, unc, X, y, eps, mask.astype(np.float32))
    assert float(n) == en and abs(float(f) - ef) < 1e-6
    G = np.concatenate([np_(px_grads["auto_loc"]), np_(px_grads["auto_scale"])], axis=1)
    scale = np.abs(eG).max()
    np.testing.assert_allclose(G, eG, rtol=PX_RTOL, atol=PX_ATOL * scale)
    np.testing.assert_allclose(np_(px_loss), eL, rtol=PX_RTOL, atol=PX_ATOL * np.abs(eL).max())
    # masked rows are exactly zero (reference tests/test_dpsvi.py:137-144)
    assert np.all(G[~mask] == 0) and np.all(np_(px_loss)[~mask] == 0)
    assert not np.allclose(G[mask], 0)


# --------------------------------------------------------------------------- reference tests/test_dpsvi.py
class TestReferenceDPSVI:
    batch_size, num_elements, num_obs_total = 10, 8, 100
    dp_scale, clipping_threshold = 1.0, 2.0

    def setup_method(self, _):
        import d3p_amd.random as strong
        from d3p_amd.models import SGD
        from d3p_amd.svi import DPSVI
        self.rng_suite = strong
        self.rng = strong.PRNGKey(9782346)
        self.mask = torch.arange(self.batch_size, device="cuda") < self.num_elements
        self.rescale_factor = self.batch_size / self.num_elements
        self.px_grads = (torch.ones((self.batch_size, 10000), device="cuda"),
                         torch.ones((self.batch_size, 10000), device="cuda"))
        self.masked_px_grads = tuple(g * self.mask.reshape(-1, 1) for g in self.px_grads)
        self.px_loss = torch.arange(self.batch_size, dtype=torch.float32, device="cuda") * self.mask
        self.svi = DPSVI(None, None, SGD(1.0), None, self.clipping_threshold, self.dp_scale,
                         num_obs_total=self.num_obs_total, rng_suite=self.rng_suite)

    def _logreg_svi(self, unscale=True):
        from d3p_amd.models import SGD, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
        from d3p_amd.svi import DPSVI
        model = LogisticRegression()
        return DPSVI(model, AutoDiagonalNormal(model), SGD(1.0), Trace_ELBO(), self.clipping_threshold,
                     self.dp_scale, rng_suite=self.rng_suite, clip_unscaled_observations=unscale,
                     num_obs_total=self.num_obs_total)

    def test_init(self):  # tests/test_dpsvi.py:64-86
        svi = self._logreg_svi()
        batch = (torch.zeros((self.batch_size, 3), device="cuda"), torch.zeros(self.batch_size, device="cuda"))
        st = svi.init(self.rng, *batch)
        assert st.observation_scale == self.num_obs_total
        assert torch.equal(st.rng_key, self.rng)
        p = svi.get_params(st)
        assert tuple(p["auto_loc"].shape) == (3,) and torch.allclose(p["auto_scale"], torch.full((3,), 0.1, device="cuda"))
        assert float(p["auto_loc"].abs().max()) <= 2.0

    def test_init_no_unscaling(self):  # tests/test_dpsvi.py:88-110
        svi = self._logreg_svi(unscale=False)
        batch = (torch.zeros((self.batch_size, 3), device="cuda"), torch.zeros(self.batch_size, device="cuda"))
        st = svi.init(self.rng, *batch)
        assert st.observation_scale == 1.0 and torch.equal(st.rng_key, self.rng)

    def test_compute_px_gradients_masking(self):  # tests/test_dpsvi.py:112-144
        svi = self._logreg_svi()
        batch = (torch.ones((self.batch_size, 3), device="cuda"), torch.ones(self.batch_size, device="cuda"))
        st = svi.init(self.rng, *batch)
        new_st, px_losses, px_grads, num_elements, factor = svi._compute_per_example_gradients(
            st, st.rng_key, *batch, mask=self.mask)
        assert new_st.observation_scale == st.observation_scale
        assert float(num_elements) == self.num_elements
        assert abs(float(factor) - self.batch_size / self.num_elements) < 1e-6
        ne = self.num_elements
        assert not np.allclose(np_(px_losses)[:ne], 0) and np.allclose(np_(px_losses)[ne:], 0)
        for site in ("auto_loc", "auto_scale"):
            g = np_(px_grads[site])
            assert not np.allclose(g[:ne], 0) and np.allclose(g[ne:], 0)

    def test_px_gradient_clipping(self):  # tests/test_dpsvi.py:146-173
        from d3p_amd.svi import DPSVIState, full_norm
        st = DPSVIState(None, self.rng, 0.8)
        px_grads = (torch.tensor([1.0, 0.0], device="cuda").repeat_interleave(10).reshape(2, 10),
                    torch.tensor([0.0, 1.0], device="cuda").repeat_interleave(2).reshape(2, 2))
        norms = [float(full_norm(tuple(g[i] for g in px_grads))) for i in range(2)]
        assert np.allclose(norms, (np.sqrt(10), np.sqrt(2)))
        new_st, clipped = self.svi._clip_gradients(st, px_grads)
        assert new_st == st and isinstance(clipped, tuple) and len(clipped) == 2
        assert [tuple(c.shape) for c in clipped] == [(2, 10), (2, 2)]
        cn = [float(full_norm(tuple(g[i] for g in clipped))) for i in range(2)]
        assert np.allclose(cn, [2.0, np.sqrt(2)])
        _, cg = self.svi._combine_gradients(clipped, torch.ones(2, device="cuda"))
        assert float(full_norm(cg)) < self.clipping_threshold

    def test_px_gradient_aggregation(self):  # tests/test_dpsvi.py:175-191
        np.random.seed(0)
        px = [np.random.normal(1, 1, size=(self.batch_size, 10000)).astype(np.float32) for _ in range(2)]
        loss, grads = self.svi._combine_gradients([torch.tensor(p).cuda() for p in px], self.px_loss)
        assert np.allclose(float(loss), np_(self.px_loss).mean())
        for g, p in zip(grads, px):
            assert np.allclose(np_(g), p.mean(axis=0), atol=1e-6)

    def test_dp_noise_perturbation(self):  # tests/test_dpsvi.py:193-216
        from d3p_amd.svi import DPSVIState
        st = DPSVIState(None, self.rng, 0.3)
        grads = tuple(g.mean(dim=0) for g in self.px_grads)
        masked = tuple(g.mean(dim=0) for g in self.masked_px_grads)
        new_st, pert = self.svi._perturb_and_reassemble_gradients(st, self.rng, masked, self.num_elements,
                                                                 self.rescale_factor)
        assert new_st.optim_state is st.optim_state and isinstance(pert, tuple)
        expected_std = self.dp_scale * (self.clipping_threshold / self.num_elements) * 0.3 * self.rescale_factor
        assert abs(expected_std - 0.09375) < 1e-9
        for p, site in zip(pert, grads):
            assert p.shape == site.shape
            assert abs(float(p.std()) - expected_std) < 1e-2
            assert abs(float(site.mean()) * 0.3 - float(p.mean())) < 5e-3

    def test_dp_noise_not_deterministic_over_rngs_and_sites(self):  # tests/test_dpsvi.py:218-258
        from d3p_amd.svi import DPSVIState
        st = DPSVIState(None, self.rng, 0.3)
        k1, k2 = self.rng_suite.split(self.rng)
        grads = tuple(g.mean(dim=0) for g in self.px_grads)
        _, a = self.svi._perturb_and_reassemble_gradients(st, k1, grads, self.num_elements, self.rescale_factor)
        _, b = self.svi._perturb_and_reassemble_gradients(st, k2, grads, self.num_elements, self.rescale_factor)
        assert not any(torch.allclose(x, y) for x, y in zip(a, b))
        assert not torch.allclose(a[0], a[1])

    def test_perturbation_matches_oracle(self, O):
        from d3p_amd.svi import DPSVIState
        st = DPSVIState(None, self.rng, 0.3)
        avg = [np.linspace(-1, 1, 1000).astype(np.float32), np.linspace(2, 3, 37).astype(np.float32)]
        _, pert = self.svi._perturb_and_reassemble_gradients(st, self.rng, [torch.tensor(a).cuda() for a in avg],
                                                             self.num_elements, self.rescale_factor)
        exp = O.perturb(O.PRNGKey(9782346), np.concatenate(avg), [1000, 37], self.dp_scale, self.clipping_threshold,
                        self.num_elements, 0.3, self.rescale_factor)
        np.testing.assert_allclose(np.concatenate([np_(p) for p in pert]), exp, rtol=1e-5, atol=1e-7)


# --------------------------------------------------------------------------- tests/test_gradient_manipulators.py
def test_gradient_manipulators(gpu):
    from d3p_amd.svi import clip_gradient, full_norm, normalize_gradient
    t = lambda *a: torch.tensor(a, dtype=torch.float32, device="cuda")
    ones = lambda *shape: torch.ones(shape, device="cuda")
    ref_tree = (ones(17, 2, 3), ones(2, 54), (ones(2, 3), ones(3, 4, 5)), ())
    assert abs(float(full_norm(ref_tree)) - 16.613247) < 1e-5   # sqrt(276), tests/test_gradient_manipulators.py:70-79
    tree = ([t(1, 2), torch.zeros((0,), device="cuda")], {"a": t(3, 4, 5).reshape(3, 1), "b": (t(6), t(7, 8, 9, 10))})
    assert abs(float(full_norm(tree)) - np.sqrt(385.0)) < 1e-5
    for empty in (None, [], ()):
        assert full_norm(empty) == 0.0                       # :62-68
    with pytest.raises(ValueError):
        clip_gradient(tree, 0.0)                             # :101-103
    for c in (100.0, float("inf")):                          # identity when C >= norm (:81-99)
        out = clip_gradient(tree, c)
        assert abs(float(full_norm(out)) - np.sqrt(385.0)) < 1e-5
    out = clip_gradient(tree, 3.0)
    assert float(full_norm(out)) <= 3.0 + 1e-6
    assert torch.allclose(out[1]["b"][1] / out[1]["b"][1][0], t(7, 8, 9, 10) / 7)   # direction preserved
    assert abs(float(full_norm(normalize_gradient(tree))) - 1.0) < 1e-6              # :105-109


@pytest.mark.parametrize("order", [0, 1, 2, 3, 0.5, -1.5, float("inf"), float("-inf"), None])
def test_full_norm_takes_any_order_of_numpy_linalg_norm(gpu, order):
    """`ord` of full_norm / normalize_gradient (d3p/svi.py:68-103: "any value possible for numpy.linalg.norm") on the
    concatenation of a ragged tree; the float64 numpy norm of the same float32 values is the reference, rtol 1e-5 (fp32 sums of 176 powers)."""
    from d3p_amd.svi import full_norm, normalize_gradient
    g = np.random.default_rng(11)
    parts = [g.standard_normal(s).astype(np.float32) for s in ((7, 3), (1,), (130,), (2, 2, 5))]
    parts[2][::9] = 0.0                                        # (order 0 counts the non-zero entries)
    tree = (torch.tensor(parts[0]).cuda(), {"b": torch.tensor(parts[1]).cuda(), "c": (torch.tensor(parts[2]).cuda(),)},
            [torch.tensor(parts[3]).cuda()])
    flat = np.concatenate([p.reshape(-1) for p in parts]).astype(np.float64)
    want = np.linalg.norm(flat, ord=order)
    got = float(full_norm(tree, ord=order))
    assert abs(got - want) <= 1e-5 * abs(want), (order, got, want)
    if order != 0 and want > 0:
        assert abs(float(full_norm(normalize_gradient(tree, ord=order), ord=order)) - 1.0) < 1e-5
    with pytest.raises(ValueError):
        full_norm(tree, ord="fro")


def test_constructor_validation(gpu):
    from d3p_amd.models import SGD
    from d3p_amd.svi import DPSVI
    with pytest.raises(ValueError):
        DPSVI(None, None, SGD(1.0), None, float("inf"), 1.0)   # svi.py:187-188
    svi = DPSVI(None, None, SGD(1.0), None, 1.0, 1.0)
    with pytest.raises(ValueError):
        svi.get_epsilon(1e-5, 0.01)                            # svi.py:454-455


# --------------------------------------------------------------------------- fused update vs oracle
@pytest.mark.parametrize("B,d,icpt,masked", [(16, 8, False, False), (16, 8, True, True), (50, 512, False, True),
                                             (256, 512, False, False), (37, 100, True, True), (8, 1024, False, False),
                                             (70, 512, True, True), (40, 300, False, False), (33, 128, True, False),
                                             # 16-byte-aligned rows that are NOT full tiles take the scalar-load form since round 4
                                             # (two / four columns per lane and half: 16- / 8-wave workgroups)
                                             (300, 256, False, True), (130, 192, False, False), (200, 384, False, True),
                                             (90, 64, False, False), (150, 640, False, False),
                                             # wide rows: the column-chunked kernel of d3p_logreg_wide.h (2048 < d <= 4096)
                                             (21, 3000, True, True), (12, 4096, False, False)])
@pytest.mark.parametrize("onchip", [False, True])
def test_fused_update_vs_oracle(rng, O, B, d, icpt, masked, onchip):
    N = 5000
    X, y, loc, unc, mask = make_problem(B, d, icpt, 17 * B + d, N, mask_frac=0.7 if masked else None)
    D = d + int(icpt)
    svi = make_svi(d, icpt, N, C=0.7, sigma=1.3, prior=1.5, lr=1e-2)
    key = rng.PRNGKey(4242)
    st = state_with(svi, key, loc, unc)
    Xt, yt = torch.tensor(X).cuda(), torch.tensor(y).cuda()
    mt = torch.tensor(mask).cuda() if masked else True
    eps = None if onchip else np.random.default_rng(2).normal(size=(B, D)).astype(np.float32)
    gout = torch.empty(2 * D, device="cuda")
    new_st, loss = svi._update_fused(st, Xt, yt, mask=mt, _eps=None if onchip else torch.tensor(eps).cuda(), _grad_out=gout)

    spec = O.logreg_spec(d, icpt, 1.5, 3.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(0.7, 1.3, 1e-2, 0.9, 0.999, 1e-8)
    ost = O.LogregState(O.PRNGKey(4242), D, loc, unc)
    eloss, egrad = O.logreg_update(spec, hy, ost, X, y, None if not masked else mask.astype(np.float32), eps)
    close(np_(gout), egrad)
    assert abs(float(loss) - eloss) <= 2e-5 * abs(eloss) + 1e-6
    assert np.array_equal(np_(new_st.rng_key).ravel(), ost.key)          # next state key: bit-exact
    step, params, m, v = new_st.optim_state
    assert int(step) == 1
    close(np_(m), ost.m)
    close(np_(v), ost.v, rtol=2e-4)
    np.testing.assert_allclose(np_(params), ost.params, rtol=1e-5, atol=1e-6)
    # the input state is untouched (functional update like the reference)
    assert int(st.optim_state[0]) == 0 and np.array_equal(np_(st.optim_state[1]), np.concatenate([loc, unc]))


@pytest.mark.parametrize("seed", [1, 2])
def test_fused_update_random_shapes_vs_oracle(rng, O, seed):
    """Thirty random shapes per seed -- 1 .. 1100 features with and without intercept, batches of 1 .. 3000, masked or not -- through
    the one-step resident run against the oracle: the width decides the kernel (lean d = 512 forms, the generic kernel's scalar-load
    tiles of 1 / 2 / 4 / 8 columns per lane and half, its full 16-byte-load tiles), the batch the workgroups and waves that get items."""
    rs = np.random.default_rng(100 + seed)
    N = 5000
    for _ in range(30):
        d = int(rs.choice([1, 3, 4, 8, 31, 64, 65, 128, 129, 200, 256, 257, 384, 500, 512, 513, 700, 1024, 1100]))
        icpt = bool(rs.integers(2))
        B = int(rs.choice([1, 2, 31, 64, 65, 200, 700, 1500, 3000]))
        if d > 600 and B > 700:
            B = 300   # (the oracle materialises B x P gradients)
        masked = bool(rs.integers(2))
        X, y, loc, unc, mask = make_problem(B, d, icpt, 19 * B + d, N, mask_frac=0.7 if masked else None)
        D = d + int(icpt)
        svi = make_svi(d, icpt, N, C=0.7, sigma=1.3, prior=1.5, lr=1e-2)
        st = state_with(svi, rng.PRNGKey(4243), loc, unc)
        gout = torch.empty(2 * D, device="cuda")
        new_st, loss = svi._update_fused(st, torch.tensor(X).cuda(), torch.tensor(y).cuda(), mask=torch.tensor(mask).cuda() if masked else True,
                                         _grad_out=gout)
        spec = O.logreg_spec(d, icpt, 1.5, 3.0, lik_scale=N, obs_scale=N)
        hy = O.Hyper(0.7, 1.3, 1e-2, 0.9, 0.999, 1e-8)
        ost = O.LogregState(O.PRNGKey(4243), D, loc, unc)
        eloss, egrad = O.logreg_update(spec, hy, ost, X, y, None if not masked else mask.astype(np.float32), None)
        what = f"d={d} icpt={icpt} B={B} masked={masked}"
        np.testing.assert_allclose(np_(gout), egrad, rtol=G_RTOL, atol=1e-6 * max(1e-30, float(np.abs(egrad).max())), err_msg=what)
        assert abs(float(loss) - eloss) <= 2e-5 * abs(eloss) + 1e-6, what
        assert np.array_equal(np_(new_st.rng_key).ravel(), ost.key), what
        np.testing.assert_allclose(np_(new_st.optim_state[1]), ost.params, rtol=1e-5, atol=1e-6, err_msg=what)


def test_fused_equals_staged_pipeline(rng):
    """update() (fused kernels) and the reference's literal five-stage composition agree."""
    B, d, N = 128, 512, 10**5
    X, y, loc, unc, mask = make_problem(B, d, False, 3, N, mask_frac=0.9)
    svi = make_svi(d, False, N, C=1.0, sigma=1.0)
    key = rng.PRNGKey(77)
    st = state_with(svi, key, loc, unc)
    Xt, yt, mt = torch.tensor(X).cuda(), torch.tensor(y).cuda(), torch.tensor(mask).cuda()
    s1, l1 = svi._update_fused(st, Xt, yt, mask=mt)
    s2, l2 = svi._update_staged(st, Xt, yt, mask=mt)
    assert torch.equal(s1.rng_key, s2.rng_key)
    assert abs(float(l1) - float(l2)) <= 2e-5 * abs(float(l2))
    np.testing.assert_allclose(np_(s1.optim_state[1]), np_(s2.optim_state[1]), rtol=1e-5, atol=1e-6)
    close(np_(s1.optim_state[2]), np_(s2.optim_state[2]))


def test_empty_batch_gives_nan_like_reference(rng):
    """n == 0 -> sensitivity inf, inf * 0 = NaN gradients (svi.py:305, :365, :375)."""
    B, d = 8, 16
    X, y, loc, unc, _ = make_problem(B, d, False, 5)
    svi = make_svi(d, False, 100)
    st = state_with(svi, rng.PRNGKey(1), loc, unc)
    gout = torch.empty(2 * d, device="cuda")
    _, loss = svi._update_fused(st, torch.tensor(X).cuda(), torch.tensor(y).cuda(),
                                mask=torch.zeros(B, dtype=torch.bool, device="cuda"), _grad_out=gout)
    assert float(loss) == 0.0 and bool(torch.isnan(gout).all())


def test_update_is_deterministic_full_size(rng):
    """BASELINE config 2 shape: bitwise-reproducible step (fixed reduction order, no float atomics)."""
    B, d, N = 4096, 512, 10**6
    X, y, loc, unc, _ = make_problem(B, d, False, 9, N)
    svi = make_svi(d, False, N)
    st = state_with(svi, rng.PRNGKey(5), loc, unc)
    Xt, yt = torch.tensor(X).cuda(), torch.tensor(y).cuda()
    outs = []
    for _ in range(3):
        g = torch.empty(2 * d, device="cuda")
        s, l = svi._update_fused(st, Xt, yt, _grad_out=g)
        outs.append((np_(g).copy(), float(l), np_(s.optim_state[1]).copy()))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and o[1] == outs[0][1] and np.array_equal(o[2], outs[0][2])


def test_full_size_update_vs_oracle(rng, O):
    """BASELINE config 2 per-step shape (B=4096, d=512, AutoDiagonalNormal) against the oracle."""
    B, d, N = 4096, 512, 10**6
    X, y, loc, unc, _ = make_problem(B, d, False, 10, N)
    svi = make_svi(d, False, N, C=1.0, sigma=1.0, lr=1e-3)
    st = state_with(svi, rng.PRNGKey(0), loc, unc)
    g = torch.empty(2 * d, device="cuda")
    s, loss = svi._update_fused(st, torch.tensor(X).cuda(), torch.tensor(y).cuda(), _grad_out=g)
    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    ost = O.LogregState(O.PRNGKey(0), d, loc, unc)
    eloss, egrad = O.logreg_update(spec, O.Hyper(1.0, 1.0, 1e-3, 0.9, 0.999, 1e-8), ost, X, y)
    close(np_(g), egrad)
    assert abs(float(loss) - eloss) <= 2e-5 * abs(eloss)
    np.testing.assert_allclose(np_(s.optim_state[1]), ost.params, rtol=1e-5, atol=1e-6)
    # size-independent property: the noise-free part is bounded by the clipping threshold
    svi0 = make_svi(d, False, N, C=1.0, sigma=0.0)
    g0 = torch.empty(2 * d, device="cuda")
    svi0._update_fused(st, torch.tensor(X).cuda(), torch.tensor(y).cuda(), _grad_out=g0)
    assert float(g0.norm()) / N <= 1.0 + 1e-5    # ||mean of clipped|| * obs_scale / N <= C


# --------------------------------------------------------------------------- fused multi-step loop
@pytest.mark.parametrize("sampler", ["feistel", "poisson"])
def test_run_steps_vs_oracle(rng, O, sampler):
    from d3p_amd.minibatch import poisson_batchify_data, subsample_batchify_data
    N, d, B, steps = 3000, 64, 48, 6
    r = np.random.default_rng(0)
    X = r.normal(size=(N, d)).astype(np.float32)
    y = (r.random(N) < 0.5).astype(np.float32)
    loc = np.zeros(d, np.float32)
    unc = np.full(d, -2.0, np.float32)
    Xt, yt = torch.tensor(X).cuda(), torch.tensor(y).cuda()
    svi = make_svi(d, False, N, C=1.0, sigma=0.5, lr=1e-2)
    st = state_with(svi, rng.PRNGKey(100), loc, unc)
    bkey = rng.PRNGKey(200)
    q = B / N
    if sampler == "feistel":
        init, get_batch = subsample_batchify_data((Xt, yt), B)
        maxB = B
    else:
        maxB = 70
        init, get_batch = poisson_batchify_data((Xt, yt), q, maxB)
    nb, bstate = init(bkey)
    first = 2
    new_st, losses = svi.run_steps(st, get_batch, bstate, first, steps)

    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.5, 1e-2, 0.9, 0.999, 1e-8)
    ost = O.LogregState(O.PRNGKey(100), d, loc, unc)
    elosses = []
    for t in range(steps):
        bk = O.fold_in(O.PRNGKey(200), first + t)
        if sampler == "feistel":
            idx = O.feistel_sample(bk, N, B)
            mask = None
        else:
            idx, nsel, nvalid = O.poisson_select(bk, np.float32(q), N, maxB)
            mask = (np.arange(maxB) < nvalid).astype(np.float32)
        el, _ = O.logreg_update(spec, hy, ost, X[idx], y[idx], mask)
        elosses.append(el)
    np.testing.assert_allclose(np_(losses), elosses, rtol=5e-5)
    assert np.array_equal(np_(new_st.rng_key).ravel(), ost.key)
    assert int(new_st.optim_state[0]) == steps
    np.testing.assert_allclose(np_(new_st.optim_state[1]), ost.params, rtol=2e-4, atol=2e-5)
    # the API-parity path (get_batch + update, one step at a time) walks the same trajectory
    st2 = st
    for t in range(steps):
        out = get_batch(first + t, bstate)
        (bx, by), m = (out, True) if sampler == "feistel" else out
        st2, l2 = svi.update(st2, bx, by, mask=m)
        assert abs(float(l2) - elosses[t]) <= 5e-5 * abs(elosses[t])
    assert torch.equal(st2.rng_key, new_st.rng_key)
    np.testing.assert_allclose(np_(st2.optim_state[1]), np_(new_st.optim_state[1]), rtol=1e-5, atol=1e-6)


def test_run_steps_is_a_function_of_its_input_state(rng):
    """DPSVI.update / the epoch body return a NEW state (svi.py:395-434): the state handed to run_steps is bit for bit what it
    was afterwards (the native run copies it inside its first kernel, d3p_dpvi_logreg_run_from), running from it twice gives
    the same result, and zero steps return an equal state."""
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, B = 2000, 64, 32
    r = np.random.default_rng(5)
    Xt = torch.tensor(r.normal(size=(N, d)).astype(np.float32)).cuda()
    yt = torch.tensor((r.random(N) < 0.5).astype(np.float32)).cuda()
    svi = make_svi(d, False, N, C=1.0, sigma=0.5, lr=1e-2)
    st = state_with(svi, rng.PRNGKey(7), np.zeros(d, np.float32), np.full(d, -2.0, np.float32))
    init, get_batch = subsample_batchify_data((Xt, yt), B)
    nb, bstate = init(rng.PRNGKey(8))
    st, _ = svi.run_steps(st, get_batch, bstate, 0, 3)
    before = [t.clone() for t in st.optim_state] + [st.rng_key.clone()]
    a, la = svi.run_steps(st, get_batch, bstate, 3, 9)
    b, lb = svi.run_steps(st, get_batch, bstate, 3, 9)
    for x, y0 in zip(list(st.optim_state) + [st.rng_key], before):
        assert torch.equal(x, y0)
    assert torch.equal(la, lb) and torch.equal(a.rng_key, b.rng_key) and int(a.optim_state[0]) == 12
    for x, y0 in zip(a.optim_state, b.optim_state):
        assert torch.equal(x, y0)
    z, lz = svi.run_steps(st, get_batch, bstate, 3, 0)
    assert lz.numel() == 0 and torch.equal(z.rng_key, st.rng_key)
    for x, y0 in zip(z.optim_state, st.optim_state):
        assert torch.equal(x, y0)


def test_staged_update_with_debug_rng_suite(gpu, O):
    """rng_suite=d3p_amd.random.debug routes DPSVI through the threefry suite (d3p/random/debug.py)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import d3p_amd.random.debug as dbg
    B, d, N = 32, 24, 400
    X, y, loc, unc, _ = make_problem(B, d, False, 21, N)
    svi = make_svi(d, False, N, rng_suite=dbg)
    st = state_with(svi, dbg.PRNGKey(3), loc, unc)
    s1, l1 = svi.update(st, torch.tensor(X).cuda(), torch.tensor(y).cuda())
    keys = O.tf_split([0, 3], 3)
    assert np.array_equal(np_(s1.rng_key), keys[0])
    eps = O.px_eps(keys[1], B, d)
    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    L, G, n, f = O.logreg_px_grads(spec, loc, unc, X, y, eps)
    G = O.clip_rows(G, 1.0)
    eloss, avg = O.combine(G, L)
    assert abs(float(l1) - eloss) <= 2e-5 * abs(eloss)
    sk = O.tf_split(keys[2], 2)
    noise = np.concatenate([O.tf_normal(sk[0], d), O.tf_normal(sk[1], d)])
    g = (avg + noise * (1.0 * 1.0 / B)) * N * 1.0
    x, m, v = O.adam(np.concatenate([loc, unc]), np.zeros(2 * d), np.zeros(2 * d), g, 0)
    np.testing.assert_allclose(np_(s1.optim_state[1]), x, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,d,icpt", [(40, 16, False), (25, 7, True), (300, 512, False)])
def test_evaluate_vs_oracle(rng, O, B, d, icpt):
    """DPSVI.evaluate (svi.py:436-449): -ELBO of the batch with one guide draw, key = split(state.rng_key, 1)[0]."""
    N = 10**4
    X, y, loc, unc, _ = make_problem(B, d, icpt, 31 * B + d, N)
    svi = make_svi(d, icpt, N, prior=1.5)
    st = state_with(svi, rng.PRNGKey(99), loc, unc)
    got = float(svi.evaluate(st, torch.tensor(X).cuda(), torch.tensor(y).cuda()))
    spec = O.logreg_spec(d, icpt, 1.5, 3.0, lik_scale=N, obs_scale=1.0)
    jax_key = O.convert_to_jax_rng_key(O.split(O.PRNGKey(99), 1)[0])
    exp = O.logreg_evaluate(spec, loc, unc, X, y, jax_key)
    assert abs(got - exp) <= 2e-5 * abs(exp)
    # independent float64 restatement of the ELBO with the oracle's eps
    k = O.tf_split(O.tf_split(O.tf_split(jax_key, 2)[1], 2)[1], 2)[1]
    D = d + int(icpt)
    eps = O.tf_normal(k, D).astype(np.float64)
    s = np.log1p(np.exp(unc.astype(np.float64)))
    z = loc + s * eps
    ps = np.array([1.5] * d + [3.0] * (D - d))
    logq = np.sum(-0.5 * eps**2 - np.log(s) - 0.5 * np.log(2 * np.pi))
    logp = np.sum(-0.5 * (z / ps)**2 - np.log(ps) - 0.5 * np.log(2 * np.pi))
    t = X.astype(np.float64) @ z[:d] + (z[d] if icpt else 0.0)
    ll = np.sum(y * t - np.logaddexp(0, t))
    assert abs(exp - (-(logp + N / B * ll - logq))) <= 1e-5 * abs(exp)


@pytest.mark.parametrize("B,d", [(16384, 64), (4096, 512)])
def test_run_steps_with_more_workgroups_than_cus_matches_stepwise_updates(rng, B, d):
    """The device-resident loop applies every update exactly once in every workgroup, also when a launch has more
    workgroups than the chip has CUs (late workgroups start after the first ones have published the new
    parameters): run_steps and the one-step-at-a-time update() walk the same trajectory, and run_steps is
    bitwise reproducible."""
    from d3p_amd.minibatch import subsample_batchify_data
    N, steps = 10**5, 7
    g = torch.Generator().manual_seed(3)
    X = torch.randn(N, d, generator=g).cuda()
    y = (torch.rand(N, generator=g) < 0.5).float().cuda()
    svi = make_svi(d, False, N, C=1.0, sigma=0.3, lr=5e-2)
    st = state_with(svi, rng.PRNGKey(11), np.zeros(d, np.float32), np.full(d, -1.0, np.float32))
    init, get_batch = subsample_batchify_data((X, y), B)
    _, bstate = init(rng.PRNGKey(12))
    runs = [svi.run_steps(st, get_batch, bstate, 0, steps) for _ in range(3)]
    for s2, l2 in runs[1:]:
        assert torch.equal(s2.optim_state[1], runs[0][0].optim_state[1]) and torch.equal(l2, runs[0][1])
    ref = st
    ref_losses = []
    for t in range(steps):
        ref, l = svi.update(ref, *get_batch(t, bstate))
        ref_losses.append(float(l))
    np.testing.assert_allclose(np_(runs[0][1]), ref_losses, rtol=2e-5)
    np.testing.assert_allclose(np_(runs[0][0].optim_state[1]), np_(ref.optim_state[1]), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(np_(runs[0][0].optim_state[2]), np_(ref.optim_state[2]), rtol=2e-4, atol=1e-9)


def test_chained_launch_matches_one_launch_per_step(rng):
    """The default run loop executes the <= 128 steps of a prepared batch as ONE launch: resident workgroups that loop over
    the steps (MODE 4, d = 512, D3P_PERSISTENT_STEPS=1) or one workgroup set per step waiting on arrival counters (MODE 3); D3P_NO_CHAINED_STEPS=1 (switches are read once per process, so checked in child
    processes) selects one launch per step.  All walk the same trajectory bit for bit -- the sums are exact integer sums
    in each -- when their workgroups hold the same examples; the default chained form is pipelined (8-wave workgroups, two
    examples per wave) and agrees to fp32 rounding.  No wait hit its bound."""
    import ctypes as C
    import subprocess
    import sys
    import d3p_amd._lib as L
    from d3p_amd.minibatch import subsample_batchify_data
    code = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
import d3p_amd.random as rng
from d3p_amd.minibatch import subsample_batchify_data
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState
N, d, B, steps = 50000, 512, 4096, 150
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g).cuda(); y = (torch.rand(N, generator=g) < 0.5).float().cuda()
model = LogisticRegression(d)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-2), Trace_ELBO(), 1.0, 0.7, num_obs_total=N)
st = DPSVIState(svi.optim.init(torch.cat([torch.zeros(d), torch.full((d,), -2.0)]).cuda()), rng.PRNGKey(3), float(N))
_, gb = subsample_batchify_data((X, y), B)
s2, losses = svi.run_steps(st, gb, rng.PRNGKey(4), 0, steps)
np.save(sys.argv[1], np.concatenate([losses.cpu().numpy(), s2.optim_state[1].cpu().numpy()]))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    import tempfile
    outs = []
    # persistent launch (MODE 4, this geometry qualifies); the chained form (MODE 3) with the workgroup shape of the
    # other two; one launch per step; then the default chained form (pipelined: 8-wave workgroups), twice
    for env_extra in ({"D3P_PERSISTENT_STEPS": "1"}, {"D3P_NO_PIPELINED_STEPS": "1"}, {"D3P_NO_CHAINED_STEPS": "1"}, {}, {}):
        with tempfile.NamedTemporaryFile(suffix=".npy") as f:
            env = dict(os.environ, **env_extra)
            subprocess.run([sys.executable, "-c", code, f.name], check=True, env=env, timeout=300)
            outs.append(np.load(f.name))
    steps = 150
    # losses (every one the exact integer sum of the same fp32 partials) and the trajectory they imply: bit for bit.  The very
    # last update is applied by k_flush for the chained / one-launch forms and inside the launch by the persistent form:
    # the same formula compiled in two kernels (fma contraction may differ), so the final parameters agree to 1 ulp-ish.
    assert np.array_equal(outs[1], outs[2])
    assert np.array_equal(outs[0][:steps], outs[1][:steps])
    np.testing.assert_allclose(outs[0][steps:], outs[1][steps:], rtol=1e-6, atol=1e-7)
    assert np.all(np.isfinite(outs[0]))
    # The pipelined form groups the examples differently into workgroups (two per wave), so its fp32 workgroup partials
    # round differently: reproducible bit for bit run to run, and equal to the others to fp32 rounding over the 150 steps (two launches: 128 + 22)
    assert np.array_equal(outs[3], outs[4])
    steps = 150
    np.testing.assert_allclose(outs[3][:steps], outs[0][:steps], rtol=2e-6)
    np.testing.assert_allclose(outs[3][steps:], outs[0][steps:], rtol=1e-4, atol=2e-6)
    # abort flag of the bounded waits after a run in this process (d = 512: persistent form; d = 64: chained form)
    for N, d, B in ((20000, 512, 4096), (20000, 64, 1024)):
        _check_no_wait_hit_its_bound(rng, N, d, B)


def test_chained_launch_with_poisson_batches_matches_one_launch_per_step(rng):
    """Poisson batches are padded to max_batch_size and processed through dense lists of the valid positions; the chained
    launch walks those lists too and is bitwise identical to one launch per step (same workgroup shape; the pipelined
    default agrees to fp32 rounding)."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
import d3p_amd.random as rng
from d3p_amd.minibatch import poisson_batchify_data
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState
N, d, steps = 60000, 512, 150
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g).cuda(); y = (torch.rand(N, generator=g) < 0.5).float().cuda()
model = LogisticRegression(d)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-2), Trace_ELBO(), 1.0, 0.7, num_obs_total=N)
st = DPSVIState(svi.optim.init(torch.cat([torch.zeros(d), torch.full((d,), -2.0)]).cuda()), rng.PRNGKey(3), float(N))
_, gb = poisson_batchify_data((X, y), 4096 / N, 0.99)
s2, losses = svi.run_steps(st, gb, rng.PRNGKey(4), 0, steps)
np.save(sys.argv[1], np.concatenate([losses.cpu().numpy(), s2.optim_state[1].cpu().numpy()]))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    outs = []
    for env_extra in ({"D3P_NO_PIPELINED_STEPS": "1"}, {"D3P_NO_CHAINED_STEPS": "1"}, {}):
        with tempfile.NamedTemporaryFile(suffix=".npy") as f:
            subprocess.run([sys.executable, "-c", code, f.name], check=True, env=dict(os.environ, **env_extra), timeout=300)
            outs.append(np.load(f.name))
    assert np.array_equal(outs[0], outs[1]) and np.all(np.isfinite(outs[0]))
    # default (pipelined) chained form: other grouping of the examples into workgroups => equal to fp32 rounding
    np.testing.assert_allclose(outs[2][:150], outs[0][:150], rtol=2e-6)
    np.testing.assert_allclose(outs[2][150:], outs[0][150:], rtol=1e-4, atol=2e-6)


def _check_no_wait_hit_its_bound(rng, N, d, B):
    import ctypes as C
    import d3p_amd._lib as L
    from d3p_amd.minibatch import subsample_batchify_data
    X = torch.randn(N, d).cuda()
    y = (torch.rand(N) < 0.5).float().cuda()
    svi = make_svi(d, False, N)
    st = state_with(svi, rng.PRNGKey(1), np.zeros(d, np.float32), np.full(d, -2.0, np.float32))
    _, gb = subsample_batchify_data((X, y), B)
    svi.run_steps(st, gb, rng.PRNGKey(2), 0, 40)
    lib = L.load()
    model = svi._model_struct(d, {}, float(N))
    src = L.BatchSource(L.D3P_BATCH_FEISTEL, B, 0.0, 0, None, None, None, N, 0, N)
    ws = svi._workspace(0, X.device)          # (the run's buffer: the same purpose, device and stream)
    flag = C.c_int32(-1)
    L.check(lib.d3p_dpvi_logreg_chain_status(L.stream_ptr(), C.byref(model), C.byref(src), L.ptr(ws), ws.numel(), C.byref(flag)))
    assert flag.value == 0


def test_kernel_timing_hook_counts_launches_and_steps(rng):
    """d3p_dpvi_logreg_kernel_timing_*: while enabled the run loop brackets its step-kernel launches with HIP events; the read
    returns the summed kernel time, the launches and the DP-VI steps they covered, and clears the record (bench.py's roofline
    block is built from exactly these three numbers)."""
    import ctypes as C
    import d3p_amd._lib as L
    from d3p_amd.minibatch import subsample_batchify_data
    lib = L.load()
    N, d, B, steps = 20000, 64, 1024, 300
    X = torch.randn(N, d).cuda()
    y = (torch.rand(N) < 0.5).float().cuda()
    svi = make_svi(d, False, N)
    st = state_with(svi, rng.PRNGKey(1), np.zeros(d, np.float32), np.full(d, -2.0, np.float32))
    _, gb = subsample_batchify_data((X, y), B)
    us, launches, nsteps = C.c_double(-1.0), C.c_uint32(99), C.c_uint32(99)
    L.check(lib.d3p_dpvi_logreg_kernel_timing_read(C.byref(us), C.byref(launches), C.byref(nsteps)))  # clear
    svi.run_steps(st, gb, rng.PRNGKey(2), 0, steps)                                                 # hook off: nothing recorded
    L.check(lib.d3p_dpvi_logreg_kernel_timing_read(C.byref(us), C.byref(launches), C.byref(nsteps)))
    assert (us.value, launches.value, nsteps.value) == (0.0, 0, 0)
    L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(1))
    try:
        svi.run_steps(st, gb, rng.PRNGKey(2), 0, steps)
    finally:
        L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(0))
    L.check(lib.d3p_dpvi_logreg_kernel_timing_read(C.byref(us), C.byref(launches), C.byref(nsteps)))
    assert nsteps.value == steps
    assert launches.value == 3                       # chained launches of 128 + 128 + 44 steps
    assert 0.0 < us.value < 1e6 and us.value / steps > 1.0   # a step takes microseconds, not nanoseconds or seconds
    L.check(lib.d3p_dpvi_logreg_kernel_timing_read(C.byref(us), C.byref(launches), C.byref(nsteps)))
    assert (launches.value, nsteps.value) == (0, 0)


def test_wide_rows_run_steps_matches_stepwise_updates(rng):
    """d = 3000 (the column-chunked kernel, two-kernel steps): the device-resident loop and one update() per step walk the same
    trajectory, bitwise reproducibly."""
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, B, steps = 3000, 3000, 512, 5
    g = torch.Generator().manual_seed(4)
    X = torch.randn(N, d, generator=g).cuda()
    y = (torch.rand(N, generator=g) < 0.5).float().cuda()
    svi = make_svi(d, False, N, C=1.0, sigma=0.4, lr=2e-2)
    st = state_with(svi, rng.PRNGKey(21), np.zeros(d, np.float32), np.full(d, -1.5, np.float32))
    init, get_batch = subsample_batchify_data((X, y), B)
    _, bstate = init(rng.PRNGKey(22))
    runs = [svi.run_steps(st, get_batch, bstate, 0, steps) for _ in range(2)]
    assert torch.equal(runs[0][0].optim_state[1], runs[1][0].optim_state[1]) and torch.equal(runs[0][1], runs[1][1])
    ref, ref_losses = st, []
    for t in range(steps):
        ref, l = svi.update(ref, *get_batch(t, bstate))
        ref_losses.append(float(l))
    np.testing.assert_allclose(np_(runs[0][1]), ref_losses, rtol=2e-5)
    np.testing.assert_allclose(np_(runs[0][0].optim_state[1]), np_(ref.optim_state[1]), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("d", [512, 1024])
def test_intercept_tile_run_steps_matches_stepwise_updates_and_scalar_form(rng, d):
    """d = 512 / 1024 WITH an intercept (examples/logistic_regression.py:49-66): k_logreg_main's TAIL form -- a full tile of
    features, the intercept as the last second-half column, the left-over middle column computed by every lane.  The chained
    device-resident loop (140 steps: two launches) and one update() per step walk the same trajectory, reproducibly; a child
    process with D3P_NO_TAIL_TILE=1 (the scalar-load form, other summation order) agrees to fp32 rounding."""
    import subprocess
    import sys
    import tempfile
    from d3p_amd.minibatch import subsample_batchify_data
    N, B, steps = 6000, 700, 140
    g = torch.Generator().manual_seed(5)
    X = torch.randn(N, d, generator=g).cuda()
    y = (torch.rand(N, generator=g) < 0.5).float().cuda()
    svi = make_svi(d, True, N, C=1.0, sigma=0.4, lr=2e-2)
    st = state_with(svi, rng.PRNGKey(31), np.zeros(d + 1, np.float32), np.full(d + 1, -1.5, np.float32))
    init, get_batch = subsample_batchify_data((X, y), B)
    _, bstate = init(rng.PRNGKey(32))
    runs = [svi.run_steps(st, get_batch, bstate, 0, steps) for _ in range(2)]
    assert torch.equal(runs[0][0].optim_state[1], runs[1][0].optim_state[1]) and torch.equal(runs[0][1], runs[1][1])
    ref, ref_losses = st, []
    for t in range(steps):
        ref, l = svi.update(ref, *get_batch(t, bstate))
        ref_losses.append(float(l))
    np.testing.assert_allclose(np_(runs[0][1]), ref_losses, rtol=2e-5)
    np.testing.assert_allclose(np_(runs[0][0].optim_state[1]), np_(ref.optim_state[1]), rtol=1e-5, atol=2e-6)
    code = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import d3p_amd.random as rng
from d3p_amd.minibatch import subsample_batchify_data
from test_gpu_dpsvi import make_svi, state_with
d = int(sys.argv[2]); N, B, steps = 6000, 700, 140
g = torch.Generator().manual_seed(5)
X = torch.randn(N, d, generator=g).cuda(); y = (torch.rand(N, generator=g) < 0.5).float().cuda()
svi = make_svi(d, True, N, C=1.0, sigma=0.4, lr=2e-2)
st = state_with(svi, rng.PRNGKey(31), np.zeros(d + 1, np.float32), np.full(d + 1, -1.5, np.float32))
init, get_batch = subsample_batchify_data((X, y), B)
_, bstate = init(rng.PRNGKey(32))
s2, losses = svi.run_steps(st, get_batch, bstate, 0, steps)
np.save(sys.argv[1], np.concatenate([losses.cpu().numpy(), s2.optim_state[1].cpu().numpy()]))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    with tempfile.NamedTemporaryFile(suffix=".npy") as f:
        subprocess.run([sys.executable, "-c", code, f.name, str(d)], check=True, env=dict(os.environ, D3P_NO_TAIL_TILE="1"),
                       timeout=300)
        other = np.load(f.name)
    np.testing.assert_allclose(other[:steps], np_(runs[0][1]), rtol=2e-5)
    np.testing.assert_allclose(other[steps:], np_(runs[0][0].optim_state[1]), rtol=1e-4, atol=5e-6)


def test_chained_launch_is_reproducible_under_uneven_load(rng):
    """The cross-workgroup hand-offs of the chained launch (arrival counters, group flags, agent-scope accumulator reads) must
    not depend on timing: a run made while matrix products of changing size on a second stream compete for the CUs ends bit
    for bit like the undisturbed one, and no bounded wait hits its bound (tools/soak_chained.py is the long form)."""
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, B, steps = 60000, 512, 4096, 3000
    g = torch.Generator().manual_seed(7)
    X = torch.randn(N, d, generator=g).cuda()
    y = (torch.rand(N, generator=g) < 0.5).float().cuda()
    svi = make_svi(d, False, N)
    st = state_with(svi, rng.PRNGKey(41), np.zeros(d, np.float32), np.full(d, -2.0, np.float32))
    _, gb = subsample_batchify_data((X, y), B)
    quiet_state, quiet_losses = svi.run_steps(st, gb, rng.PRNGKey(42), 0, steps)
    torch.cuda.synchronize()
    from d3p_amd.dist import concurrent_streams
    side = concurrent_streams(1)[0]
    mats = [torch.randn(n, n, device="cuda") for n in (512, 1024, 3072)]
    done = torch.cuda.Event()
    loud_state, loud_losses = svi.run_steps(st, gb, rng.PRNGKey(42), 0, steps, check_status=False)   # stays asynchronous
    done.record()
    k = 0
    with torch.cuda.stream(side):
        while not done.query():
            a = mats[k % 3]
            (a @ a).sum()
            k += 1
    torch.cuda.synchronize()
    assert k > 0
    assert svi.last_run_status() == (False, False)
    assert torch.equal(loud_losses, quiet_losses) and torch.equal(loud_state.optim_state[1], quiet_state.optim_state[1])
    assert bool(torch.isfinite(quiet_losses).all())
    _check_no_wait_hit_its_bound(rng, 20000, 512, 4096)


@pytest.mark.parametrize("kind", ["with_replacement", "split"])
def test_run_steps_serves_batchifiers_without_a_native_loop(rng, kind):
    """Sampling with replacement (minibatch.py:195-214) and split_batchify_data's epoch batches (minibatch.py:242-312) have no
    device-resident loop: run_steps then makes the same steps through get_batch + update and returns what that loop returns."""
    from d3p_amd.minibatch import split_batchify_data, subsample_batchify_data
    N, d, B, steps = 1500, 24, 50, 5
    g = torch.Generator().manual_seed(9)
    X = torch.randn(N, d, generator=g).cuda()
    y = (torch.rand(N, generator=g) < 0.5).float().cuda()
    svi = make_svi(d, False, N, C=1.0, sigma=0.5, lr=1e-2)
    st = state_with(svi, rng.PRNGKey(51), np.zeros(d, np.float32), np.full(d, -1.0, np.float32))
    if kind == "with_replacement":
        init, get_batch = subsample_batchify_data((X, y), B, with_replacement=True)
    else:
        init, get_batch = split_batchify_data((X, y), B)
    _, bstate = init(rng.PRNGKey(52))
    new_st, losses = svi.run_steps(st, get_batch, bstate, 1, steps)
    ref, ref_losses = st, []
    for t in range(steps):
        ref, l = svi.update(ref, *get_batch(1 + t, bstate))
        ref_losses.append(float(l))
    assert losses.shape == (steps,)
    np.testing.assert_array_equal(np_(losses), np.asarray(ref_losses, np.float32))
    assert torch.equal(new_st.optim_state[1], ref.optim_state[1]) and torch.equal(new_st.rng_key, ref.rng_key)


def test_run_steps_falls_back_to_one_launch_per_step_when_the_chained_launch_is_stopped(rng, monkeypatch):
    """A chained launch that a bounded wait stopped (its workgroups did not make progress together: a GPU shared with other
    work) must cost the caller time, not the result: run_steps re-runs the steps from the untouched input state with one launch
    per step (d3p_dpvi_logreg_set_run_form(1)) and warns.  The stop is simulated at the point where run_steps reads the status."""
    import d3p_amd._lib as L
    from d3p_amd.minibatch import subsample_batchify_data
    from d3p_amd.svi import DPSVI
    N, d, B, steps = 20000, 512, 4096, 9
    g = torch.Generator().manual_seed(17)
    X = torch.randn(N, d, generator=g).cuda()
    y = (torch.rand(N, generator=g) < 0.5).float().cuda()
    svi = make_svi(d, False, N, C=1.0, sigma=0.5, lr=1e-2)
    st = state_with(svi, rng.PRNGKey(71), np.zeros(d, np.float32), np.full(d, -2.0, np.float32))
    _, gb = subsample_batchify_data((X, y), B)
    want_state, want_losses = svi.run_steps(st, gb, rng.PRNGKey(72), 0, steps)
    real = DPSVI.last_run_status
    calls = {"n": 0}

    def stopped_once(self):
        calls["n"] += 1
        aborted, nonfinite = real(self)
        if calls["n"] == 1:
            self._last_abort_code = 4 | (3 << 8)      # "the previous step was not released", step 3
            return True, nonfinite
        return aborted, nonfinite
    monkeypatch.setattr(DPSVI, "last_run_status", stopped_once)
    forms = []
    lib = L.load()
    real_set = lib.d3p_dpvi_logreg_set_run_form

    class Spy:
        def __call__(self, form):
            forms.append(int(form))
            return real_set(form)
    monkeypatch.setattr(lib, "d3p_dpvi_logreg_set_run_form", Spy(), raising=False)
    with pytest.warns(RuntimeWarning, match="one launch per step"):
        got_state, got_losses = svi.run_steps(st, gb, rng.PRNGKey(72), 0, steps)
    assert forms == [1, 0] and calls["n"] == 2
    # (the two forms group the examples differently into workgroups: equal to fp32 rounding, same keys)
    np.testing.assert_allclose(np_(got_losses), np_(want_losses), rtol=2e-6)
    np.testing.assert_allclose(np_(got_state.optim_state[1]), np_(want_state.optim_state[1]), rtol=1e-4, atol=2e-6)
    assert torch.equal(got_state.rng_key, want_state.rng_key) and int(got_state.optim_state[0]) == steps
    # and a run that is stopped in the fallback form too raises
    monkeypatch.setattr(DPSVI, "last_run_status", lambda self: (True, False))
    with pytest.warns(RuntimeWarning), pytest.raises(L.D3PError):
        svi.run_steps(st, gb, rng.PRNGKey(72), 0, steps)


# --------------------------------------------------------------------------- the example's own guide: two sample sites, four leaves
@pytest.mark.parametrize("B,d", [(64, 512), (37, 9)])
def test_example_guide_with_two_sites_vs_oracle(rng, O, B, d):
    """examples/logistic_regression.py:49-86 as it stands in the reference: model with sites 'w' (d) and 'intercept', the hand-written
    guide with exp-parametrised scales -- four parameter leaves in sorted-name order, ONE perturbation key per leaf (svi.py:487-491), every
    sample site's eps from its own key.  Three masked updates at d = 512 + intercept through DPSVI.update (the fused clipped sums between
    d3p_dpvi_leaves_begin and d3p_dpvi_leaves_finalize) against O.meanfield_logreg_update: losses 2e-5, parameters / Adam moments 1e-4, keys bit-exact; the per-site eps kernel
    (d3p_px_eps_sites) against the oracle's stream (rtol 2e-6, like rng.normal); evaluate; get_params' leaf names and shapes."""
    import ctypes as C
    import d3p_amd._lib as L
    from d3p_amd.models import Adam, LogisticRegression, MeanFieldGuide, Trace_ELBO
    from d3p_amd.svi import DPSVI
    N, C_, sigma, lr = 5000, 0.8, 1.1, 1e-2
    r = np.random.default_rng(100 + d)
    X = r.normal(size=(B, d)).astype(np.float32)
    y = (r.random(B) < 0.5).astype(np.float32)
    mask = r.random(B) < 0.8
    model = LogisticRegression(d, prior_scale=1.0, intercept=True, intercept_prior_scale=1.0)
    svi = DPSVI(model, MeanFieldGuide(model), Adam(lr), Trace_ELBO(), C_, sigma, num_obs_total=N)
    Xt, yt, mt = torch.tensor(X).cuda(), torch.tensor(y).cuda(), torch.tensor(mask).cuda()
    st = svi.init(rng.PRNGKey(21), Xt, yt)
    assert st.optim_state[1].numel() == 2 * d + 2 and float(st.optim_state[1].abs().max()) == 0.0 and st.observation_scale == N
    par = svi.get_params(st)
    assert sorted(par) == ["intercept_loc", "intercept_std_log", "w_loc", "w_std_log"]
    assert par["w_loc"].shape == (d,) and par["intercept_std_log"].shape == ()

    # the per-site eps stream
    jk = O.convert_to_jax_rng_key(O.split(O.PRNGKey(21), 3)[1])
    eps = torch.empty((B, d + 1), device="cuda")
    sizes = (C.c_int32 * 2)(d, 1)
    jk_dev = torch.from_numpy(np.ascontiguousarray(jk, np.uint32).view(np.int32)).cuda().view(torch.uint32)
    L.check(L.load().d3p_px_eps_sites(L.stream_ptr(), L.ptr(jk_dev), B, 0, B, sizes, 2, L.ptr(eps)))
    want = O.px_eps_sites(jk, B, [d, 1])
    np.testing.assert_allclose(np_(eps), want, rtol=2e-6, atol=1e-7)
    # (a shard of the batch: rows pos0 .. of the same stream)
    part = torch.empty((5, d + 1), device="cuda")
    L.check(L.load().d3p_px_eps_sites(L.stream_ptr(), L.ptr(jk_dev), B, 7, 5, sizes, 2, L.ptr(part)))
    assert torch.equal(part, eps[7:12])

    spec = O.logreg_spec(d, True, 1.0, 1.0, lik_scale=N, obs_scale=N, guide_exp=True)
    hy = O.Hyper(C_, sigma, lr, 0.9, 0.999, 1e-8)
    ost = O.MeanFieldLogregState(O.PRNGKey(21), d)
    for step in range(3):
        st, loss = svi.update(st, Xt, yt, mask=mt)
        eloss, _ = O.meanfield_logreg_update(spec, hy, ost, X, y, mask.astype(np.float32))
        assert abs(float(loss) - eloss) <= 2e-5 * abs(eloss), (step, float(loss), eloss)
        assert np.array_equal(np_(st.rng_key).ravel(), ost.key)
        close(np_(st.optim_state[2]), ost.m)
        np.testing.assert_allclose(np_(st.optim_state[1]), ost.params, rtol=1e-4, atol=1e-6)
    assert int(st.optim_state[0]) == 3
    got = float(svi.evaluate(st, Xt, yt))
    spec_e = O.logreg_spec(d, True, 1.0, 1.0, lik_scale=N, obs_scale=1.0, guide_exp=True)
    exp = O.meanfield_logreg_evaluate(spec_e, np_(st.optim_state[1]), X, y, O.convert_to_jax_rng_key(O.split(ost.key, 1)[0]))
    assert abs(got - exp) <= 2e-5 * abs(exp)


@pytest.mark.parametrize("B,d,masked", [(200, 4, False), (64, 512, True), (4096, 512, False), (33, 70, True)])
def test_example_guide_fused_update_vs_stage_composition(rng, B, d, masked):
    """DPSVI.update with the example's guide runs around the FUSED clipped sums (d3p_dpvi_leaves_begin -> d3p_px_eps_sites ->
    d3p_dpvi_logreg_local_sums -> d3p_dpvi_leaves_finalize, nine launches); the reference's five-stage composition (svi.py:413-434) on the
    same state is its check: keys and step counter bit-exact, the perturbed gradient (the noise words are the same, the clipped sums
    differ in fp32 summation order) 2e-5 of its largest element, loss 2e-5, state 1e-4.  Four updates, so that Adam moments and a moved
    state key enter; then a batch with no valid example (svi.py:305: factor 0, C / 0 = inf -> NaN state in both)."""
    from d3p_amd.models import Adam, LogisticRegression, MeanFieldGuide, Trace_ELBO
    from d3p_amd.svi import DPSVI
    N = 20000
    r = np.random.default_rng(7 * d + B)
    Xt = torch.tensor(r.normal(size=(B, d)).astype(np.float32)).cuda()
    yt = torch.tensor((r.random(B) < 0.5).astype(np.float32)).cuda()
    mt = torch.tensor(r.random(B) < 0.7).cuda() if masked else True
    model = LogisticRegression(d, prior_scale=2.0, intercept=True, intercept_prior_scale=3.0)
    svi = DPSVI(model, MeanFieldGuide(model), Adam(5e-2), Trace_ELBO(), 1.3, 0.9, num_obs_total=N)
    assert svi._leaves_fusable()
    st = svi.init(rng.PRNGKey(5), Xt, yt)
    P = 2 * d + 2
    for step in range(4):
        g = torch.empty(P, device="cuda")
        new, loss = svi.update(st, Xt, yt, mask=mt, _grad_out=g)
        ref, ref_loss = svi._update_staged(st, Xt, yt, mask=mt)
        assert torch.equal(new.rng_key, ref.rng_key) and int(new.optim_state[0]) == int(ref.optim_state[0]) == step + 1
        assert abs(float(loss) - float(ref_loss)) <= 2e-5 * abs(float(ref_loss))
        # the staged path's gradient: what moved m from the old state
        m_old, m_ref = np_(st.optim_state[2]), np_(ref.optim_state[2])
        g_ref = (m_ref - 0.9 * m_old) / np.float32(0.1)
        np.testing.assert_allclose(np_(g), g_ref, rtol=0, atol=3e-5 * np.abs(g_ref).max())
        for k in (1, 2, 3):
            a, b = np_(new.optim_state[k]), np_(ref.optim_state[k])
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5 * max(np.abs(b).max(), 1e-30))
        assert new.optim_state[1].data_ptr() != st.optim_state[1].data_ptr()      # (functional: the old state is untouched)
        st = new
    empty, loss0 = svi.update(st, Xt, yt, mask=False)
    ref, ref_loss0 = svi._update_staged(st, Xt, yt, mask=False)
    assert float(loss0) == float(ref_loss0) == 0.0
    assert torch.equal(empty.rng_key, ref.rng_key)
    assert np.array_equal(np.isnan(np_(empty.optim_state[1])), np.isnan(np_(ref.optim_state[1])))
    assert np.isnan(np_(empty.optim_state[1])).any()
    _, loss1 = svi.update(empty, Xt, yt, mask=False)       # (a masked sum over NaN parameters: NaN * 0 = NaN in the reference)
    _, ref_loss1 = svi._update_staged(empty, Xt, yt, mask=False)
    assert np.isnan(float(loss1)) and np.isnan(float(ref_loss1))


def test_example_guide_four_leaves_get_four_noise_streams(rng, O):
    """svi.py:487-491 with the example's guide: split(key, 4), leaf k += normal(site_key_k, shape_k) * scale -- the two scalar leaves take
    word 0 of their own keys, not elements of a shared vector (bit-exact against the oracle's ChaCha stream)."""
    from d3p_amd.models import Adam, LogisticRegression, MeanFieldGuide, Trace_ELBO
    from d3p_amd.svi import DPSVI
    d = 12
    model = LogisticRegression(d, intercept=True)
    svi = DPSVI(model, MeanFieldGuide(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, num_obs_total=100)
    zeros = {"intercept_loc": torch.zeros((), device="cuda"), "intercept_std_log": torch.zeros((), device="cuda"),
             "w_loc": torch.zeros(d, device="cuda"), "w_std_log": torch.zeros(d, device="cuda")}
    out = DPSVI.perturbation_function(rng, rng.PRNGKey(8), zeros, 1.0)
    keys = O.split(O.PRNGKey(8), 4)
    for k, name in enumerate(sorted(zeros)):
        want = O.normal(keys[k], tuple(zeros[name].shape))
        np.testing.assert_allclose(np_(out[name]), want, rtol=2e-6, atol=1e-7)


# --------------------------------------------------------------------------- host arrays at the Python surface
def test_host_arrays_are_moved_to_the_device_not_handed_to_kernels(rng):
    """The reference takes numpy arrays at every call (jax moves them, float64 as float32 with x64 off).  Here a host pointer inside a
    kernel would be a GPU memory fault: update / evaluate / init move host tensors and numpy arrays (batch, labels, mask) to the device
    and give bit for bit what the same call gives on device tensors; a batchifier over a HOST table is refused by run_steps with an
    error, not run."""
    import d3p_amd._lib as L
    from d3p_amd.minibatch import subsample_batchify_data
    from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, MeanFieldGuide, Trace_ELBO
    from d3p_amd.svi import DPSVI
    B, d, N = 96, 24, 3000
    r = np.random.default_rng(4)
    X64 = r.normal(size=(B, d))                        # float64 on the host
    y_int = (r.random(B) < 0.5).astype(np.int64)
    mask_np = r.random(B) < 0.8
    Xd, yd, md = torch.tensor(X64, dtype=torch.float32).cuda(), torch.tensor(y_int, dtype=torch.float32).cuda(), torch.tensor(mask_np).cuda()
    for guide_cls, icpt in ((AutoDiagonalNormal, False), (MeanFieldGuide, True)):
        model = LogisticRegression(d, intercept=icpt)
        svi = DPSVI(model, guide_cls(model), Adam(1e-2), Trace_ELBO(), 1.0, 0.7, num_obs_total=N)
        st_dev = svi.init(rng.PRNGKey(2), Xd, yd)
        st_host = svi.init(rng.PRNGKey(2), X64, y_int)
        assert st_host.optim_state[1].is_cuda and torch.equal(st_host.optim_state[1], st_dev.optim_state[1])
        want, want_loss = svi.update(st_dev, Xd, yd, mask=md)
        for Xa, ya, ma in ((X64, y_int, mask_np), (torch.tensor(X64), torch.tensor(y_int), torch.tensor(mask_np)),
                           (Xd, yd, list(mask_np))):
            got, loss = svi.update(st_dev, Xa, ya, mask=ma)
            assert torch.equal(got.optim_state[1], want.optim_state[1]) and float(loss) == float(want_loss)
            assert torch.equal(got.rng_key, want.rng_key)
        assert float(svi.evaluate(want, X64, y_int)) == float(svi.evaluate(want, Xd, yd))
        # a float64 batch that already lives on the device is converted too (its bytes are not float32 rows)
        got64, loss64 = svi.update(st_dev, torch.tensor(X64).cuda(), torch.tensor(y_int).cuda().double(), mask=md)
        assert torch.equal(got64.optim_state[1], want.optim_state[1]) and float(loss64) == float(want_loss)
    # the stage-wise path too
    st2, px_loss, px_grads, n_el, factor = svi._compute_per_example_gradients(st_dev, rng.PRNGKey(9), X64, y_int, mask=mask_np)
    st3, px_loss_d, px_grads_d, _, _ = svi._compute_per_example_gradients(st_dev, rng.PRNGKey(9), Xd, yd, mask=md)
    assert torch.equal(px_loss, px_loss_d) and all(torch.equal(px_grads[k], px_grads_d[k]) for k in px_grads)
    # a table on the host behind a batchifier: refused
    model = LogisticRegression(d)
    svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-2), Trace_ELBO(), 1.0, 0.7, num_obs_total=N)
    table = (torch.tensor(r.normal(size=(N, d)), dtype=torch.float32), torch.zeros(N))
    init_b, get_b = subsample_batchify_data(table, batch_size=B, rng_suite=rng)
    _, bstate = init_b(rng.PRNGKey(1))
    with pytest.raises(L.D3PError):
        svi.run_steps(st_dev, get_b, bstate, 0, 3)
    # a device table that is not contiguous float32 (float64 features, integer labels): run_steps walks get_batch + update, which
    # convert per step -- the same trajectory as the native loop over the float32 table
    Xt = torch.tensor(r.normal(size=(N, d)), dtype=torch.float32)
    yt = torch.tensor(r.integers(0, 2, size=N))
    _, get_f32 = subsample_batchify_data((Xt.cuda(), yt.float().cuda()), batch_size=B, rng_suite=rng)
    _, get_f64 = subsample_batchify_data((Xt.double().cuda(), yt.cuda()), batch_size=B, rng_suite=rng)
    st_plain = svi.init(rng.PRNGKey(2), Xd, yd)
    a, la = svi.run_steps(st_plain, get_f32, rng.PRNGKey(1), 0, 3)
    b, lb = svi.run_steps(st_plain, get_f64, rng.PRNGKey(1), 0, 3)
    assert torch.equal(a.rng_key, b.rng_key)
    np.testing.assert_allclose(np_(lb), np_(la), rtol=2e-5)
    np.testing.assert_allclose(np_(b.optim_state[1]), np_(a.optim_state[1]), rtol=1e-4, atol=1e-5)
    # a batchifier state that is not a device key (here: a host copy of one): a TypeError, not a host address in a kernel
    with pytest.raises(TypeError):
        svi.run_steps(st_plain, get_f32, rng.PRNGKey(1).cpu(), 0, 3)
    # a float64 parameter vector: the optimiser's state is float32 whatever it is given (jax without x64); a state BUILT with float64
    # arrays is refused (its bytes are not float32 parameters)
    p64 = torch.tensor(np.concatenate([np.zeros(d), np.full(d, -2.0)]))           # float64, host
    st64 = svi.optim.init(p64.cuda())
    assert st64[1].dtype == torch.float32 and st64[2].dtype == torch.float32
    assert svi.optim.init(p64.numpy())[1].is_cuda
    from d3p_amd.svi import DPSVIState as _S
    with pytest.raises(L.D3PError):
        svi.update(_S((st64[0], p64.cuda(), st64[2], st64[3]), st_plain.rng_key, st_plain.observation_scale), Xd, yd)
    # a state that went to the host (e.g. to be saved): refused until it is moved back
    from d3p_amd.svi import DPSVIState
    host_state = DPSVIState(tuple(t.cpu() for t in st_dev.optim_state), st_dev.rng_key.cpu(), st_dev.observation_scale)
    with pytest.raises(L.D3PError):
        svi.update(host_state, Xd, yd)
    with pytest.raises(L.D3PError):
        svi.evaluate(host_state, Xd, yd)


def test_shapes_that_do_not_fit_the_state_are_python_errors(rng):
    """The kernels take sizes from the batch and addresses from the state; a batch with another feature count, labels or a mask of
    another length, or a state made for another model would be read out of bounds on the GPU: every entry point refuses them first."""
    from d3p_amd.minibatch import subsample_batchify_data
    from d3p_amd.models import (Adam, AutoDiagonalNormal, GaussianMixtureGuide, GaussianMixtureModel, LogisticRegression, MeanFieldGuide,
                                Trace_ELBO, VAEGuide, VAEModel)
    from d3p_amd.svi import DPSVI
    B, d, N = 32, 12, 1000
    X = torch.randn(B, d, device="cuda")
    y = (torch.rand(B, device="cuda") < 0.5).float()
    for guide_cls, icpt in ((AutoDiagonalNormal, False), (MeanFieldGuide, True)):
        model = LogisticRegression(d, intercept=icpt)
        svi = DPSVI(model, guide_cls(model), Adam(1e-2), Trace_ELBO(), 1.0, 0.7, num_obs_total=N)
        st = svi.init(rng.PRNGKey(2), X, y)
        svi.update(st, X, y)
        wider = torch.randn(B, d + 3, device="cuda")
        for call in (lambda: svi.update(st, wider, y), lambda: svi.evaluate(st, wider, y),
                     lambda: svi.update(st, X, y[:-1]), lambda: svi.update(st, X, y, mask=torch.ones(B + 1, dtype=torch.bool)),
                     lambda: svi._compute_per_example_gradients(st, rng.PRNGKey(1), wider, y)):
            with pytest.raises(ValueError):
                call()
    table = (torch.randn(N, d + 1, device="cuda"), torch.zeros(N, device="cuda"))
    model = LogisticRegression(d)
    svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-2), Trace_ELBO(), 1.0, 0.7, num_obs_total=N)
    st = svi.init(rng.PRNGKey(2), X, y)
    init_b, get_b = subsample_batchify_data(table, batch_size=B, rng_suite=rng)
    with pytest.raises(ValueError):
        svi.run_steps(st, get_b, init_b(rng.PRNGKey(1))[1], 0, 3)
    gmm = GaussianMixtureModel(3)
    gsvi = DPSVI(gmm, GaussianMixtureGuide(gmm), Adam(1e-2), Trace_ELBO(), 1.0, 0.7, k=3, num_obs_total=N)
    gst = gsvi.init(rng.PRNGKey(3), X)
    gsvi.update(gst, X)
    with pytest.raises(ValueError):
        gsvi.update(gst, torch.randn(B, d + 1, device="cuda"))
    with pytest.raises(ValueError):
        gsvi.evaluate(gst, torch.randn(B, d + 1, device="cuda"))
    vae = VAEModel(z_dim=4, hidden_dim=16, scale=1.0 / N)
    vsvi = DPSVI(vae, VAEGuide(vae), Adam(1e-2), Trace_ELBO(), 1.0, 0.7, num_obs_total=N)
    Xv = (torch.rand(B, 24, device="cuda") < 0.5).float()
    vst = vsvi.init(rng.PRNGKey(4), Xv)
    vsvi.update(vst, Xv)
    with pytest.raises(ValueError):
        vsvi.update(vst, (torch.rand(B, 28, device="cuda") < 0.5).float())
    with pytest.raises(ValueError):
        vsvi.evaluate(vst, (torch.rand(B, 28, device="cuda") < 0.5).float())


def test_stage_methods_check_the_sizes_they_hand_to_kernels(rng):
    """The five stage methods are public (the reference's tests call them): a gradient leaf with another number of rows than there
    are losses, or a gradient that is shorter than the parameter vector, is a ValueError before a kernel reads past it."""
    from d3p_amd.models import SGD, Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.optimizers import ADADP
    from d3p_amd.svi import DPSVI
    d, B = 6, 10
    model = LogisticRegression(d)
    X, y = torch.randn(B, d, device="cuda"), torch.zeros(B, device="cuda")
    for optim in (Adam(1e-2), SGD(1e-2), ADADP(1e-2)):
        svi = DPSVI(model, AutoDiagonalNormal(model), optim, Trace_ELBO(), 1.0, 0.5, num_obs_total=100)
        st = svi.init(rng.PRNGKey(0), X, y)
        with pytest.raises(ValueError):
            svi._apply_gradient(st, {"auto_loc": torch.zeros(d, device="cuda"), "auto_scale": torch.zeros(d - 1, device="cuda")})
        svi._apply_gradient(st, {"auto_loc": torch.zeros(d, device="cuda"), "auto_scale": torch.zeros(d, device="cuda")})
    with pytest.raises(ValueError):
        svi._combine_gradients({"auto_loc": torch.zeros(B + 2, d, device="cuda")}, torch.zeros(B, device="cuda"))
    loss, avg = svi._combine_gradients({"auto_loc": torch.ones(B, d, device="cuda")}, torch.full((B,), 2.0, device="cuda"))
    assert float(loss) == 2.0 and torch.equal(avg["auto_loc"], torch.ones(d, device="cuda"))


def test_run_is_num_steps_updates_on_the_same_arguments(rng):
    """numpyro.infer.SVI.run, inherited by the reference's DPSVI: init, then num_steps x update on the same arguments;
    SVIRunResult(params, state, losses).  stable_update=True would bypass the private pipeline in the reference: refused."""
    from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.svi import DPSVI, SVIRunResult
    B, d = 40, 7
    X, y = torch.randn(B, d, device="cuda"), (torch.rand(B, device="cuda") < 0.5).float()
    model = LogisticRegression(d)
    svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-2), Trace_ELBO(), 1.0, 0.5, num_obs_total=500)
    res = svi.run(rng.PRNGKey(3), 5, X, y)
    assert isinstance(res, SVIRunResult) and tuple(res.losses.shape) == (5,) and int(res.state.optim_state[0]) == 5
    st = svi.init(rng.PRNGKey(3), X, y)
    ls = []
    for _ in range(5):
        st, l = svi.update(st, X, y)
        ls.append(float(l))
    assert torch.equal(res.state.optim_state[1], st.optim_state[1]) and torch.equal(res.state.rng_key, st.rng_key)
    assert [float(v) for v in res.losses] == ls
    assert sorted(res.params) == sorted(svi.get_params(st)) and torch.equal(res.params["auto_loc"], svi.get_params(st)["auto_loc"])
    more = svi.run(rng.PRNGKey(9), 2, X, y, init_state=res.state)
    assert int(more.state.optim_state[0]) == 7
    with pytest.raises(NotImplementedError):
        svi.run(rng.PRNGKey(3), 2, X, y, stable_update=True)


def test_one_dpsvi_object_driven_from_two_streams(rng):
    """The reference is functional (nothing shared between calls); here a DPSVI object owns scratch buffers.  They are per STREAM: two
    streams that enqueue updates of different batches through the same object, beside each other, give what the same calls give one
    after the other."""
    from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.svi import DPSVI
    B, d, N = 4096, 512, 100000
    g = torch.Generator().manual_seed(1)
    Xa, Xb = torch.randn(B, d, generator=g).cuda(), torch.randn(B, d, generator=g).cuda()
    ya, yb = (torch.rand(B, generator=g) < 0.5).float().cuda(), (torch.rand(B, generator=g) < 0.5).float().cuda()
    model = LogisticRegression(d)
    svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-2), Trace_ELBO(), 1.0, 0.5, num_obs_total=N)
    st = svi.init(rng.PRNGKey(0), Xa, ya)
    want_a, la = svi.update(st, Xa, ya)
    want_b, lb = svi.update(st, Xb, yb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(30):
        with torch.cuda.stream(s1):
            got_a, ga = svi.update(st, Xa, ya)
        with torch.cuda.stream(s2):
            got_b, gb = svi.update(st, Xb, yb)
        torch.cuda.synchronize()
        assert torch.equal(got_a.optim_state[1], want_a.optim_state[1]) and float(ga) == float(la)
        assert torch.equal(got_b.optim_state[1], want_b.optim_state[1]) and float(gb) == float(lb)


@pytest.mark.parametrize("d", [24, 512, 2600])
def test_small_guide_scales_keep_their_relative_accuracy(rng, O, d):
    """softplus(u) as a SCALE: `max(u, 0) + log(1 + exp(-|u|))` loses it below u ~ -4 (1e-7 absolute: 3e-5 relative at u = -6, 20 % at
    u = -15, exactly 0 below -16.6 -- 0 / 0 and log 0 in the step).  Posterior standard deviations of 1e-3 .. 1e-7 are such u.  Per-example
    gradients (those of the unconstrained scales carry 1 / s), the loss, evaluate and a fused update at u in [-60, -4] against the oracle
    (jax.nn.softplus = logaddexp; log1pf there): 2e-5."""
    B, N = 12, 1000
    r = np.random.default_rng(d)
    X = r.normal(size=(B, d)).astype(np.float32)
    y = (r.random(B) < 0.5).astype(np.float32)
    loc = (r.normal(size=d) * 0.3).astype(np.float32)
    unc = r.choice(np.array([-4.5, -6.0, -9.0, -12.0, -14.9, -15.1, -16.7, -20.0, -40.0, -60.0], np.float32), size=d).astype(np.float32)
    svi = make_svi(d, False, N, C=1e6, sigma=0.0, lr=1e-3)
    st = state_with(svi, rng.PRNGKey(3), loc, unc)
    eps = r.normal(size=(B, d)).astype(np.float32)
    Xt, yt = torch.tensor(X).cuda(), torch.tensor(y).cuda()
    _, px_loss, px_grads, n, f = svi._compute_per_example_gradients(st, rng.PRNGKey(4), Xt, yt, _eps=torch.tensor(eps).cuda())
    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    eL, eG, _, _ = O.logreg_px_grads(spec, loc, unc, X, y, eps)
    G = np.concatenate([np_(px_grads["auto_loc"]), np_(px_grads["auto_scale"])], axis=1)
    np.testing.assert_allclose(G, eG, rtol=2e-5, atol=2e-6 * np.abs(eG).max())
    np.testing.assert_allclose(np_(px_loss), eL, rtol=2e-5)
    assert np.isfinite(G).all() and np.abs(G[:, d:]).min() > 0          # (the entropy gradient -1/obs * ds/s is never 0 or NaN)
    got = float(svi.evaluate(st, Xt, yt))
    want = O.logreg_evaluate(O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=1.0), loc, unc, X, y,
                             O.convert_to_jax_rng_key(O.split(O.PRNGKey(3), 1)[0]))
    assert abs(got - want) <= 2e-5 * abs(want)
    new, loss = svi.update(st, Xt, yt)
    ost = O.LogregState(O.PRNGKey(3), d, loc, unc)
    eloss, _ = O.logreg_update(spec, O.Hyper(1e6, 0.0, 1e-3, 0.9, 0.999, 1e-8), ost, X, y)
    assert abs(float(loss) - eloss) <= 2e-5 * abs(eloss)
    np.testing.assert_allclose(np_(new.optim_state[1]), ost.params, rtol=1e-4, atol=1e-6)
