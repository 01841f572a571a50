"""d3p.optimizers.ADADP on the device: the reference's tests/test_adadp_optimizer.py re-expressed against
d3p_amd.optimizers.ADADP (tree-structured state, numpyro optimiser protocol), the kernels against the CPU oracle on a
long random trajectory, and DPSVI driving it through the stage-wise update."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def np_(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def template():
    return (torch.ones(7, 10), torch.ones(7), (torch.ones(2, 7), torch.ones(2)))


def with_value(v):
    return (torch.full((7, 10), v), torch.full((7,), v), (torch.full((2, 7), v), torch.full((2,), v)))


def leaves(t):
    from d3p_amd.svi import _tree_flatten
    return _tree_flatten(t)[0]


def tree_close(expected, actual):
    le, la = leaves(expected), leaves(actual)
    assert len(le) == len(la)
    for e, a in zip(le, la):
        assert tuple(e.shape) == tuple(a.shape)
        np.testing.assert_allclose(np_(a), np_(e), rtol=1e-6, atol=1e-7)


def test_init(gpu):
    from d3p_amd.optimizers import ADADP
    adadp = ADADP(1., 1.)
    value = template()
    i, (x, lr, x_stepped, x_prev) = adadp.init(value)
    assert i == 0 and lr == 1.
    tree_close(value, x)
    tree_close(with_value(0.), x_stepped)
    tree_close(value, x_prev)
    tree_close(value, adadp.get_params((i, (x, lr, x_stepped, x_prev))))


def test_update_step_1(gpu):
    from d3p_amd.optimizers import ADADP
    adadp = ADADP(1., 1.)
    value = with_value(0.)
    i, (x, lr, x_stepped, x_prev) = adadp.update(with_value(1.), (0, (value, 1., value, value)))
    assert i == 1 and float(lr) == 1.
    tree_close(with_value(-0.5), x)
    tree_close(with_value(-1.), x_stepped)
    tree_close(value, x_prev)


def test_update_step_2_no_stability_check(gpu):
    from d3p_amd.optimizers import ADADP
    adadp = ADADP(1., tol=5., stability_check=False)
    state = (1, (with_value(-0.5), 1., with_value(-1.), with_value(0.)))
    i, (x, lr, x_stepped, x_prev) = adadp.update(with_value(2.), state)
    assert i == 2
    tree_close(with_value(-1.5), x)
    assert abs(float(lr) - 1.018308251) < 1e-6


def test_update_step_2_with_stability_check(gpu):
    from d3p_amd.optimizers import ADADP
    adadp = ADADP(1., tol=5., stability_check=True)
    state = (1, (with_value(-0.5), 1., with_value(-1.), with_value(0.)))
    i, (x, lr, x_stepped, x_prev) = adadp.update(with_value(3.), state)
    assert i == 2
    tree_close(with_value(0.), x)                   # update rejected
    assert abs(float(lr) - 0.9) < 1e-7              # 0.72005267 clipped by the lower bound


def test_adadp_triple_is_the_class_without_the_step_counter(gpu):
    """d3p/optimizers.py:29-112: `adadp(...)` returns (init_fun, update_fun(i, g, state), get_params) over the bare state
    (x, lr, x_stepped, x_prev); the reference's own two-step known answers (tests/test_adadp_optimizer.py) through it."""
    from d3p_amd.optimizers import adadp
    init, update, get_params = adadp(1., tol=5., stability_check=False)
    x, lr, x_stepped, x_prev = init(with_value(0.))
    assert lr == 1.
    tree_close(with_value(0.), x_stepped)
    state = update(0, with_value(1.), (x, lr, x_stepped, x_prev))
    tree_close(with_value(-0.5), get_params(state))
    tree_close(with_value(-1.), state[2])
    state = update(1, with_value(2.), state)
    tree_close(with_value(-1.5), get_params(state))
    assert abs(float(state[1]) - 1.018308251) < 1e-6


@pytest.mark.parametrize("P,stability", [(1, True), (1000, True), (300000, False)])
def test_trajectory_vs_oracle(gpu, O, P, stability):
    from d3p_amd.optimizers import ADADP
    r = np.random.default_rng(P)
    adadp = ADADP(0.05, tol=0.3, stability_check=stability)
    x0 = (r.normal(size=P) * 2).astype(np.float32)
    state = adadp.init(torch.tensor(x0).cuda())
    ox, olr, oxs, oxp = x0.copy(), 0.05, np.zeros(P, np.float32), x0.copy()
    rejected = 0
    for i in range(12):
        g = (r.normal(size=P) * (3.0 if i % 5 == 3 else 0.5)).astype(np.float32)
        state = adadp.update(torch.tensor(g).cuda(), state)
        before = ox.copy()
        ox, olr, oxs, oxp = O.adadp(ox, olr, oxs, oxp, g, i, tol=0.3, stability_check=stability)
        rejected += int(i % 2 == 1 and np.array_equal(ox, oxp) and not np.array_equal(before, ox))
        assert state[0] == i + 1
        np.testing.assert_allclose(np_(state[1][0]), ox, rtol=1e-6, atol=1e-7)
        assert abs(float(state[1][1]) - olr) <= 2e-6 * olr
        np.testing.assert_allclose(np_(state[1][2]), oxs, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(np_(state[1][3]), oxp, rtol=1e-6, atol=1e-7)   # copies of x: same 1-ulp fma differences
    if stability and P > 1:
        assert rejected > 0        # the trajectory exercises the rejection branch


def test_dpsvi_with_adadp(gpu, O):
    """DPSVI accepts the optimiser (stage-wise path): one even and one odd step against the oracle's stages."""
    import d3p_amd.random as rng
    from d3p_amd.models import AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.optimizers import ADADP
    from d3p_amd.svi import DPSVI, DPSVIState
    B, d, N = 24, 16, 500
    r = np.random.default_rng(0)
    X = r.normal(size=(B, d)).astype(np.float32)
    y = (r.random(B) < 0.5).astype(np.float32)
    model = LogisticRegression(d)
    svi = DPSVI(model, AutoDiagonalNormal(model), ADADP(1e-2, tol=1.0), Trace_ELBO(), 1.0, 0.5, N=N)
    p0 = np.concatenate([np.zeros(d), np.full(d, -2.0)]).astype(np.float32)
    st = DPSVIState(svi.optim.init(torch.tensor(p0).cuda()), rng.PRNGKey(1), float(N))
    spec = O.logreg_spec(d, False, 1.0, 1.0, lik_scale=N, obs_scale=N)
    key = O.PRNGKey(1)
    ox, olr, oxs, oxp = p0.copy(), 1e-2, np.zeros(2 * d, np.float32), p0.copy()
    for i in range(2):
        st, loss = svi.update(st, torch.tensor(X).cuda(), torch.tensor(y).cuda())
        ks = O.split(key, 3)
        key = ks[0]
        eps = O.px_eps(O.convert_to_jax_rng_key(ks[1]), B, d)
        L, G, n, f = O.logreg_px_grads(spec, ox[:d], ox[d:], X, y, eps)
        eloss, avg = O.combine(O.clip_rows(G, 1.0), L)
        g = O.perturb(ks[2], avg, [d, d], 0.5, 1.0, n, N, f)
        ox, olr, oxs, oxp = O.adadp(ox, olr, oxs, oxp, g, i, tol=1.0)
        assert abs(float(loss) - eloss) <= 2e-5 * abs(eloss)
        np.testing.assert_allclose(np_(svi.get_params(st)["auto_loc"]), ox[:d], rtol=1e-4, atol=1e-5)
    assert np.array_equal(np_(st.rng_key), key)
