"""Host-side logic and the C-ABI boundary, without a GPU: the shared library loads, exports every
symbol include/d3p_hip.h declares, the Python surface validates arguments like the reference, and
compute entry points fail loudly when no device is present (there is no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_GPU = not torch.cuda.is_available()


def test_library_loads_and_exports_every_declared_symbol():
    import d3p_amd._lib as L
    L.build()
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "d3p_hip.h")).read()
    declared = set(re.findall(r"\b(d3p_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    raw = ctypes.CDLL(L._SO)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/d3p_hip.h but not exported"
    assert declared == set(L.SIGNATURES), "ctypes SIGNATURES out of sync with the header"
    assert lib.d3p_abi_version() == 9
    assert isinstance(lib.d3p_device_count(), int)


def test_struct_layouts_match_the_header():
    import d3p_amd._lib as L
    assert ctypes.sizeof(L.LogregModel) == 36 and L.LogregModel.family.offset == 24
    assert ctypes.sizeof(L.DpsviHyper) == 24
    assert ctypes.sizeof(L.DpsviState) == 48 and L.DpsviState.params.offset == 16
    assert ctypes.sizeof(L.BatchSource) == 64 and L.BatchSource.batch_key.offset == 16
    assert L.BatchSource.n_rows.offset == 40


def test_error_codes_without_touching_the_device():
    import d3p_amd._lib as L
    lib = L.load()
    assert lib.d3p_rng_split(None, None, 2, None) == -1
    assert b"null pointer" in lib.d3p_last_error()
    assert lib.d3p_clip_rows(None, None, 0, 0, 0.0) == -1          # svi.py:119-120
    assert b"clipping threshold" in lib.d3p_last_error()
    with pytest.raises(ValueError):
        L.check(-1)
    with pytest.raises(L.D3PError):
        L.check(-2)
    assert lib.d3p_poisson_select_workspace(10**6) > 10**6 // 8
    m = L.LogregModel(512, 0, 1.0, 1.0, 1e6, 1e-6)
    src = L.BatchSource(L.D3P_BATCH_FEISTEL, 4096, 0.0, 0, None, None, None, 10**6, 0, 10**6)
    ws = lib.d3p_dpvi_logreg_workspace(ctypes.byref(m), ctypes.byref(src))
    assert 8 * 2**20 < ws < 64 * 2**20


@pytest.mark.skipif(not NO_GPU, reason="checks the no-device behaviour")
def test_compute_fails_loudly_without_a_gpu():
    import d3p_amd._lib as L
    import d3p_amd.random as rng
    from d3p_amd.util import sample_from_array
    with pytest.raises(L.D3PError, match="no CPU fallback"):
        rng.PRNGKey(0)
    with pytest.raises(L.D3PError):
        sample_from_array(torch.zeros(4, 4), torch.zeros(10), 3)
    with pytest.raises(L.D3PError):
        from d3p_amd.svi import full_norm
        full_norm([torch.ones(3)])


def test_product_package_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "d3p_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "libd3p_oracle" not in text, f


def test_seed_packing_matches_oracle_layout(O):
    import d3p_amd.random as rng
    for seed in (0, 7, 2**255 + 3, b"xyz", [1, 2, 3]):
        assert np.array_equal(rng._state_words(seed).reshape(4, 4), O.PRNGKey(seed))
    with pytest.raises(ValueError):
        rng._state_words(bytes(40))


def test_batchifier_argument_validation_and_counts():
    """reference d3p/minibatch.py:75-91, :169-179, :192 and tests/test_minibatch.py:117-135, :242-250"""
    from d3p_amd.minibatch import (batch_size_to_q, poisson_batchify_data, q_to_batch_size,
                                   split_batchify_data, subsample_batchify_data)
    data = (torch.zeros(105, 3), torch.zeros(105))
    for factory in (subsample_batchify_data, split_batchify_data):
        with pytest.raises(ValueError):
            factory(data)
        with pytest.raises(ValueError):
            factory(data, batch_size=10, q=0.1)
        with pytest.raises(ValueError):
            factory(())
        with pytest.raises(ValueError):
            factory((torch.zeros(5, 2), torch.zeros(6)), batch_size=2)
    init, _ = subsample_batchify_data(data, batch_size=10)
    key = object()
    assert init(key) == (10, key)                       # N // B, key returned unchanged
    init, _ = subsample_batchify_data(data, q=0.1)
    assert init(key)[0] == 10
    with pytest.raises(ValueError):
        poisson_batchify_data((), 0.1, 10)
    with pytest.raises(ValueError):
        poisson_batchify_data([torch.zeros(5)], 0.1, 10)          # must be a tuple
    with pytest.raises(ValueError):
        poisson_batchify_data(data, 1.5, 10)
    with pytest.raises(ValueError):
        poisson_batchify_data(data, 0.1, -1)
    init, get_batch = poisson_batchify_data((torch.zeros(100, 2),), 0.1, 100)
    assert init(key) == (10, key)
    with pytest.raises(ZeroDivisionError):
        poisson_batchify_data((torch.zeros(5, 2),), 0.1, 5)[0](key)   # int(q * N) == 0, as the reference
    _, gb = poisson_batchify_data((torch.zeros(105, 1),), 0.3, 0.9)
    assert gb.source.batch_size == 39                   # tests/test_minibatch.py:341-351
    assert q_to_batch_size(0.1, 105) == 10 and batch_size_to_q(10, 100) == 0.1


def test_dpsvi_constructor_and_accounting_validation():
    from d3p_amd.models import SGD, AutoDiagonalNormal, LogisticRegression
    from d3p_amd.svi import DPSVI, DPSVIState, clip_gradient, full_norm
    with pytest.raises(ValueError):
        DPSVI(None, None, SGD(1.0), None, float("inf"), 1.0)
    with pytest.raises(ValueError):
        DPSVI(None, None, SGD(1.0), None, float("nan"), 1.0)
    svi = DPSVI(None, None, SGD(1.0), None, 1.0, 1.0, num_obs_total=100)
    with pytest.raises(ValueError):
        svi.get_epsilon(1e-5, 0.01)
    assert svi._validate_epochs_and_iter(2, None, 0.01) == 200
    assert full_norm(None) == 0.0 and full_norm([]) == 0.0 and full_norm(()) == 0.0
    with pytest.raises(ValueError):
        clip_gradient([torch.ones(2)], 0.0)
    st = DPSVIState(None, "k", 3.0)
    assert svi._update_state_rng(st, "k2") == DPSVIState(None, "k2", 3.0)
    with pytest.raises(ValueError):
        AutoDiagonalNormal(LogisticRegression(4), init_scale=0.0)
    g = AutoDiagonalNormal(LogisticRegression(4))
    assert abs(np.log1p(np.exp(g.unconstrained_init_scale())) - 0.1) < 1e-12


def test_tree_flatten_order_is_jax_like():
    from d3p_amd.svi import _tree_flatten, _tree_unflatten
    tree = {"b": (1, [2, 3]), "a": 0, "c": None}
    leaves, td = _tree_flatten(tree)
    assert leaves == [0, 1, 2, 3]                        # dict keys sorted, None has no leaves
    assert _tree_unflatten(td, [10, 11, 12, 13]) == {"a": 10, "b": (11, [12, 13]), "c": None}


def test_shard_rows_partition():
    from d3p_amd.dist import shard_rows
    for n, w in [(10**8, 8), (1000003, 8), (7, 8), (100, 1), (5, 2)]:
        parts = [shard_rows(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1


# ---------------------------------------------------------------------------------------------- numpyro adapter
def _site(name, dist, shape, observed=False, plate=None, **params):
    return {"name": name, "dist": dist, "shape": tuple(shape), "event_dim": 1 if len(shape) else 0, "is_observed": observed,
            "scale": None, "plate_sizes": [("batch", plate)] if plate else [], "params": params}


def test_numpyro_adapter_maps_trace_records_to_the_built_families():
    """spec_from_sites is the pure-Python half of the adapter (trace_model, the numpyro-facing half, cannot run here):
    site records written the way numpyro's trace writes them -> model spec + flat layout.  AutoDiagonalNormal packs the
    latent sites in SORTED name order (ravel_pytree of a dict), so the example's 'intercept' precedes 'w' in numpyro's flat
    vector while this build keeps the intercept as the last column: the layout carries the permutation."""
    from d3p_amd._lib import D3PError
    from d3p_amd.models import GaussianMean, GaussianMixtureModel, LogisticRegression
    from d3p_amd.numpyro_adapter import spec_from_sites
    # examples/logistic_regression.py:49-66
    spec, lay, n = spec_from_sites([_site("w", "Normal", (5,), loc=0.0, scale=4.0), _site("intercept", "Normal", (), loc=0.0, scale=4.0),
                                    _site("ys", "Bernoulli", (7,), observed=True, plate=1000)])
    assert isinstance(spec, LogisticRegression) and spec.d == 5 and spec.intercept and spec.prior_scale == 4.0 and n == 1000
    assert [s for s, _ in lay.sites] == ["intercept", "w"] and lay.build_order == ["w", "intercept"] and lay.D == 6
    numpyro_flat = [100, 0, 1, 2, 3, 4]                       # [intercept | w0..w4]
    cols = lay.to_build_order(numpyro_flat)
    assert cols == [0, 1, 2, 3, 4, 100] and lay.from_build_order(cols) == numpyro_flat
    # README.md:89-99 (no intercept)
    spec, lay, n = spec_from_sites([_site("w", "Normal", (3,), loc=0.0, scale=1.0), _site("obs", "Bernoulli", (4,), observed=True, plate=50)])
    assert isinstance(spec, LogisticRegression) and not spec.intercept and lay.to_build == [0, 1, 2] and n == 50
    # examples/simple_gaussian_posterior.py:51-65
    spec, lay, _ = spec_from_sites([_site("mu", "Normal", (4,), loc=0.0, scale=1.0), _site("obs", "Normal", (10, 4), observed=True, plate=1000, scale=0.1)])
    assert isinstance(spec, GaussianMean) and spec.d == 4 and spec.obs_scale == 0.1
    # examples/gaussian_mixture_model.py:51-68
    spec, lay, _ = spec_from_sites([_site("pis", "Dirichlet", (3,)), _site("mus", "Normal", (3, 2), loc=0.0, scale=10.0),
                                    _site("sigs", "InverseGamma", (3, 2)), _site("obs", "GaussianMixture", (9, 2), observed=True, plate=100)])
    assert isinstance(spec, GaussianMixtureModel) and (spec.k, spec.d) == (3, 2) and [s for s, _ in lay.sites] == ["mus", "pis", "sigs"]
    with pytest.raises(D3PError):                             # anything else is refused, naming the sites
        spec_from_sites([_site("a", "Gamma", (2,)), _site("obs", "Poisson", (3,), observed=True, plate=10)])
    with pytest.raises(D3PError):                             # and tracing itself needs numpyro
        from d3p_amd.numpyro_adapter import trace_model
        trace_model(lambda: None)


def test_numpyro_adapter_recognises_the_examples_hand_written_guides():
    """guide_spec_from_sites: the records of a traced hand-written guide (param and sample statements in program order) -> the guide
    spec.  The logistic-regression example's OWN guide (examples/logistic_regression.py:67-86) -> MeanFieldGuide; the one-site guide of
    examples/simple_gaussian_posterior.py:67-82 -> DiagonalNormalGuide; anything else is refused with the reason."""
    from d3p_amd._lib import D3PError
    from d3p_amd.models import DiagonalNormalGuide, GaussianMean, LogisticRegression, MeanFieldGuide
    from d3p_amd.numpyro_adapter import guide_spec_from_sites

    def par(name, shape):
        return {"name": name, "type": "param", "dist": None, "shape": tuple(shape)}

    def smp(name, shape, dist="Normal"):
        return {"name": name, "type": "sample", "dist": dist, "shape": tuple(shape)}
    d = 5
    lr = LogisticRegression(d, intercept=True)
    example = [par("w_loc", (d,)), par("w_std_log", (d,)), smp("w", (d,)), par("intercept_loc", ()), par("intercept_std_log", ()),
               smp("intercept", ())]
    g = guide_spec_from_sites(lr, example)
    assert isinstance(g, MeanFieldGuide) and g.param_names() == ("intercept_loc", "intercept_std_log", "w_loc", "w_std_log")
    assert g.leaf_sizes(d) == [1, 1, d, d] and g.sites(d) == [("w", d), ("intercept", 1)]
    assert MeanFieldGuide.tree_from_kernel(2).tolist() == [2, 5, 0, 1, 3, 4]      # kernel order [w0 w1 b | s0 s1 sb] -> tree order
    with pytest.raises(D3PError, match="program order"):      # intercept sampled first: the sites would take each other's keys
        guide_spec_from_sites(lr, [par("intercept_loc", ()), par("intercept_std_log", ()), smp("intercept", ()), par("w_loc", (d,)),
                                   par("w_std_log", (d,)), smp("w", (d,))])
    with pytest.raises(D3PError, match="parameters"):         # another parametrisation (a softplus scale, say)
        guide_spec_from_sites(lr, [par("w_loc", (d,)), par("w_scale", (d,)), smp("w", (d,)), par("intercept_loc", ()),
                                   par("intercept_std_log", ()), smp("intercept", ())])
    with pytest.raises(D3PError, match="not Normal"):
        guide_spec_from_sites(lr, [par("w_loc", (d,)), par("w_std_log", (d,)), smp("w", (d,), "Laplace"), par("intercept_loc", ()),
                                   par("intercept_std_log", ()), smp("intercept", ())])
    gm = GaussianMean(4)
    g1 = guide_spec_from_sites(gm, [par("mu_loc", (4,)), par("mu_std_log", (4,)), smp("mu", (4,))])
    assert isinstance(g1, DiagonalNormalGuide) and g1.param_names() == ("mu_loc", "mu_std_log")
    with pytest.raises(ValueError):                           # MeanFieldGuide needs more than one site
        MeanFieldGuide(LogisticRegression(3))


# ---- d3p.util shape / type predicates (known answers of the reference's tests/test_util.py:30-327)

def test_util_map_over_secondary_dims_with_sum():
    from d3p_amd import util
    x = torch.tensor([[[.3, .4], [2., 1.], [6., 1.5]], [[1., 2.], [2.4, -1], [3.2, 1.]]])
    expected = torch.tensor([[1.3, 2.4], [4.4, 0], [9.2, 2.5]])
    assert torch.allclose(util.map_over_secondary_dims(torch.sum)(x), expected)


def test_util_predicates():
    from d3p_amd import util
    one = torch.ones(1, 1, 1)
    many = torch.ones(3, 2)
    for arr in (one, many, np.ones((1,)), np.ones((2, 2))):
        assert util.has_shape(arr) and util.is_array(arr)
    for not_arr in (3., 3, None, (1, 2)):
        assert not util.is_array(not_arr)
    assert not util.has_shape(None) and not util.has_shape((1, 2)) and not util.has_shape(2.)
    assert util.is_scalar(one) and util.is_scalar(np.ones((1, 1))) and util.is_scalar(5.) and util.is_scalar(torch.tensor(2))
    assert not util.is_scalar(many) and not util.is_scalar((1, 2)) and not util.is_scalar(None)
    assert util.is_integer(3) and util.is_integer(torch.arange(4)) and util.is_integer(np.arange(4)) and util.is_integer(np.int32(2))
    assert not util.is_integer(3.) and not util.is_integer(torch.ones(3)) and not util.is_integer(np.ones(3))
    assert util.is_int_scalar(3) and util.is_int_scalar(torch.tensor([[4]])) and util.is_int_scalar(np.array([7]))
    assert not util.is_int_scalar(3.) and not util.is_int_scalar(torch.tensor([4.])) and not util.is_int_scalar(torch.arange(3))


def test_util_normalize():
    from d3p_amd import util
    assert abs(float(torch.linalg.vector_norm(util.normalize(torch.arange(7)))) - 1.) < 1e-6
    assert abs(float(util.normalize(8.)) - 1.) < 1e-7
    assert abs(float(np.linalg.norm(util.normalize(np.arange(7)))) - 1.) < 1e-6


def test_util_unvectorize_shape():
    from d3p_amd import util
    a = torch.tensor([[3, 4], [5, 6], [7, 8]])
    assert util.unvectorize_shape(a, 2) == (3, 2)
    assert util.unvectorize_shape(a, 1) == (3, 2)
    assert util.unvectorize_shape(a, 3) == (1, 3, 2)
    assert util.unvectorize_shape(a, 4) == (1, 1, 3, 2)
    assert util.unvectorize_shape(3, 1) == (1,)
    assert util.unvectorize_shape(3, 2) == (1, 1)
    assert util.unvectorize_shape_1d(3) == (1,) and util.unvectorize_shape_2d(a) == (3, 2) and util.unvectorize_shape_3d(a) == (1, 3, 2)
    assert util.example_count(a) == 3 and util.example_count(torch.tensor([1])) == 1 and util.example_count(3.) == 1


def test_get_observations_scale_for_the_declared_models():
    """d3p/svi.py:43-65: the scale of the observed site for a given call -- plate(N, subsample_size = batch) gives N / batch; no total
    count: 1; the VAE's handlers.scale factor multiplies either (examples/vae.py:95-105).  DPSVI.init asks for a one-element batch."""
    import numpy as np
    from d3p_amd.models import GaussianMean, GaussianMixtureModel, LogisticRegression, VAEModel
    from d3p_amd.svi import get_observations_scale
    X, y = np.zeros((8, 3), np.float32), np.zeros(8, np.float32)
    lr = LogisticRegression(3)
    assert get_observations_scale(lr, (X, y), {"N": 1000}, None) == 125.0
    assert get_observations_scale(lr, (X[:1], y[:1]), {"num_obs_total": 1000}, None) == 1000.0
    assert get_observations_scale(lr, (X, y, 40), {}, None) == 5.0
    assert get_observations_scale(lr, (X, y), {}, None) == 1.0
    assert get_observations_scale(GaussianMean(3), (X,), {"num_obs_total": 64}, {"mu_loc": 0}) == 8.0
    assert get_observations_scale(GaussianMixtureModel(2), (X,), {"N": 16}) == 2.0
    vae = VAEModel(z_dim=2, hidden_dim=4, scale=0.5)
    assert get_observations_scale(vae, (X,), {"num_obs_total": 80}) == 5.0
    assert get_observations_scale(vae, (X,), {}) == 0.5


def test_gaussian_mixture_mean_and_variance_as_the_reference_writes_them():
    """d3p/gmm.py:97-103, literally (the weights broadcast against the LAST axis of locs); `component_mean` is the mixture's mean."""
    import numpy as np
    from d3p_amd.gmm import GaussianMixture
    locs = np.array([[1.0, 2.0], [3.0, 5.0]], np.float32)
    scales = np.array([[1.0, 1.0], [2.0, 0.5]], np.float32)
    pis = np.array([0.25, 0.75], np.float32)
    g = GaussianMixture(locs, scales, pis)
    assert abs(float(g.mean) - float((pis * locs).sum())) < 1e-6
    want = pis * (scales ** 2 + locs ** 2) - float((pis * locs).sum()) ** 2
    np.testing.assert_allclose(g.variance.numpy(), want, rtol=1e-6)
    np.testing.assert_allclose(g.component_mean.numpy(), pis @ locs, rtol=1e-6)
    assert g.num_components == 2
