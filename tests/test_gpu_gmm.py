"""GaussianMixture.log_prob (d3p/gmm.py:71-86): HIP kernel vs the oracle, the reference's known-answer test
(tests/test_gmm.py:85-107) and scipy."""
import numpy as np
import pytest
import torch
from scipy.special import logsumexp
from scipy.stats import norm

pytestmark = pytest.mark.gpu


def test_log_prob_reference_known_answer(gpu, O):
    from d3p_amd.gmm import GaussianMixture
    locs = np.array([[-5., -5.], [0., 0.], [5., 5.]], np.float32)
    scales = np.ones_like(locs) * 0.1
    pis = np.array([.5, .3, .2], np.float32)
    mix = GaussianMixture(locs, scales, pis)
    x = np.array([[-4, -3], [1, .5]], np.float32)
    log_phis = np.stack([norm(locs[k], scales[k]).logpdf(x).sum(axis=-1) for k in range(3)])
    expected = logsumexp(np.log(pis).reshape(3, 1) + log_phis, axis=0)
    actual = mix.log_prob(torch.tensor(x)).cpu().numpy()
    assert actual.shape == (2,) and np.allclose(expected, actual, rtol=1e-5)
    assert np.allclose(O.gmm_log_prob(x, locs, scales, pis), expected, rtol=1e-5)
    assert mix.num_components == 3
    assert abs(float(mix.log_prob(torch.tensor(x[0]))) - expected[0]) < 1e-3 * abs(expected[0])
    with pytest.raises(ValueError):
        GaussianMixture(locs, scales, np.ones(3), validate_args=True)      # tests/test_gmm.py:27-32


@pytest.mark.parametrize("B,d,K", [(1, 1, 1), (33, 64, 16), (1000, 64, 16), (17, 200, 3), (9, 5, 40)])
def test_log_prob_vs_oracle(gpu, O, B, d, K):
    from d3p_amd.gmm import GaussianMixture
    r = np.random.default_rng(B + d + K)
    locs = r.normal(size=(K, d)).astype(np.float32) * 3
    scales = (0.2 + r.random((K, d))).astype(np.float32)
    pis = r.dirichlet(np.ones(K)).astype(np.float32)
    x = (r.normal(size=(B, d)) * 3).astype(np.float32)
    got = GaussianMixture(locs, scales, pis).log_prob(torch.tensor(x)).cpu().numpy()
    exp = O.gmm_log_prob(x, locs, scales, pis)
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=1e-4)


def test_sample_shapes_and_mixture_statistics_like_the_reference_tests(gpu, O):
    """tests/test_gmm.py:36-83 on the d3p_amd surface (key = a jax.random key there, a threefry key here): sample shapes, components in
    range, component frequencies within 3 standard deviations of pi, per-component means within 3 standard errors -- and the draws
    themselves against the oracle's restatement of d3p/gmm.py:91-95 on the same key (components bit-exact, values rtol 2e-6)."""
    import warnings
    from d3p_amd.gmm import GaussianMixture
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import d3p_amd.random.debug as tf
    locs = np.array([[-5., -5.], [0., 0.], [5., 5.]], np.float32)
    scales = np.ones_like(locs) * 0.1
    mix = GaussianMixture(locs, scales, np.ones(3, np.float32) / 3)
    key = tf.PRNGKey(2963)
    assert tuple(mix.sample(key).shape) == (2,)
    assert tuple(mix.sample(key, sample_shape=(5, 4)).shape) == (5, 4, 2)
    pis = np.array([.5, .3, .2], np.float32)
    mix = GaussianMixture(locs, scales, pis)
    n_total = 1000
    vals, (zs,) = mix.sample_with_intermediates(key, sample_shape=(10, n_total // 10))
    assert tuple(zs.shape) == (10, n_total // 10) and tuple(vals.shape) == (10, n_total // 10, 2)
    zs_h, vals_h = zs.cpu().numpy(), vals.cpu().numpy()
    assert zs_h.min() >= 0 and zs_h.max() < 3
    freq = np.bincount(zs_h.ravel(), minlength=3) / n_total
    assert np.allclose(freq, pis, atol=3 * np.sqrt(pis * (1 - pis) / n_total))
    for i in range(3):
        assert np.allclose(locs[i], vals_h[zs_h == i].mean(axis=0), atol=3 * scales[i] / np.sqrt(freq[i] * n_total) + 1e-3)
    exp_x, exp_z = O.gmm_sample_with_intermediates(np.array([0, 2963], np.uint32), locs, scales, pis, (10, n_total // 10))
    assert np.array_equal(zs_h, exp_z)
    np.testing.assert_allclose(vals_h, exp_x, rtol=2e-6, atol=1e-6)
