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
