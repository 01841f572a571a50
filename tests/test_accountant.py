"""CPU tests of the privacy calibration rows: the restated Fourier accountant (d3p_amd.accountant, stands in
for the absent `fourier-accountant` package) pinned against independent truth, and d3p_amd.dputil (d3p/dputil.py) on
top of it, including the reference's own test case (tests/test_dputil.py:26-44)."""
import numpy as np
import pytest
from scipy.integrate import quad
from scipy.stats import norm

import d3p_amd.accountant as A
from d3p_amd.dputil import approximate_sigma, approximate_sigma_remove_relation, get_bracketing_bounds


def gaussian_mechanism_delta(eps, sigma, sensitivity=1.0):
    """analytic Gaussian mechanism (Balle & Wang 2018, Theorem 8)"""
    s = sigma / sensitivity
    return norm.cdf(0.5 / s - eps * s) - np.exp(eps) * norm.cdf(-0.5 / s - eps * s)


@pytest.mark.parametrize("k,sigma,eps", [(1, 2.0, 0.5), (10, 3.0, 1.0), (1000, 30.0, 1.0)])
def test_without_subsampling_the_composition_is_the_analytic_gaussian_mechanism(k, sigma, eps):
    # k compositions of N(., sigma^2) = one Gaussian mechanism with sigma / sqrt(k); substitution doubles the sensitivity
    ref = gaussian_mechanism_delta(eps, sigma / np.sqrt(k))
    assert abs(A.get_delta_R(eps, sigma, 1.0, k, nx=4e5) - ref) <= 1e-7 * ref
    ref2 = gaussian_mechanism_delta(eps, sigma / np.sqrt(k), sensitivity=2.0)
    assert abs(A.get_delta_S(eps, sigma, 1.0, k, nx=4e5) - ref2) <= 1e-7 * ref2


@pytest.mark.parametrize("q,sigma,eps", [(0.01, 1.0, 0.3), (0.1, 0.8, 1.0), (0.5, 2.0, 0.2)])
def test_one_composition_matches_direct_quadrature_of_the_privacy_profile(q, sigma, eps):
    def phi(t, m):
        return np.exp(-(t - m) ** 2 / (2 * sigma * sigma)) / np.sqrt(2 * np.pi * sigma * sigma)

    def f_x(t):
        return q * phi(t, 1.0) + (1 - q) * phi(t, 0.0)

    ref_r = quad(lambda t: max(0.0, f_x(t) - np.exp(eps) * phi(t, 0.0)), -30, 30, limit=500, epsabs=1e-14)[0]
    ref_s = quad(lambda t: max(0.0, f_x(t) - np.exp(eps) * (q * phi(t, -1.0) + (1 - q) * phi(t, 0.0))), -30, 30, limit=500,
                 epsabs=1e-14)[0]
    assert abs(A.get_delta_R(eps, sigma, q, 1, nx=1e6) - ref_r) <= 1e-6 * ref_r
    assert abs(A.get_delta_S(eps, sigma, q, 1, nx=1e6) - ref_s) <= 1e-6 * ref_s


def test_epsilon_inverts_delta_and_is_monotone_in_sigma():
    for get_eps, get_delta in ((A.get_epsilon_R, A.get_delta_R), (A.get_epsilon_S, A.get_delta_S)):
        eps = [get_eps(1e-5, s, 0.01, 1000, nx=4e5) for s in (0.9, 1.2, 2.0)]
        assert eps[0] > eps[1] > eps[2] > 0
        assert abs(get_delta(eps[1], 1.2, 0.01, 1000, nx=4e5) - 1e-5) < 1e-9
    assert A.get_epsilon_S(1e-5, 1.2, 0.01, 1000, nx=4e5) > A.get_epsilon_R(1e-5, 1.2, 0.01, 1000, nx=4e5)


def test_bad_arguments_and_the_unstable_range_raise_value_error():
    with pytest.raises(ValueError):
        A.get_epsilon_R(1e-5, -1.0, 0.01, 10)
    with pytest.raises(ValueError):
        A.get_delta_R(25.0, 1.0, 0.01, 10, L=20.0)
    with pytest.raises(ValueError):  # the failing query of tests/test_dputil.py:33-34
        A.get_epsilon_R(1e-5, 0.625, 0.001, 100000, nx=int(1e6))


def test_approximate_sigma_recovers_accountant_failure():
    """tests/test_dputil.py:26-44"""
    sigma, reached_eps, num_evals = approximate_sigma_remove_relation(1.0, 1e-5, 0.001, 100000, maxeval=40, tol=1e-4)
    assert np.allclose(reached_eps, 1.0, atol=1e-4)
    assert np.isfinite(sigma)
    assert num_evals <= 40
    assert abs(A.get_epsilon_R(1e-5, sigma, 0.001, 100000) - reached_eps) < 1e-9


def test_approximate_sigma_options():
    sigma, eps, n = approximate_sigma(1.0, 1e-5, 0.01, 1000, maxeval=20)
    assert abs(eps - 1.0) <= 1e-4 and n <= 20
    assert abs(A.get_epsilon_S(1e-5, sigma, 0.01, 1000) - eps) < 1e-9
    sigma_r, eps_r, _ = approximate_sigma_remove_relation(1.0, 1e-5, 0.01, 1000, maxeval=20, force_smaller=True)
    assert eps_r < 1.0 and sigma_r < sigma  # add/remove needs less noise than substitution
    with pytest.raises(RuntimeError):  # no bracket within two evaluations
        approximate_sigma(1.0, 1e-5, 0.01, 1000, maxeval=2)

    calls = []

    def fake_eps(sig, precision=1.0):  # eps = 4 / sigma: bracket found by walking sigma up by factors of 4
        calls.append(sig)
        return 4.0 / sig

    bounds, bound_eps, n = get_bracketing_bounds(fake_eps, 0.5, 10, initial_sigma=1.0)
    assert list(bounds) == [1.0, 16.0] and list(bound_eps) == [4.0, 0.25] and n == len(calls) == 4


def test_dpsvi_accounting_methods():
    from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.svi import DPSVI
    model = LogisticRegression(4)
    svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.3, num_obs_total=100)
    with pytest.raises(ValueError):  # d3p/svi.py:454-455
        svi.get_epsilon(1e-5, 0.01)
    eps = svi.get_epsilon(1e-5, 0.01, num_iter=500)
    assert abs(eps - A.get_epsilon_R(1e-5, 1.3, 0.01, 500)) < 1e-12
    assert abs(svi.get_delta(eps, 0.01, num_epochs=5) - 1e-5) < 1e-8  # 5 epochs at q = 0.01 = 500 iterations
