"""Fourier accountant for the subsampled Gaussian mechanism (host side, numpy).

The reference imports ``get_epsilon_R / get_epsilon_S`` (d3p/dputil.py:17, d3p/svi.py:31) and ``get_delta_R``
(d3p/svi.py:32) from the third-party package ``fourier-accountant >= 0.12, < 1`` (setup.py:48), which is not
vendored and not installed here.  This module restates the published algorithm -- Koskela, Jaelkoe, Honkela,
"Computing Tight Differential Privacy Guarantees Using FFT", AISTATS 2020, Algorithm 1 with the privacy loss
distributions of sections 5.1 (remove/add neighbours, suffix ``_R``) and 5.2 (substitution, suffix ``_S``):

1. discretise the privacy loss distribution (PLD) ``omega`` of one mechanism invocation on the grid
   ``x_i = -L + i dx``, ``dx = 2 L / nx``;
2. the PLD of ``ncomp`` compositions is the ``ncomp``-fold convolution: swap the halves of the grid vector, FFT, raise to
   the power ``ncomp``, inverse FFT, swap back;
3. ``delta(eps) = sum_{x_i > eps} (1 - exp(eps - x_i)) C_i``; ``eps(delta)`` by Newton's method on that function.

Same call signatures and failure behaviour as the package (``ValueError`` when the result leaves the window ``[-L, L]`` or is
not finite -- d3p/dputil.py:62-63 relies on that, and tests/test_dputil.py:33-34 pins one failing case).
PARITY: the package itself cannot be run here; the implementation is pinned against independent truth instead
(tests/test_accountant.py): the analytic Gaussian mechanism for q = 1 with composition, and direct quadrature of
``int max(0, f_X - e^eps f_Y)`` for one composition of the subsampled mechanisms.
"""
import numpy as np

__all__ = ["get_delta_R", "get_delta_S", "get_epsilon_R", "get_epsilon_S"]


def _grid(nx, L):
    nx = int(nx)
    if nx < 4 or nx % 2:
        raise ValueError("nx must be an even integer >= 4")
    if not L > 0:
        raise ValueError("L must be positive")
    dx = 2.0 * L / nx
    return nx, dx, -L + dx * np.arange(nx)


def _mixture_density(t, sigma, q):
    """density of q N(1, sigma^2) + (1 - q) N(0, sigma^2) at t"""
    norm = 1.0 / np.sqrt(2.0 * np.pi * sigma * sigma)
    return norm * ((1.0 - q) * np.exp(-t * t / (2.0 * sigma * sigma)) + q * np.exp(-(t - 1.0) ** 2 / (2.0 * sigma * sigma)))


def _pld_remove(x, sigma, q):
    """omega(s) for f_X = q N(1) + (1-q) N(0) against f_Y = N(0) (section 5.1): the loss
    s(t) = log(q exp((2t - 1) / (2 sigma^2)) + 1 - q) is increasing, defined for s > log(1 - q)."""
    w = np.zeros_like(x)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        ok = x > np.log1p(-q) if q < 1.0 else np.ones_like(x, dtype=bool)
        es = np.exp(x[ok])
        arg = (es - (1.0 - q)) / q
        t = sigma * sigma * np.log(arg) + 0.5
        dt = sigma * sigma * es / (es - (1.0 - q))
        w[ok] = _mixture_density(t, sigma, q) * dt
    return w


def _pld_substitute(x, sigma, q):
    """omega(s) for f_X = q N(1) + (1-q) N(0) against f_Y = q N(-1) + (1-q) N(0) (section 5.2).  With
    c = q exp(-1 / (2 sigma^2)) and u = exp(t / sigma^2) the loss satisfies c u^2 + (1-q)(1 - e^s) u - c e^s = 0."""
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        c = q * np.exp(-1.0 / (2.0 * sigma * sigma))
        es = np.exp(x)
        a = (1.0 - q) * (1.0 - es)
        disc = np.sqrt(a * a + 4.0 * c * c * es)
        u = (disc - a) / (2.0 * c)
        # for a > 0 the difference cancels: use the conjugate form u = 2 c e^s / (disc + a)
        u = np.where(a > 0, 2.0 * c * es / (disc + a), u)
        da = -(1.0 - q) * es
        ddisc = (a * da + 2.0 * c * c * es) / disc
        du = np.where(a > 0,
                      (2.0 * c * es * (disc + a) - 2.0 * c * es * (ddisc + da)) / (disc + a) ** 2,
                      (ddisc - da) / (2.0 * c))
        t = sigma * sigma * np.log(u)
        dt = sigma * sigma * du / u
        w = _mixture_density(t, sigma, q) * dt
    return w


def _composed_masses(pld, sigma, q, ncomp, nx, L):
    """probability masses C_i of the ncomp-fold composed PLD on the grid (steps 1 and 2 of the module docstring)"""
    if not (sigma > 0 and 0 < q <= 1):
        raise ValueError("sigma must be positive and q in (0, 1]")
    ncomp = int(ncomp)
    if ncomp < 1:
        raise ValueError("ncomp must be >= 1")
    nx, dx, x = _grid(nx, L)
    w = pld(x, sigma, q) * dx
    if not np.all(np.isfinite(w)):
        raise ValueError("the privacy loss distribution could not be evaluated on the grid (try a larger sigma or L)")
    half = nx // 2
    swapped = np.concatenate([w[half:], w[:half]])
    conv = np.fft.ifft(np.fft.fft(swapped) ** ncomp)
    conv = np.concatenate([conv[half:], conv[:half]])
    masses = np.real(conv)
    if not np.all(np.isfinite(masses)):
        raise ValueError("numerical failure in the FFT composition (try a larger sigma or L)")
    return x, dx, masses


def _delta_at(eps, x, masses):
    """delta(eps) and its derivative with respect to eps"""
    first = int(np.searchsorted(x, eps, side="right"))
    tail_x, tail_m = x[first:], masses[first:]
    e = np.exp(eps - tail_x)
    return float(np.sum((1.0 - e) * tail_m)), float(np.sum(-e * tail_m))


def _get_delta(pld, target_eps, sigma, q, ncomp, nx, L):
    if not abs(target_eps) < L:
        raise ValueError("target_eps must lie inside the window (-L, L)")
    x, _, masses = _composed_masses(pld, sigma, q, ncomp, nx, L)
    delta, _ = _delta_at(float(target_eps), x, masses)
    if not np.isfinite(delta):
        raise ValueError("numerical failure while evaluating delta")
    return delta


def _get_epsilon(pld, target_delta, sigma, q, ncomp, nx, L):
    if not 0 < target_delta < 1:
        raise ValueError("target_delta must lie in (0, 1)")
    x, _, masses = _composed_masses(pld, sigma, q, ncomp, nx, L)
    eps = 0.0
    delta, slope = _delta_at(eps, x, masses)
    for _ in range(200):
        if abs(delta - target_delta) <= 1e-10:
            break
        if not (np.isfinite(delta) and np.isfinite(slope)) or slope == 0.0:
            raise ValueError("numerical failure in the Newton iteration for epsilon")
        eps -= (delta - target_delta) / slope
        if not abs(eps) < L:
            raise ValueError("epsilon left the window [-L, L]: the parameters are outside the accountant's stable range")
        delta, slope = _delta_at(eps, x, masses)
    if not np.isfinite(eps) or abs(delta - target_delta) > 1e-8:
        raise ValueError("the Newton iteration for epsilon did not converge")
    return eps


def get_delta_R(target_eps=1.0, sigma=2.0, q=0.01, ncomp=1e4, nx=1e6, L=20.0):
    """delta(target_eps) after ncomp compositions, remove/add neighbouring relation (d3p/svi.py:467)."""
    return _get_delta(_pld_remove, target_eps, sigma, q, ncomp, nx, L)


def get_delta_S(target_eps=1.0, sigma=2.0, q=0.01, ncomp=1e4, nx=1e6, L=20.0):
    """delta(target_eps) after ncomp compositions, substitution neighbouring relation."""
    return _get_delta(_pld_substitute, target_eps, sigma, q, ncomp, nx, L)


def get_epsilon_R(target_delta=1e-6, sigma=2.0, q=0.01, ncomp=1e4, nx=1e6, L=20.0):
    """epsilon(target_delta) after ncomp compositions, remove/add relation (d3p/svi.py:461, d3p/dputil.py:325)."""
    return _get_epsilon(_pld_remove, target_delta, sigma, q, ncomp, nx, L)


def get_epsilon_S(target_delta=1e-6, sigma=2.0, q=0.01, ncomp=1e4, nx=1e6, L=20.0):
    """epsilon(target_delta) after ncomp compositions, substitution relation (d3p/dputil.py:277)."""
    return _get_epsilon(_pld_substitute, target_delta, sigma, q, ncomp, nx, L)
