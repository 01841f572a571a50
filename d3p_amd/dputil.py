"""sigma calibration: the noise scale that yields a target epsilon (d3p/dputil.py:149-330).

``approximate_sigma`` (substitution relation) and ``approximate_sigma_remove_relation`` (add/remove relation) search the
sigma whose accountant epsilon hits ``target_eps``: a rough bracket around the target is established first (the
accountant is only reliable in part of the parameter space and signals failure with ``ValueError``), then the bracket is
shrunk by evaluating the model ``sigma = a - b log(eps)`` fitted through both ends, with a midpoint probe whenever one end
has been replaced more than twice in a row.  Same arguments, return tuple ``(sigma, eps, num_evals)`` and stopping rules as
the reference; the accountant is ``d3p_amd.accountant`` (the ``fourier-accountant`` package is not available here).
"""
import numpy as np

from d3p_amd.accountant import get_epsilon_R, get_epsilon_S

__all__ = ["approximate_sigma", "approximate_sigma_remove_relation"]

_NO_BOUNDS = "Could not establish bounds in given evaluation limit"


class _Budget:
    """counts accountant evaluations against maxeval"""

    def __init__(self, fn, maxeval):
        self.fn, self.maxeval, self.used = fn, maxeval, 0

    @property
    def exhausted(self):
        return self.used >= self.maxeval

    def __call__(self, sigma, **kw):
        self.used += 1
        return self.fn(sigma, **kw)


def get_bracketing_bounds(compute_eps_fn, target_eps, maxeval, initial_sigma=1.0):
    """(bounds, bound_eps, num_evals) with bound_eps[0] >= target_eps >= bound_eps[1]  (d3p/dputil.py:24-108)."""
    assert initial_sigma > 0.0 and target_eps > 0 and isinstance(maxeval, int) and maxeval > 0
    ev = compute_eps_fn if isinstance(compute_eps_fn, _Budget) else _Budget(compute_eps_fn, maxeval)

    # a starting sigma at which the accountant answers AND the answer is stable under doubled precision (within 10 %)
    sigma, eps = initial_sigma, None
    while not ev.exhausted:
        try:
            eps = ev(sigma, precision=1.0)
            finer = ev(sigma, precision=2.0)
            if abs(1.0 - eps / finer) <= 0.1:
                break
        except ValueError:
            pass
        sigma *= 10.0
    if ev.exhausted:
        raise RuntimeError(_NO_BOUNDS)

    anchor_sigma, anchor_eps = sigma, eps
    grow = eps >= target_eps  # the anchor is the LOWER sigma bound: walk up by factors of 4; otherwise walk down
    while (eps >= target_eps) if grow else (eps < target_eps):
        sigma = sigma * 4.0 if grow else sigma / 4.0
        while True:
            if ev.exhausted:
                raise RuntimeError(_NO_BOUNDS)
            try:
                eps = ev(sigma)
                break
            except ValueError:
                # step back towards the anchor, where the accountant is known to work
                sigma = 0.9 * 0.5 * (sigma + anchor_sigma) if grow else 1.2 * sigma
                if (grow and sigma <= anchor_sigma) or (not grow and sigma >= anchor_sigma):
                    raise RuntimeError(_NO_BOUNDS)
            finally:
                if ev.exhausted:
                    raise RuntimeError(_NO_BOUNDS)
    if grow:
        return np.array([anchor_sigma, sigma]), np.array([anchor_eps, eps]), ev.used
    return np.array([sigma, anchor_sigma]), np.array([eps, anchor_eps]), ev.used


def update_bounds(sig, eps, target_eps, bounds, bound_eps, consecutive_updates):
    """replaces the lower sigma bound if eps is still above the target, else the upper one (d3p/dputil.py:111-146)."""
    assert bound_eps[1] <= eps <= bound_eps[0]
    side = 0 if eps > target_eps else 1
    bounds[side], bound_eps[side] = sig, eps
    streak = [0, 0]
    streak[side] = consecutive_updates[side] + 1
    return bounds, bound_eps, streak


def _approximate_sigma(compute_eps_fn, target_eps, q, tol=1e-4, force_smaller=False, maxeval=10):
    """d3p/dputil.py:149-234."""
    ev = _Budget(compute_eps_fn, maxeval)
    # for q = 0.01, sigma = 1 lies in the accountant's stable range: scale the first guess with q
    bounds, bound_eps, _ = get_bracketing_bounds(ev, target_eps, maxeval, initial_sigma=q / 0.01)
    sigma, eps = bounds[1], bound_eps[1]
    streak = [0, 0]

    def probe(s):
        nonlocal sigma, eps, bounds, bound_eps, streak
        sigma, eps = s, ev(s)
        bounds, bound_eps, streak = update_bounds(s, eps, target_eps, bounds, bound_eps, streak)

    while abs(target_eps - eps) > tol and not ev.exhausted:
        assert bound_eps[0] >= target_eps >= bound_eps[1]
        # sigma = a - b log(eps) through both ends of the bracket (shape found empirically for this accountant)
        b = (bounds[1] - bounds[0]) / (np.log(bound_eps[0]) - np.log(bound_eps[1]))
        a = np.mean(bounds + b * np.log(bound_eps))
        guess = a - b * np.log(target_eps)
        assert bounds[0] <= guess <= bounds[1]
        probe(guess)
        if not ev.exhausted and max(streak) > 2:  # the neglected end gets a midpoint probe
            probe(np.mean(bounds))

    if force_smaller and eps > target_eps:
        below = bound_eps < target_eps
        sigma, eps = bounds[below][0], bound_eps[below][0]
    assert not force_smaller or eps < target_eps
    return sigma, eps, ev.used


def _calibrate(accountant_eps, target_eps, delta, q, num_iter, tol, force_smaller, maxeval):
    window = max(20, target_eps * 2)

    def compute_eps(sigma, precision=1.0):
        L = window * precision
        return accountant_eps(delta, sigma, q, ncomp=num_iter, L=L, nx=1e6 * L / 20)

    return _approximate_sigma(compute_eps, target_eps, q, tol, force_smaller, maxeval)


def approximate_sigma(target_eps, delta, q, num_iter, tol=1e-4, force_smaller=False, maxeval=10):
    """sigma for (target_eps, delta) after num_iter iterations at sampling ratio q, substitution relation
    (d3p/dputil.py:237-282).  Returns (sigma, reached epsilon, accountant evaluations)."""
    return _calibrate(get_epsilon_S, target_eps, delta, q, num_iter, tol, force_smaller, maxeval)


def approximate_sigma_remove_relation(target_eps, delta, q, num_iter, tol=1e-4, force_smaller=False, maxeval=10):
    """the same for the add/remove relation (d3p/dputil.py:285-330)."""
    return _calibrate(get_epsilon_R, target_eps, delta, q, num_iter, tol, force_smaller, maxeval)
