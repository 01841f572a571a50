"""Gaussian mixture density: mirror of ``d3p.gmm.GaussianMixture`` (reference d3p/gmm.py:26-107) for the part
that is on the DP-VI path, ``log_prob`` (``d3p_gmm_log_prob``).  The DP-VI step of the mixture MODEL of
examples/gaussian_mixture_model.py (BASELINE config 3) lives in ``d3p_amd/csrc/d3p_gmm.hip`` behind
``d3p_amd.models.GaussianMixtureModel`` / ``DPSVI``.  Ancestral sampling (``sample`` /
``sample_with_intermediates``, d3p/gmm.py:88-95) is off the update path; it takes a ``jax.random`` key in the reference, i.e. a
threefry key here (``d3p_amd.random.debug`` keys), and runs on the threefry stream kernels."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


class GaussianMixture:
    """Mixture of diagonal Gaussians with shared weights (d3p/gmm.py:53-69).

    :param locs: (k, d) component locations.
    :param scales: (k, d) component scales.
    :param mixture_probabilities: (k,) weights; must lie on the simplex when ``validate_args`` is set."""

    def __init__(self, locs, scales, mixture_probabilities, validate_args=None):
        self.locs = torch.as_tensor(locs, dtype=torch.float32)
        self.scales = torch.as_tensor(scales, dtype=torch.float32)
        self.mixture_probabilities = torch.as_tensor(mixture_probabilities, dtype=torch.float32)
        if self.locs.dim() == 1:
            self.locs, self.scales = self.locs.reshape(-1, 1), self.scales.reshape(-1, 1)
        self._validate_args = bool(validate_args)
        if self._validate_args:
            p = self.mixture_probabilities
            if bool((p < 0).any()) or abs(float(p.sum()) - 1.0) > 1e-6:
                raise ValueError("GaussianMixture: mixture_probabilities must lie on the simplex")
            if bool((self.scales <= 0).any()):
                raise ValueError("GaussianMixture: scales must be positive")
        self.event_shape = tuple(self.locs.shape[1:])

    @property
    def mean(self):
        """d3p/gmm.py:97-99, as written there: ``(mixture_probabilities * locs).sum()`` -- the weights broadcast against the LAST axis of
        ``locs`` (k, d), so the expression only evaluates where d == k or d == 1 (and is a scalar, not the mixture's mean vector); kept
        literally for callers that rely on it.  ``component_mean`` is the mixture's mean."""
        return (self.mixture_probabilities * self.locs).sum()

    @property
    def variance(self):
        """d3p/gmm.py:101-103 calls the ``mean`` PROPERTY's value (``self.mean()``), which raises TypeError in the reference whenever it is
        read; here the same expression with the value itself: ``pi * (scales^2 + locs^2) - mean^2`` (same broadcasting caveat as ``mean``)."""
        return (self.mixture_probabilities * (self.scales ** 2 + self.locs ** 2)) - self.mean ** 2

    @property
    def component_mean(self):
        """sum_k pi_k loc_k, shape (d,) (no reference counterpart: what ``mean`` presumably meant)."""
        return (self.mixture_probabilities.reshape(-1, *([1] * (self.locs.dim() - 1))) * self.locs).sum(dim=0)

    @property
    def num_components(self):
        return self.mixture_probabilities.shape[-1]

    def log_prob(self, value):
        """logsumexp_k(log pi_k + sum_d log N(value_d; loc_kd, scale_kd)) (d3p/gmm.py:71-86) on the GPU."""
        _lib.require_device()
        x = torch.as_tensor(value, dtype=torch.float32)
        single = x.dim() == len(self.event_shape)
        k = self.locs.shape[0]
        xs = x.reshape(-1, self.locs[0].numel()).cuda().contiguous()
        locs = self.locs.reshape(k, -1).cuda().contiguous()
        scales = self.scales.reshape(k, -1).cuda().contiguous()
        pis = self.mixture_probabilities.cuda().contiguous()
        out = torch.empty(xs.shape[0], dtype=torch.float32, device=xs.device)
        check(_lib.load().d3p_gmm_log_prob(stream_ptr(), ptr(xs), xs.shape[0], xs.shape[1], ptr(locs), ptr(scales),
                                           ptr(pis), k, ptr(out)))
        batch_shape = tuple(x.shape[:x.dim() - len(self.event_shape)])
        return out[0] if single else out.reshape(batch_shape)

    def sample(self, key, sample_shape=()):
        """d3p/gmm.py:88-89."""
        return self.sample_with_intermediates(key, sample_shape)[0]

    def sample_with_intermediates(self, key, sample_shape=()):
        """d3p/gmm.py:91-95: ``component_key, samples_key = split(key)``; ``zs ~ Categorical(pi)`` of shape ``sample_shape``;
        ``xs = locs[zs] + scales[zs] * normal(samples_key, sample_shape + event_shape)``.  ``key``: a threefry key (2 x uint32 CUDA
        tensor, ``d3p_amd.random.debug.PRNGKey``), as the reference takes a ``jax.random`` key.  The component draw restates numpyro's
        ``CategoricalProbs.sample`` for probabilities: one uniform per draw against the cumulative sums, ``sum(cumsum(p) < u)``
        (numpyro's plumbing: UNPINNED like the rest of it).  Returns ``(xs, (zs,))``."""
        _lib.require_device()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")      # (the debug suite warns on import: sampling from a density is not a privacy mechanism)
            from .random import debug as tf
        shape = tuple(int(s) for s in tuple(sample_shape))
        keys = tf.split(key, 2)
        component_key, samples_key = keys[0], keys[1]
        dev = keys.device
        cum = torch.cumsum(self.mixture_probabilities.to(dev), dim=-1)
        u = tf.uniform(component_key, shape + (1,))
        zs = (cum < u).sum(dim=-1)                                        # in [0, k - 1] (k when u > the last cumulative sum: clamped)
        zs = zs.clamp_(max=self.num_components - 1)
        locs, scales = self.locs.to(dev), self.scales.to(dev)
        eps = tf.normal(samples_key, shape + self.event_shape)
        xs = locs[zs] + scales[zs] * eps
        return xs, (zs,)
