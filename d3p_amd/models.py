"""Declarative stand-ins for the NumPyro objects the reference passes to DPSVI.

numpyro / jax cannot be imported in this build, so "tracing the model once" is
replaced by a small model specification whose flat parameter layout is exactly what tracing the
reference's model + ``numpyro.infer.autoguide.AutoDiagonalNormal`` yields:

    params = {'auto_loc': (D,), 'auto_scale': (D,)}     D = number of latent scalars

Reference workloads covered: README.md:89-99 (logistic regression, no intercept, prior N(0, 4)),
examples/logistic_regression.py:49-66 (with intercept, prior N(0, 1)) and
examples/simple_gaussian_posterior.py:51-81 (Gaussian observations with a latent mean, hand-written
guide with an exp-transformed scale; BASELINE config 1).
"""
import math

import torch


class LogisticRegression:
    """w ~ Normal(0, prior_scale)^d [, intercept ~ Normal(0, intercept_prior_scale)];
    ys ~ Bernoulli(logits = xs @ w + intercept) inside ``plate('batch', N, batch_size)``.

    Call signature of the reference models: ``model(xs, ys, N)`` / ``model(batch_X, batch_y,
    num_obs_total=)``; the total count is taken from the keyword ``N`` or ``num_obs_total``."""

    has_labels = True

    def __init__(self, d=None, prior_scale=1.0, intercept=False, intercept_prior_scale=1.0):
        self.d = d
        self.prior_scale = float(prior_scale)
        self.intercept = bool(intercept)
        self.intercept_prior_scale = float(intercept_prior_scale)

    def latent_dim(self, d):
        return d + (1 if self.intercept else 0)

    def site_names(self):
        return ("w", "intercept") if self.intercept else ("w",)

    @staticmethod
    def num_obs_total(args, kwargs):
        for k in ("N", "num_obs_total"):
            if kwargs.get(k) is not None:
                return float(kwargs[k])
        if len(args) >= 3 and args[2] is not None:
            return float(args[2])
        return None


class GaussianMean:
    """mu ~ Normal(0, prior_scale)^d;  obs ~ Normal(mu, obs_scale).to_event(1) inside
    ``plate('batch', num_obs_total, batch_size)`` (examples/simple_gaussian_posterior.py:51-65; the example
    passes its ``x_var = .1`` as the *scale* of the Normal, so ``obs_scale`` is a standard deviation).

    Call signature of the reference model: ``model(obs, num_obs_total=)``; there are no labels."""

    has_labels = False

    def __init__(self, d=None, prior_scale=1.0, obs_scale=0.1):
        self.d = d
        self.prior_scale = float(prior_scale)
        self.obs_scale = float(obs_scale)
        self.intercept = False
        self.intercept_prior_scale = float(prior_scale)

    def latent_dim(self, d):
        return d

    def site_names(self):
        return ("mu",)

    @staticmethod
    def num_obs_total(args, kwargs):
        for k in ("N", "num_obs_total"):
            if kwargs.get(k) is not None:
                return float(kwargs[k])
        if len(args) >= 2 and args[1] is not None and not hasattr(args[1], "shape"):
            return float(args[1])
        return None

    @staticmethod
    def analytical_solution(obs, prior_scale=1.0, obs_scale=0.1):
        """Exact posterior of mu: (loc (d,), std scalar).  Conjugate normal-normal update; the reference's
        helper (examples/simple_gaussian_posterior.py:84-92) is the same formula with its x_var read as a
        variance."""
        N = obs.shape[0]
        var = 1.0 / (N / obs_scale ** 2 + 1.0 / prior_scale ** 2)
        return var * obs.sum(0) / obs_scale ** 2, math.sqrt(var)


class GaussianMixtureModel:
    """pis ~ Dirichlet(ones(k)); mus ~ Normal(0, prior_mu_scale)^(k x d); sigs ~ InverseGamma(1, 1)^(k x d);
    obs ~ GaussianMixture(mus, sigs, pis) inside ``plate('batch', num_obs_total, batch_size)``
    (examples/gaussian_mixture_model.py:51-68; BASELINE config 3).  Call signature of the reference model:
    ``model(k, obs, num_obs_total=)``; ``k`` is passed to DPSVI as a static keyword like in the example."""

    has_labels = False
    family = "gmm"

    def __init__(self, k=None, d=None, prior_mu_scale=10.0):
        self.k = k
        self.d = d
        self.prior_mu_scale = float(prior_mu_scale)

    def site_names(self):
        return ("pis", "mus", "sigs")

    @staticmethod
    def num_obs_total(args, kwargs):
        for key in ("N", "num_obs_total"):
            if kwargs.get(key) is not None:
                return float(kwargs[key])
        return None


class GaussianMixtureGuide:
    """The example's guide (examples/gaussian_mixture_model.py:70-85): pis ~ Dirichlet(exp(alpha_log)),
    mus ~ Normal(mus_loc, 1), sigs ~ InverseGamma(1, 1); params 'alpha_log' (k,) and 'mus_loc' (k, d), zeros at init."""

    def __init__(self, model):
        self.model = model

    def param_names(self):
        return ("alpha_log", "mus_loc")


class VAEModel:
    """The decoder side of examples/vae.py:104-135: z ~ Normal(0, I_z) inside the plate, decoder
    z -> Dense(hidden, softplus) -> Dense(out, sigmoid), obs ~ Bernoulli.  ``scale`` is the factor of an enclosing
    ``numpyro.handlers.scale`` (the example uses 1 / num_samples, vae.py:194-195)."""

    has_labels = False
    family = "vae"

    def __init__(self, z_dim=None, hidden_dim=None, scale=1.0, hidden_dim2=None):
        """``hidden_dim2``: width of a second hidden layer on each side (encoder x -> hidden -> hidden2 -> heads, decoder
        z -> hidden2 -> hidden -> out): BASELINE config 5's 784 -> [400, 200] -> 50 variant.  ``hidden_dim`` may also be a
        pair ``(400, 200)``.  The reference's network has one hidden layer (examples/vae.py:80-85)."""
        if isinstance(hidden_dim, (tuple, list)):
            if len(hidden_dim) not in (1, 2):
                raise ValueError("VAEModel: one or two hidden layers")
            hidden_dim, hidden_dim2 = hidden_dim[0], (hidden_dim[1] if len(hidden_dim) == 2 else hidden_dim2)
        self.z_dim = z_dim
        self.hidden_dim = hidden_dim
        self.hidden_dim2 = int(hidden_dim2) if hidden_dim2 else 0
        self.scale = float(scale)

    @staticmethod
    def num_obs_total(args, kwargs):
        for key in ("N", "num_obs_total"):
            if kwargs.get(key) is not None:
                return float(kwargs[key])
        return None


class VAEGuide:
    """The encoder side of examples/vae.py:138-153: x -> Dense(hidden, softplus) -> (Dense(z), exp(Dense(z))),
    z ~ Normal(z_loc, z_std).  Parameters live in the numpyro.module trees 'decoder$params' / 'encoder$params'."""

    def __init__(self, model):
        self.model = model


class DiagonalNormalGuide:
    """The hand-written mean-field guides of the reference's examples: one sample site
    ``Normal(<site>_loc, exp(<site>_std_log))`` over the model's latent vector
    (examples/simple_gaussian_posterior.py:67-82, 'mu_loc' / 'mu_std_log' initialised to zeros).
    The parameter dict sorts as [<site>_loc, <site>_std_log], the same flat order as AutoDiagonalNormal."""

    transform = "exp"

    def __init__(self, model, site=None, init_loc=0.0, init_std_log=0.0):
        self.model = model
        self.site = site if site is not None else model.site_names()[0]
        self.init_loc = init_loc
        self.init_std_log = float(init_std_log)

    def param_names(self):
        return (self.site + "_loc", self.site + "_std_log")


class MeanFieldGuide:
    """The hand-written mean-field guide of examples/logistic_regression.py:67-86: ONE sample site per latent site of the model, in the
    model's order -- `sample('w', Normal(w_loc, exp(w_std_log)))`, then `sample('intercept', Normal(intercept_loc, exp(intercept_std_log)))`
    -- i.e. four parameter leaves (zeros at init, :77-83).  What differs from a one-site guide over [w, intercept]:
      * the parameter dict flattens in sorted-name order (svi.py:490): intercept_loc (1), intercept_std_log (1), w_loc (d), w_std_log (d);
      * the Gaussian mechanism draws ONE KEY PER LEAF, split(key, 4) (svi.py:487-491);
      * numpyro's seed handler gives every sample site its own key (rng, site_key = split(rng) per sample statement), so 'w' and
        'intercept' take their eps from different threefry streams (d3p_px_eps_sites; this plumbing is UNPINNED).
    The joint density is the same as the one-site guide's, so the per-example gradient kernels are shared; DPSVI.update runs this guide
    around the fused clipped sums (DPSVI._update_leaves; the five-stage composition, _update_staged, is its check)."""

    transform = "exp"

    def __init__(self, model):
        if not getattr(model, "intercept", False):
            raise ValueError("MeanFieldGuide: the model must have more than one latent site (LogisticRegression(intercept=True)); "
                             "use DiagonalNormalGuide for a one-site model")
        self.model = model

    def sites(self, d):
        """[(site name, size)] in the guide's program order."""
        return [("w", int(d)), ("intercept", 1)]

    def param_names(self):
        return ("intercept_loc", "intercept_std_log", "w_loc", "w_std_log")     # tree_flatten order of the parameter dict

    def leaf_sizes(self, d):
        return [1, 1, int(d), int(d)]

    @staticmethod
    def tree_from_kernel(d, device=None):
        """Index tensor: tree-ordered flat vector = kernel-ordered [w_loc, intercept_loc | w_std_log, intercept_std_log][index]."""
        D = d + 1
        return torch.cat([torch.tensor([d, D + d]), torch.arange(d), D + torch.arange(d)]).to(device=device, dtype=torch.long)


class init_to_uniform:
    """numpyro.infer.init_to_uniform(radius=2): auto_loc ~ U(-radius, radius).  The draw uses the
    threefry key of ``DPSVI.init`` (numpyro's exact key plumbing is unpinned)."""

    def __init__(self, radius=2.0):
        self.radius = float(radius)


class init_to_value:
    def __init__(self, values):
        self.values = values


class AutoDiagonalNormal:
    """numpyro.infer.autoguide.AutoDiagonalNormal: q(z) = Normal(auto_loc, auto_scale), with
    auto_scale constrained by softplus (unconstrained value optimised), init_scale 0.1."""

    def __init__(self, model, init_loc_fn=None, init_scale=0.1):
        if init_scale <= 0:
            raise ValueError("Expected init_scale > 0. but got {}".format(init_scale))
        self.model = model
        self.init_loc_fn = init_loc_fn if init_loc_fn is not None else init_to_uniform()
        self.init_scale = float(init_scale)

    transform = "softplus"

    def param_names(self):
        return ("auto_loc", "auto_scale")

    def unconstrained_init_scale(self):
        # softplus^-1(init_scale)
        return math.log(math.expm1(self.init_scale))


class Trace_ELBO:
    """Marker for numpyro.infer.Trace_ELBO (num_particles = 1)."""

    def __init__(self, num_particles=1):
        if num_particles != 1:
            raise NotImplementedError("only num_particles=1 is supported")
        self.num_particles = 1


def _flat_params(params):
    """The optimisers' flat parameter vector: float32 (jax without x64 holds float32 whatever it is given; the kernels read float32),
    on the device it is on -- numpy arrays and lists go to the current GPU."""
    if not isinstance(params, torch.Tensor):
        import numpy as np
        params = torch.as_tensor(np.asarray(params, dtype=np.float32)).cuda()
    return params if params.dtype == torch.float32 else params.to(torch.float32)


class Adam:
    """numpyro.optim.Adam(step_size, b1=0.9, b2=0.999, eps=1e-8)."""

    def __init__(self, step_size, b1=0.9, b2=0.999, eps=1e-8):
        self.step_size, self.b1, self.b2, self.eps = float(step_size), float(b1), float(b2), float(eps)

    def init(self, params: torch.Tensor):
        params = _flat_params(params)
        dev = params.device
        return (torch.zeros((), dtype=torch.int32, device=dev), params,
                torch.zeros_like(params), torch.zeros_like(params))

    def get_params(self, optim_state):
        return optim_state[1]


class SGD:
    """numpyro.optim.SGD(step_size) (used by the reference's tests, tests/test_dpsvi.py:57)."""

    def __init__(self, step_size):
        self.step_size = float(step_size)

    def init(self, params):
        params = _flat_params(params)
        return (torch.zeros((), dtype=torch.int32, device=params.device), params)

    def get_params(self, optim_state):
        return optim_state[1]
