"""DP-VI with per-example gradient clipping and Gaussian perturbation on MI355X.

Mirror of ``d3p.svi`` (reference d3p/svi.py): ``DPSVI``, ``DPSVIState``, ``full_norm``,
``clip_gradient``, ``normalize_gradient`` with the same names, argument order, return tuples and
exceptions.  ``DPSVI.update`` runs the fused HIP path (per-example gradient -> clip -> sum ->
noise -> Adam, the B x P gradient tensor is never materialised); the five stage methods the
reference's tests call directly (tests/test_dpsvi.py:128-129, :158-159, :171, :185, :199-202)
are kept and operate on materialised device tensors through their own kernels.
"""
import ctypes as C
from typing import Any, NamedTuple

import numpy as np
import torch

from . import _lib
from . import random as strong_rng
from ._lib import BatchSource, DpsviHyper, DpsviState, GmmModel, LogregModel, VaeModel, check, ptr, stream_ptr
from .optimizers import ADADP
from .models import (SGD, Adam, AutoDiagonalNormal, DiagonalNormalGuide, GaussianMean, GaussianMixtureGuide, MeanFieldGuide,
                     GaussianMixtureModel, LogisticRegression, VAEGuide, VAEModel,
                     init_to_uniform, init_to_value)
from .util import example_count

PRNGState = Any


def _fresh_optim_state(optim_state):
    """Copies of (step, params, m, v) for a functional update (svi.py:395-434 returns a new state): the three float
    vectors are copied by ONE kernel into one buffer (views of it are returned) -- per-call host time matters for
    step-by-step loops."""
    step, params, m, v = optim_state
    n = params.numel()
    if m.numel() == n and v.numel() == n and params.dtype == m.dtype == v.dtype:
        flat = torch.cat((params.reshape(-1), m.reshape(-1), v.reshape(-1)))
        return step.clone(), flat[:n].view_as(params), flat[n:2 * n].view_as(m), flat[2 * n:].view_as(v)
    return step.clone(), params.clone(), m.clone(), v.clone()


class SVIRunResult(NamedTuple):
    """numpyro.infer.svi.SVIRunResult: what ``SVI.run`` returns."""
    params: Any
    state: Any
    losses: Any


class DPSVIState(NamedTuple):
    """d3p/svi.py:37-40."""
    optim_state: Any
    rng_key: PRNGState
    observation_scale: float


# ------------------------------------------------------------------ pytree helpers (jax.tree_util order)
def _tree_flatten(tree):
    if tree is None:
        return [], ("none",)
    if isinstance(tree, dict):
        keys = sorted(tree)
        leaves, defs = [], []
        for k in keys:
            l, d = _tree_flatten(tree[k])
            leaves += l
            defs.append((k, d, len(l)))
        return leaves, ("dict", defs)
    if isinstance(tree, (tuple, list)):
        leaves, defs = [], []
        for x in tree:
            l, d = _tree_flatten(x)
            leaves += l
            defs.append((d, len(l)))
        return leaves, ("tuple" if isinstance(tree, tuple) else "list", defs)
    return [tree], ("leaf",)


def _tree_unflatten(treedef, leaves):
    kind = treedef[0]
    if kind == "none":
        return None
    if kind == "leaf":
        return leaves[0]
    out, pos = [], 0
    if kind == "dict":
        d = {}
        for k, sub, n in treedef[1]:
            d[k] = _tree_unflatten(sub, leaves[pos:pos + n])
            pos += n
        return d
    for sub, n in treedef[1]:
        out.append(_tree_unflatten(sub, leaves[pos:pos + n]))
        pos += n
    return tuple(out) if kind == "tuple" else out


def _as_device_f32(x, device=None):
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.as_tensor(np.asarray(x, dtype=np.float32))
    if not t.is_cuda:
        t = t.cuda() if device is None else t.to(device)
    return t.to(torch.float32)


def _batch_array(x):
    """A batch argument as a contiguous CUDA tensor.  Device tensors pass through; host tensors and numpy arrays are moved to the
    current device, floating types as float32 -- what jax does with a numpy argument at every call (x64 off) -- so that a HOST pointer
    never reaches a kernel (a GPU memory fault, not a Python error)."""
    if isinstance(x, torch.Tensor) and x.is_cuda:
        if x.is_floating_point() and x.dtype != torch.float32:
            x = x.to(torch.float32)        # (the kernels read float32; a float64 / half batch is converted like a host one)
        return x.contiguous()
    _lib.require_device()
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    if t.is_floating_point():
        t = t.to(torch.float32)
    return t.cuda().contiguous()


def _mask_array(mask, device, B):
    """An example mask (tensor, numpy array, list) as a contiguous uint8 tensor on ``device``: one entry per example of the batch."""
    t = mask if isinstance(mask, torch.Tensor) else torch.as_tensor(np.asarray(mask))
    if t.numel() != int(B):
        raise ValueError(f"mask: {t.numel()} entries for a batch of {int(B)} examples")
    return t.to(device=device, dtype=torch.uint8).reshape(-1).contiguous()


# ------------------------------------------------------------------ module-level gradient manipulators
def get_observations_scale(model, model_args, model_kwargs, params=None):
    """d3p/svi.py:43-65 for the declared model families: the scale numpyro applies to the observed site's log-likelihood for THESE
    arguments.  The reference traces the model and reads ``msg['scale']`` of the observed sites; the declared models have exactly one,
    inside ``plate('batch', N, subsample_size=len(batch))``, so the scale is N / len(batch), or 1.0 when the call names no total count
    (either way times the VAE's ``handlers.scale`` factor) -- read from the model's declaration (``num_obs_total``), not from a trace.  ``params`` is
    accepted for the reference's signature (it substitutes them before tracing; the scale does not depend on them).  ``DPSVI.init`` calls
    this on a one-element batch (svi.py:225-234): the scale is then N itself."""
    kwargs = dict(model_kwargs or {})
    n_total = model.num_obs_total(tuple(model_args), kwargs)
    batch = example_count(model_args[0]) if len(model_args) else 1
    scale = 1.0 if n_total is None else float(n_total) / float(max(int(batch), 1))
    if isinstance(model, VAEModel):
        scale *= float(model.scale)
    return scale


def full_norm(vector_parts, ord=2):
    """d3p/svi.py:68-87: norm over all leaves of a tree treated as one vector (0. if empty); `ord` as for
    numpy.linalg.norm of a vector (None = 2, 0, 1, +-inf, any other p)."""
    if ord is None:
        ord = 2
    if isinstance(ord, str):
        raise ValueError(f"Invalid order '{ord}' for vector norm.")   # ('fro' / 'nuc' are matrix norms; numpy raises too)
    leaves, _ = _tree_flatten(vector_parts)
    if len(leaves) == 0:
        return 0.0
    _lib.require_device()
    flat = [_as_device_f32(g).reshape(-1) for g in leaves]
    v = torch.cat(flat) if len(flat) > 1 else flat[0].contiguous()
    if v.numel() == 0:
        return 0.0
    out = torch.empty(1, dtype=torch.float32, device=v.device)
    if ord == 2:
        check(_lib.load().d3p_full_norm(stream_ptr(), ptr(v), v.numel(), ptr(out), None, 0))
    else:
        check(_lib.load().d3p_full_norm_ord(stream_ptr(), ptr(v), v.numel(), float(ord), ptr(out)))
    return out[0]


def _clip_flat_rows(rows: torch.Tensor, c: float) -> torch.Tensor:
    check(_lib.load().d3p_clip_rows(stream_ptr(), ptr(rows), rows.shape[0], rows.shape[1], float(c)))
    return rows


def clip_gradient(gradient_parts, c):
    """d3p/svi.py:106-124: scale every leaf by 1/max(1, ||g||/c); c == 0 -> ValueError."""
    if c == 0.0:
        raise ValueError("The clipping threshold must be greater than 0.")
    leaves, treedef = _tree_flatten(gradient_parts)
    if len(leaves) == 0:
        return gradient_parts
    _lib.require_device()
    flat = [_as_device_f32(g) for g in leaves]
    row = torch.cat([g.reshape(-1) for g in flat]).reshape(1, -1).contiguous()
    if np.isinf(c):
        return _tree_unflatten(treedef, flat)
    _clip_flat_rows(row, c)
    out, pos = [], 0
    for g in flat:
        out.append(row[0, pos:pos + g.numel()].reshape(g.shape))
        pos += g.numel()
    return _tree_unflatten(treedef, out)


def normalize_gradient(gradient_parts, ord=2):
    """d3p/svi.py:90-103."""
    norm = full_norm(gradient_parts, ord=ord)
    leaves, treedef = _tree_flatten(gradient_parts)
    inv = 1.0 / norm
    return _tree_unflatten(treedef, [_as_device_f32(g) * inv for g in leaves])


# ------------------------------------------------------------------ DPSVI
class DPSVI:
    """Differentially-private SVI (d3p/svi.py:127-498) for the model families of d3p_amd.models.

    :param model: a ``d3p_amd.models.LogisticRegression`` specification (stands in for the NumPyro
        model function; numpyro is not importable in this build).
    :param guide: ``d3p_amd.models.AutoDiagonalNormal(model)``.
    :param optim: ``d3p_amd.models.Adam`` / ``SGD``.
    :param per_example_loss: ``d3p_amd.models.Trace_ELBO()``.
    :param clipping_threshold: C, must be finite (ValueError otherwise, svi.py:187-188).
    :param dp_scale: sigma of the Gaussian mechanism.
    :param rng_suite: ``d3p_amd.random`` (default, ChaCha20) or ``d3p_amd.random.debug``.
    :param clip_unscaled_observations: svi.py:159-164.
    :param static_kwargs: constant model arguments, e.g. ``N=`` / ``num_obs_total=``.
    """

    def __init__(self, model, guide, optim, per_example_loss, clipping_threshold, dp_scale,
                 rng_suite=strong_rng, clip_unscaled_observations=True, **static_kwargs):
        self._clipping_threshold = clipping_threshold
        self._dp_scale = dp_scale
        self._rng_suite = rng_suite
        self._clip_unscaled_observations = clip_unscaled_observations
        if not np.isfinite(clipping_threshold):
            raise ValueError("clipping_threshold must be finite!")
        # (the reference only rejects C == 0, at the first clip_gradient call, svi.py:119-120; a negative C would disable
        # clipping there -- this build refuses C <= 0 and a negative / non-finite noise scale up front)
        if clipping_threshold < 0:
            raise ValueError("The clipping threshold must be greater than 0.")
        if not (np.isfinite(dp_scale) and dp_scale >= 0):
            raise ValueError("dp_scale must be finite and >= 0")
        self.model = model
        self.guide = guide
        self.optim = optim
        self.loss = per_example_loss
        self.static_kwargs = static_kwargs
        self._ws = {}

    # ---------------------------------------------------------------- state helpers (svi.py:192-211)
    @staticmethod
    def _update_state_rng(dp_svi_state, rng_key):
        return DPSVIState(dp_svi_state.optim_state, rng_key, dp_svi_state.observation_scale)

    @staticmethod
    def _update_state_optim_state(dp_svi_state, optim_state):
        return DPSVIState(optim_state, dp_svi_state.rng_key, dp_svi_state.observation_scale)

    def _split_rng_key(self, dp_svi_state, count=1):
        split_keys = self._rng_suite.split(dp_svi_state.rng_key, count + 1)
        return DPSVI._update_state_rng(dp_svi_state, split_keys[0]), split_keys[1:]

    # ---------------------------------------------------------------- model plumbing
    # ---------------------------------------------------------------- VAE (BASELINE config 5)
    def _is_vae(self):
        return isinstance(self.model, VAEModel) and isinstance(self.guide, VAEGuide)

    def _vae_struct(self, D, kwargs, observation_scale):
        kw = dict(self.static_kwargs)
        kw.update(kwargs)
        z, h, h2 = self._vae_dims(kw)
        n_total = self.model.num_obs_total((), kw)
        site_scale = (1.0 if n_total is None else n_total) * self.model.scale   # plate(N, 1) x handlers.scale
        return VaeModel(int(D), int(h), int(z), float(site_scale), 1.0 / float(observation_scale), int(h2))

    def _vae_dims(self, kw):
        """(z, hidden, hidden2) from the call's keyword arguments or the model; hidden2 = 0: one hidden layer."""
        z = kw.get("z_dim") or self.model.z_dim
        h = kw.get("hidden_dim") or self.model.hidden_dim
        h2 = kw.get("hidden_dim2") or getattr(self.model, "hidden_dim2", 0) or 0
        if isinstance(h, (tuple, list)):
            h, h2 = h[0], (h[1] if len(h) > 1 else h2)
        if z is None or h is None:
            raise ValueError("VAEModel: z_dim and hidden_dim must be given")
        return int(z), int(h), int(h2)

    @staticmethod
    def _vae_flat(X):
        return X.reshape(X.shape[0], -1).contiguous().to(torch.float32)

    def _vae_tree(self, flat, vm):
        """Flat parameter vector -> the numpyro.module trees of stax.serial (empty tuples for the parameter-free layers)."""
        D, H, Z, H2 = vm.D, vm.H, vm.Z, vm.H2
        hs = [H] + ([H2] if H2 else [])
        dec_dims, enc_dims = [Z] + hs[::-1] + [D], [D] + hs
        shapes = []
        for i, o in zip(dec_dims[:-1], dec_dims[1:]):
            shapes += [(i, o), (o,)]
        for i, o in zip(enc_dims[:-1], enc_dims[1:]):
            shapes += [(i, o), (o,)]
        shapes += [(hs[-1], Z), (Z,), (hs[-1], Z), (Z,)]
        leaves, pos = [], 0
        for shp in shapes:
            n = int(np.prod(shp))
            leaves.append(flat[pos:pos + n].reshape(shp))
            pos += n
        n_dec, n_enc = len(dec_dims) - 1, len(enc_dims) - 1
        dense = [(leaves[2 * k], leaves[2 * k + 1]) for k in range(n_dec + n_enc)]
        (Wl, bl, Ws, bs) = leaves[-4:]
        decoder, encoder = [], []
        for lay in dense[:n_dec]:
            decoder += [lay, ()]            # stax.serial(Dense, Softplus, .., Dense, Sigmoid): parameter-free layers are ()
        for lay in dense[n_dec:]:
            encoder += [lay, ()]
        encoder += [(), ((Wl, bl), ((Ws, bs), ()))]   # FanOut(2), parallel(Dense, serial(Dense, Exp))
        return {"decoder$params": decoder, "encoder$params": encoder}

    def _init_vae(self, rng_key, *args, **kwargs):
        _lib.require_device()
        lib = _lib.load()
        X = self._vae_flat(_batch_array(args[0]))
        vm = self._vae_struct(X.shape[1], kwargs, 1.0)
        P = int(lib.d3p_vae_num_params(C.byref(vm)))
        # stax.Dense(W_init=stax.randn(), b_init=normal()): every leaf ~ N(0, 0.01^2); the reference's key plumbing
        # through numpyro.module / stax init_fun is unpinned, this build draws the flat vector from the init key
        jax_rng_key = self._rng_suite.convert_to_jax_rng_key(rng_key).contiguous()
        params = torch.empty(P, dtype=torch.float32, device=X.device)
        check(lib.d3p_tf_normal(stream_ptr(), ptr(jax_rng_key), P, ptr(params)))
        params.mul_(1e-2)
        observation_scale = 1.0
        if self._clip_unscaled_observations:
            observation_scale = vm.scale        # get_observations_scale: scale of the 'obs' site (svi.py:43-65)
        return DPSVIState(self.optim.init(params), rng_key, observation_scale)

    def _update_vae(self, svi_state, *args, mask=True, _eps=None, _grad_out=None, **kwargs):
        if not (isinstance(self.optim, Adam) and self._rng_suite is strong_rng):
            raise _lib.D3PError("VAE step: needs numpyro-style Adam and rng_suite=d3p_amd.random")
        _lib.require_device()
        lib = _lib.load()
        X = self._vae_flat(_batch_array(args[0]))
        B, D = X.shape
        dev = X.device
        vm = self._vae_struct(D, kwargs, svi_state.observation_scale)
        hyper = self._hyper()
        mask_t = None
        if not isinstance(mask, bool):
            mask_t = _mask_array(mask, X.device, B)
        elif mask is False:
            mask_t = torch.zeros(B, dtype=torch.uint8, device=dev)
        ws = self._workspace(lib.d3p_dpvi_vae_workspace(C.byref(vm), B), dev, "vae_step")
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        keybuf = torch.empty((2, 16), dtype=torch.uint32, device=dev)
        step0, params0, m0, v0 = svi_state.optim_state
        key0 = svi_state.rng_key.reshape(16)
        n = params0.numel()
        self._require_sizes(n, lib.d3p_vae_num_params(C.byref(vm)))
        if (params0.dtype == m0.dtype == v0.dtype == torch.float32 and m0.numel() == n and v0.numel() == n
                and params0.is_contiguous() and m0.is_contiguous() and v0.is_contiguous() and key0.is_contiguous()
                and key0.dtype == torch.uint32 and step0.dtype == torch.int32):
            # out of place: the kernels read the old state and write the new one (no copy of the 3 x P floats)
            flat = torch.empty(3 * n, dtype=torch.float32, device=dev)
            step, params, m, v = torch.empty_like(step0), flat[:n].view_as(params0), flat[n:2 * n].view_as(m0), flat[2 * n:].view_as(v0)
            st = self._state_struct(keybuf, 0, (step, params, m, v))
            frm = self._state_struct(key0, 0, (step0, params0, m0, v0))
            check(lib.d3p_dpvi_vae_update_from(stream_ptr(), C.byref(vm), C.byref(hyper), C.byref(st), C.byref(frm), ptr(X),
                                               ptr(mask_t), B, ptr(_eps), ptr(loss), ptr(_grad_out), ptr(ws), ws.numel()))
        else:
            step, params, m, v = _fresh_optim_state(svi_state.optim_state)
            keybuf[0].copy_(key0)
            st = self._state_struct(keybuf, 0, (step, params, m, v))
            check(lib.d3p_dpvi_vae_update(stream_ptr(), C.byref(vm), C.byref(hyper), C.byref(st), ptr(X), ptr(mask_t), B,
                                          ptr(_eps), ptr(loss), ptr(_grad_out), ptr(ws), ws.numel()))
        return DPSVIState((step, params, m, v), keybuf[1].reshape(4, 4), svi_state.observation_scale), loss[0]

    def _is_gmm(self):
        return isinstance(self.model, GaussianMixtureModel) and isinstance(self.guide, GaussianMixtureGuide)

    def _gmm_struct(self, d, kwargs, observation_scale):
        kw = dict(self.static_kwargs)
        kw.update(kwargs)
        k = kw.get("k") or self.model.k
        if k is None:
            raise ValueError("GaussianMixtureModel: the number of components k must be given")
        n_total = self.model.num_obs_total((), kw)
        return GmmModel(int(k), int(d), self.model.prior_mu_scale, 1.0 if n_total is None else n_total,
                        1.0 / float(observation_scale))

    def _require_logreg(self):
        if (not isinstance(self.model, (LogisticRegression, GaussianMean))
                or not isinstance(self.guide, (AutoDiagonalNormal, DiagonalNormalGuide, MeanFieldGuide))):
            raise _lib.D3PError("DPSVI: model must be d3p_amd.models.LogisticRegression or GaussianMean with an "
                                "AutoDiagonalNormal, DiagonalNormalGuide or MeanFieldGuide guide (the model families built so far)")

    def _labels(self, args):
        """The label vector of the batch, or None for families without labels (GaussianMean)."""
        if not self.model.has_labels:
            return None
        return _batch_array(args[1]).to(torch.float32)

    def _model_struct(self, d, kwargs, observation_scale):
        kw = dict(self.static_kwargs)
        kw.update(kwargs)
        n_total = self.model.num_obs_total((), kw)
        # a per-example batch has size 1, so plate(N, 1) scales the likelihood by N (svi.py:277)
        lik_scale = 1.0 if n_total is None else n_total
        m = self.model
        gauss = isinstance(m, GaussianMean)
        return LogregModel(int(d), int(m.intercept), m.prior_scale, m.intercept_prior_scale,
                           float(lik_scale), 1.0 / float(observation_scale),
                           _lib.D3P_FAMILY_GAUSS_MEAN if gauss else _lib.D3P_FAMILY_LOGREG,
                           _lib.D3P_GUIDE_EXP if self.guide.transform == "exp" else _lib.D3P_GUIDE_SOFTPLUS,
                           m.obs_scale if gauss else 0.0)

    def _hyper(self):
        o = self.optim
        if isinstance(o, Adam):
            return DpsviHyper(float(self._clipping_threshold), float(self._dp_scale), o.step_size, o.b1, o.b2, o.eps)
        return DpsviHyper(float(self._clipping_threshold), float(self._dp_scale), getattr(o, "step_size", 0.0),
                          0.9, 0.999, 1e-8)

    def _workspace(self, nbytes, device, tag="ws"):
        # one buffer per (purpose, device, STREAM): two streams that drive the same DPSVI object enqueue kernels that run beside each
        # other -- they must not share scratch memory (the reference is functional: nothing is shared between calls)
        key = (tag, str(device), torch.cuda.current_stream(device).cuda_stream)
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._ws[key] = buf
        return buf

    # ---------------------------------------------------------------- init (svi.py:213-236)
    def init(self, rng_key, *args, **kwargs):
        if self._is_gmm():
            return self._init_gmm(rng_key, *args, **kwargs)
        if self._is_vae():
            return self._init_vae(rng_key, *args, **kwargs)
        self._require_logreg()
        _lib.require_device()
        X = _batch_array(args[0])
        d = int(X.shape[1])
        D = self.model.latent_dim(d)
        jax_rng_key = self._rng_suite.convert_to_jax_rng_key(rng_key)
        if isinstance(self.guide, MeanFieldGuide):
            # param("w_loc", zeros(d)), param("w_std_log", zeros(d)), param("intercept_loc", 0.), param("intercept_std_log", 0.)
            # (examples/logistic_regression.py:77-83), flat in tree_flatten order of the dict
            loc = torch.zeros(D, dtype=torch.float32, device=X.device)
            unc = torch.zeros(D, dtype=torch.float32, device=X.device)
        elif isinstance(self.guide, DiagonalNormalGuide):
            # param(<site>_loc, zeros(d)), param(<site>_std_log, zeros(d)) (simple_gaussian_posterior.py:77-79)
            loc = torch.zeros(D, dtype=torch.float32, device=X.device) + _as_device_f32(self.guide.init_loc, X.device)
            unc = torch.full((D,), self.guide.init_std_log, dtype=torch.float32, device=X.device)
        else:
            fn = self.guide.init_loc_fn
            if isinstance(fn, init_to_uniform):
                loc = _dbg_uniform(jax_rng_key, D, -fn.radius, fn.radius)
            elif isinstance(fn, init_to_value):
                loc = _as_device_f32(fn.values, X.device).reshape(D).clone()
            else:
                raise ValueError("unsupported init_loc_fn")
            unc = torch.full((D,), self.guide.unconstrained_init_scale(), dtype=torch.float32, device=X.device)
        params = torch.cat([loc, unc]).contiguous()
        optim_state = self.optim.init(params)

        observation_scale = 1.0
        if self._clip_unscaled_observations:
            kw = dict(self.static_kwargs)
            kw.update(kwargs)
            # a one-element batch: plate(N, subsample_size=1) -> N (svi.py:225-234)
            observation_scale = get_observations_scale(self.model, (args[0][:1],) + tuple(args[1:]), kw)
        return DPSVIState(optim_state, rng_key, observation_scale)

    def _init_gmm(self, rng_key, *args, **kwargs):
        _lib.require_device()
        X = _batch_array(args[0])
        gm = self._gmm_struct(int(X.shape[1]), kwargs, 1.0)
        # param('alpha_log', zeros(k)), param('mus_loc', zeros((k, d)))  (gaussian_mixture_model.py:79-83)
        params = torch.zeros(gm.K + gm.K * gm.d, dtype=torch.float32, device=X.device)
        observation_scale = 1.0
        if self._clip_unscaled_observations:
            kw = dict(self.static_kwargs)
            kw.update(kwargs)
            observation_scale = get_observations_scale(self.model, (args[0][:1],) + tuple(args[1:]), kw)
        self._gmm_shape = (gm.K, gm.d)
        return DPSVIState(self.optim.init(params), rng_key, observation_scale)

    def get_params(self, svi_state):
        """Constrained parameters (numpyro SVI.get_params): auto_scale = softplus(unconstrained) for
        AutoDiagonalNormal; the hand-written guides keep their ``*_std_log`` unconstrained."""
        p = self.optim.get_params(svi_state.optim_state)
        if self._is_vae():
            z, h, h2 = self._vae_dims(dict(self.static_kwargs))
            he = h2 if h2 else h
            # P = (z he + he) + [h2 h + h] + (h D + D) + (D h + h) + [h h2 + h2] + 2 (he z + z): solve for D
            rest = (z * he + he) + 2 * (he * z + z) + h + (2 * h * h2 + h + h2 if h2 else 0)
            D = (p.numel() - rest) // (2 * h + 1)
            return self._vae_tree(p.clone(), VaeModel(D, h, z, 1.0, 1.0, h2))
        if self._is_gmm():
            K = int(self.static_kwargs.get("k") or self.model.k)
            return {"alpha_log": p[:K].clone(), "mus_loc": p[K:].reshape(K, -1).clone()}
        D = p.numel() // 2
        if isinstance(self.guide, MeanFieldGuide):
            out, pos = {}, 0
            for name, size in zip(self.guide.param_names(), self.guide.leaf_sizes(D - 1)):
                out[name] = p[pos:pos + size].clone() if size > 1 else p[pos].clone()
                pos += size
            return out
        n_loc, n_scale = self.guide.param_names()
        if isinstance(self.guide, DiagonalNormalGuide):
            return {n_loc: p[:D].clone(), n_scale: p[D:].clone()}
        return {n_loc: p[:D].clone(), n_scale: torch.nn.functional.softplus(p[D:])}

    # ---------------------------------------------------------------- stage 1 (svi.py:238-308)
    def _compute_per_example_gradients_gmm(self, dp_svi_state, step_rng_key, *args, mask=True, **kwargs):
        _lib.require_device()
        lib = _lib.load()
        X = _batch_array(args[0])
        B, d = X.shape
        gm = self._gmm_struct(d, kwargs, dp_svi_state.observation_scale)
        K, P = gm.K, gm.K + gm.K * d
        jax_rng_key = self._rng_suite.convert_to_jax_rng_key(step_rng_key).contiguous()
        params = self.optim.get_params(dp_svi_state.optim_state).contiguous()
        if params.numel() != P:
            raise ValueError("GaussianMixtureModel: parameter vector does not match k and the data dimension")
        mask_t = None
        if not isinstance(mask, bool):
            mask_t = _mask_array(mask, X.device, B)
        elif mask is False:
            mask_t = torch.zeros(B, dtype=torch.uint8, device=X.device)
        px_loss = torch.empty(B, dtype=torch.float32, device=X.device)
        px_grads = torch.empty((B, P), dtype=torch.float32, device=X.device)
        meta = torch.empty(2, dtype=torch.float32, device=X.device)
        ws = self._workspace(lib.d3p_gmm_px_grads_workspace(K, B), X.device, "gmm")
        lat = kwargs.get("_latents_out")
        check(lib.d3p_gmm_px_grads(stream_ptr(), C.byref(gm), ptr(params), ptr(X), ptr(mask_t), B, ptr(jax_rng_key),
                                   ptr(px_loss), ptr(px_grads), ptr(meta), ptr(lat), ptr(ws), ws.numel()))
        grads = {"alpha_log": px_grads[:, :K], "mus_loc": px_grads[:, K:].reshape(B, K, d)}
        return dp_svi_state, px_loss, grads, meta[0], meta[1]

    def _compute_per_example_gradients(self, dp_svi_state, step_rng_key, *args, mask=True, **kwargs):
        if self._is_gmm():
            return self._compute_per_example_gradients_gmm(dp_svi_state, step_rng_key, *args, mask=mask, **kwargs)
        self._require_logreg()
        _lib.require_device()
        lib = _lib.load()
        X = _batch_array(args[0])
        y = self._labels(args)
        B, d = X.shape
        D = self.model.latent_dim(d)
        jax_rng_key = self._rng_suite.convert_to_jax_rng_key(step_rng_key).contiguous()
        params = self.optim.get_params(dp_svi_state.optim_state).contiguous()
        self._require_sizes(params.numel(), 2 * D, B, y)
        model = self._model_struct(d, kwargs, dp_svi_state.observation_scale)
        mask_t = None
        if not isinstance(mask, bool):
            mask_t = _mask_array(mask, X.device, B)
        elif mask is False:
            mask_t = torch.zeros(B, dtype=torch.uint8, device=X.device)
        px_loss = torch.empty(B, dtype=torch.float32, device=X.device)
        px_grads = torch.empty((B, 2 * D), dtype=torch.float32, device=X.device)
        meta = torch.empty(2, dtype=torch.float32, device=X.device)
        ws = self._workspace(lib.d3p_logreg_px_grads_workspace(C.byref(model), B), X.device, "px")
        eps = kwargs.get("_eps")
        multi = isinstance(self.guide, MeanFieldGuide)
        if multi:
            # the state's parameters are flat in tree order (four leaves); the kernels take [loc (D) | unconstrained scale (D)] with the
            # intercept last; every sample site draws its eps from its own key (d3p_px_eps_sites)
            perm = MeanFieldGuide.tree_from_kernel(d, X.device)
            kern = torch.empty_like(params)
            kern[perm] = params
            params = kern
            if eps is None:
                sizes = (C.c_int32 * 2)(d, 1)
                eps = torch.empty((B, D), dtype=torch.float32, device=X.device)
                check(lib.d3p_px_eps_sites(stream_ptr(), ptr(jax_rng_key), B, 0, B, sizes, 2, ptr(eps)))
        check(lib.d3p_logreg_px_grads(stream_ptr(), C.byref(model), ptr(params), ptr(X), ptr(y), ptr(mask_t), B,
                                      ptr(eps), ptr(jax_rng_key), ptr(px_loss), ptr(px_grads), ptr(meta),
                                      ptr(ws), ws.numel()))
        if multi:
            grads = {"intercept_loc": px_grads[:, d], "intercept_std_log": px_grads[:, D + d],
                     "w_loc": px_grads[:, :d], "w_std_log": px_grads[:, D:D + d]}
            return dp_svi_state, px_loss, grads, meta[0], meta[1]
        n_loc, n_scale = self.guide.param_names()
        grads = {n_loc: px_grads[:, :D], n_scale: px_grads[:, D:]}
        return dp_svi_state, px_loss, grads, meta[0], meta[1]

    # ---------------------------------------------------------------- stage 2 (svi.py:310-325)
    def _clip_gradients(self, dp_svi_state, px_grads):
        if self._clipping_threshold == 0.0:
            raise ValueError("The clipping threshold must be greater than 0.")
        _lib.require_device()
        leaves, treedef = _tree_flatten(px_grads)
        flat = [_as_device_f32(g) for g in leaves]
        B = flat[0].shape[0]
        rows = torch.cat([g.reshape(B, -1) for g in flat], dim=1).contiguous()
        _clip_flat_rows(rows, self._clipping_threshold)
        out, pos = [], 0
        for g in flat:
            w = g[0].numel()
            out.append(rows[:, pos:pos + w].reshape(g.shape))
            pos += w
        return dp_svi_state, _tree_unflatten(treedef, out)

    # ---------------------------------------------------------------- stage 3 (svi.py:327-348)
    def _combine_gradients(self, px_clipped_grads, px_loss):
        _lib.require_device()
        lib = _lib.load()
        leaves, treedef = _tree_flatten(px_clipped_grads)
        px_loss = _as_device_f32(px_loss).contiguous()
        B = px_loss.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=px_loss.device)
        outs = []
        for k, g in enumerate(leaves):
            g = _as_device_f32(g, px_loss.device)
            if g.dim() < 1 or g.shape[0] != B:
                raise ValueError(f"_combine_gradients: a gradient leaf with {g.shape[0] if g.dim() else 0} rows beside {B} per-example losses")
            rows = g.reshape(g.shape[0], -1).contiguous()
            avg = torch.empty(rows.shape[1], dtype=torch.float32, device=rows.device)
            check(lib.d3p_combine(stream_ptr(), ptr(rows), ptr(px_loss) if k == 0 else None, rows.shape[0],
                                  rows.shape[1], ptr(avg), ptr(loss) if k == 0 else None))
            outs.append(avg.reshape(g.shape[1:]))
        if not leaves:
            check(lib.d3p_combine(stream_ptr(), ptr(px_loss.reshape(B, 1)), ptr(px_loss), B, 1,
                                  ptr(torch.empty(1, device=px_loss.device)), ptr(loss)))
        return loss[0], _tree_unflatten(treedef, outs)

    # ---------------------------------------------------------------- stage 4 (svi.py:350-377)
    def _perturb_and_reassemble_gradients(self, dp_svi_state, step_rng_key, avg_clipped_grads, num_elements,
                                          batch_mask_scaling_factor):
        _lib.require_device()
        lib = _lib.load()
        leaves, treedef = _tree_flatten(avg_clipped_grads)
        dev = step_rng_key.device
        if isinstance(num_elements, torch.Tensor) or isinstance(batch_mask_scaling_factor, torch.Tensor):
            meta = torch.stack([_as_device_f32(num_elements, dev).reshape(()),
                                _as_device_f32(batch_mask_scaling_factor, dev).reshape(())]).contiguous()
        else:
            meta = torch.tensor([float(num_elements), float(batch_mask_scaling_factor)], dtype=torch.float32,
                                device=dev)
        obs_scale = float(dp_svi_state.observation_scale)
        per_site_rngs = self._rng_suite.split(step_rng_key, len(leaves))  # svi.py:491
        outs = []
        for g, site_rng in zip(leaves, per_site_rngs):
            g = _as_device_f32(g, dev).contiguous()
            noise = self._rng_suite.normal(site_rng, g.shape).contiguous()  # svi.py:487
            out = torch.empty_like(g)
            check(lib.d3p_perturb_apply(stream_ptr(), ptr(g), ptr(noise), g.numel(), float(self._dp_scale),
                                        float(self._clipping_threshold), ptr(meta), obs_scale, ptr(out)))
            outs.append(out)
        return dp_svi_state, _tree_unflatten(treedef, outs)

    @staticmethod
    def perturbation_function(rng_suite, rng, values, perturbation_scale):
        """d3p/svi.py:470-498: each leaf += normal(site_key) * perturbation_scale."""
        _lib.require_device()
        lib = _lib.load()
        leaves, treedef = _tree_flatten(values)
        if isinstance(perturbation_scale, torch.Tensor):
            perturbation_scale = float(perturbation_scale)
        meta = torch.tensor([1.0, 1.0], dtype=torch.float32, device=rng.device)
        per_site_rngs = rng_suite.split(rng, len(leaves))
        outs = []
        for g, site_rng in zip(leaves, per_site_rngs):
            g = _as_device_f32(g, rng.device).contiguous()
            noise = rng_suite.normal(site_rng, g.shape).contiguous()
            out = torch.empty_like(g)
            check(lib.d3p_perturb_apply(stream_ptr(), ptr(g), ptr(noise), g.numel(), float(perturbation_scale), 1.0,
                                        ptr(meta), 1.0, ptr(out)))
            outs.append(out)
        return _tree_unflatten(treedef, outs)

    # ---------------------------------------------------------------- stage 5 (svi.py:379-393)
    def _apply_gradient(self, dp_svi_state, perturbed_grads):
        _lib.require_device()
        lib = _lib.load()
        leaves, _ = _tree_flatten(perturbed_grads)
        g = torch.cat([_as_device_f32(l).reshape(-1) for l in leaves]).contiguous()
        st = dp_svi_state.optim_state
        if isinstance(self.optim, ADADP):
            return self._update_state_optim_state(dp_svi_state, self.optim.update(g, st))
        self._require_device_state(dp_svi_state)
        if g.numel() != st[1].numel():
            raise ValueError(f"_apply_gradient: a gradient of {g.numel()} elements for {st[1].numel()} parameters")
        step = st[0].clone()
        params = st[1].clone()
        if isinstance(self.optim, Adam):
            m, v = st[2].clone(), st[3].clone()
            o = self.optim
            check(lib.d3p_adam_step(stream_ptr(), ptr(params), ptr(m), ptr(v), ptr(step), ptr(g), params.numel(),
                                    o.step_size, o.b1, o.b2, o.eps))
            new = (step, params, m, v)
        elif isinstance(self.optim, SGD):
            check(lib.d3p_sgd_step(stream_ptr(), ptr(params), ptr(step), ptr(g), params.numel(),
                                   self.optim.step_size))
            new = (step, params)

        else:
            raise _lib.D3PError("unsupported optimiser")
        return self._update_state_optim_state(dp_svi_state, new)

    # ---------------------------------------------------------------- update (svi.py:395-434)
    def _fusable(self):
        return (isinstance(self.model, (LogisticRegression, GaussianMean))
                and isinstance(self.guide, (AutoDiagonalNormal, DiagonalNormalGuide))
                and isinstance(self.optim, Adam) and self._rng_suite is strong_rng)

    def _gmm_fusable(self):
        return self._is_gmm() and isinstance(self.optim, Adam) and self._rng_suite is strong_rng

    def _update_gmm_fused(self, svi_state, *args, mask=True, _grad_out=None, **kwargs):
        """DPSVI.update for the mixture model through ``d3p_dpvi_gmm_update`` (one C call, no B x P tensor)."""
        _lib.require_device()
        lib = _lib.load()
        X = _batch_array(args[0])
        B, d = X.shape
        dev = X.device
        gm = self._gmm_struct(d, kwargs, svi_state.observation_scale)
        hyper = self._hyper()
        self._require_sizes(svi_state.optim_state[1].numel(), gm.K + gm.K * d)
        step, params, m, v = _fresh_optim_state(svi_state.optim_state)
        keybuf = torch.empty((2, 16), dtype=torch.uint32, device=dev)
        keybuf[0].copy_(svi_state.rng_key.reshape(16))
        mask_t = None
        if not isinstance(mask, bool):
            mask_t = _mask_array(mask, X.device, B)
        elif mask is False:
            mask_t = torch.zeros(B, dtype=torch.uint8, device=dev)
        st = self._state_struct(keybuf, 0, (step, params, m, v))
        ws = self._workspace(lib.d3p_dpvi_gmm_workspace(C.byref(gm), B), dev, "gmm_step")
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        check(lib.d3p_dpvi_gmm_update(stream_ptr(), C.byref(gm), C.byref(hyper), C.byref(st), ptr(X), ptr(mask_t), B,
                                      ptr(loss), ptr(_grad_out), ptr(ws), ws.numel()))
        return DPSVIState((step, params, m, v), keybuf[1].reshape(4, 4), svi_state.observation_scale), loss[0]

    def _run_steps_gmm(self, svi_state, get_batch, batchifier_state, first_batch, num_steps, **kwargs):
        info = getattr(get_batch, "source", None)
        if info is None or info.rng_suite is not strong_rng or info.kind != _lib.D3P_BATCH_FEISTEL:
            raise _lib.D3PError("run_steps (mixture model): get_batch must come from subsample_batchify_data "
                                "(without replacement) with rng_suite=d3p_amd.random")
        _lib.require_device()
        lib = _lib.load()
        X = info.dataset[0]
        if not (X.is_cuda and X.is_contiguous() and X.dtype == torch.float32):
            raise _lib.D3PError("run_steps: dataset arrays must be contiguous float32 CUDA tensors")
        N, d = X.shape
        dev = X.device
        gm = self._gmm_struct(d, kwargs, svi_state.observation_scale)
        hyper = self._hyper()
        self._require_sizes(svi_state.optim_state[1].numel(), gm.K + gm.K * d)
        step, params, m, v = _fresh_optim_state(svi_state.optim_state)
        keybuf = torch.empty((2, 16), dtype=torch.uint32, device=dev)
        keybuf[0].copy_(svi_state.rng_key.reshape(16))
        bkey = strong_rng._key(batchifier_state)          # (a 16-word CUDA key: anything else is a TypeError, not an address)
        st = self._state_struct(keybuf, 0, (step, params, m, v))
        B = int(info.batch_size)
        ws = self._workspace(lib.d3p_dpvi_gmm_workspace(C.byref(gm), B), dev, "gmm_step")
        losses = torch.empty(max(num_steps, 1), dtype=torch.float32, device=dev)
        check(lib.d3p_dpvi_gmm_run(stream_ptr(), C.byref(gm), C.byref(hyper), C.byref(st), ptr(bkey), int(first_batch),
                                   ptr(X), int(N), B, int(num_steps), ptr(losses), ptr(ws), ws.numel()))
        return (DPSVIState((step, params, m, v), keybuf[num_steps & 1].reshape(4, 4), svi_state.observation_scale),
                losses[:num_steps])

    def _run_steps_stepwise(self, svi_state, get_batch, batchifier_state, first_batch, num_steps, **kwargs):
        """``num_steps`` x (``get_batch`` -> ``update``) through the public calls: what ``run_steps`` does for batchifiers and
        model / optimiser combinations that have no native loop.  Returns ``(new_state, losses[num_steps])``."""
        losses = []
        for t in range(int(num_steps)):
            out = get_batch(int(first_batch) + t, batchifier_state)
            if isinstance(out, tuple) and len(out) == 2 and isinstance(out[0], tuple):   # (batch_tuple, mask)
                svi_state, loss = self.update(svi_state, *out[0], mask=out[1], **kwargs)
            else:
                svi_state, loss = self.update(svi_state, *out, **kwargs)
            losses.append(loss.reshape(()))
        dev = losses[0].device if losses else svi_state.optim_state[1].device
        return svi_state, (torch.stack(losses) if losses else torch.empty(0, dtype=torch.float32, device=dev))

    def _run_steps_vae(self, svi_state, info, batchifier_state, first_batch, num_steps, **kwargs):
        """The VAE's epoch body (examples/vae.py:227-246) as ONE native call: per step fold_in, the Feistel indices, the row
        gather and the update (~35 launches), enqueued back to back by d3p_dpvi_vae_run."""
        _lib.require_device()
        lib = _lib.load()
        X = self._vae_flat(info.dataset[0])
        if not (X.is_cuda and X.is_contiguous() and X.dtype == torch.float32):
            raise _lib.D3PError("run_steps: dataset arrays must be contiguous float32 CUDA tensors")
        N, D = X.shape
        B = int(info.batch_size)
        dev = X.device
        vm = self._vae_struct(D, kwargs, svi_state.observation_scale)
        hyper = self._hyper()
        self._require_sizes(svi_state.optim_state[1].numel(), lib.d3p_vae_num_params(C.byref(vm)))
        step, params, m, v = _fresh_optim_state(svi_state.optim_state)
        keybuf = torch.empty((2, 16), dtype=torch.uint32, device=dev)
        keybuf[0].copy_(svi_state.rng_key.reshape(16))
        bkey = strong_rng._key(batchifier_state)          # (a 16-word CUDA key: anything else is a TypeError, not an address)
        st = self._state_struct(keybuf, 0, (step, params, m, v))
        ws = self._workspace(lib.d3p_dpvi_vae_workspace(C.byref(vm), B), dev, "vae_step")
        xb = torch.empty((B, D), dtype=torch.float32, device=dev)
        idx = torch.empty(B + 16, dtype=torch.uint32, device=dev)
        losses = torch.empty(max(int(num_steps), 1), dtype=torch.float32, device=dev)
        check(lib.d3p_dpvi_vae_run(stream_ptr(), C.byref(vm), C.byref(hyper), C.byref(st), ptr(bkey), int(first_batch), ptr(X), int(N), B,
                                   int(num_steps), ptr(losses), ptr(xb), ptr(idx), ptr(ws), ws.numel()))
        return (DPSVIState((step, params, m, v), keybuf[int(num_steps) & 1].reshape(4, 4), svi_state.observation_scale),
                losses[:int(num_steps)])

    @staticmethod
    def _require_sizes(n_params, expected, B=None, y=None):
        """The kernels take shapes from the batch and addresses from the state: a state for another model or feature count, or labels
        of another length, would be read out of bounds."""
        if int(n_params) != int(expected):
            raise ValueError(f"the state holds {int(n_params)} parameters, this model on a batch of this shape has {int(expected)}")
        if y is not None and B is not None and y.numel() != int(B):
            raise ValueError(f"labels: {y.numel()} entries for a batch of {int(B)} examples")

    @staticmethod
    def _require_device_state(svi_state):
        """The state's key and flat optimiser arrays are handed to kernels by address: they must be device tensors (a state that was
        moved to the host, e.g. to save it, goes back with ``.cuda()`` first)."""
        for t in (svi_state.rng_key,) + tuple(svi_state.optim_state):
            if isinstance(t, torch.Tensor) and not t.is_cuda:
                raise _lib.D3PError("DPSVI: the state's tensors must live on the GPU (found a host tensor in the state)")
            if isinstance(t, torch.Tensor) and t.is_floating_point() and t.dtype != torch.float32:
                raise _lib.D3PError(f"DPSVI: the state's float arrays must be float32 (found {t.dtype}): the kernels read them by address")

    def update(self, svi_state, *args, mask=True, **kwargs):
        """One DP-VI step on a batch; returns ``(new_state, loss)`` (svi.py:395-434)."""
        self._require_device_state(svi_state)
        if self._gmm_fusable():
            return self._update_gmm_fused(svi_state, *args, mask=mask, **kwargs)
        if self._is_vae():
            return self._update_vae(svi_state, *args, mask=mask, **kwargs)
        if self._fusable():
            return self._update_fused(svi_state, *args, mask=mask, **kwargs)
        if self._leaves_fusable() and "_eps" not in kwargs:
            return self._update_leaves(svi_state, *args, mask=mask, **kwargs)
        return self._update_staged(svi_state, *args, mask=mask, **kwargs)

    def _update_staged(self, svi_state, *args, mask=True, **kwargs):
        """The reference's five-stage composition, literally (svi.py:413-434)."""
        svi_state, update_rng_keys = self._split_rng_key(svi_state, 2)
        gradient_rng_key, perturbation_rng_key = update_rng_keys[0], update_rng_keys[1]
        svi_state, px_losses, px_grads, num_elements, factor = self._compute_per_example_gradients(
            svi_state, gradient_rng_key, *args, mask=mask, **kwargs)
        svi_state, px_clipped_grads = self._clip_gradients(svi_state, px_grads)
        loss, avg_clipped_grads = self._combine_gradients(px_clipped_grads, px_losses)
        svi_state, perturbed_grads = self._perturb_and_reassemble_gradients(
            svi_state, perturbation_rng_key, avg_clipped_grads, num_elements, factor)
        svi_state = self._apply_gradient(svi_state, perturbed_grads)
        return svi_state, loss

    def _leaves_fusable(self):
        return (isinstance(self.model, LogisticRegression) and isinstance(self.guide, MeanFieldGuide)
                and isinstance(self.optim, Adam) and self._rng_suite is strong_rng)

    def _update_leaves(self, svi_state, *args, mask=True, _grad_out=None, **kwargs):
        """DPSVI.update for the example's own guide (four parameter leaves, two sample sites) around the FUSED clipped sums: the step's
        keys and the parameters in the kernels' column order (d3p_dpvi_leaves_begin), every sample site's eps from its own key
        (d3p_px_eps_sites), per-example gradient -> clip -> sum without a B x P tensor (d3p_dpvi_logreg_local_sums), then mean, one noise
        key per leaf, rescaling and Adam (d3p_dpvi_leaves_finalize): nine launches, the same result as ``_update_staged`` (~95) to fp32
        summation order."""
        _lib.require_device()
        lib = _lib.load()
        X = _batch_array(args[0])
        y = self._labels(args)
        B, d = X.shape
        D = self.model.latent_dim(d)
        P = 2 * D
        dev = X.device
        step0, params0, m0, v0 = svi_state.optim_state
        key0 = svi_state.rng_key.reshape(16)
        self._require_sizes(params0.numel(), P, B, y)
        if not (X.dtype == torch.float32 and (y is None or y.dtype == torch.float32)
                and params0.dtype == m0.dtype == v0.dtype == torch.float32 and params0.numel() == m0.numel() == v0.numel() == P
                and params0.is_contiguous() and m0.is_contiguous() and v0.is_contiguous() and key0.is_contiguous()
                and key0.dtype == torch.uint32 and step0.dtype == torch.int32):
            return self._update_staged(svi_state, *args, mask=mask, **kwargs)
        model = self._model_struct(d, kwargs, svi_state.observation_scale)
        hyper = self._hyper()
        mask_t = None
        if not isinstance(mask, bool):
            mask_t = _mask_array(mask, X.device, B)
        elif mask is False:
            mask_t = torch.zeros(B, dtype=torch.uint8, device=dev)
        src = BatchSource(_lib.D3P_BATCH_EXPLICIT, B, 0.0, 0, None, None,
                          None if mask_t is None else mask_t.data_ptr(), B, 0, B)
        col_of = self._ws.get(("col_of", d, dev))
        if col_of is None:
            col_of = MeanFieldGuide.tree_from_kernel(d, dev).to(torch.int32).contiguous()
            self._ws[("col_of", d, dev)] = col_of
        sites = self.guide.sites(d)
        leaves = self.guide.leaf_sizes(d)
        n_leaves = len(leaves)
        site_sizes = (C.c_int32 * len(sites))(*[n for _, n in sites])
        leaf_sizes = (C.c_int32 * n_leaves)(*leaves)
        keys = torch.empty(16 * (2 + n_leaves), dtype=torch.uint32, device=dev)   # next key | jax key (2 of 16 words) | leaf keys
        next_key, jax_key, leaf_keys = keys[:16], keys[16:18], keys[32:]
        # kernel-order parameters + [sums | loss sum | n] + the new state + the loss in one allocation
        kern, sums, params, m, v, loss = torch.empty(5 * P + 3, dtype=torch.float32, device=dev).split((P, P + 2, P, P, P, 1))
        step = torch.empty_like(step0)
        eps = torch.empty((B, D), dtype=torch.float32, device=dev)
        s = stream_ptr()
        check(lib.d3p_dpvi_leaves_begin(s, ptr(key0), n_leaves, ptr(params0), ptr(col_of), P, ptr(next_key), ptr(jax_key), ptr(leaf_keys),
                                        ptr(kern)))
        check(lib.d3p_px_eps_sites(s, ptr(jax_key), B, 0, B, site_sizes, len(sites), ptr(eps)))
        st = self._state_struct(key0, 0, (step0, kern, m0, v0))      # (read only: the sums need the parameters, nothing else)
        ws = self._workspace(lib.d3p_dpvi_logreg_workspace(C.byref(model), C.byref(src)), dev)
        check(lib.d3p_dpvi_logreg_local_sums(s, C.byref(model), C.byref(hyper), C.byref(st), C.byref(src), ptr(X), ptr(y), ptr(eps),
                                             ptr(sums), ptr(ws), ws.numel()))
        check(lib.d3p_dpvi_leaves_finalize(s, C.byref(hyper), ptr(sums), ptr(col_of), ptr(leaf_keys), leaf_sizes, n_leaves, B,
                                           float(svi_state.observation_scale), ptr(params0), ptr(m0), ptr(v0), ptr(step0), ptr(params),
                                           ptr(m), ptr(v), ptr(step), ptr(loss), ptr(_grad_out)))
        return DPSVIState((step, params, m, v), next_key.reshape(4, 4), svi_state.observation_scale), loss[0]

    def _state_struct(self, keybuf, slot, optim_state):
        step, params, m, v = optim_state
        return DpsviState(keybuf.data_ptr(), int(slot), params.data_ptr(), m.data_ptr(), v.data_ptr(),
                          step.data_ptr())

    def _update_fused(self, svi_state, *args, mask=True, _eps=None, _grad_out=None, **kwargs):
        _lib.require_device()
        lib = _lib.load()
        X = _batch_array(args[0])
        y = self._labels(args)
        B, d = X.shape
        D = self.model.latent_dim(d)
        dev = X.device
        model = self._model_struct(d, kwargs, svi_state.observation_scale)
        hyper = self._hyper()
        mask_t = None
        if not isinstance(mask, bool):
            mask_t = _mask_array(mask, X.device, B)
        elif mask is False:
            mask_t = torch.zeros(B, dtype=torch.uint8, device=dev)
        src = BatchSource(_lib.D3P_BATCH_EXPLICIT, B, 0.0, 0, None, None,
                          None if mask_t is None else mask_t.data_ptr(), B, 0, B)
        step0, params0, m0, v0 = svi_state.optim_state
        key0 = svi_state.rng_key.reshape(16)
        n = params0.numel()
        self._require_sizes(n, 2 * D, B, y)
        if (_eps is None and _grad_out is None and X.dtype == torch.float32 and (y is None or y.dtype == torch.float32)
                and params0.dtype == m0.dtype == v0.dtype == torch.float32 and m0.numel() == n and v0.numel() == n
                and params0.is_contiguous() and m0.is_contiguous() and v0.is_contiguous() and key0.is_contiguous()
                and key0.dtype == torch.uint32 and step0.dtype == torch.int32):
            # One step of the device-resident run on the caller's batch: the new state is written by the run itself (no
            # copies of the old one), four launches instead of the nine of the two-call form below -- 59 -> 40 us per call
            params, m, v, loss = torch.empty(3 * n + 1, dtype=torch.float32, device=dev).split((n, n, n, 1))
            if params0.dim() != 1:
                params, m, v = params.view_as(params0), m.view_as(m0), v.view_as(v0)
            step = torch.empty_like(step0)
            keybuf = torch.empty((2, 16), dtype=torch.uint32, device=dev)
            st = self._state_struct(keybuf, 0, (step, params, m, v))
            frm = self._state_struct(key0, 0, (step0, params0, m0, v0))
            ws = self._workspace(lib.d3p_dpvi_logreg_workspace(C.byref(model), C.byref(src)), dev)
            check(lib.d3p_dpvi_logreg_run_from(stream_ptr(), C.byref(model), C.byref(hyper), C.byref(st), C.byref(frm), C.byref(src),
                                               0, ptr(X), ptr(y), 1, ptr(loss), ptr(ws), ws.numel()))
            return DPSVIState((step, params, m, v), keybuf[1].reshape(4, 4), svi_state.observation_scale), loss[0]
        step, params, m, v = _fresh_optim_state(svi_state.optim_state)
        keybuf = torch.empty((2, 16), dtype=torch.uint32, device=dev)
        keybuf[0].copy_(key0)
        st = self._state_struct(keybuf, 0, (step, params, m, v))
        ws = self._workspace(lib.d3p_dpvi_logreg_workspace(C.byref(model), C.byref(src)), dev)
        sums = torch.empty(2 * D + 2, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        s = stream_ptr()
        check(lib.d3p_dpvi_logreg_local_sums(s, C.byref(model), C.byref(hyper), C.byref(st), C.byref(src), ptr(X),
                                             ptr(y), ptr(_eps), ptr(sums), ptr(ws), ws.numel()))
        check(lib.d3p_dpvi_logreg_finalize(s, C.byref(model), C.byref(hyper), C.byref(st), C.byref(src), ptr(sums),
                                           ptr(loss), ptr(_grad_out), ptr(ws), ws.numel()))
        new_state = DPSVIState((step, params, m, v), keybuf[1].reshape(4, 4), svi_state.observation_scale)
        return new_state, loss[0]

    # ---------------------------------------------------------------- fused multi-step loop
    def run_steps(self, svi_state, get_batch, batchifier_state, first_batch, num_steps, check_status=True, _no_fallback=False,
                  **kwargs):
        """``num_steps`` x (get_batch(i, batchifier_state) -> update) for i = first_batch..., enqueued
        back to back on the device: the body of the reference's ``jit(lax.fori_loop(...))`` epoch
        (examples/logistic_regression.py:149-160, examples/vae.py:227-246).  Batchifiers of ``subsample_batchify_data``
        (without replacement) and ``poisson_batchify_data`` run as native device-resident loops (logistic regression /
        Gaussian mean: chained launches; mixture model and VAE: one native call for the whole run); everything else
        (sampling with replacement, ``split_batchify_data``, other rng suites or optimisers) runs the same steps through
        ``get_batch`` + ``update``.  Returns ``(new_state, losses[num_steps])``.

        ``check_status`` (default): synchronise the stream once after the run and read the run's status words
        (``d3p_dpvi_logreg_run_status``); a run whose chained launch was aborted by a bounded wait raises ``D3PError``
        instead of returning a silently truncated trajectory.  ``check_status=False`` keeps the call asynchronous; the
        caller then checks with ``DPSVI.last_run_status()`` before trusting the result.  A run that was stopped is run again,
        from the same (untouched) input state, with one launch per step (``d3p_dpvi_logreg_set_run_form``) before anything is
        raised: a stalled chained launch costs the caller time, not the result."""
        self._require_device_state(svi_state)
        info = getattr(get_batch, "source", None)
        if info is not None and info.kind == _lib.D3P_BATCH_POISSON and int(info.batch_size) > example_count(info.dataset[0]):
            raise AssertionError("poisson_batchify_data: max_batch_size exceeds the number of records")   # (as get_batch, minibatch.py:116)
        if info is not None:
            if any(not (isinstance(a, torch.Tensor) and a.is_cuda) for a in info.dataset):
                raise _lib.D3PError("run_steps: the batchifier's dataset arrays must be CUDA tensors")
            if any(a.dtype != torch.float32 or not a.is_contiguous() for a in info.dataset):
                # (float64 / integer-labelled / strided tables: the native loops read contiguous float32 rows in place; get_batch + update
                #  gather and convert per step)
                return self._run_steps_stepwise(svi_state, get_batch, batchifier_state, first_batch, num_steps, **kwargs)
        if info is None or info.rng_suite is not strong_rng or not (self._gmm_fusable() or self._is_vae() or self._fusable()):
            # no native loop for this combination (sampling with replacement, split_batchify_data's epochs, another rng_suite,
            # the stage-wise optimisers): the same steps through the API-parity path -- get_batch + update, one call each,
            # still without a host synchronisation between them
            return self._run_steps_stepwise(svi_state, get_batch, batchifier_state, first_batch, num_steps, **kwargs)
        if self._gmm_fusable():
            if info.kind != _lib.D3P_BATCH_FEISTEL:
                return self._run_steps_stepwise(svi_state, get_batch, batchifier_state, first_batch, num_steps, **kwargs)
            return self._run_steps_gmm(svi_state, get_batch, batchifier_state, first_batch, num_steps, **kwargs)
        if self._is_vae():
            if info.kind != _lib.D3P_BATCH_FEISTEL or not (isinstance(self.optim, Adam)):
                return self._run_steps_stepwise(svi_state, get_batch, batchifier_state, first_batch, num_steps, **kwargs)
            return self._run_steps_vae(svi_state, info, batchifier_state, first_batch, num_steps, **kwargs)
        _lib.require_device()
        lib = _lib.load()
        X = info.dataset[0]
        y = info.dataset[1] if self.model.has_labels else None
        if not (X.is_cuda and X.is_contiguous() and X.dtype == torch.float32
                and (y is None or (y.is_cuda and y.is_contiguous() and y.dtype == torch.float32))):
            raise _lib.D3PError("run_steps: dataset arrays must be contiguous float32 CUDA tensors")
        N, d = X.shape
        dev = X.device
        model = self._model_struct(d, kwargs, svi_state.observation_scale)
        hyper = self._hyper()
        bkey = strong_rng._key(batchifier_state)          # (a 16-word CUDA key: anything else is a TypeError, not an address)
        step0, params0, m0, v0 = svi_state.optim_state
        key0 = svi_state.rng_key.reshape(16)
        n = params0.numel()
        self._require_sizes(n, 2 * self.model.latent_dim(d), N, y)
        nl = max(num_steps, 1)
        if (params0.dtype == m0.dtype == v0.dtype == torch.float32 and m0.numel() == n
                and v0.numel() == n and params0.is_contiguous() and m0.is_contiguous() and v0.is_contiguous()
                and key0.is_contiguous() and key0.dtype == torch.uint32 and step0.dtype == torch.int32):
            # the new state is written by the run itself (d3p_dpvi_logreg_run_from copies the old one inside its first kernel):
            # no copy / fill launches on the host's enqueue path -- they were ~80 us of a 20-step run.  Everything the run
            # writes -- the three state arrays, the losses, the step counter, the two key slots -- is ONE allocation, handed
            # to the library as raw addresses; the tensors the caller gets back are views of it made AFTER the enqueue, while
            # the device is already running (what the host does in front of the first launch is wall time of a short run: 17 us
            # of a 20-step run's 205 before, 9 now)
            # layout in 4-byte words: [params n | m n | v n | losses nl | step 1 | pad to 4 | key slots 2 x 16]
            koff = (3 * n + nl + 1 + 3) & ~3
            buf = torch.empty(koff + 32, dtype=torch.float32, device=dev)
            base = buf.data_ptr()
            bidx = None
            src = BatchSource(info.kind, info.batch_size, float(info.q), int(info.suppress), bkey.data_ptr(), None, None, N, 0, N)
            st = DpsviState(base + 4 * koff, 0, base, base + 4 * n, base + 8 * n, base + 4 * (3 * n + nl))
            frm = DpsviState(key0.data_ptr(), 0, params0.data_ptr(), m0.data_ptr(), v0.data_ptr(), step0.data_ptr())
            ws = self._workspace(lib.d3p_dpvi_logreg_workspace(C.byref(model), C.byref(src)), dev)
            check(lib.d3p_dpvi_logreg_run_from(stream_ptr(), C.byref(model), C.byref(hyper), C.byref(st), C.byref(frm), C.byref(src),
                                               int(first_batch), ptr(X), ptr(y), int(num_steps), base + 12 * n, ptr(ws), ws.numel()))
            params, m, v, losses, tail = buf.split((n, n, n, nl, koff + 32 - 3 * n - nl))
            if params0.dim() != 1:
                params, m, v = params.view_as(params0), m.view_as(m0), v.view_as(v0)
            step = tail[0].view(torch.int32)
            keybuf = tail[koff - 3 * n - nl:].view(torch.uint32).view(2, 16)
        else:
            losses = torch.empty(nl, dtype=torch.float32, device=dev)
            step, params, m, v = _fresh_optim_state(svi_state.optim_state)
            keybuf = torch.empty((2, 16), dtype=torch.uint32, device=dev)
            keybuf[0].copy_(key0)
            bidx = torch.full((1,), int(first_batch), dtype=torch.int32, device=dev)   # (a fill kernel: no host-to-device copy + sync)
            src = BatchSource(info.kind, info.batch_size, float(info.q), int(info.suppress), bkey.data_ptr(),
                              bidx.data_ptr(), None, N, 0, N)
            st = self._state_struct(keybuf, 0, (step, params, m, v))
            ws = self._workspace(lib.d3p_dpvi_logreg_workspace(C.byref(model), C.byref(src)), dev)
            check(lib.d3p_dpvi_logreg_run(stream_ptr(), C.byref(model), C.byref(hyper), C.byref(st), C.byref(src),
                                          ptr(X), ptr(y), int(num_steps), ptr(losses), ptr(ws), ws.numel()))
        self._last_run = (model, src, ws, (bkey, bidx))   # what last_run_status() needs (keeps the buffers alive)
        if check_status:
            aborted, _ = self.last_run_status()
            if aborted and not _no_fallback:
                # The chained launch was stopped by a bounded wait (its workgroups did not make progress together: a GPU shared
                # with other work).  The input state is untouched (functional update), so the same steps are run again with one
                # launch per step -- no cross-workgroup waits, about twice the time -- instead of failing the caller: the
                # reference's jit(fori_loop) cannot stall either.
                import warnings
                why = self.last_abort_code()
                warnings.warn("run_steps: the chained launch was stopped by a bounded wait (" + why + "); re-running the "
                              f"{int(num_steps)} steps with one launch per step", RuntimeWarning)
                check(lib.d3p_dpvi_logreg_set_run_form(1))
                try:
                    return self.run_steps(svi_state, get_batch, batchifier_state, first_batch, num_steps, check_status=True,
                                          _no_fallback=True, **kwargs)
                finally:
                    check(lib.d3p_dpvi_logreg_set_run_form(0))
            if aborted:
                raise _lib.D3PError("run_steps: a bounded wait ran out (the step kernel's workgroups did not make progress); the "
                                    "run was stopped and its state and losses are invalid -- " + self.last_abort_code())
        new_key = keybuf[num_steps & 1].reshape(4, 4)
        return DPSVIState((step, params, m, v), new_key, svi_state.observation_scale), (losses if num_steps == nl else losses[:num_steps])

    def last_run_status(self):
        """(aborted, nonfinite) of the last ``run_steps`` call, after synchronising the stream: ``aborted`` -- a bounded
        wait of the chained launch ran out and the run stopped advancing; ``nonfinite`` -- a partial sum was NaN / Inf, so
        parameters and losses turned NaN from that step on (the reference's float sums do the same)."""
        last = getattr(self, "_last_run", None)
        if last is None:
            return False, False
        model, src, ws, _ = last
        aborted, nonfinite = C.c_int32(0), C.c_int32(0)
        check(_lib.load().d3p_dpvi_logreg_run_status(stream_ptr(), C.byref(model), C.byref(src), ptr(ws), ws.numel(),
                                                     C.byref(aborted), C.byref(nonfinite)))
        self._last_abort_code = int(aborted.value) & 0xffffffff
        return bool(aborted.value), bool(nonfinite.value)

    def last_abort_code(self):
        """The code of the bounded wait that stopped the last run (0: none), as text: which wait, at which step of its
        launch (d3p_logreg_kernel.h, D3P_ABORT_*).  Valid after ``last_run_status()``."""
        return _lib.describe_abort(getattr(self, "_last_abort_code", 0))

    # ---------------------------------------------------------------- numpyro.infer.SVI.run (inherited by the reference's DPSVI)
    def run(self, rng_key, num_steps, *args, progress_bar=False, stable_update=False, init_state=None, **kwargs):
        """``numpyro.infer.SVI.run``, which the reference's DPSVI inherits: ``num_steps`` x ``update`` on the SAME arguments, from
        ``init(rng_key, *args)`` or ``init_state``; returns ``SVIRunResult(params, state, losses)``.  ``progress_bar`` is accepted and
        ignored (nothing is printed).  ``stable_update=True`` is refused: numpyro's ``stable_update`` evaluates the plain ELBO through
        the optimiser (``eval_and_stable_update``) -- in the reference it would BYPASS the clip / noise pipeline, i.e. not be private."""
        if stable_update:
            raise NotImplementedError("DPSVI.run(stable_update=True): numpyro's stable_update steps on the unclipped, noise-free ELBO "
                                      "gradient (it does not go through DPSVI.update); refused")
        state = self.init(rng_key, *args, **kwargs) if init_state is None else init_state
        losses = []
        for _ in range(int(num_steps)):
            state, loss = self.update(state, *args, **kwargs)
            losses.append(loss.reshape(()))
        dev = state.optim_state[1].device if isinstance(state.optim_state[1], torch.Tensor) else None
        return SVIRunResult(self.get_params(state), state, torch.stack(losses) if losses else torch.empty(0, dtype=torch.float32, device=dev))

    # ---------------------------------------------------------------- evaluate / accounting
    def evaluate(self, svi_state, *args, **kwargs):
        """ELBO loss of a batch at the current parameters (d3p/svi.py:436-449 -> numpyro SVI.evaluate)."""
        self._require_device_state(svi_state)
        if self._is_vae():
            _lib.require_device()
            lib = _lib.load()
            X = self._vae_flat(_batch_array(args[0]))
            B, D = X.shape
            jax_rng_key = self._rng_suite.convert_to_jax_rng_key(self._rng_suite.split(svi_state.rng_key, 1)[0]).contiguous()
            params = self.optim.get_params(svi_state.optim_state).contiguous()
            vm = self._vae_struct(D, kwargs, 1.0)
            vm.scale = vm.scale / B                  # plate(N, B) scales every site by N / B instead of N
            self._require_sizes(params.numel(), lib.d3p_vae_num_params(C.byref(vm)))
            ws = self._workspace(lib.d3p_dpvi_vae_workspace(C.byref(vm), B), X.device, "vae_step")
            loss = torch.empty(1, dtype=torch.float32, device=X.device)
            check(lib.d3p_vae_evaluate(stream_ptr(), C.byref(vm), ptr(params), ptr(X), B, ptr(jax_rng_key),
                                       ptr(kwargs.get("_eps")), ptr(loss), ptr(ws), ws.numel()))
            return loss[0]
        if self._is_gmm():
            _lib.require_device()
            lib = _lib.load()
            X = _batch_array(args[0])
            B, d = X.shape
            jax_rng_key = self._rng_suite.convert_to_jax_rng_key(self._rng_suite.split(svi_state.rng_key, 1)[0]).contiguous()
            params = self.optim.get_params(svi_state.optim_state).contiguous()
            gm = self._gmm_struct(d, kwargs, 1.0)
            self._require_sizes(params.numel(), gm.K + gm.K * d)
            ws = self._workspace(lib.d3p_gmm_evaluate_workspace(C.byref(gm), B), X.device, "gmm_eval")
            loss = torch.empty(1, dtype=torch.float32, device=X.device)
            check(lib.d3p_gmm_evaluate(stream_ptr(), C.byref(gm), ptr(params), ptr(X), B, ptr(jax_rng_key), ptr(loss),
                                       ptr(ws), ws.numel()))
            return loss[0]
        self._require_logreg()
        _lib.require_device()
        lib = _lib.load()
        X = _batch_array(args[0])
        y = self._labels(args)
        B, d = X.shape
        # we split to have the same seed as `update` given an svi_state (svi.py:446-447)
        jax_rng_key = self._rng_suite.convert_to_jax_rng_key(self._rng_suite.split(svi_state.rng_key, 1)[0]).contiguous()
        params = self.optim.get_params(svi_state.optim_state).contiguous()
        model = self._model_struct(d, kwargs, 1.0)
        self._require_sizes(params.numel(), 2 * self.model.latent_dim(d), B, y)
        ws = self._workspace(lib.d3p_logreg_evaluate_workspace(C.byref(model), B), X.device, "eval")
        loss = torch.empty(1, dtype=torch.float32, device=X.device)
        if isinstance(self.guide, MeanFieldGuide):   # two sample sites, each with its own key; parameters into the kernels' order
            kern = torch.empty_like(params)
            kern[MeanFieldGuide.tree_from_kernel(d, X.device)] = params
            sizes = (C.c_int32 * 2)(d, 1)
            check(lib.d3p_logreg_evaluate_sites(stream_ptr(), C.byref(model), ptr(kern), ptr(X), ptr(y), B, ptr(jax_rng_key), sizes, 2,
                                                ptr(loss), ptr(ws), ws.numel()))
            return loss[0]
        check(lib.d3p_logreg_evaluate(stream_ptr(), C.byref(model), ptr(params), ptr(X), ptr(y), B, ptr(jax_rng_key),
                                      ptr(loss), ptr(ws), ws.numel()))
        return loss[0]

    def _validate_epochs_and_iter(self, num_epochs, num_iter, q):
        """d3p/svi.py:451-456."""
        if num_epochs is not None:
            num_iter = num_epochs / q
        if num_iter is None:
            raise ValueError("A value must be supplied for either num_iter or num_epochs")
        return num_iter

    def get_epsilon(self, target_delta, q, num_epochs=None, num_iter=None):
        """d3p/svi.py:458-462 (accountant: d3p_amd.accountant, the restated Fourier accountant)."""
        num_iter = self._validate_epochs_and_iter(num_epochs, num_iter, q)
        from d3p_amd.accountant import get_epsilon_R
        return get_epsilon_R(target_delta, self._dp_scale, q, ncomp=num_iter)

    def get_delta(self, target_epsilon, q, num_epochs=None, num_iter=None):
        """d3p/svi.py:464-468 (accountant: d3p_amd.accountant, the restated Fourier accountant)."""
        num_iter = self._validate_epochs_and_iter(num_epochs, num_iter, q)
        from d3p_amd.accountant import get_delta_R
        return get_delta_R(target_epsilon, self._dp_scale, q, ncomp=num_iter)


def _dbg_uniform(jax_key, n, lo, hi):
    """uniform(lo, hi) from a threefry key without importing the warning-emitting debug module."""
    out = torch.empty(max(n, 1), dtype=torch.float32, device=jax_key.device)
    check(_lib.load().d3p_tf_uniform(stream_ptr(), ptr(jax_key.contiguous()), n, float(lo), float(hi), ptr(out)))
    return out[:n]
