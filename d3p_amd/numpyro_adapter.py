"""Import-guarded NumPyro front-end: "host Python traces the NumPyro model / guide once to a flat parameter layout"
(BASELINE.json north_star; d3p/svi.py:213-236 traces through numpyro's ``SVI.init``).

numpyro and jax are absent from the build container, so this module is written against numpyro's PUBLIC
tracing API (``numpyro.handlers.seed`` / ``trace`` and the site dictionaries they yield) and is split in two:

* ``trace_model(model, *args, **kwargs)`` -- the only function that imports numpyro: runs the model once under
  ``handlers.trace(handlers.seed(model, 0))`` and reduces every sample site to a plain record
  ``{name, dist, shape, event_dim, is_observed, scale, plate_sizes, params}``;
* ``spec_from_sites(records)`` -- pure Python: recognises the model families this build has kernels for and returns the
  declarative spec (``d3p_amd.models``) + the flat latent layout.  Tested without numpyro by feeding it records written
  the way numpyro's trace writes them (tests/test_host_logic.py).

``from_numpyro(model, guide, ...)`` = both steps + the guide check; models outside the built families raise
``D3PError`` naming the sites that could not be mapped (the reference accepts ANY model / guide, d3p/svi.py:213-236,
:265-281: that generality needs a tracing compiler and is out of this build's scope).

Flat layout (numpyro's, restated): AutoDiagonalNormal packs the latent sites with ``jax.flatten_util.ravel_pytree`` of a
dict, i.e. in SORTED order of the site names; for ``{'intercept', 'w'}`` (examples/logistic_regression.py:61-62) that puts
the intercept FIRST, while this build's kernels keep it as the last latent column.  ``FlatLayout.to_build_order`` /
``from_build_order`` are the permutation between the two; the guide noise of latent j in numpyro's order is word j of
``jax.random.normal(key, (D,))`` (UNPINNED -- DESIGN.md section 2 -- until the capture script has been run)."""
from . import _lib
from .models import (AutoDiagonalNormal, DiagonalNormalGuide, GaussianMean, GaussianMixtureGuide,
                     GaussianMixtureModel, LogisticRegression, MeanFieldGuide)


class FlatLayout:
    """Latent sites in numpyro's flat order (sorted names) and the permutation to this build's column order."""

    def __init__(self, sites, build_order):
        self.sites = list(sites)                  # [(name, numel)] in numpyro's (sorted) order
        self.build_order = list(build_order)      # site names in this build's column order
        self.D = sum(n for _, n in self.sites)
        off, start = 0, {}
        for name, n in self.sites:
            start[name] = (off, n)
            off += n
        idx = []
        for name in self.build_order:
            o, n = start[name]
            idx.extend(range(o, o + n))
        self.to_build = idx                       # build column c holds numpyro latent to_build[c]

    def to_build_order(self, flat):
        """numpyro-ordered flat vector (list / tensor) -> this build's column order."""
        return flat[self.to_build] if hasattr(flat, "shape") else [flat[i] for i in self.to_build]

    def from_build_order(self, cols):
        inv = [0] * self.D
        for c, i in enumerate(self.to_build):
            inv[i] = c
        return cols[inv] if hasattr(cols, "shape") else [cols[i] for i in inv]


def trace_model(model, *args, **kwargs):
    """Run ``model`` once under numpyro's seed + trace handlers and return its sample sites as plain records."""
    try:
        import numpyro  # noqa: F401
        from numpyro import handlers
    except ImportError as e:
        raise _lib.D3PError("d3p_amd.numpyro_adapter.trace_model needs numpyro (absent here): " + str(e))
    tr = handlers.trace(handlers.seed(model, 0)).get_trace(*args, **kwargs)
    records = []
    for name, site in tr.items():
        if site["type"] != "sample":
            continue
        fn = site["fn"]
        base = fn
        while hasattr(base, "base_dist"):         # Independent / to_event / ExpandedDistribution wrappers
            base = base.base_dist
        params = {}
        for p in ("loc", "scale", "concentration", "rate", "logits", "probs"):
            v = getattr(base, p, None)
            if v is not None and getattr(v, "size", 2) == 1:
                params[p] = float(v.reshape(()))
            elif v is not None:
                try:
                    import numpy as np
                    a = np.asarray(v)
                    if a.size and float(a.max()) == float(a.min()):
                        params[p] = float(a.ravel()[0])
                except Exception:  # noqa: BLE001 -- traced / abstract values stay unrecorded
                    pass
        records.append({
            "name": name, "dist": type(base).__name__, "shape": tuple(site["value"].shape),
            "event_dim": int(getattr(fn, "event_dim", 0)), "is_observed": bool(site["is_observed"]),
            "scale": None if site.get("scale") is None else float(site["scale"]),
            "plate_sizes": [(f.name, int(f.size)) for f in site.get("cond_indep_stack", ())],
            "params": params,
        })
    return records


def _numel(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


def spec_from_sites(records):
    """Site records -> (model spec, FlatLayout, num_obs_total or None).  Raises D3PError for unknown structures."""
    latent = [r for r in records if not r["is_observed"]]
    observed = [r for r in records if r["is_observed"]]
    if len(observed) != 1:
        raise _lib.D3PError(f"numpyro adapter: expected exactly one observed site, found {[r['name'] for r in observed]}")
    obs = observed[0]
    n_total = obs["plate_sizes"][0][1] if obs["plate_sizes"] else None
    names = sorted(r["name"] for r in latent)
    by_name = {r["name"]: r for r in latent}

    def normal_zero(r):
        return r["dist"] == "Normal" and r["params"].get("loc", None) in (0.0, None) and "scale" in r["params"]

    # ---- Bayesian logistic regression (README.md:89-99; examples/logistic_regression.py:49-66)
    if obs["dist"] in ("Bernoulli", "BernoulliLogits", "BernoulliProbs") and 1 <= len(latent) <= 2:
        vec = [r for r in latent if len(r["shape"]) == 1 and r["shape"][0] >= 1 and normal_zero(r)]
        sca = [r for r in latent if r["shape"] == () and normal_zero(r)]
        if len(vec) == 1 and len(vec) + len(sca) == len(latent):
            w = vec[0]
            spec = LogisticRegression(d=w["shape"][0], prior_scale=w["params"]["scale"], intercept=bool(sca),
                                      intercept_prior_scale=sca[0]["params"]["scale"] if sca else 1.0)
            build = [w["name"]] + ([sca[0]["name"]] if sca else [])        # this build: features first, intercept last
            return spec, FlatLayout([(n, _numel(by_name[n]["shape"])) for n in names], build), n_total
    # ---- Gaussian observations with a latent mean (examples/simple_gaussian_posterior.py:51-65)
    if obs["dist"] == "Normal" and len(latent) == 1 and normal_zero(latent[0]) and len(latent[0]["shape"]) == 1 \
            and obs["shape"][-1:] == latent[0]["shape"] and "scale" in obs["params"]:
        mu = latent[0]
        spec = GaussianMean(d=mu["shape"][0], prior_scale=mu["params"]["scale"], obs_scale=obs["params"]["scale"])
        return spec, FlatLayout([(mu["name"], mu["shape"][0])], [mu["name"]]), n_total
    # ---- mixture model of examples/gaussian_mixture_model.py:51-68
    if obs["dist"] == "GaussianMixture" and set(names) == {"pis", "mus", "sigs"} and by_name["pis"]["dist"] == "Dirichlet" \
            and by_name["sigs"]["dist"] == "InverseGamma" and normal_zero(by_name["mus"]):
        k, d = by_name["mus"]["shape"]
        spec = GaussianMixtureModel(k=k, d=d, prior_mu_scale=by_name["mus"]["params"]["scale"])
        return spec, FlatLayout([(n, _numel(by_name[n]["shape"])) for n in names], ["pis", "mus", "sigs"]), n_total
    raise _lib.D3PError("numpyro adapter: the traced model is none of the families this build has kernels for "
                        "(logistic regression, Gaussian mean, Gaussian mixture; the VAE is declared with VAEModel); "
                        f"latent sites {[(r['name'], r['dist'], r['shape']) for r in latent]}, observed "
                        f"{(obs['name'], obs['dist'], obs['shape'])}")


def trace_guide(guide, *args, **kwargs):
    """Run a hand-written guide once under numpyro's seed + trace handlers: its ``param`` and ``sample`` statements IN PROGRAM ORDER as
    plain records ``{name, type, dist, shape}`` (the order of the sample statements is the order in which numpyro's seed handler hands
    out the sites' keys: DESIGN.md section 4)."""
    try:
        import numpyro  # noqa: F401
        from numpyro import handlers
    except ImportError as e:
        raise _lib.D3PError("d3p_amd.numpyro_adapter.trace_guide needs numpyro (absent here): " + str(e))
    tr = handlers.trace(handlers.seed(guide, 0)).get_trace(*args, **kwargs)
    records = []
    for name, site in tr.items():
        if site["type"] not in ("sample", "param"):
            continue
        base = site.get("fn")
        while hasattr(base, "base_dist"):
            base = base.base_dist
        records.append({"name": name, "type": site["type"], "dist": type(base).__name__ if site["type"] == "sample" else None,
                        "shape": tuple(getattr(site["value"], "shape", ()))})
    return records


def guide_spec_from_sites(spec, guide_records):
    """The records of a hand-written guide (trace_guide) -> the guide spec this build has for it, or D3PError.  Recognised: the
    examples' mean-field guides, whose every sample site ``s`` is ``Normal(param(s + '_loc'), exp(param(s + '_std_log')))``:
    one site over the model's latent vector -> DiagonalNormalGuide (examples/simple_gaussian_posterior.py:67-82); the sites ``w``
    THEN ``intercept`` of the logistic regression -> MeanFieldGuide (examples/logistic_regression.py:67-86: four leaves, one
    perturbation key per leaf, the sites' eps keys in this program order)."""
    samples = [r for r in guide_records if r["type"] == "sample"]
    params = {r["name"]: r for r in guide_records if r["type"] == "param"}
    names = [r["name"] for r in samples]
    want = set()
    for r in samples:
        if r["dist"] != "Normal":
            raise _lib.D3PError(f"numpyro adapter: guide site {r['name']} is {r['dist']}, not Normal")
        want |= {r["name"] + "_loc", r["name"] + "_std_log"}
    if set(params) != want:
        raise _lib.D3PError(f"numpyro adapter: guide parameters {sorted(params)} are not the <site>_loc / <site>_std_log of its sites {names}")
    for r in samples:
        if params[r["name"] + "_loc"]["shape"] != r["shape"] or params[r["name"] + "_std_log"]["shape"] != r["shape"]:
            raise _lib.D3PError(f"numpyro adapter: parameter shapes of guide site {r['name']} do not match the site's shape {r['shape']}")
    if sorted(names) != sorted(spec.site_names()):
        raise _lib.D3PError(f"numpyro adapter: guide sites {names} are not the model's latent sites {list(spec.site_names())}")
    if len(names) == 1:
        return DiagonalNormalGuide(spec, site=names[0])
    if isinstance(spec, LogisticRegression) and spec.intercept:
        gspec = MeanFieldGuide(spec)
        order = [n for n, _ in gspec.sites(spec.d)]
        # the model's OWN names may differ from 'w' / 'intercept': the vector site first, the scalar site second
        vec_first = len(samples[0]["shape"]) == 1 and samples[1]["shape"] == ()
        if not vec_first:
            raise _lib.D3PError(f"numpyro adapter: the guide samples {names}; this build draws the sites' eps keys in the order "
                                f"{order} (vector site, then intercept) -- another program order gives the sites other keys")
        return gspec
    raise _lib.D3PError(f"numpyro adapter: no built guide for the sample sites {names}")


def from_numpyro(model, guide, *args, **kwargs):
    """numpyro model (+ ``numpyro.infer.autoguide.AutoDiagonalNormal`` instance or one of the examples' hand-written
    guides, traced and recognised by their sites and parameters) -> ``(model_spec, guide_spec, FlatLayout, num_obs_total)`` for
    ``d3p_amd.svi.DPSVI``."""
    records = trace_model(model, *args, **kwargs)
    spec, layout, n_total = spec_from_sites(records)
    gname = type(guide).__name__
    if isinstance(spec, GaussianMixtureModel):
        gspec = GaussianMixtureGuide(spec)
    elif gname == "AutoDiagonalNormal":
        gspec = AutoDiagonalNormal(spec, init_scale=float(getattr(guide, "_init_scale", 0.1)))
    elif callable(guide):
        gspec = guide_spec_from_sites(spec, trace_guide(guide, *args, **kwargs))
    else:
        raise _lib.D3PError(f"numpyro adapter: guide {gname} is not AutoDiagonalNormal nor one of the examples' guides")
    return spec, gspec, layout, n_total
