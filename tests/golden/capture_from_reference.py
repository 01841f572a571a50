#!/usr/bin/env python3
"""Capture golden vectors FROM THE REAL REFERENCE (d3p + jax-chacha-prng + numpyro + jax) for the pieces this build could
only restate (DESIGN.md section 2 "parity unpinned"): the ChaCha20 key / nonce / counter layout of
`chacha.random` (PRNGKey, split, fold_in, random_bits, uniform), d3p.random.normal / randint on top of it, the Feistel
sampler's indices under that stream, Poisson selection, numpyro's AutoDiagonalNormal initialisation and one DPSVI.update.

This container has none of those packages (and no network), so the script has NEVER been run here and no file it writes is
committed.  It exists so that whoever has an environment with the reference's dependencies can run

    pip install d3p==0.2.0        # pulls jax, numpyro, jax-chacha-prng, fourier-accountant
    python tests/golden/capture_from_reference.py            # writes tests/golden/reference_vectors.json

after which `pytest tests/test_reference_vectors.py` compares the CPU oracle (oracle/) with the captured values, item by
item, and names every place where this build's own layout differs.  Adopting the real layout is then a change in ONE place
per side: the "key layout" block of oracle/d3p_oracle.c (d3po_key_from_bytes, derive_child, d3po_random_words) and the
block of the same name in d3p_amd/csrc/d3p_device.h; the numpyro key chain is d3po_px_sample_key / px_sample_key.

The script only CALLS the reference; nothing of its source is copied here.  Output: JSON (lists of ints / floats)."""
import json
import os
import sys


def main():
    try:
        import jax
        import jax.numpy as jnp
        import numpy as np
        import numpyro
        import numpyro.distributions as dist
        from numpyro.infer import Trace_ELBO
        from numpyro.infer.autoguide import AutoDiagonalNormal
        from numpyro.optim import Adam
        import d3p.random as rng
        from d3p.minibatch import poisson_sample_idxs
        from d3p.svi import DPSVI
        from d3p.util import sample_from_array
    except ImportError as e:                      # the expected outcome in the build container
        print(f"reference stack not importable here ({e}); nothing captured", file=sys.stderr)
        return 2

    def ints(a):
        return [int(v) for v in np.asarray(a).ravel()]

    def floats(a):
        return [float(v) for v in np.asarray(a, dtype=np.float64).ravel()]

    out = {"versions": {"jax": jax.__version__, "numpyro": numpyro.__version__}}
    try:
        import chacha
        out["versions"]["jax-chacha-prng"] = getattr(chacha, "__version__", "unknown")
    except ImportError:
        pass

    # ---- rng_suite on the ChaCha20 stream (d3p/random/__init__.py:28-155)
    k0 = rng.PRNGKey(0)
    k = rng.PRNGKey(98734)
    r = {"PRNGKey_0": ints(k0), "PRNGKey_98734": ints(k),
         "PRNGKey_bytes_00_1f": ints(rng.PRNGKey(bytes(range(32)))),
         "split_3": ints(rng.split(k, 3)), "fold_in_5": ints(rng.fold_in(k, 5)),
         "convert_to_jax_rng_key": ints(rng.convert_to_jax_rng_key(k))}
    for w in (8, 16, 32, 64):
        try:
            r[f"random_bits_{w}_x20"] = ints(rng.random_bits(k, w, (20,)))
        except Exception as e:  # noqa: BLE001 -- e.g. 64-bit without jax_enable_x64
            r[f"random_bits_{w}_x20"] = f"error: {e}"
    r["uniform_x20"] = floats(rng.uniform(k, (20,)))
    r["uniform_m1_1_x20"] = floats(rng.uniform(k, (20,), jnp.float32, -1.0, 1.0))
    r["normal_x20"] = floats(rng.normal(k, (20,)))
    for name, lo, hi, dt in (("int32_0_10", 0, 10, np.int32), ("int32_8_1033", 8, 8 + 2**10 + 1, np.int32),
                             ("int8_full", -2**7, 2**7, np.int8), ("int16_0_32768", 0, 2**15, np.int16)):
        r[f"randint_{name}_x40"] = ints(rng.randint(k, (40,), lo, hi, dt))
    out["rng"] = r

    # ---- Feistel sampler and Poisson selection under that stream (d3p/util.py:216-301, d3p/minibatch.py:29-39)
    s = {}
    for N, n in ((1, 1), (2, 2), (100, 100), (105, 30), (10**6, 32), (10**8, 32)):
        s[f"feistel_N{N}_n{n}"] = ints(sample_from_array(k, jnp.arange(N), n, 0, rng_suite=rng))
    sel, cnt = poisson_sample_idxs(k, 0.1, 1000, rng, 150)           # (rng_key, q, N, rng_suite, cutoff_size)
    s["poisson_N1000_q0.1_max150"] = {"idxs": ints(sel), "count": int(cnt)}
    out["sampling"] = s

    # ---- one DPSVI.update: logistic regression + AutoDiagonalNormal (README.md:89-99), B = 16, d = 8
    d, B, N = 8, 16, 1000
    g = np.random.default_rng(0)
    X = g.normal(size=(B, d)).astype(np.float32)
    y = (g.random(B) < 0.5).astype(np.float32)

    def model(X, y, N):
        w = numpyro.sample("w", dist.Normal(jnp.zeros(d), 1.0).to_event(1))
        with numpyro.plate("batch", N, subsample_size=X.shape[0]):
            numpyro.sample("y", dist.Bernoulli(logits=X @ w), obs=y)

    guide = AutoDiagonalNormal(model)
    svi = DPSVI(model, guide, Adam(1e-3), Trace_ELBO(), clipping_threshold=1.0, dp_scale=1.0, N=N)
    state = svi.init(k0, jnp.asarray(X), jnp.asarray(y))
    u = {"X": floats(X), "y": floats(y), "shape": [B, d, N],
         "init_params_unconstrained": {n: floats(v) for n, v in svi.optim.get_params(state.optim_state).items()},
         "init_params_constrained": {n: floats(v) for n, v in svi.get_params(state).items()},
         "observation_scale": float(state.observation_scale)}
    state1, _keys = svi._split_rng_key(state, 2)
    _, px_loss, px_grads, n_el, factor = svi._compute_per_example_gradients(state1, _keys[0], jnp.asarray(X), jnp.asarray(y))
    u["px_loss"] = floats(px_loss)
    u["px_grads"] = {n: floats(v) for n, v in px_grads.items()}
    new_state, loss = svi.update(state, jnp.asarray(X), jnp.asarray(y))
    u["loss"] = float(loss)
    u["params_after_unconstrained"] = {n: floats(v) for n, v in svi.optim.get_params(new_state.optim_state).items()}
    u["rng_key_after"] = ints(new_state.rng_key)
    out["update_logreg_B16_d8"] = u

    # ---- the same update under the logistic-regression example's OWN model and guide (examples/logistic_regression.py:49-86: sample
    # sites 'w' and 'intercept', four parameter leaves with exp scales): pins the order of the leaves, the one-key-per-leaf perturbation
    # and the seed handler's per-site keys (oracle: O.meanfield_logreg_update / O.px_eps_sites)
    def model2(batch_X, batch_y=None, num_obs_total=None):
        z_w = numpyro.sample("w", dist.Normal(jnp.zeros((d,)), jnp.ones((d,))))
        z_intercept = numpyro.sample("intercept", dist.Normal(0, 1))
        logits = batch_X.dot(z_w) + z_intercept
        with numpyro.plate("batch", num_obs_total, batch_X.shape[0]):
            return numpyro.sample("obs", dist.Bernoulli(logits=logits), obs=batch_y)

    def guide2(batch_X, batch_y=None, num_obs_total=None):
        w_loc = numpyro.param("w_loc", jnp.zeros((d,)))
        w_std = jnp.exp(numpyro.param("w_std_log", jnp.zeros((d,))))
        numpyro.sample("w", dist.Normal(w_loc, w_std))
        i_loc = numpyro.param("intercept_loc", 0.)
        i_std = jnp.exp(numpyro.param("intercept_std_log", 0.))
        numpyro.sample("intercept", dist.Normal(i_loc, i_std))

    svi2 = DPSVI(model2, guide2, Adam(1e-2), Trace_ELBO(), clipping_threshold=1.0, dp_scale=1.0, num_obs_total=N)
    state = svi2.init(k0, jnp.asarray(X), jnp.asarray(y))
    u2 = {"X": floats(X), "y": floats(y), "shape": [B, d, N], "observation_scale": float(state.observation_scale),
          "leaf_order": sorted(svi2.optim.get_params(state.optim_state)),   # tree_flatten of a dict: sorted keys (svi.py:490)
          "init_params": {n: floats(v) for n, v in svi2.optim.get_params(state.optim_state).items()}}
    state1, _keys = svi2._split_rng_key(state, 2)
    _, px_loss, px_grads, n_el, factor = svi2._compute_per_example_gradients(state1, _keys[0], jnp.asarray(X), jnp.asarray(y))
    u2["px_loss"] = floats(px_loss)
    u2["px_grads"] = {n: floats(v) for n, v in px_grads.items()}
    new_state, loss = svi2.update(state, jnp.asarray(X), jnp.asarray(y))
    u2["loss"] = float(loss)
    u2["params_after"] = {n: floats(v) for n, v in svi2.optim.get_params(new_state.optim_state).items()}
    u2["rng_key_after"] = ints(new_state.rng_key)
    u2["evaluate_after"] = float(svi2.evaluate(new_state, jnp.asarray(X), jnp.asarray(y)))
    out["update_logreg_example_guide_B16_d8"] = u2

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst)
    return 0


if __name__ == "__main__":
    sys.exit(main())
