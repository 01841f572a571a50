#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

The reference (DPBayes/d3p) cannot be imported in the build container: jax, numpyro and
jax-chacha-prng are not installed and there is no network, so no vector can be
captured from the reference itself.  The fixtures therefore have two sources, kept apart:

  external_vectors.json   published known answers that do NOT come from this repository:
                          RFC 8439 (ChaCha20 block function), Random123 (threefry2x32-20 KATs),
                          values printed in the JAX documentation / asserted in JAX's own tests
                          (jax.random.split / random_bits / normal), the reference's own
                          known-answer tests (d3p tests/, cited per entry).
  *.npz                   outputs of oracle/ (the CPU restatement) on fixed seeds.  They freeze the
                          build's own ChaCha key layout and the full update dataflow so that the HIP
                          path and any later change of the oracle are checked against committed data.

Run:  python tests/golden/make_golden.py      (rewrites the .npz files; the json is hand-written)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def rng_vectors():
    out = {}
    for seed in (0, 1, 9782346):
        k = O.PRNGKey(seed)
        out[f"key_{seed}"] = k
        out[f"split3_{seed}"] = O.split(k, 3)
        out[f"fold_in5_{seed}"] = O.fold_in(k, 5)
        out[f"bits32_{seed}"] = O.random_bits(k, 32, (40,))
        out[f"bits8_{seed}"] = O.random_bits(k, 8, (7,))
        out[f"bits16_{seed}"] = O.random_bits(k, 16, (5,))
        out[f"bits64_{seed}"] = O.random_bits(k, 64, (3,))
        out[f"uniform_{seed}"] = O.uniform(k, (33,))
        out[f"normal_{seed}"] = O.normal(k, (33,))
        out[f"randint_0_100_{seed}"] = O.randint(k, (50,), 0, 100)
        out[f"jaxkey_{seed}"] = O.convert_to_jax_rng_key(k)
    out["px_eps_B4_D6"] = O.px_eps(O.convert_to_jax_rng_key(O.PRNGKey(0)), 4, 6)
    np.savez(os.path.join(HERE, "rng_vectors.npz"), **out)


def sampler_vectors():
    out = {}
    for cap, n in [(1, 1), (2, 2), (100, 100), (105, 50), (10**6, 64), (10**8, 64)]:
        out[f"feistel_{cap}_{n}"] = O.feistel_sample(O.PRNGKey(cap + n), cap, n)
    for N, q, cutoff, suppress in [(100, 0.1, 100, 0), (105, 0.3, 39, 0), (105, 0.3, 20, 1), (4097, 0.05, 260, 0)]:
        idx, nsel, nvalid = O.poisson_select(O.PRNGKey(N + cutoff), q, N, cutoff, bool(suppress))
        out[f"poisson_{N}_{cutoff}_{suppress}_idx"] = idx
        out[f"poisson_{N}_{cutoff}_{suppress}_counts"] = np.array([nsel, nvalid], np.uint32)
    np.savez(os.path.join(HERE, "sampler_vectors.npz"), **out)


def update_vectors():
    """One full DPSVI.update at B=16, d=8 (+ intercept) with every intermediate of svi.py:395-434."""
    r = np.random.default_rng(20261003)
    B, d, N = 16, 8, 1000
    for name, icpt in (("nointercept", False), ("intercept", True)):
        D = d + int(icpt)
        X = r.normal(size=(B, d)).astype(np.float32)
        y = (r.random(B) < 0.5).astype(np.float32)
        mask = (np.arange(B) < 13).astype(np.float32)
        loc = (0.3 * r.normal(size=D)).astype(np.float32)
        unc = (0.5 * r.normal(size=D) - 1).astype(np.float32)
        spec = O.logreg_spec(d, icpt, 1.5, 3.0, lik_scale=N, obs_scale=N)
        key = O.PRNGKey(4242)
        ks = O.split(key, 3)
        jax_key = O.convert_to_jax_rng_key(ks[1])
        eps = O.px_eps(jax_key, B, D)
        px_loss, px_grads, n, factor = O.logreg_px_grads(spec, loc, unc, X, y, eps, mask)
        clipped = O.clip_rows(px_grads, 0.7)
        loss, avg = O.combine(clipped, px_loss)
        pert = O.perturb(ks[2], avg, [D, D], 1.3, 0.7, n, N, factor)
        st = O.LogregState(key, D, loc, unc)
        hy = O.Hyper(0.7, 1.3, 1e-2, 0.9, 0.999, 1e-8)
        loss2, grad2 = O.logreg_update(spec, hy, st, X, y, mask)
        assert np.allclose(loss, loss2) and np.allclose(pert, grad2)
        np.savez(os.path.join(HERE, f"update_B16_d8_{name}.npz"), X=X, y=y, mask=mask, loc=loc, unc=unc,
                 key=key, next_key=st.key.reshape(4, 4), jax_key=jax_key, eps=eps, px_loss=px_loss,
                 px_grads=px_grads, num_elements=np.float32(n), factor=np.float32(factor), clipped=clipped,
                 avg=avg, loss=np.float32(loss), perturbed=pert, new_params=st.params, adam_m=st.m, adam_v=st.v,
                 hyper=np.array([0.7, 1.3, 1e-2, 0.9, 0.999, 1e-8], np.float32),
                 model=np.array([d, int(icpt), 1.5, 3.0, N, N], np.float32))


if __name__ == "__main__":
    rng_vectors()
    sampler_vectors()
    update_vectors()
    print("golden fixtures written to", HERE)
