"""Pins the CPU oracle (oracle/) before it is trusted as the checker of the HIP path.

Sources of truth used here (none of them is this repository): RFC 8439, OpenSSL's chacha20,
Random123 / JAX-published threefry values, scipy.special.erfinv (float64), torch autograd on an
ELBO written with torch.distributions, and every known-answer test the reference holds for the
path (tests/golden/external_vectors.json cites them).  What remains UNPINNED (the ChaCha key
layout of jax-chacha-prng, numpyro's key plumbing) is stated in oracle/d3p_oracle.c's header.
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest
import scipy.stats
from scipy.special import erfinv

HERE = os.path.dirname(os.path.abspath(__file__))
EXT = json.load(open(os.path.join(HERE, "golden", "external_vectors.json")))


def _state(key_bytes, counter, nonce_bytes):
    st = np.zeros(16, np.uint32)
    st[:4] = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574]
    st[4:12] = np.frombuffer(key_bytes, "<u4")
    st[12] = counter
    st[13:16] = np.frombuffer(nonce_bytes, "<u4")
    return st


def test_chacha20_block_rfc8439(O):
    v = EXT["rfc8439_2_3_2"]
    out = O.chacha20_block(_state(bytes.fromhex(v["key_bytes"]), v["counter"], bytes.fromhex(v["nonce_bytes"])))
    assert [f"{int(w):08x}" for w in out] == v["output_words"]
    ks = O.chacha20_block(_state(bytes(32), 0, bytes(12))).astype("<u4").tobytes()
    assert ks[:16].hex() == EXT["rfc8439_a1_1"]["keystream_prefix"]
    # the keystream of PRNGKey(0) (zero key, counter 0, nonce 0) starts with the same bytes
    assert O.random_bits(O.PRNGKey(0), 8, (16,)).tobytes().hex() == EXT["rfc8439_a1_1"]["keystream_prefix"]


@pytest.mark.skipif(shutil.which("openssl") is None, reason="openssl not installed")
def test_chacha20_keystream_vs_openssl(O):
    r = np.random.default_rng(0)
    for _ in range(3):
        key = r.integers(0, 256, 32, dtype=np.uint8).tobytes()
        nonce = r.integers(0, 256, 12, dtype=np.uint8).tobytes()
        ctr = int(r.integers(0, 2**31))
        st = _state(key, ctr, nonce).reshape(4, 4)
        mine = O.random_bits(st, 32, (16 * 5,)).astype("<u4").tobytes()
        iv = ctr.to_bytes(4, "little") + nonce
        p = subprocess.run(["openssl", "enc", "-chacha20", "-K", key.hex(), "-iv", iv.hex()], input=bytes(len(mine)),
                           capture_output=True)
        if p.returncode != 0:
            pytest.skip("openssl has no chacha20")
        assert p.stdout == mine


def test_threefry_and_jax_layouts(O):
    for c in EXT["threefry2x32_20"]["cases"]:
        k, ctr = [int(x, 16) for x in c["key"]], [int(x, 16) for x in c["ctr"]]
        assert [f"{int(w):08x}" for w in O.threefry2x32(k[0], k[1], ctr[0], ctr[1])] == c["out"]
    assert O.tf_split([0, 0], 2).tolist() == EXT["jax_split_prngkey0"]["value"]
    assert O.tf_random_words([0, 1701], 3).tolist() == EXT["jax_random_bits_1701"]["value"]
    assert abs(float(O.tf_normal([0, 0], 1)[0]) - EXT["jax_normal_prngkey0"]["value"]) < 1e-7
    # jax fold_in(key, data) = threefry_2x32(key, [0, data])
    assert O.tf_fold_in([7, 9], 5).tolist() == O.threefry2x32(7, 9, 0, 5).tolist()


def test_erfinv_and_normal_against_float64(O):
    u = np.concatenate([np.linspace(-0.999999, 0.999999, 20001), [np.nextafter(np.float32(-1), np.float32(0))]])
    u = u.astype(np.float32)
    got = O.erfinv_f32(u)
    exp = erfinv(u.astype(np.float64))
    assert np.max(np.abs(got - exp) / np.maximum(np.abs(exp), 1e-3)) < 1e-6
    key = O.PRNGKey(3)
    lo = np.nextafter(np.float32(-1), np.float32(0))
    z = O.normal(key, (100000,))
    t = np.sqrt(2.0) * erfinv(O.uniform(key, (100000,), lo, 1.0).astype(np.float64))
    assert np.max(np.abs(z - t) / np.maximum(np.abs(t), 1e-3)) < 1e-6
    assert scipy.stats.kstest(z, "norm").pvalue > 0.01 and abs(z.mean()) < 5 / np.sqrt(z.size)


def test_random_bits_widths_are_views_of_one_keystream(O):
    k = O.PRNGKey(11)
    raw = O.random_bits(k, 8, (64,)).tobytes()
    assert O.random_bits(k, 16, (32,)).astype("<u2").tobytes() == raw
    assert O.random_bits(k, 32, (16,)).astype("<u4").tobytes() == raw
    assert O.random_bits(k, 64, (8,)).astype("<u8").tobytes() == raw
    # split / fold_in / random_bits never collide (domain separation by nonce tag)
    kids = O.split(k, 4).reshape(4, 16)
    assert len({bytes(c) for c in kids} | {bytes(O.fold_in(k, 0).ravel())}) == 5
    assert not np.array_equal(kids[0, 4:12], O.random_bits(k, 32, (8,)))
    assert np.all(kids[:, 12:] == 0) and np.all(kids[:, :4] == k.ravel()[:4])


def test_prngkey_seed_forms(O):
    assert np.array_equal(O.PRNGKey(5), O.PRNGKey((5).to_bytes(32, "big")))
    assert np.array_equal(O.PRNGKey(b"ab"), O.PRNGKey(b"ab" + bytes(30)))
    assert np.array_equal(O.PRNGKey(2**256 + 7), O.PRNGKey(7))
    with pytest.raises(ValueError):
        O.PRNGKey(bytes(33))


def test_randint_reference_behaviour(O):
    ka = EXT["d3p_known_answers"]["randint_single_value"]
    assert np.all(O.randint(O.PRNGKey(1), (100,), ka["minval"], ka["maxval"]) == ka["value"])
    x = O.randint(O.PRNGKey(2), (20000,), 0, 10)
    assert x.min() == 0 and x.max() == 9
    assert scipy.stats.chisquare(np.bincount(x, minlength=10)).pvalue > 0.01
    x = O.randint(O.PRNGKey(3), (5000,), -5, 1 << 15)
    assert x.min() >= -5 and x.max() < (1 << 15)


def test_feistel_permutation_properties(O):
    """reference tests/test_util.py:331-373 + the structural quirks of util.py:229-301"""
    k = O.PRNGKey(0)
    assert np.unique(O.feistel_sample(k, 10**6, 978)).size == 978
    for n in (100, 99, 1):
        s = O.feistel_sample(O.PRNGKey(n), 100, n)
        assert np.unique(s).size == n and s.max() < 100
    assert np.array_equal(np.sort(O.feistel_sample(k, 105, 105)), np.arange(105))
    assert O.feistel_sample(k, 1, 1).tolist() == [0]
    assert O.feistel_sample(k, 2, 2).tolist() == [0, 1]   # bits = 1: lower half empty -> identity
    full = O.feistel_sample(O.PRNGKey(9), 1 << 12, 1 << 12)
    assert np.array_equal(np.sort(full), np.arange(1 << 12))
    assert not np.array_equal(full, np.arange(1 << 12))


def test_poisson_select_semantics(O):
    ka = EXT["d3p_known_answers"]
    N, q = 100, 0.1
    key = O.PRNGKey(0)
    u = O.uniform(key, (N,))
    sel = np.nonzero(u <= np.float32(q))[0]
    idx, nsel, nvalid = O.poisson_select(key, q, N, N)
    assert nsel == sel.size == nvalid
    assert np.array_equal(idx[:nsel], sel[::-1])                      # descending selected indices
    assert np.array_equal(idx[nsel:], np.setdiff1d(np.arange(N), sel)[::-1])
    # argsort(bool)[::-1][:cutoff] with a stable sort (minibatch.py:37)
    assert np.array_equal(idx, np.argsort(u <= np.float32(q), kind="stable")[::-1])
    idx, nsel, nvalid = O.poisson_select(key, 0.9, N, 10)
    assert nvalid == 10 and np.array_equal(idx, np.nonzero(O.uniform(key, (N,)) <= np.float32(0.9))[0][::-1][:10])
    assert O.poisson_select(key, 0.9, N, 10, suppress=True)[2] == 0
    assert N // int(q * N) == ka["poisson_num_batches"]["num_batches"]
    pq = ka["poisson_quantile"]
    assert int(scipy.stats.poisson(pq["N"] * pq["q"]).ppf(pq["mass"])) == pq["max_batch_size"]
    sizes = [O.poisson_select(O.fold_in(key, i), q, N, N)[1] for i in range(1000)]
    assert abs(np.mean(sizes) - q * N) < 4 * np.sqrt(q * (1 - q) * N / 1000)


def test_gradient_manipulator_known_answers(O):
    ka = EXT["d3p_known_answers"]
    tree = [np.ones(s, np.float32) for s in ka["full_norm_tree"]["shapes"]] + [()]
    assert abs(O.full_norm(tree) - ka["full_norm_tree"]["value"]) < 1e-5
    assert O.full_norm(None) == 0 and O.full_norm([]) == 0 and O.full_norm(()) == 0
    with pytest.raises(ValueError):
        O.clip_rows(np.ones((2, 3), np.float32), 0.0)
    c = ka["clip_norms"]
    g = np.concatenate([np.repeat([1.0, 0.0], 10).reshape(2, 10), np.repeat([0.0, 1.0], 2).reshape(2, 2)], axis=1)
    assert np.allclose(np.linalg.norm(g, axis=1), c["norms"])
    assert np.allclose(np.linalg.norm(O.clip_rows(g, c["c"]), axis=1), c["clipped"])
    big = np.full((3, 7), 5.0, np.float32)
    assert np.array_equal(O.clip_rows(big, 1e9), big)      # identity when C >= norm
    loss, avg = O.combine(g.astype(np.float32), np.array([1.0, 3.0], np.float32))
    assert loss == 2.0 and np.allclose(avg, g.mean(axis=0))


def test_perturbation_known_answer(O):
    ka = EXT["d3p_known_answers"]["perturbation_std"]
    avg = np.full(20000, 0.8, np.float32)
    out = O.perturb(O.PRNGKey(9782346), avg, [10000, 10000], ka["dp_scale"], ka["c"], ka["num_elements"],
                    ka["obs_scale"], ka["factor"])
    for site in (out[:10000], out[10000:]):
        assert abs(site.std() - ka["std"]) < 1e-2
        assert abs(site.mean() - 0.8 * ka["obs_scale"] * ka["factor"]) < 5e-3
    assert not np.allclose(out[:10000], out[10000:])        # different key per site (svi.py:491)
    out2 = O.perturb(O.split(O.PRNGKey(9782346), 2)[1], avg, [10000, 10000], 1.0, 2.0, 8, 0.3, 1.25)
    assert not np.allclose(out, out2)


def test_masking_known_answer(O):
    ka = EXT["d3p_known_answers"]["mask_factor"]
    B, d, N = ka["batch"], 3, 100
    spec = O.logreg_spec(d, False, lik_scale=N, obs_scale=N)
    mask = (np.arange(B) < ka["num_elements"]).astype(np.float32)
    r = np.random.default_rng(0)
    L, G, n, f = O.logreg_px_grads(spec, np.zeros(d), np.full(d, -2.25), np.ones((B, d)), np.ones(B),
                                   r.normal(size=(B, d)), mask)
    assert n == ka["num_elements"] and abs(f - ka["factor"]) < 1e-7
    assert np.all(L[8:] == 0) and np.all(G[8:] == 0) and not np.allclose(L[:8], 0) and not np.allclose(G[:8], 0)


def test_logreg_gradient_vs_torch_autograd(O):
    """The hand-derived per-example gradient equals autodiff of the ELBO written with
    torch.distributions (AutoDiagonalNormal: z = loc + softplus(u) * eps)."""
    import torch
    import torch.distributions as dist
    from torch.func import grad, vmap
    for (d, icpt, N, pw, pb, unscale) in [(8, False, 1000, 4.0, 1.0, True), (5, True, 250, 1.0, 2.0, True),
                                          (16, True, 100, 1.0, 1.0, False)]:
        D, B = d + int(icpt), 7
        r = np.random.default_rng(d)
        X = r.normal(size=(B, d)).astype(np.float32)
        y = (r.random(B) < 0.5).astype(np.float32)
        eps = r.normal(size=(B, D)).astype(np.float32)
        loc = (0.3 * r.normal(size=D)).astype(np.float32)
        unc = r.normal(size=D).astype(np.float32)
        mask = np.array([1, 1, 0, 1, 1, 1, 0], np.float32)
        obs = float(N) if unscale else 1.0
        spec = O.logreg_spec(d, icpt, pw, pb, lik_scale=N, obs_scale=obs)
        L, G, n, f = O.logreg_px_grads(spec, loc, unc, X, y, eps, mask)

        def px_loss(params, x, yv, e, m):
            l, u = params
            s = torch.nn.functional.softplus(u)
            z = l + s * e
            logq = dist.Normal(l, s).log_prob(z).sum()
            ps = torch.cat([torch.full((d,), pw), torch.full((D - d,), pb)]).double()
            logp = dist.Normal(torch.zeros(D).double(), ps).log_prob(z).sum()
            logit = x @ z[:d] + (z[d] if icpt else 0.0)
            ll = -torch.nn.functional.binary_cross_entropy_with_logits(logit, yv, reduction="sum")
            return (1 / obs) * (-(logp + N * ll - logq)) * m

        t = lambda a: torch.tensor(a, dtype=torch.float64)  # noqa: E731
        params = (t(loc), t(unc))
        gl, gu = vmap(grad(px_loss), in_dims=(None, 0, 0, 0, 0))(params, t(X), t(y), t(eps), t(mask))
        Lt = vmap(px_loss, in_dims=(None, 0, 0, 0, 0))(params, t(X), t(y), t(eps), t(mask)) * obs * f
        Gt = torch.cat([gl, gu], 1).numpy()
        assert np.abs(G - Gt).max() / np.abs(Gt).max() < 1e-6
        assert np.abs(L - Lt.numpy()).max() / np.abs(Lt.numpy()).max() < 1e-6


def test_example_guide_with_two_sites_vs_torch_autograd_and_the_stage_composition(O):
    """The logistic-regression example's OWN model and guide (examples/logistic_regression.py:49-86): sites 'w' (d) and 'intercept'
    (scalar), four parameter leaves w_loc / w_std_log / intercept_loc / intercept_std_log with exp-parametrised scales.  The oracle's
    update (O.meanfield_logreg_update) against an independent composition: the ELBO written site by site with torch.distributions,
    per-example gradients of the FOUR leaves by autograd, leaves concatenated in tree_flatten order of the parameter dict (sorted
    names, svi.py:490), joint clip, mean, ONE perturbation key per leaf (split(key, 4), svi.py:491), Adam.  The per-site guide noise
    is the oracle's (numpyro's seed-handler plumbing: UNPINNED); everything behind it is pinned here."""
    import torch
    import torch.distributions as dist
    from torch.func import grad, vmap
    d, B, N, C_, sigma, lr = 6, 9, 500, 0.7, 1.3, 1e-2
    r = np.random.default_rng(11)
    X = r.normal(size=(B, d)).astype(np.float32)
    y = (r.random(B) < 0.5).astype(np.float32)
    mask = np.array([1, 1, 0, 1, 1, 1, 1, 0, 1], np.float32)
    spec = O.logreg_spec(d, True, 1.0, 1.0, lik_scale=N, obs_scale=N, guide_exp=True)
    hy = O.Hyper(C_, sigma, lr, 0.9, 0.999, 1e-8)
    st = O.MeanFieldLogregState(O.PRNGKey(4), d)
    st.params[:] = (0.2 * r.normal(size=2 * d + 2)).astype(np.float32)
    p0, key0 = st.params.copy(), st.key.copy()
    loss, g = O.meanfield_logreg_update(spec, hy, st, X, y, mask)

    ks = O.split(key0, 3)
    eps = O.px_eps_sites(O.convert_to_jax_rng_key(ks[1]), B, [d, 1])          # [eps_w (d) | eps_intercept (1)]
    assert np.array_equal(eps[3, :d], O.tf_normal(O.px_site_keys(O.convert_to_jax_rng_key(ks[1]), B, 3, 2)[0], d))
    t = lambda a: torch.tensor(a, dtype=torch.float64)  # noqa: E731
    leaves = {"intercept_loc": t(p0[0:1]), "intercept_std_log": t(p0[1:2]), "w_loc": t(p0[2:2 + d]), "w_std_log": t(p0[2 + d:])}

    def px_loss(par, x, yv, e, m):
        w = par["w_loc"] + torch.exp(par["w_std_log"]) * e[:d]
        b = par["intercept_loc"] + torch.exp(par["intercept_std_log"]) * e[d:]
        logq = dist.Normal(par["w_loc"], torch.exp(par["w_std_log"])).log_prob(w).sum() + \
            dist.Normal(par["intercept_loc"], torch.exp(par["intercept_std_log"])).log_prob(b).sum()
        logp = dist.Normal(0.0, 1.0).log_prob(w).sum() + dist.Normal(0.0, 1.0).log_prob(b).sum()
        ll = -torch.nn.functional.binary_cross_entropy_with_logits(x @ w + b[0], yv, reduction="sum")
        return (1.0 / N) * (-(logp + N * ll - logq)) * m

    gs = vmap(grad(px_loss), in_dims=(None, 0, 0, 0, 0))(leaves, t(X), t(y), t(eps), t(mask))
    n = float(mask.sum())
    f = B / n
    L = vmap(px_loss, in_dims=(None, 0, 0, 0, 0))(leaves, t(X), t(y), t(eps), t(mask)) * N * f
    rows = torch.cat([gs[k].reshape(B, -1) for k in sorted(gs)], 1)            # tree_flatten: sorted dict keys
    norms = rows.norm(dim=1)
    rows = rows / torch.clamp(norms / C_, min=1.0)[:, None]
    avg = rows.mean(0).numpy()
    want_g = O.perturb(ks[2], avg.astype(np.float32), [1, 1, d, d], sigma, C_, n, float(N), f)
    np.testing.assert_allclose(g, want_g, rtol=2e-5, atol=1e-6 * np.abs(want_g).max())
    assert abs(loss - float(L.mean())) <= 2e-6 * abs(float(L.mean()))
    x, m, v = O.adam(p0, np.zeros_like(p0), np.zeros_like(p0), want_g, 0, lr=lr)
    np.testing.assert_allclose(st.params, x, rtol=1e-5, atol=1e-7)
    assert st.step == 1 and np.array_equal(st.key, np.asarray(ks[0]).ravel())
    # the four leaves get FOUR different noise streams: the two scalar leaves' noise is word 0 of their own keys
    site_keys = O.split(ks[2], 4)
    scale = sigma * C_ / n
    noise = (want_g / (N * f) - avg) / scale
    np.testing.assert_allclose(noise[0], O.normal(site_keys[0], (1,))[0], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(noise[2:2 + d], O.normal(site_keys[2], (d,)), rtol=1e-3, atol=1e-4)


def test_adam_matches_jax_optimizers_formula(O):
    r = np.random.default_rng(1)
    x, m, v = r.normal(size=9), np.zeros(9), np.zeros(9)
    xs, ms, vs = x.astype(np.float32), m.astype(np.float32), v.astype(np.float32)
    for i in range(4):
        g = r.normal(size=9)
        m = 0.1 * g + 0.9 * m
        v = 0.001 * g * g + 0.999 * v
        x = x - 1e-2 * (m / (1 - 0.9 ** (i + 1))) / (np.sqrt(v / (1 - 0.999 ** (i + 1))) + 1e-8)
        xs, ms, vs = O.adam(xs, ms, vs, g.astype(np.float32), i, lr=1e-2)
    assert np.allclose(xs, x, rtol=1e-5, atol=1e-6)


def test_full_update_is_the_stage_composition(O):
    """logreg_update == split -> px_grads -> clip -> combine -> perturb -> adam (svi.py:413-434)."""
    B, d, N = 12, 6, 300
    r = np.random.default_rng(4)
    X = r.normal(size=(B, d)).astype(np.float32)
    y = (r.random(B) < 0.5).astype(np.float32)
    loc, unc = np.zeros(d, np.float32), np.full(d, -2.0, np.float32)
    spec = O.logreg_spec(d, False, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    st = O.LogregState(O.PRNGKey(5), d, loc, unc)
    loss, grad = O.logreg_update(spec, hy, st, X, y)
    ks = O.split(O.PRNGKey(5), 3)
    eps = O.px_eps(O.convert_to_jax_rng_key(ks[1]), B, d)
    L, G, n, f = O.logreg_px_grads(spec, loc, unc, X, y, eps)
    l2, avg = O.combine(O.clip_rows(G, 1.0), L)
    g2 = O.perturb(ks[2], avg, [d, d], 0.7, 1.0, n, N, f)
    assert abs(loss - l2) < 1e-6 * abs(l2) and np.allclose(grad, g2, rtol=1e-6)
    assert np.array_equal(st.key, ks[0].ravel()) and st.step.value == 1
    x2, _, _ = O.adam(np.concatenate([loc, unc]), np.zeros(2 * d), np.zeros(2 * d), g2, 0, lr=1e-2)
    assert np.allclose(st.params, x2)


def test_config1_simple_posterior_plumbing(O):
    """BASELINE configs[0]-style plumbing on the CPU restatement: a few hundred DP-VI steps on a small
    table move the variational mean towards the non-private optimum (no GPU involved)."""
    N, d, B = 1000, 4, 100
    X, y = O.synth_logreg(7, 0, N, d)
    w = O.synth_wtrue(7, d)[:d]
    spec = O.logreg_spec(d, False, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.1, 5e-2, 0.9, 0.999, 1e-8)
    st = O.LogregState(O.PRNGKey(0), d, np.zeros(d, np.float32), np.full(d, -2.25, np.float32))
    bkey = O.PRNGKey(1)
    for i in range(300):
        idx = O.feistel_sample(O.fold_in(bkey, i), N, B)
        O.logreg_update(spec, hy, st, X[idx], y[idx])
    loc = st.params[:d]
    cos = float(loc @ w / (np.linalg.norm(loc) * np.linalg.norm(w)))
    assert cos > 0.9


@pytest.mark.parametrize("guide_exp", [False, True])
def test_gauss_mean_gradient_vs_finite_differences(O, guide_exp):
    """Gaussian-mean family (examples/simple_gaussian_posterior.py:51-81) and the exp-transformed hand-written
    guide: the oracle's analytic per-example gradient against central differences of the loss in float64."""
    d, N = 6, 50.0
    r = np.random.default_rng(3)
    spec = O.gauss_mean_spec(d, prior=1.3, lik_sigma=0.7, lik_scale=N, obs_scale=2.0, guide_exp=guide_exp)
    loc = r.normal(size=d).astype(np.float32)
    unc = (r.normal(size=d) * 0.3).astype(np.float32)
    X = r.normal(size=(1, d)).astype(np.float32)
    eps = r.normal(size=(1, d)).astype(np.float32)
    loss, g, n, f = O.logreg_px_grads(spec, loc, unc, X, None, eps)

    def L(p):
        s = np.exp(p[d:]) if guide_exp else np.log1p(np.exp(p[d:]))
        z = p[:d] + s * eps[0]
        lq = np.sum(-0.5 * eps[0] ** 2 - np.log(s) - 0.5 * np.log(2 * np.pi))
        lp = np.sum(-0.5 * (z / 1.3) ** 2 - np.log(1.3) - 0.5 * np.log(2 * np.pi))
        ll = np.sum(-0.5 * ((X[0] - z) / 0.7) ** 2 - np.log(0.7) - 0.5 * np.log(2 * np.pi))
        return 0.5 * ((lq - lp) - N * ll)      # inv_obs = 1 / 2

    p = np.concatenate([loc, unc]).astype(np.float64)
    fd = np.array([(L(p + 1e-6 * e) - L(p - 1e-6 * e)) / 2e-6 for e in np.eye(2 * d)])
    assert n == 1 and f == 1.0
    assert abs(loss[0] - 2.0 * L(p)) < 1e-5 * abs(2.0 * L(p))      # px_loss is rescaled by obs_scale (svi.py:306)
    np.testing.assert_allclose(g[0], fd, rtol=1e-4, atol=1e-4)


def test_config1_gauss_mean_converges_to_the_analytical_posterior(O):
    """BASELINE configs[0] (examples/simple_gaussian_posterior.py: N = 1000 toy rows around mu = 1, hand-written
    guide initialised at the prior): non-private-ish DP-VI on the CPU restatement reaches the conjugate posterior."""
    N, d, B = 1000, 4, 100
    r = np.random.default_rng(1234)
    X = (1.0 + 0.1 * r.normal(size=(N, d))).astype(np.float32)
    spec = O.gauss_mean_spec(d, prior=1.0, lik_sigma=0.1, lik_scale=N, obs_scale=N)
    hy = O.Hyper(20.0, 0.01, 2e-2, 0.9, 0.999, 1e-8)
    st = O.LogregState(O.PRNGKey(0), d, np.zeros(d, np.float32), np.zeros(d, np.float32))
    bkey = O.PRNGKey(1)
    for i in range(1500):
        idx = O.feistel_sample(O.fold_in(bkey, i), N, B)
        O.logreg_update(spec, hy, st, X[idx], None)
    var = 1.0 / (N / 0.1 ** 2 + 1.0)
    loc_exact, std_exact = var * X.sum(0) / 0.1 ** 2, np.sqrt(var)
    assert np.abs(st.params[:d] - loc_exact).max() < 0.02
    assert np.all(np.exp(st.params[d:]) < 0.05)          # started at 1.0, exact value is 0.0032


def test_adadp_known_answers_from_the_reference_tests(O):
    """tests/test_adadp_optimizer.py:66-131 on the flat vector of the reference's template tree
    ((7, 10), (7,), ((2, 7), (2,)) -> 93 scalars)."""
    P = 93
    zeros, ones = np.zeros(P, np.float32), np.ones(P, np.float32)
    # test_update_step_1: even step from 0 with g = 1, lr = 1
    x, lr, xs, xp = O.adadp(zeros, 1.0, zeros, zeros, ones, 0, tol=1.0)
    assert np.all(x == -0.5) and lr == 1.0 and np.all(xs == -1.0) and np.all(xp == 0.0)
    # test_update_step_2_no_stability_check: expected lr 1.018308251, x = -1.5
    x, lr, xs, xp = O.adadp(-0.5 * ones, 1.0, -ones, zeros, 2 * ones, 1, tol=5.0, stability_check=False)
    assert np.all(x == -1.5) and abs(lr - 1.018308251) < 1e-6
    # test_update_step_2_with_stability_check: update rejected, lr 0.72005267 clipped to 0.9
    x, lr, xs, xp = O.adadp(-0.5 * ones, 1.0, -ones, zeros, 3 * ones, 1, tol=5.0, stability_check=True)
    assert np.all(x == 0.0) and abs(lr - 0.9) < 1e-7
    # a zero error estimate gives sqrt(tol / 0) = inf -> factor 1.1
    x, lr, xs, xp = O.adadp(zeros, 2.0, zeros, zeros, zeros, 1, tol=1.0)
    assert abs(lr - 2.2) < 1e-6 and np.all(x == 0.0)


def test_special_functions_vs_scipy(O):
    """digamma and the implicit-reparametrisation derivative of the Gamma quantile (the JVP of jax.random.gamma)."""
    import scipy.special as sp
    for x in (0.01, 0.3, 1.0, 2.5, 10.0, 123.4):
        assert abs(O.digamma(x) - sp.digamma(x)) < 1e-12 * max(1.0, abs(sp.digamma(x)))
    for a in (0.2, 0.7, 1.0, 1.5, 3.0, 10.0, 40.0):
        for u in (1e-6, 0.01, 0.3, 0.5, 0.9, 0.999):
            x = sp.gammaincinv(a, u)
            h = 1e-5 * a
            fd = (sp.gammaincinv(a + h, u) - sp.gammaincinv(a - h, u)) / (2 * h)
            assert abs(O.gamma_grad(a, x) - fd) < 1e-6 * abs(fd)


def test_gamma_sampler_distribution(O):
    import scipy.stats as st
    for a in (0.3, 1.0, 4.2):
        xs = np.array([O.gamma_sample(np.array([i, 77], np.uint32), 3, a) for i in range(6000)])
        assert st.kstest(xs, "gamma", args=(a,)).pvalue > 1e-3
        assert abs(xs.mean() - a) < 5 * np.sqrt(a / 6000)


def test_gmm_model_gradient_vs_finite_differences(O):
    """BASELINE config 3's model and guide (examples/gaussian_mixture_model.py:51-85): the oracle's analytic
    per-example gradient against central differences of a float64 numpy restatement of the loss in which the
    Dirichlet draw is re-parametrised through the inverse Gamma CDF at fixed uniforms (the same pathwise derivative)."""
    import scipy.special as sp
    K, d, N = 4, 3, 50.0
    r = np.random.default_rng(5)
    spec = O.gmm_spec(K, d, prior_mu_scale=10.0, lik_scale=N, obs_scale=2.0)
    alpha_log = (r.normal(size=K) * 0.4).astype(np.float32)
    mus_loc = r.normal(size=(K, d)).astype(np.float32)
    x = r.normal(size=d).astype(np.float32)
    g, eps, sigs = O.gmm_px_latents(spec, alpha_log, np.array([11, 22], np.uint32), 5, 2)
    sigs = np.minimum(sigs, 50.0).astype(np.float32)      # keep the finite-difference problem well scaled
    u = sp.gammainc(np.exp(alpha_log.astype(np.float64)), g)
    loss, grad = O.gmm_px_loss_grad_given(spec, alpha_log, mus_loc, x, g, eps, sigs)

    def L(p):
        alpha = np.exp(p[:K])
        gg = sp.gammaincinv(alpha, u)
        pis = gg / gg.sum()
        mus = p[K:].reshape(K, d) + eps
        lq = sp.gammaln(alpha.sum()) - sp.gammaln(alpha).sum() + ((alpha - 1) * np.log(pis)).sum() + (-0.5 * eps ** 2).sum()
        lp = sp.gammaln(K) + (-0.5 * (mus / 10.0) ** 2 - np.log(10.0)).sum()
        comp = np.log(pis) + (-0.5 * ((x - mus) / sigs) ** 2 - np.log(sigs) - 0.5 * np.log(2 * np.pi)).sum(1)
        return 0.5 * ((lq - lp) - N * sp.logsumexp(comp))

    p = np.concatenate([alpha_log, mus_loc.ravel()]).astype(np.float64)
    assert abs(loss - L(p)) < 1e-5 * abs(L(p))
    fd = np.array([(L(p + 1e-5 * e) - L(p - 1e-5 * e)) / 2e-5 for e in np.eye(p.size)])
    np.testing.assert_allclose(grad, fd, rtol=2e-4, atol=2e-4 * np.abs(fd).max())


def test_cpu_baseline_loop_walks_the_checker_trajectory(O):
    """bench.py's cpu_baseline times d3po_logreg_run_feistel (Feistel sampling + gather + the reference's staged dataflow
    in one C loop, float32, per-column guide terms hoisted).  It must walk the same trajectory as the step-by-step
    composition of the checker functions (float64 accumulation inside): same keys, same indices, parameters to 1e-5."""
    d, B, rows, steps = 64, 96, 5000, 5
    X, y = O.synth_logreg(123, 0, rows, d)
    spec = O.logreg_spec(d, False, 1.0, 1.0, lik_scale=rows, obs_scale=rows)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    bkey = O.PRNGKey(1)
    ref = O.LogregState(O.PRNGKey(0), d, np.zeros(d, np.float32), np.full(d, -2.0, np.float32))
    for t in range(steps):
        idx = O.feistel_sample(O.fold_in(bkey, 3 + t), rows, B)
        ref_loss, _ = O.logreg_update(spec, hy, ref, X[idx], y[idx])
    for threads in (1, 3):
        st = O.LogregState(O.PRNGKey(0), d, np.zeros(d, np.float32), np.full(d, -2.0, np.float32))
        loss = O.logreg_run_feistel(spec, hy, st, X, y, bkey, 3, B, steps, threads)
        assert np.array_equal(st.key, ref.key) and st.step.value == steps
        assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss)
        np.testing.assert_allclose(st.params, ref.params, rtol=1e-5, atol=1e-6)


def test_randint_integer_dtypes_like_the_reference_tests(O):
    """d3p/random/__init__.py:108-146 for every dtype of its table (:115-123); the cases are the reference's own
    (tests/test_random.py:74-146): int32 with 2^10 + 1 values hits both bounds and is uniform, int8 over its FULL range
    (delta = 255 wraps in uint8, the mask is all ones), int16 up to 2^15, a single-value support gives that value."""
    import scipy.stats
    n = 72000
    x = O.randint(O.PRNGKey(8025111), (1000, 8, 9), 8, 8 + 2**10 + 1, np.int32)
    assert x.dtype == np.int32 and x.shape == (1000, 8, 9) and x.max() == 8 + 2**10 and x.min() == 8
    assert scipy.stats.chisquare(np.bincount(x.ravel() - 8, minlength=2**10 + 1)).pvalue >= 0.01
    assert np.array_equal(x, O.randint32_legacy(O.PRNGKey(8025111), (1000, 8, 9), 8, 8 + 2**10 + 1))   # the 32-bit restatement
    x = O.randint(O.PRNGKey(802511), (n,), -2**7, 2**7, np.int8)
    assert x.dtype == np.int8 and x.min() == -128 and x.max() == 127
    assert scipy.stats.chisquare(np.bincount(x.astype(np.int64) + 128, minlength=256)).pvalue >= 0.01
    x = O.randint(O.PRNGKey(8025111), (n,), 0, 2**15, np.int16)
    assert x.dtype == np.int16 and x.min() >= 0 and 2**15 - 64 <= x.max() < 2**15
    assert scipy.stats.chisquare(np.bincount(x.astype(np.int64) // 128, minlength=256)).pvalue >= 0.01
    x = O.randint(O.PRNGKey(5), (n,), -(2**40), 2**40 + 3, np.int64)
    assert x.dtype == np.int64 and x.min() >= -(2**40) and x.max() < 2**40 + 3 and (np.abs(x) > 2**33).any()
    for dt in (np.int8, np.int16, np.int32, np.int64):
        assert np.all(O.randint(O.PRNGKey(8025111), (100,), -4, -3, dt) == -4)
    with pytest.raises(TypeError):
        O.randint(O.PRNGKey(1), (3,), 0, 5, np.float32)


# ---- the oracle's VAE restatement (one and two hidden layers) against torch autograd on an ELBO written with torch.distributions
@pytest.mark.parametrize("H2", [0, 5])
def test_vae_per_example_gradients_vs_torch_autograd(O, H2):
    """oracle d3po_vae_step_sums materialises every per-example gradient by hand-written backpropagation; here the same
    per-example loss  inv_obs * scale * (log q(z|x) - log p(z) - log p(x|z))  is written with torch.distributions on
    float64 tensors (stax.Dense layers, softplus, Normal / Bernoulli(logits), examples/vae.py:65-153) and differentiated
    by autograd.  Norms, losses and the clipped sums must agree to float32 rounding -- for the reference's one-hidden-layer
    network and for the 784 -> [400, 200] -> 50 style variant (H2 > 0), which has no counterpart in the reference."""
    import torch
    B, D, H, Z = 7, 11, 6, 3
    r = np.random.default_rng(5)
    spec = O.vae_spec(D, H, Z, scale=0.7, obs_scale=2.0, H2=H2)
    sizes = O.vae_leaf_sizes(D, H, Z, H2)
    P = O.vae_num_params(spec)
    assert P == sum(sizes) and len(sizes) == (14 if H2 else 10)
    params = (r.normal(size=P) * 0.4).astype(np.float32)
    X = (r.random((B, D)) < 0.4).astype(np.float32)
    eps = r.normal(size=(B, Z)).astype(np.float32)
    _, norms0, _ = O.vae_step_sums(spec, params, X, eps, 1e30)
    clip = float(np.median(norms0))
    sums, norms, px_loss = O.vae_step_sums(spec, params, X, eps, clip)

    hs = [H] + ([H2] if H2 else [])
    dec_dims, enc_dims = [Z] + hs[::-1] + [D], [D] + hs
    shapes = []
    for i, o in zip(dec_dims[:-1], dec_dims[1:]):
        shapes += [(i, o), (o,)]
    for i, o in zip(enc_dims[:-1], enc_dims[1:]):
        shapes += [(i, o), (o,)]
    shapes += [(hs[-1], Z), (Z,), (hs[-1], Z), (Z,)]
    flat = torch.tensor(params, dtype=torch.float64, requires_grad=True)

    def loss_of(i):
        leaves, pos = [], 0
        for shp in shapes:
            n = int(np.prod(shp))
            leaves.append(flat[pos:pos + n].reshape(shp))
            pos += n
        n_dec = len(dec_dims) - 1
        dec, enc, heads = leaves[:2 * n_dec], leaves[2 * n_dec:-4], leaves[-4:]
        x = torch.tensor(X[i], dtype=torch.float64)
        h = x
        for k in range(0, len(enc), 2):
            h = torch.nn.functional.softplus(h @ enc[k] + enc[k + 1])
        z_loc, z_std = h @ heads[0] + heads[1], torch.exp(h @ heads[2] + heads[3])
        z = z_loc + z_std * torch.tensor(eps[i], dtype=torch.float64)
        log_q = torch.distributions.Normal(z_loc, z_std).log_prob(z).sum()
        log_p = torch.distributions.Normal(0.0, 1.0).log_prob(z).sum()
        h = z
        for k in range(0, len(dec) - 2, 2):
            h = torch.nn.functional.softplus(h @ dec[k] + dec[k + 1])
        logits = h @ dec[-2] + dec[-1]
        log_lik = torch.distributions.Bernoulli(logits=logits).log_prob(x).sum()
        return (1.0 / 2.0) * 0.7 * (log_q - log_p - log_lik)

    acc = np.zeros(P)
    for i in range(B):
        li = loss_of(i)
        (g,) = torch.autograd.grad(li, flat)
        g = g.numpy()
        nrm = np.linalg.norm(g)
        assert abs(li.item() - px_loss[i]) <= 2e-6 * abs(li.item()) + 1e-6
        assert abs(nrm - norms[i]) <= 2e-6 * nrm
        acc += g / max(1.0, nrm / clip)
    np.testing.assert_allclose(sums[:P], acc, rtol=2e-5, atol=2e-6 * np.abs(acc).max())
    assert sums[P + 1] == B and abs(sums[P] - px_loss.sum()) <= 1e-5 * abs(px_loss.sum())
    assert (norms > clip).any() and (norms < clip).any()
