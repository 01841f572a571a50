#!/usr/bin/env python3
"""Randomised sweep of the update path against the CPU oracle (test infrastructure: the oracle is the checker).

A case = one seed -> a random model family (logistic regression with / without intercept, Gaussian mean), guide (AutoDiagonalNormal,
the exp-parametrised hand-written one, the example's two-site guide), shape (d from 1 to wide rows, B from 1 to 5000), table size,
batch source (Feistel subsampling or Poisson sampling through `run_steps`, or `update` on an explicit batch with a random mask),
clipping threshold, noise scale, step size, number of steps and first batch index.  The HIP trajectory is compared with the oracle's
restatement of d3p/svi.py:395-434 driven by the oracle's own samplers: state key bit-exact, every loss (rtol 1e-4 + 1e-6 (D + N): the loss is a difference of large sums), the NaN pattern of
losses and parameters (empty batches: svi.py:305, :365), final parameters
(rtol 5e-4, atol 5e-5 of the largest) and step counter.

    python tests/fuzz_vs_oracle.py [update|big|stepwise|staged|gmm|vae|rng|batches|shards|posshards] [first_seed=0] [count=40] [out.jsonl]

`gmm`: the mixture model's update (explicit batches with masks, Feistel runs) vs the oracle's stage composition; `rng`: split / fold_in /
random_bits / randint / uniform / normal / Feistel / Poisson selection at random arguments, bit-exact (normal: 2e-6).

`tests/test_gpu_fuzz.py` runs a fixed handful of seeds inside the suite; a long sweep is run by hand on a GPU box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LOSS_RTOL, PARAM_RTOL, PARAM_ATOL = 1e-4, 5e-4, 5e-5
DIMS = [1, 2, 3, 5, 8, 17, 31, 64, 100, 255, 256, 257, 511, 512, 513, 700, 1024, 2048, 2049, 2600]
BATCHES = [1, 2, 7, 32, 63, 64, 65, 200, 1000, 4096, 5000]


def _frac(flags):
    """Fraction of True entries (0 for an empty array)."""
    return float(np.mean(flags)) if np.size(flags) else 0.0


def _far(got, want, tol):
    """Elementwise: True where got and want differ by more than tol (equal infinities are equal; NaNs are compared separately)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    with np.errstate(invalid="ignore"):
        return ~((got == want) | (np.abs(got - want) <= tol))


def draw_case(seed):
    r = np.random.default_rng(100_003 * seed + 17)
    c = {"seed": int(seed)}
    c["family"] = str(r.choice(["logreg", "logreg", "logreg_icpt", "gauss"]))
    guides = {"logreg": ["auto", "auto"], "logreg_icpt": ["auto", "meanfield"], "gauss": ["auto", "exp"]}[c["family"]]
    c["guide"] = str(r.choice(guides))
    c["d"] = int(r.choice(DIMS))
    c["B"] = int(r.choice(BATCHES))
    # bound the oracle's work (single thread, ~1e8 element-updates per second): B * d * steps <= 6e7
    budget = 6e7 / (c["B"] * c["d"])
    steps_pool = [s for s in (1, 2, 5, 20, 130) if s <= max(budget, 1)]
    c["steps"] = int(r.choice(steps_pool))
    c["source"] = str(r.choice(["feistel", "feistel", "poisson", "explicit"])) if c["guide"] != "meanfield" else \
        str(r.choice(["feistel", "explicit"]))
    c["N"] = int(min(max(c["B"] * float(r.choice([1.0, 1.5, 10.0, 100.0])), c["B"]), 200_000, 4e8 / (4 * c["d"])))
    c["N"] = max(c["N"], c["B"])
    c["clip"] = float(r.choice([0.1, 1.0, 10.0, 1e6]))
    c["sigma"] = float(r.choice([0.0, 0.5, 2.0]))
    c["lr"] = float(r.choice([1e-3, 1e-2, 1e-1]))
    c["first"] = int(r.choice([0, 7, 1000]))
    c["quantile"] = float(r.choice([0.5, 0.99]))
    c["suppress"] = bool(r.random() < 0.3)
    c["mask_keep"] = float(r.choice([1.0, 0.8, 0.3]))
    c["init_scale"] = float(r.choice([0.0, 0.3]))
    c["key"], c["bkey"] = int(r.integers(0, 2**31)), int(r.integers(0, 2**31))
    c["unscale"] = bool(r.random() < 0.8)        # clip_unscaled_observations (svi.py:225-234): False = observation_scale 1
    c["split_at"] = int(r.integers(1, c["steps"])) if (c["steps"] > 1 and r.random() < 0.5) else 0   # the run as two consecutive run_steps calls
    r3 = np.random.default_rng(1_200_007 * seed + 71)      # (later additions draw from their own stream: the earlier fields of a seed stay)
    c["unc_center"] = float(r3.choice([-2.0, -2.0, -6.0, -10.0, -14.0]))   # unconstrained scales around it: posterior std 0.13 .. 8e-7
    return c


def run_case(c, O, dump=False):
    """Runs one case on the GPU and in the oracle; returns the case dict extended by the comparison ("ok": bool, "why": text)."""
    import scipy.stats
    import torch
    import d3p_amd.random as rng
    from d3p_amd.minibatch import poisson_batchify_data, subsample_batchify_data
    from d3p_amd.models import (Adam, AutoDiagonalNormal, DiagonalNormalGuide, GaussianMean, LogisticRegression, MeanFieldGuide,
                                Trace_ELBO)
    from d3p_amd.svi import DPSVI, DPSVIState
    r = np.random.default_rng(c["seed"] + 5)
    d, B, N, steps = c["d"], c["B"], c["N"], c["steps"]
    gauss, icpt = c["family"] == "gauss", c["family"] == "logreg_icpt"
    D = d + (1 if icpt else 0)
    X = r.normal(size=(N, d)).astype(np.float32)
    if gauss:
        X = (1.0 + 0.5 * X).astype(np.float32)
    y = None if gauss else (r.random(N) < 0.5).astype(np.float32)
    obs = float(N) if c.get("unscale", True) else 1.0
    pw, pb, ls = c.get("prior_w", 1.0), c.get("prior_b", 2.0), c.get("lik_sigma", 0.7)
    if gauss:
        pw = c.get("prior_w", 1.5)
        model = GaussianMean(d, prior_scale=pw, obs_scale=ls)
        guide = AutoDiagonalNormal(model) if c["guide"] == "auto" else DiagonalNormalGuide(model)
        spec = O.gauss_mean_spec(d, prior=pw, lik_sigma=ls, lik_scale=N, obs_scale=obs, guide_exp=c["guide"] != "auto")
    else:
        model = LogisticRegression(d, prior_scale=pw, intercept=icpt, intercept_prior_scale=pb)
        guide = MeanFieldGuide(model) if c["guide"] == "meanfield" else AutoDiagonalNormal(model)
        spec = O.logreg_spec(d, icpt, pw, pb, lik_scale=N, obs_scale=obs, guide_exp=c["guide"] == "meanfield")
    svi = DPSVI(model, guide, Adam(c["lr"]), Trace_ELBO(), c["clip"], c["sigma"], num_obs_total=N,
                clip_unscaled_observations=c.get("unscale", True), **({"d": d} if gauss else {}))
    hy = O.Hyper(c["clip"], c["sigma"], c["lr"], 0.9, 0.999, 1e-8)
    meanfield = c["guide"] == "meanfield"
    if meanfield:
        ost = O.MeanFieldLogregState(O.PRNGKey(c["key"]), d)
        params0 = np.zeros(2 * D, np.float32)
        if c["init_scale"]:
            params0 = (r.normal(size=2 * D) * c["init_scale"]).astype(np.float32)
            ost.params[:] = params0
    else:
        loc0 = (r.normal(size=D) * c["init_scale"]).astype(np.float32)
        unc0 = (r.normal(size=D) * c["init_scale"] + c.get("unc_center", -2.0)).astype(np.float32)
        ost = O.LogregState(O.PRNGKey(c["key"]), D, loc0, unc0)
        params0 = np.concatenate([loc0, unc0])
    st = DPSVIState(svi.optim.init(torch.tensor(params0).cuda()), rng.PRNGKey(c["key"]), obs)
    Xd = torch.tensor(X).cuda()
    yd = None if gauss else torch.tensor(y).cuda()
    table = (Xd,) if gauss else (Xd, yd)
    # (DPSVI.init computes the same observation scale for these arguments)
    if float(svi.init(rng.PRNGKey(1), *(a[:max(B, 1)] for a in table)).observation_scale) != obs:
        raise AssertionError("observation scale of init")
    upd = O.meanfield_logreg_update if meanfield else O.logreg_update
    el = []
    t0 = time.time()
    if c["source"] == "explicit":
        mask = r.random(B) < c["mask_keep"]
        losses = []
        for t in range(steps):
            idx = (np.arange(B) * 7 + 13 * t) % N
            args = (Xd[idx],) if gauss else (Xd[idx], yd[idx])
            st, loss = svi.update(st, *args, mask=torch.tensor(mask).cuda() if c["mask_keep"] < 1.0 else True)
            losses.append(loss.reshape(()))
            el.append(upd(spec, hy, ost, X[idx], None if gauss else y[idx], mask.astype(np.float32) if c["mask_keep"] < 1.0 else None)[0])
        losses = torch.stack(losses)
    elif c["source"] == "feistel":
        _, gb = subsample_batchify_data(table, B)
        k = c.get("split_at", 0)
        if k:
            st, l1 = svi.run_steps(st, gb, rng.PRNGKey(c["bkey"]), c["first"], k)
            st, l2 = svi.run_steps(st, gb, rng.PRNGKey(c["bkey"]), c["first"] + k, steps - k)
            losses = torch.cat([l1, l2])
        else:
            st, losses = svi.run_steps(st, gb, rng.PRNGKey(c["bkey"]), c["first"], steps)
        for t in range(steps):
            idx = O.feistel_sample(O.fold_in(O.PRNGKey(c["bkey"]), c["first"] + t), N, B)
            el.append(upd(spec, hy, ost, X[idx], None if gauss else y[idx])[0])
    elif c["source"] in ("with_replacement", "split"):
        # batchifiers without a native loop: run_steps walks get_batch + update (DPSVI._run_steps_stepwise)
        from d3p_amd.minibatch import split_batchify_data
        if c["source"] == "split":
            init_b, gb = split_batchify_data(table, B)
            nb, bstate = init_b(rng.PRNGKey(c["bkey"]))
            first = c["first"] % max(nb - steps + 1, 1)
            steps = min(steps, nb - first)
            perm = O.feistel_sample(O.PRNGKey(c["bkey"]), N, N)
            batches = [perm[(first + t) * B:(first + t + 1) * B] for t in range(steps)]
        else:
            _, gb = subsample_batchify_data(table, B, with_replacement=True)
            bstate, first = rng.PRNGKey(c["bkey"]), c["first"]
            batches = [np.asarray(O.randint(O.fold_in(O.PRNGKey(c["bkey"]), first + t), (B,), 0, N)).astype(np.int64) for t in range(steps)]
        st, losses = svi.run_steps(st, gb, bstate, first, steps)
        for idx in batches:
            el.append(upd(spec, hy, ost, X[idx], None if gauss else y[idx])[0])
    else:
        q = B / N
        if int(q * N) == 0:
            q = 1.0 / N
        maxB = min(max(int(scipy.stats.poisson(N * q).ppf(c["quantile"])), 1), N)   # (max_batch_size > N is an error, minibatch.py:116)
        _, gb = poisson_batchify_data(table, q, maxB, handle_oversized_batch="suppress" if c["suppress"] else "truncate")
        k = c.get("split_at", 0)
        if k:
            st, l1 = svi.run_steps(st, gb, rng.PRNGKey(c["bkey"]), c["first"], k)
            st, l2 = svi.run_steps(st, gb, rng.PRNGKey(c["bkey"]), c["first"] + k, steps - k)
            losses = torch.cat([l1, l2])
        else:
            st, losses = svi.run_steps(st, gb, rng.PRNGKey(c["bkey"]), c["first"], steps)
        for t in range(steps):
            idx, nsel, nvalid = O.poisson_select(O.fold_in(O.PRNGKey(c["bkey"]), c["first"] + t), np.float32(q), N, maxB, c["suppress"])
            mask = (np.arange(maxB) < nvalid).astype(np.float32)
            el.append(upd(spec, hy, ost, X[idx], None if gauss else y[idx], mask)[0])
    torch.cuda.synchronize()
    c["seconds"] = round(time.time() - t0, 2)
    got_l, want_l = losses.detach().cpu().numpy().astype(np.float64), np.asarray(el, np.float64)
    got_p, want_p = st.optim_state[1].detach().cpu().numpy(), ost.params
    why = []
    # (two float32 trajectories with different summation orders drift apart with the number of steps: the tolerances grow with it)
    drift = 1.0 + steps / 16.0
    both_nan = np.isnan(got_l) & np.isnan(want_l)
    if not np.array_equal(np.isnan(got_l), np.isnan(want_l)):
        why.append("losses: NaN pattern differs")
    else:
        # (the loss is a difference of sums over D latent terms and N-scaled likelihood terms: an absolute fp32 floor beside the rtol)
        ok_l = both_nan | (np.abs(got_l - want_l) <= drift * LOSS_RTOL * np.abs(want_l) + 1e-6 * (D + N))
        if not ok_l.all():
            k = int(np.argmin(ok_l))
            why.append(f"loss {k}: {got_l[k]!r} vs {want_l[k]!r}")
    if not np.array_equal(st.rng_key.cpu().numpy().ravel(), ost.key):
        why.append("state key differs")
    if int(st.optim_state[0]) != steps:
        why.append(f"step counter {int(st.optim_state[0])} != {steps}")
    if not np.array_equal(np.isnan(got_p), np.isnan(want_p)):
        why.append("parameters: NaN pattern differs")
    else:
        fin = ~np.isnan(want_p)
        scale = np.abs(want_p[fin]).max() if fin.any() else 0.0
        bad = np.abs(got_p[fin] - want_p[fin]) > drift * (PARAM_RTOL * np.abs(want_p[fin]) + PARAM_ATOL * max(scale, 1e-30))
        # (Adam's early steps are lr g / (|g| + 1e-8): a gradient component near 0 carries its own relative error into the step -- a
        #  handful of parameters a few per cent of a step apart is that)
        small = bad & (np.abs(got_p[fin] - want_p[fin]) <= 0.05 * c["lr"] * steps)
        if (bad & ~small).any() or _frac(small) > 0.005:
            k = int(np.argmax(np.abs(got_p[fin] - want_p[fin])))
            why.append(f"parameter: {got_p[fin][k]!r} vs {want_p[fin][k]!r} (largest {scale:.3g}); {int(bad.sum())} of {int(fin.sum())} out of tolerance")
    # evaluate (svi.py:436-449) on a fresh batch at the state the trajectory reached (not in the corner where a scale underflows)
    if not why and np.isfinite(got_p).all() and np.abs(got_p).max() < 80.0:
        nb = min(B, N)
        idx = (np.arange(nb) * 3 + 1) % N
        args = (Xd[idx],) if gauss else (Xd[idx], yd[idx])
        got_e = float(svi.evaluate(st, *args))
        jk = O.convert_to_jax_rng_key(O.split(ost.key, 1)[0])
        if gauss:
            spec_e = O.gauss_mean_spec(d, prior=pw, lik_sigma=ls, lik_scale=N, obs_scale=1.0, guide_exp=c["guide"] != "auto")
        else:
            spec_e = O.logreg_spec(d, icpt, pw, pb, lik_scale=N, obs_scale=1.0, guide_exp=meanfield)
        if meanfield:
            want_e = O.meanfield_logreg_evaluate(spec_e, got_p, X[idx], y[idx], jk)
        else:
            want_e = O.logreg_evaluate(spec_e, got_p[:D], got_p[D:], X[idx], None if gauss else y[idx], jk)
        if not (got_e == want_e or abs(got_e - want_e) <= LOSS_RTOL * abs(want_e) + 1e-6 * (D + N) or (np.isnan(got_e) and np.isnan(want_e))):
            why.append(f"evaluate: {got_e!r} vs {want_e!r}")
    if dump:
        c["got_losses"], c["want_losses"] = [float(v) for v in got_l], [float(v) for v in want_l]
        c["got_nan_params"], c["want_nan_params"] = int(np.isnan(got_p).sum()), int(np.isnan(want_p).sum())
        c["mask_sum"] = int(mask.sum()) if c["source"] != "feistel" else None
    c["ok"], c["why"] = not why, "; ".join(why)
    c["final_loss"] = float(want_l[-1]) if len(want_l) else None
    return c


# ------------------------------------------------------------------ batchifiers without a native loop (sampling with replacement, epoch splits)
def draw_stepwise_case(seed):
    c = draw_case(53 * seed + 11)
    c["seed"] = int(seed)
    r = np.random.default_rng(1_100_009 * seed + 67)
    c["source"] = str(r.choice(["with_replacement", "split"]))
    c["steps"] = min(c["steps"], 5)
    if c["source"] == "split":
        c["N"] = max(c["N"], c["B"] * c["steps"])
    c["split_at"] = 0
    return c


# ------------------------------------------------------------------ large batches, random prior / likelihood scales
def draw_big_case(seed):
    c = draw_case(31 * seed + 5)
    c["seed"] = int(seed)
    r = np.random.default_rng(1_000_003 * seed + 61)
    c["B"] = int(r.choice([5000, 8192, 16384, 32768, 100_000]))
    c["d"] = int(r.choice([1, 2, 5, 17, 64, 100, 256, 512]))
    budget = 8e7 / (c["B"] * c["d"])
    c["steps"] = int(r.choice([s for s in (1, 2, 3, 5, 20) if s <= max(budget, 1)]))
    c["N"] = int(c["B"] * int(r.choice([1, 2, 10])))
    c["split_at"] = 0
    c["prior_w"], c["prior_b"] = float(r.choice([0.1, 1.0, 10.0])), float(r.choice([0.5, 3.0]))
    c["lik_sigma"] = float(r.choice([0.1, 0.7, 3.0]))
    c["mask_keep"] = float(r.choice([1.0, 0.5]))
    return c


# ------------------------------------------------------------------ the five-stage composition (SGD: no fused path) and evaluate
def draw_staged_case(seed):
    c = draw_case(7919 * seed + 3)
    c["seed"], c["family_name"] = int(seed), "staged"
    if c["guide"] == "meanfield":
        c["guide"] = "auto"
    c["source"] = "explicit"
    # the stage-wise path materialises B x P per-example gradients on both sides
    while c["B"] * c["d"] > 2e5:
        c["B"] = max(c["B"] // 4, 1)
    c["steps"] = min(c["steps"], 3)
    c["N"] = max(c["N"], c["B"])
    c["lr"] = min(c["lr"], 1e-2)
    r = np.random.default_rng(800_003 * seed + 53)
    c["optim"] = str(r.choice(["sgd", "sgd", "adadp"]))
    c["tol"] = float(r.choice([0.1, 1.0, 10.0]))
    c["stability"] = bool(r.random() < 0.5)
    if c["optim"] == "adadp":
        c["steps"] = int(r.choice([2, 3, 4]))
        c["lr"] = float(r.choice([1e-6, 1e-5]))      # (N-scaled gradients: ADADP adapts from here)
    return c


def run_staged_case(c, O, dump=False):
    """DPSVI.update with SGD (the reference's tests' optimiser): per-example gradients -> clip -> mean -> one noise key per site -> step,
    every stage a materialised tensor (d3p_logreg_px_grads incl. its column-chunked form, d3p_clip_rows, d3p_combine, d3p_perturb_apply,
    d3p_sgd_step) vs the oracle's stage functions; then evaluate (svi.py:436-449) on the last state."""
    import torch
    import d3p_amd.random as rng
    from d3p_amd.models import SGD, AutoDiagonalNormal, DiagonalNormalGuide, GaussianMean, LogisticRegression, Trace_ELBO
    from d3p_amd.optimizers import ADADP
    from d3p_amd.svi import DPSVI, DPSVIState
    adadp = c.get("optim") == "adadp"
    r = np.random.default_rng(c["seed"] + 23)
    d, B, N, steps = c["d"], c["B"], c["N"], c["steps"]
    gauss, icpt = c["family"] == "gauss", c["family"] == "logreg_icpt"
    D = d + (1 if icpt else 0)
    X = r.normal(size=(N, d)).astype(np.float32)
    if gauss:
        X = (1.0 + 0.5 * X).astype(np.float32)
    y = None if gauss else (r.random(N) < 0.5).astype(np.float32)
    exp_guide = gauss and c["guide"] != "auto"
    if gauss:
        model = GaussianMean(d, prior_scale=1.5, obs_scale=0.7)
        guide = DiagonalNormalGuide(model) if exp_guide else AutoDiagonalNormal(model)
        mk_spec = lambda obs: O.gauss_mean_spec(d, prior=1.5, lik_sigma=0.7, lik_scale=N, obs_scale=obs, guide_exp=exp_guide)
    else:
        model = LogisticRegression(d, prior_scale=1.0, intercept=icpt, intercept_prior_scale=2.0)
        guide = AutoDiagonalNormal(model)
        mk_spec = lambda obs: O.logreg_spec(d, icpt, 1.0, 2.0, lik_scale=N, obs_scale=obs)
    spec = mk_spec(N)
    optim = ADADP(c["lr"], tol=c["tol"], stability_check=c["stability"]) if adadp else SGD(c["lr"])
    svi = DPSVI(model, guide, optim, Trace_ELBO(), c["clip"], c["sigma"], num_obs_total=N, **({"d": d} if gauss else {}))
    loc = (r.normal(size=D) * c["init_scale"]).astype(np.float32)
    unc = (r.normal(size=D) * c["init_scale"] - 2.0).astype(np.float32)
    x = np.concatenate([loc, unc])
    st = DPSVIState(svi.optim.init(torch.tensor(x).cuda()), rng.PRNGKey(c["key"]), float(N))
    Xd = torch.tensor(X).cuda()
    yd = None if gauss else torch.tensor(y).cuda()
    key = O.PRNGKey(c["key"])
    mask = r.random(B) < c["mask_keep"]
    use_mask = c["mask_keep"] < 1.0
    got_l, want_l = [], []
    chaotic = False
    for t in range(steps):
        idx = (np.arange(B) * 7 + 13 * t) % N
        args = (Xd[idx],) if gauss else (Xd[idx], yd[idx])
        st, loss = svi.update(st, *args, mask=torch.tensor(mask).cuda() if use_mask else True)
        got_l.append(float(loss))
        ks = O.split(key, 3)
        eps = O.px_eps(O.convert_to_jax_rng_key(ks[1]), B, D)
        with np.errstate(all="ignore"):
            L, G, n, f = O.logreg_px_grads(spec, x[:D], x[D:], X[idx], None if gauss else y[idx], eps, mask.astype(np.float32) if use_mask else None)
            eloss, avg = O.combine(O.clip_rows(G, c["clip"]), L)
            g = O.perturb(ks[2], avg, [D, D], c["sigma"], c["clip"], n, N, f)
            if adadp:
                if t == 0:
                    olr, oxs, oxp = c["lr"], np.zeros(2 * D, np.float32), x.copy()
                x, olr, oxs, oxp = O.adadp(x, olr, oxs, oxp, g, t, tol=c["tol"], stability_check=c["stability"])
            else:
                x = (x - np.float32(c["lr"]) * g).astype(np.float32)
            if np.isfinite(G).all() and np.abs(G).max() > 1e9 * min(c["clip"], 1e3):
                chaotic = True      # (a diverged run: per-example gradients of 1e12 clipped to C -- the directions of the clipped rows hang on the last bits)
        want_l.append(eloss)
        key = ks[0]
    torch.cuda.synchronize()
    got_l, want_l = np.asarray(got_l, np.float64), np.asarray(want_l, np.float64)
    got_p = svi.optim.get_params(st.optim_state).detach().cpu().numpy()
    why = []
    if adadp and not (abs(float(st.optim_state[1][1]) - olr) <= 1e-4 * abs(olr) or (np.isnan(float(st.optim_state[1][1])) and np.isnan(olr))):
        why.append(f"ADADP learning rate {float(st.optim_state[1][1])!r} vs {olr!r}")
    # (an overflowing loss is inf or NaN depending on the order of the last additions: one class)
    if not np.array_equal(np.isfinite(got_l), np.isfinite(want_l)):
        why.append(f"losses: finite / non-finite pattern differs ({got_l.tolist()} vs {want_l.tolist()})")
    else:
        fin = np.isfinite(want_l)
        with np.errstate(invalid="ignore"):
            bad = _far(got_l[fin], want_l[fin], LOSS_RTOL * np.abs(want_l[fin]) + 1e-6 * (D + N))
        if bad.any():
            k = int(np.argmax(bad))
            why.append(f"loss {k}: {got_l[fin][k]!r} vs {want_l[fin][k]!r}")
    if not np.array_equal(st.rng_key.cpu().numpy().ravel(), np.asarray(key).ravel()):
        why.append("state key differs")
    if int(st.optim_state[0]) != steps:
        why.append(f"step counter {int(st.optim_state[0])} != {steps}")
    if not np.array_equal(np.isnan(got_p), np.isnan(x)):
        why.append(f"parameters: NaN pattern differs ({int(np.isnan(got_p).sum())} vs {int(np.isnan(x).sum())})")
    else:
        fin = ~np.isnan(x)
        finite = x[fin][np.isfinite(x[fin])]
        scale = np.abs(finite).max() if finite.size else 0.0
        with np.errstate(invalid="ignore"):
            bad = _far(got_p[fin], x[fin], PARAM_RTOL * np.abs(x[fin]) + PARAM_ATOL * max(scale, 1e-30))
        if bad.any() and not chaotic:
            k = int(np.argmax(np.abs(got_p[fin] - x[fin])))
            why.append(f"parameter: {got_p[fin][k]!r} vs {x[fin][k]!r} (largest {scale:.3g}); {int(bad.sum())} of {int(fin.sum())} out of tolerance")
        # (scales below 1e-38 -- u < -87 -- flush to zero on the device (v_exp_f32 has no denormals): log s = -inf there, where a CPU
        #  reference still holds a denormal; a state in that corner is not asked for its loss)
        if fin.all() and np.isfinite(got_p).all() and (exp_guide or got_p[D:].min() > -80.0) and got_p[D:].max() < 80.0:
            idx = (np.arange(B) * 3 + 1) % N
            args = (Xd[idx],) if gauss else (Xd[idx], yd[idx])
            got_e = float(svi.evaluate(st, *args))
            want_e = O.logreg_evaluate(mk_spec(1.0), got_p[:D], got_p[D:], X[idx], None if gauss else y[idx],
                                       O.convert_to_jax_rng_key(O.split(key, 1)[0]))
            if not (got_e == want_e or abs(got_e - want_e) <= LOSS_RTOL * abs(want_e) + 1e-6 * (D + N) or (np.isnan(got_e) and np.isnan(want_e))):
                why.append(f"evaluate: {got_e!r} vs {want_e!r}")
    c["ok"], c["why"] = not why, "; ".join(why)
    c["mask_sum"] = int(mask.sum()) if use_mask else B
    return c


# ------------------------------------------------------------------ the mixture model (BASELINE configs[2]'s family)
def draw_gmm_case(seed):
    r = np.random.default_rng(200_003 * seed + 29)
    c = {"seed": int(seed), "family": "gmm"}
    c["K"] = int(r.choice([2, 3, 5, 16, 17, 32]))
    c["d"] = int(r.choice([1, 2, 3, 17, 64, 70, 128] + ([256] if c["K"] <= 16 else [])))
    cap = max(int(3e5 / (c["K"] * c["d"])), 1)
    c["B"] = int(min(int(r.choice([1, 2, 7, 33, 64, 200, 1000])), cap))
    c["steps"] = int(r.choice([1, 2, 4])) if c["B"] * c["K"] * c["d"] <= 1e5 else 1
    c["source"] = str(r.choice(["explicit", "feistel"]))
    c["N"] = int(max(c["B"] * float(r.choice([1.0, 3.0, 50.0])), c["B"]))
    c["clip"] = float(r.choice([1.0, 20.0, 1e6]))
    c["sigma"] = float(r.choice([0.0, 0.7]))
    c["lr"] = float(r.choice([1e-3, 1e-2]))
    c["first"] = int(r.choice([0, 5]))
    c["mask_keep"] = float(r.choice([1.0, 0.7, 0.0 if r.random() < 0.3 else 0.7]))
    c["key"], c["bkey"] = int(r.integers(0, 2**31)), int(r.integers(0, 2**31))
    r2 = np.random.default_rng(900_007 * seed + 59)     # (later additions draw from their own stream: the earlier fields of a seed stay)
    if c["source"] == "feistel" and c["B"] * c["K"] * c["d"] <= 3000 and r2.random() < 0.5:
        c["steps"] = int(r2.choice([65, 70, 130]))       # across the native loop's batches of prepared steps
    c["alpha_scale"] = float(r2.choice([0.4, 0.4, 1.5]))   # log-concentrations ~ N(0, alpha_scale^2): Dirichlet concentrations 0.05 .. 20 at 1.5
    c["mu_scale"] = float(r2.choice([2.0, 2.0, 8.0]))
    return c


def run_gmm_case(c, O, dump=False):
    import torch
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    from d3p_amd.models import Adam, GaussianMixtureGuide, GaussianMixtureModel, Trace_ELBO
    from d3p_amd.svi import DPSVI, DPSVIState
    r = np.random.default_rng(c["seed"] + 11)
    K, d, B, N, steps = c["K"], c["d"], c["B"], c["N"], c["steps"]
    X = (r.normal(size=(N, d)) * 3).astype(np.float32)
    params = np.concatenate([r.normal(size=K) * c.get("alpha_scale", 0.4), r.normal(size=K * d) * c.get("mu_scale", 2.0)]).astype(np.float32)
    model = GaussianMixtureModel()
    svi = DPSVI(model, GaussianMixtureGuide(model), Adam(c["lr"]), Trace_ELBO(), c["clip"], c["sigma"], k=K, d=d, num_obs_total=N)
    st = DPSVIState(svi.optim.init(torch.tensor(params).cuda()), rng.PRNGKey(c["key"]), float(N))
    spec = O.gmm_spec(K, d, 10.0, lik_scale=N, obs_scale=N)
    Xd = torch.tensor(X).cuda()
    key = O.PRNGKey(c["key"])
    x, m, v = params.copy(), np.zeros_like(params), np.zeros_like(params)
    el, mask = [], None

    blown = False     # a per-example gradient that is not finite (a Gamma draw of 1e-317 under a concentration of 0.05 ...)

    def oracle_step(i, Xb, mk):
        nonlocal key, x, m, v, blown
        ks = O.split(key, 3)
        L, G, n, f = O.gmm_px_grads(spec, x, Xb, O.convert_to_jax_rng_key(ks[1]), mk)
        if np.isfinite(x).all():       # (from FINITE parameters; a NaN state's NaN gradients are the empty-batch cases, held strictly)
            blown = blown or not (np.isfinite(G).all() and np.isfinite(L).all())
        eloss, avg = O.combine(O.clip_rows(G, c["clip"]), L)
        with np.errstate(all="ignore"):
            g = O.perturb(ks[2], avg, [K, K * d], c["sigma"], c["clip"], n, N, f)
        x, m, v = O.adam(x, m, v, g, i, lr=c["lr"])
        key = ks[0]
        return eloss
    if c["source"] == "explicit":
        mask = r.random(B) < c["mask_keep"]
        use_mask = c["mask_keep"] < 1.0
        losses = []
        for t in range(steps):
            idx = (np.arange(B) * 5 + 11 * t) % N
            st, loss = svi.update(st, Xd[idx], mask=torch.tensor(mask).cuda() if use_mask else True)
            losses.append(loss.reshape(()))
            el.append(oracle_step(t, X[idx], mask.astype(np.float32) if use_mask else None))
        losses = torch.stack(losses)
    else:
        _, gb = subsample_batchify_data((Xd,), B)
        st, losses = svi.run_steps(st, gb, rng.PRNGKey(c["bkey"]), c["first"], steps)
        for t in range(steps):
            idx = O.feistel_sample(O.fold_in(O.PRNGKey(c["bkey"]), c["first"] + t), N, B)
            el.append(oracle_step(t, X[idx], None))
    torch.cuda.synchronize()
    got_l, want_l = losses.detach().cpu().numpy().astype(np.float64), np.asarray(el, np.float64)
    got_p = st.optim_state[1].detach().cpu().numpy()
    why = []
    if blown:
        # The reference keeps such a step's damage to the columns the bad entries are in (inf * 0 = NaN after clipping); the fused
        # kernels poison the whole step (count column -> NaN loss, NaN state: "never finite garbage").  Held to: not finite garbage.
        c["ok"], c["why"] = bool(np.isnan(got_p).any()), "a non-finite per-example gradient: the step must not end in a finite state"
        return c
    if not np.array_equal(np.isnan(got_l), np.isnan(want_l)):
        why.append(f"losses: NaN pattern differs ({got_l.tolist()} vs {want_l.tolist()})")
    else:
        fin = ~np.isnan(want_l)
        bad = np.abs(got_l[fin] - want_l[fin]) > 1e-4 * np.abs(want_l[fin]) + 1e-6 * (K * d + N)
        if bad.any():
            k = int(np.argmax(bad))
            why.append(f"loss {k}: {got_l[fin][k]!r} vs {want_l[fin][k]!r}")
    if not np.array_equal(st.rng_key.cpu().numpy().ravel(), np.asarray(key).ravel()):
        why.append("state key differs")
    if int(st.optim_state[0]) != steps:
        why.append(f"step counter {int(st.optim_state[0])} != {steps}")
    if not np.array_equal(np.isnan(got_p), np.isnan(x)):
        why.append("parameters: NaN pattern differs")
    else:
        fin = ~np.isnan(x)
        scale = np.abs(x[fin]).max() if fin.any() else 0.0
        bad = np.abs(got_p[fin] - x[fin]) > 1e-3 * np.abs(x[fin]) + 1e-4 * max(scale, 1e-30)
        # (Adam's first steps are lr g / (|g| + 1e-8): a gradient component that is ~ 0 -- sums of float32 terms on the device, float64 in
        #  the oracle -- may take the other sign: a handful of parameters one or two steps of lr apart is that, not a defect)
        sign_flips = bad & (np.abs(got_p[fin] - x[fin]) <= 2.1 * c["lr"] * steps)
        if (bad & ~sign_flips).any() or _frac(sign_flips) > 0.005:
            k = int(np.argmax(np.abs(got_p[fin] - x[fin])))
            why.append(f"parameter: {got_p[fin][k]!r} vs {x[fin][k]!r} (largest {scale:.3g}); {int(bad.sum())} of {int(fin.sum())} out of tolerance")
    if np.isfinite(x).all() and np.isfinite(got_p).all():
        idx = (np.arange(B) * 3 + 2) % N
        got_e = float(svi.evaluate(st, Xd[idx]))
        want_e = O.gmm_evaluate(O.gmm_spec(K, d, 10.0, lik_scale=N, obs_scale=1.0), got_p, X[idx], O.convert_to_jax_rng_key(O.split(key, 1)[0]))
        if not (got_e == want_e or abs(got_e - want_e) <= 1e-4 * abs(want_e) + 1e-6 * (K * d + N) or (np.isnan(got_e) and np.isnan(want_e))):
            why.append(f"evaluate: {got_e!r} vs {want_e!r}")
    c["ok"], c["why"] = not why, "; ".join(why)
    if mask is not None:
        c["mask_sum"] = int(mask.sum())
    return c


# ------------------------------------------------------------------ the VAE (BASELINE configs[4]'s family)
def draw_vae_case(seed):
    r = np.random.default_rng(400_009 * seed + 37)
    c = {"seed": int(seed), "family": "vae"}
    c["D"] = int(r.choice([8, 12, 33, 64, 100, 200, 784]))
    c["H"] = int(r.choice([4, 7, 16, 40, 100, 400]))
    c["Z"] = int(r.choice([1, 2, 3, 8, 10, 50]))
    c["H2"] = int(r.choice([0, 0, 5, 12, 200]))
    c["B"] = int(r.choice([1, 3, 17, 64, 97, 130, 200]))
    if c["D"] * c["H"] > 200000 and c["B"] > 64:
        c["B"] = 64          # (the oracle materialises B x P gradients)
    c["steps"] = int(r.choice([1, 2, 3]))
    c["source"] = str(r.choice(["explicit", "explicit", "feistel"]))
    c["N"] = int(c["B"] * int(r.choice([1, 3, 40])))
    c["clip"] = float(r.choice([0.5, 3.0, 1e6]))
    c["sigma"] = float(r.choice([0.0, 0.8]))
    c["lr"] = float(r.choice([1e-3, 1e-2]))
    c["first"] = int(r.choice([0, 9]))
    c["grey"] = bool(r.random() < 0.4)
    c["mask_keep"] = float(r.choice([1.0, 0.7, 0.0 if r.random() < 0.3 else 0.7]))
    c["key"], c["bkey"] = int(r.integers(0, 2**31)), int(r.integers(0, 2**31))
    r2 = np.random.default_rng(1_300_021 * seed + 73)     # (later additions: their own stream)
    # (larger weights: saturated units, encoder scales exp(u) of e^+-5.  Beyond a factor of ~3 the float32 ELBO itself overflows -- losses
    #  of 1e25 .. inf, where a float64 oracle and a float32 device differ by construction: not a regime that is compared)
    #  (tried at 2, 3, 4, 12: what differs there is Adam's lr g / (|g| + 1e-8) on gradient components that saturated units make exactly
    #  0 in float32 and 1e-18 in the float64 oracle, and overflow -- nothing a trajectory comparison can hold)
    c["pscale_mult"] = 1.0
    return c


def run_vae_case(c, O, dump=False):
    import torch
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    from d3p_amd.models import Adam, Trace_ELBO, VAEGuide, VAEModel
    from d3p_amd.svi import DPSVI, DPSVIState
    r = np.random.default_rng(c["seed"] + 13)
    D, H, Z, H2, B, N, steps = c["D"], c["H"], c["Z"], c["H2"], c["B"], c["N"], c["steps"]
    spec = O.vae_spec(D, H, Z, scale=1.0, obs_scale=1.0, H2=H2)
    P = O.vae_num_params(spec)
    pscale = (0.03 if (D > 100 or H > 50 or H2 > 50) else 0.3) * c.get("pscale_mult", 1.0)
    params = (r.normal(size=P) * pscale).astype(np.float32)
    X = r.random((N, D)).astype(np.float32) if c["grey"] else (r.random((N, D)) < 0.3).astype(np.float32)
    model = VAEModel(scale=1.0 / N)
    svi = DPSVI(model, VAEGuide(model), Adam(c["lr"]), Trace_ELBO(), c["clip"], c["sigma"], num_obs_total=N, z_dim=Z,
                hidden_dim=(H, H2) if H2 else H)
    st = DPSVIState(svi.optim.init(torch.tensor(params).cuda()), rng.PRNGKey(c["key"]), 1.0)
    Xd = torch.tensor(X).cuda()
    sizes = O.vae_leaf_sizes(D, H, Z, H2)
    key = O.PRNGKey(c["key"])
    x, m, v = params.copy(), np.zeros(P, np.float32), np.zeros(P, np.float32)
    el, mask = [], None

    def oracle_step(i, Xb, mk):
        nonlocal key, x, m, v
        ks = O.split(key, 3)
        eps = O.px_eps(O.convert_to_jax_rng_key(ks[1]), Xb.shape[0], Z)
        with np.errstate(all="ignore"):
            sums, _, _ = O.vae_step_sums(spec, x, Xb, eps, c["clip"], mk)
            n = np.float32(sums[P + 1])
            f = np.float32(0.0) if n == 0 else np.float32(Xb.shape[0]) / n
            g = O.perturb(ks[2], sums[:P] / np.float32(Xb.shape[0]), sizes, c["sigma"], c["clip"], n, 1.0, f)
            eloss = float(np.float32(sums[P]) / np.float32(Xb.shape[0]) * f)
        x, m, v = O.adam(x, m, v, g, i, lr=c["lr"])
        key = ks[0]
        return eloss
    if c["source"] == "explicit":
        mask = r.random(B) < c["mask_keep"]
        use_mask = c["mask_keep"] < 1.0
        losses = []
        for t in range(steps):
            idx = (np.arange(B) * 3 + 7 * t) % N
            st, loss = svi.update(st, Xd[idx], mask=torch.tensor(mask).cuda() if use_mask else True)
            losses.append(loss.reshape(()))
            el.append(oracle_step(t, X[idx], mask.astype(np.float32) if use_mask else None))
        losses = torch.stack(losses)
    else:
        _, gb = subsample_batchify_data((Xd,), B)
        st, losses = svi.run_steps(st, gb, rng.PRNGKey(c["bkey"]), c["first"], steps)
        for t in range(steps):
            idx = O.feistel_sample(O.fold_in(O.PRNGKey(c["bkey"]), c["first"] + t), N, B)
            el.append(oracle_step(t, X[idx], None))
    torch.cuda.synchronize()
    got_l, want_l = losses.detach().cpu().numpy().astype(np.float64), np.asarray(el, np.float64)
    got_p = st.optim_state[1].detach().cpu().numpy()
    why = []
    if not np.array_equal(np.isnan(got_l), np.isnan(want_l)):
        why.append(f"losses: NaN pattern differs ({got_l.tolist()} vs {want_l.tolist()})")
    else:
        fin = ~np.isnan(want_l)
        bad = np.abs(got_l[fin] - want_l[fin]) > 1e-4 * np.abs(want_l[fin]) + 1e-5
        if bad.any():
            k = int(np.argmax(bad))
            why.append(f"loss {k}: {got_l[fin][k]!r} vs {want_l[fin][k]!r}")
    if not np.array_equal(st.rng_key.cpu().numpy().ravel(), np.asarray(key).ravel()):
        why.append("state key differs")
    if int(st.optim_state[0]) != steps:
        why.append(f"step counter {int(st.optim_state[0])} != {steps}")
    if not np.array_equal(np.isnan(got_p), np.isnan(x)):
        why.append(f"parameters: NaN pattern differs ({int(np.isnan(got_p).sum())} vs {int(np.isnan(x).sum())} of {P})")
    else:
        # Adam's early steps are lr g / (|g| + 1e-8): a component whose gradient is ~ 0 carries that component's relative error, so
        # the parameters are compared to the size of a step (the moments would be tighter; the trajectory is short)
        fin = ~np.isnan(x)
        tol = 0.02 * c["lr"] * steps + 1e-4 * np.abs(x[fin])
        bad = np.abs(got_p[fin] - x[fin]) > tol
        if _frac(bad) > 0.01:
            k = int(np.argmax(np.abs(got_p[fin] - x[fin])))
            why.append(f"parameter: {got_p[fin][k]!r} vs {x[fin][k]!r}; {int(bad.sum())} of {int(fin.sum())} out of tolerance")
    if np.isfinite(x).all() and np.isfinite(got_p).all():
        idx = (np.arange(B) * 5 + 1) % N
        got_e = float(svi.evaluate(st, Xd[idx]))
        want_e = O.vae_evaluate(O.vae_spec(D, H, Z, scale=1.0 / B, obs_scale=1.0, H2=H2), got_p, X[idx], O.convert_to_jax_rng_key(O.split(key, 1)[0]))
        if not (got_e == want_e or abs(got_e - want_e) <= 1e-4 * abs(want_e) + 1e-5 or (np.isnan(got_e) and np.isnan(want_e))):
            why.append(f"evaluate: {got_e!r} vs {want_e!r}")
    c["ok"], c["why"] = not why, "; ".join(why)
    if mask is not None:
        c["mask_sum"] = int(mask.sum())
    return c


# ------------------------------------------------------------------ the samplers and the rng suite (integer work: bit-exact)
def draw_rng_case(seed):
    r = np.random.default_rng(300_007 * seed + 31)
    c = {"seed": int(seed), "family": "rng"}
    c["capacity"] = int(r.choice([1, 2, 3, 5, 16, 17, 255, 256, 257, 1000, 65536, 65537, 10**6, 10**7 + 3]))
    c["n"] = int(min(c["capacity"], int(r.choice([1, 2, 7, 64, 1000, 4096, 50000]))))
    c["N"] = int(r.choice([1, 2, 17, 1000, 4097, 100_000, 1_000_003]))
    c["q"] = float(r.choice([0.0, 1e-4, 0.01, 0.3, 0.999, 1.0]))
    c["cutoff"] = int(max(1, min(c["N"], int(r.choice([1, 5, 100, 5000, 10**6])))))
    c["suppress"] = bool(r.random() < 0.5)
    c["shape"] = [int(v) for v in r.choice([0, 1, 2, 3, 17, 64, 1000], size=int(r.integers(0, 3)))]
    lo = int(r.choice([-5, 0, 1, -2**31, 100]))
    c["minval"], c["maxval"] = lo, int(min(lo + int(r.choice([1, 2, 3, 7, 256, 1000, 2**16 + 1, 2**31 - 1])), 2**31 - 1))
    c["fold"] = int(r.choice([0, 1, 77, 2**31, 2**32 - 1]))
    c["num"] = int(r.choice([1, 2, 3, 17, 100]))
    c["bits"] = int(r.choice([8, 16, 32, 64]))
    c["key"] = int(r.integers(0, 2**31))
    return c


def run_rng_case(c, O, dump=False):
    import torch
    import d3p_amd.random as rng
    from d3p_amd.util import feistel_indices
    why = []
    key, okey = rng.PRNGKey(c["key"]), O.PRNGKey(c["key"])
    def same(name, got, want):
        got = np.asarray(got.detach().cpu().numpy() if hasattr(got, "detach") else got)
        want = np.asarray(want)
        if got.shape != want.shape or not np.array_equal(got.astype(np.int64, casting="unsafe"), want.astype(np.int64, casting="unsafe")):
            why.append(name)
    shape = tuple(c["shape"])
    same("split", rng.split(key, c["num"]).reshape(c["num"], 16), np.asarray(O.split(okey, c["num"])).reshape(c["num"], 16))
    fk, ofk = rng.fold_in(key, c["fold"]), O.fold_in(okey, c["fold"])
    same("fold_in", fk.reshape(16), np.asarray(ofk).reshape(16))
    same("random_bits", rng.random_bits(fk, c["bits"], shape), O.random_bits(ofk, c["bits"], shape))
    same("randint", rng.randint(fk, shape, c["minval"], c["maxval"]), O.randint(ofk, shape, c["minval"], c["maxval"]))
    got_u, want_u = rng.uniform(fk, shape).cpu().numpy(), O.uniform(ofk, shape)
    if got_u.shape != want_u.shape or not np.array_equal(got_u, want_u):
        why.append("uniform")
    got_n, want_n = rng.normal(fk, shape).cpu().numpy(), O.normal(ofk, shape)
    if got_n.shape != want_n.shape or not np.allclose(got_n, want_n, rtol=2e-6, atol=1e-7):
        why.append("normal")
    same("convert_to_jax_rng_key", rng.convert_to_jax_rng_key(fk), O.convert_to_jax_rng_key(ofk))
    same("feistel", feistel_indices(fk, c["capacity"], c["n"], rng), O.feistel_sample(ofk, c["capacity"], c["n"]))
    # Poisson selection: indices of the selected rows first (ascending), padding behind; counts
    import ctypes as C
    import d3p_amd._lib as L
    lib = L.load()
    N, cutoff = c["N"], c["cutoff"]
    idxs = torch.empty(max(cutoff, 1) + 16, dtype=torch.uint32, device="cuda")
    counts = torch.empty(4, dtype=torch.uint32, device="cuda")
    ws = torch.empty(int(lib.d3p_poisson_select_workspace(N)), dtype=torch.uint8, device="cuda")
    L.check(lib.d3p_poisson_select(L.stream_ptr(), L.ptr(fk.contiguous()), float(c["q"]), N, cutoff, int(c["suppress"]), L.ptr(idxs), L.ptr(counts),
                                   L.ptr(ws), ws.numel()))
    want_idx, nsel, nvalid = O.poisson_select(ofk, np.float32(c["q"]), N, cutoff, c["suppress"])
    cn = counts.cpu().numpy()
    if int(cn[0]) != int(nsel) or int(cn[1]) != int(nvalid):
        why.append(f"poisson counts {cn[:2].tolist()} vs {[int(nsel), int(nvalid)]}")
    elif not np.array_equal(idxs[:int(nvalid)].cpu().numpy(), np.asarray(want_idx)[:int(nvalid)]):
        why.append("poisson indices")
    # the threefry suite (d3p/random/debug.py): jax.random's streams restated
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import d3p_amd.random.debug as dbg
    seed32 = c["key"] & 0x7fffffff
    tk, otk = dbg.PRNGKey(seed32), np.array([0, seed32], np.uint32)
    same("tf split", dbg.split(tk, c["num"]).reshape(c["num"], 2), np.asarray(O.tf_split(otk, c["num"])).reshape(c["num"], 2))
    tfk, otfk = dbg.fold_in(tk, c["fold"] & 0xffffffff), O.tf_fold_in(otk, c["fold"] & 0xffffffff)
    same("tf fold_in", tfk.reshape(2), np.asarray(otfk).reshape(2))
    n = int(np.prod(shape)) if shape else 1
    gu, wu = dbg.uniform(tfk, shape).cpu().numpy().ravel(), np.asarray(O.tf_uniform(otfk, n)).ravel()[:n]
    if gu.shape != wu.shape or not np.array_equal(gu, wu):
        why.append("tf uniform")
    gn, wn = dbg.normal(tfk, shape).cpu().numpy().ravel(), np.asarray(O.tf_normal(otfk, n)).ravel()[:n]
    if gn.shape != wn.shape or not np.allclose(gn, wn, rtol=2e-6, atol=1e-7):
        why.append("tf normal")
    if n and c["maxval"] > c["minval"] and c["maxval"] - c["minval"] < 2**31 and abs(c["minval"]) < 2**31 - 1:
        gi = dbg.randint(tfk, shape, c["minval"], c["maxval"]).cpu().numpy().ravel()
        wi = np.asarray(O.tf_randint(otfk, n, c["minval"], c["maxval"])).ravel()[:n]
        if gi.shape != wi.shape or not np.array_equal(gi.astype(np.int64), wi.astype(np.int64)):
            why.append("tf randint")
    # GaussianMixture.log_prob (d3p/gmm.py:71-86)
    from d3p_amd.gmm import GaussianMixture
    rr = np.random.default_rng(c["seed"] + 3)
    K, dd, nb = int(rr.choice([1, 2, 4, 5, 16, 17, 64])), int(rr.choice([1, 2, 3, 64, 100])), int(rr.choice([1, 7, 300]))
    locs, scales = rr.normal(size=(K, dd)).astype(np.float32) * 3, (0.2 + rr.random((K, dd))).astype(np.float32)
    pis = rr.dirichlet(np.ones(K)).astype(np.float32)
    xs = (rr.normal(size=(nb, dd)) * 3).astype(np.float32)
    got_lp = GaussianMixture(locs, scales, pis).log_prob(torch.tensor(xs)).cpu().numpy()
    want_lp = O.gmm_log_prob(xs, locs, scales, pis)
    if got_lp.shape != want_lp.shape or not np.allclose(got_lp, want_lp, rtol=2e-5, atol=2e-5):
        why.append("gmm log_prob")
    c["ok"], c["why"] = not why, "; ".join(why)
    return c


# ------------------------------------------------------------------ row shards: the data-parallel split, ranks emulated one after another
def draw_shards_case(seed):
    c = draw_case(104_729 * seed + 7)
    c["seed"] = int(seed)
    r = np.random.default_rng(600_011 * seed + 43)
    if c["family"] == "gauss":
        c["family"] = "logreg"
    c["guide"] = "auto"
    c["source"] = str(r.choice(["feistel", "feistel", "poisson"]))
    c["world"] = int(r.choice([1, 2, 3, 5, 8]))
    c["engine"] = str(r.choice(["fused", "two_kernel"]))
    if c["world"] == 1 and r.random() < 0.6:
        c["engine"] = "native"          # d3p_dpvi_logreg_run_dist_from without a communicator: the data-parallel native loop on one rank
    c["steps"] = min(c["steps"], 20)
    c["N"] = max(c["N"], c["world"], c["B"])
    return c


def run_shards_case(c, O, dump=False):
    """SURVEY 8(e)'s split with the ranks emulated one after another in ONE process: the table row-sharded over `world` engines, every
    engine evaluating the same sampler and processing the positions whose rows it holds, the ranks' sums added by hand (what the
    all-reduce computes), noise and Adam once per replica -- against the ORACLE's single-device trajectory; replicas bitwise equal."""
    import scipy.stats
    import torch
    import d3p_amd._lib as L
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.svi import DPSVI, DPSVIState
    r = np.random.default_rng(c["seed"] + 29)
    d, B, N, steps, world = c["d"], c["B"], c["N"], c["steps"], c["world"]
    icpt = c["family"] == "logreg_icpt"
    D = d + (1 if icpt else 0)
    P = 2 * D
    X = r.normal(size=(N, d)).astype(np.float32)
    y = (r.random(N) < 0.5).astype(np.float32)
    model = LogisticRegression(d, prior_scale=1.0, intercept=icpt, intercept_prior_scale=2.0)
    svi = DPSVI(model, AutoDiagonalNormal(model), Adam(c["lr"]), Trace_ELBO(), c["clip"], c["sigma"], num_obs_total=N)
    spec = O.logreg_spec(d, icpt, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(c["clip"], c["sigma"], c["lr"], 0.9, 0.999, 1e-8)
    loc0 = (r.normal(size=D) * c["init_scale"]).astype(np.float32)
    unc0 = (r.normal(size=D) * c["init_scale"] - 2.0).astype(np.float32)
    ost = O.LogregState(O.PRNGKey(c["key"]), D, loc0, unc0)
    st0 = DPSVIState(svi.optim.init(torch.tensor(np.concatenate([loc0, unc0])).cuda()), rng.PRNGKey(c["key"]), float(N))
    Xd, yd = torch.tensor(X).cuda(), torch.tensor(y).cuda()
    bkey = rng.PRNGKey(c["bkey"])
    poisson = c["source"] == "poisson"
    q, maxB = 0.0, B
    if poisson:
        q = B / N
        maxB = min(max(int(scipy.stats.poisson(N * q).ppf(c["quantile"])), 1), N)
    cls = ddist.HipEngine if c["engine"] == "two_kernel" else ddist.FusedHipEngine
    engines = []
    for rk in range(world):
        lo, hi = ddist.shard_rows(N, rk, world)
        engines.append(cls(svi, Xd[lo:hi], yd[lo:hi], N, lo, hi, L.D3P_BATCH_POISSON if poisson else L.D3P_BATCH_FEISTEL, maxB,
                           q=q, suppress=c["suppress"]))
    if c["engine"] == "native":
        try:
            st_n, losses_n = ddist.run_steps_native(engines[0], st0, bkey, c["first"], steps, comm=None)
        except L.D3PError as e:
            # rows too wide for the one-launch step: the native data-parallel loop says so (the torch loop's engines run them)
            c["ok"], c["why"] = "needs the fused step" in str(e) and (D > 2048 or (D > 1024 and (icpt or d % 8))), f"refused: {e}"
            return c
        torch.cuda.synchronize()
        code, _ = ddist.native_run_status(engines[0])
        if code:
            raise RuntimeError(f"native loop stopped: {L.describe_abort(code)}")
        finals, per_rank_losses = [st_n], [losses_n[:steps]]
    for e in engines if c["engine"] != "native" else []:
        e.begin(st0, bkey, c["first"])
        e.plan(steps)
    fused = c["engine"] == "fused"
    step_losses = []
    for _ in range(steps if c["engine"] != "native" else 0):
        bufs = [e.local_sums() for e in engines]
        total = torch.stack(bufs).sum(dim=0)
        outs = []
        for e, b in zip(engines, bufs):
            b.copy_(total)
            outs.append(e.finalize(b))
        if not fused:
            step_losses.append(torch.stack([o.reshape(()).clone() for o in outs]))     # (the two-kernel engine reports a step's loss at once)
    if c["engine"] != "native":
        finals = [e.end() for e in engines]
        torch.cuda.synchronize()
        per_rank_losses = [e.losses[:steps] for e in engines] if fused else list(torch.stack(step_losses).T)
    el = []
    for t in range(steps):
        fk = O.fold_in(O.PRNGKey(c["bkey"]), c["first"] + t)
        if poisson:
            idx, nsel, nvalid = O.poisson_select(fk, np.float32(q), N, maxB, c["suppress"])
            mask = (np.arange(maxB) < nvalid).astype(np.float32)
            el.append(O.logreg_update(spec, hy, ost, X[idx], y[idx], mask)[0])
        else:
            idx = O.feistel_sample(fk, N, B)
            el.append(O.logreg_update(spec, hy, ost, X[idx], y[idx])[0])
    why = []
    for f, e in zip(finals[1:], engines[1:]):
        if not (torch.equal(f.optim_state[1], finals[0].optim_state[1]) and torch.equal(f.optim_state[2], finals[0].optim_state[2])
                and torch.equal(f.rng_key, finals[0].rng_key)):
            if not (torch.isnan(f.optim_state[1]).any() and torch.equal(torch.isnan(f.optim_state[1]), torch.isnan(finals[0].optim_state[1]))):
                why.append("replicas differ")
            break
    st = finals[0]
    for pl in per_rank_losses[1:]:
        if not torch.equal(torch.nan_to_num(pl, nan=-1.0), torch.nan_to_num(per_rank_losses[0], nan=-1.0)):
            why.append("the ranks report different losses")
            break
    got_l, want_l = per_rank_losses[0].cpu().numpy().astype(np.float64), np.asarray(el, np.float64)
    got_p, want_p = st.optim_state[1].cpu().numpy(), ost.params
    if not np.array_equal(np.isnan(got_l), np.isnan(want_l)):
        why.append(f"losses: NaN pattern differs ({got_l.tolist()} vs {want_l.tolist()})")
    else:
        fin = ~np.isnan(want_l)
        bad = _far(got_l[fin], want_l[fin], LOSS_RTOL * np.abs(want_l[fin]) + 1e-6 * (D + N))
        if bad.any():
            k = int(np.argmax(bad))
            why.append(f"loss {k}: {got_l[fin][k]!r} vs {want_l[fin][k]!r}")
    if not np.array_equal(st.rng_key.cpu().numpy().ravel(), ost.key):
        why.append("state key differs")
    if int(st.optim_state[0]) != steps:
        why.append(f"step counter {int(st.optim_state[0])} != {steps}")
    if not np.array_equal(np.isnan(got_p), np.isnan(want_p)):
        why.append(f"parameters: NaN pattern differs ({int(np.isnan(got_p).sum())} vs {int(np.isnan(want_p).sum())})")
    else:
        fin = ~np.isnan(want_p)
        scale = np.abs(want_p[fin]).max() if fin.any() else 0.0
        bad = _far(got_p[fin], want_p[fin], PARAM_RTOL * np.abs(want_p[fin]) + PARAM_ATOL * max(scale, 1e-30))
        if bad.any():
            k = int(np.argmax(np.abs(got_p[fin] - want_p[fin])))
            why.append(f"parameter: {got_p[fin][k]!r} vs {want_p[fin][k]!r}; {int(bad.sum())} of {int(fin.sum())} out of tolerance")
    c["ok"], c["why"] = not why, "; ".join(why)
    return c


# ------------------------------------------------------------------ batch positions sharded over ranks (mixture model, VAE)
def draw_posshards_case(seed):
    r = np.random.default_rng(700_001 * seed + 47)
    c = draw_vae_case(seed) if r.random() < 0.5 else draw_gmm_case(seed)
    c["seed"] = int(seed)
    c["world"] = int(r.choice([1, 2, 3, 5, 8]))
    c["steps"] = min(int(c["steps"]), 2)
    c["uneven"] = bool(r.random() < 0.3)       # shards of different sizes, some of them EMPTY
    return c


def run_posshards_case(c, O, dump=False):
    """SURVEY 8(e) for the models whose BATCH is sharded by position (d3p_amd.dist.VaeHipEngine / GmmHipEngine): every emulated rank's
    local sums over its positions, added by hand, applied once per replica -- against DPSVI.update on the whole batch (itself held to the
    oracle by the `vae` / `gmm` families): keys and step bit-exact, replicas bitwise, loss and parameters to fp32 summation order."""
    import torch
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    from d3p_amd.models import Adam, GaussianMixtureGuide, GaussianMixtureModel, Trace_ELBO, VAEGuide, VAEModel
    from d3p_amd.svi import DPSVI, DPSVIState
    r = np.random.default_rng(c["seed"] + 31)
    B, N, world = c["B"], c["N"], c["world"]
    vae = c["family"] == "vae"
    if vae:
        D, H, Z, H2 = c["D"], c["H"], c["Z"], c["H2"]
        spec = O.vae_spec(D, H, Z, scale=1.0, obs_scale=1.0, H2=H2)
        P = O.vae_num_params(spec)
        params = (r.normal(size=P) * (0.03 if (D > 100 or H > 50 or H2 > 50) else 0.3)).astype(np.float32)
        X = r.random((B, D)).astype(np.float32) if c["grey"] else (r.random((B, D)) < 0.3).astype(np.float32)
        model = VAEModel(scale=1.0 / N)
        svi = DPSVI(model, VAEGuide(model), Adam(c["lr"]), Trace_ELBO(), c["clip"], c["sigma"], num_obs_total=N, z_dim=Z,
                    hidden_dim=(H, H2) if H2 else H)
        obs = 1.0
        make_engine = lambda: ddist.VaeHipEngine(svi)
    else:
        K, d = c["K"], c["d"]
        P = K + K * d
        X = (r.normal(size=(B, d)) * 3).astype(np.float32)
        params = np.concatenate([r.normal(size=K) * 0.4, r.normal(size=K * d) * 2]).astype(np.float32)
        model = GaussianMixtureModel()
        svi = DPSVI(model, GaussianMixtureGuide(model), Adam(c["lr"]), Trace_ELBO(), c["clip"], c["sigma"], k=K, d=d, num_obs_total=N)
        obs = float(N)
        make_engine = lambda: ddist.GmmHipEngine(svi)
    Xd = torch.tensor(X).cuda()
    mask = r.random(B) < c["mask_keep"]
    use_mask = c["mask_keep"] < 1.0
    md = torch.tensor(mask).cuda() if use_mask else None
    if c["uneven"] and world > 1:
        cuts = np.sort(r.integers(0, B + 1, size=world - 1))
        bounds = [0] + [int(v) for v in cuts] + [B]
    else:
        bounds = [ddist.shard_batch(B, rk, world)[0] for rk in range(world)] + [B]
    st_ref = DPSVIState(svi.optim.init(torch.tensor(params).cuda()), rng.PRNGKey(c["key"]), obs)
    states = [st_ref] * world
    why = []
    for t in range(c["steps"]):
        st_ref, loss_ref = svi.update(st_ref, Xd, mask=md if use_mask else True)
        engines = [make_engine() for _ in range(world)]
        live = []
        for rk, e in enumerate(engines):
            lo, hi = bounds[rk], bounds[rk + 1]
            if hi == lo:
                continue          # (a rank without positions contributes nothing; d3p_amd.dist.vae_run_steps skips its kernels the same way)
            e.begin(states[rk], Xd[lo:hi], B, lo, mask=None if md is None else md[lo:hi])
            live.append((rk, e))
        total = torch.stack([e.local_sums().clone() for _, e in live]).sum(dim=0)
        outs = {}
        for rk, e in live:
            outs[rk] = e.apply(total.clone())
        first = outs[live[0][0]]
        for rk, (ns, ls) in outs.items():
            if not (torch.equal(ns.rng_key, first[0].rng_key) and torch.equal(torch.nan_to_num(ns.optim_state[1], nan=7.0), torch.nan_to_num(first[0].optim_state[1], nan=7.0))):
                why.append(f"step {t}: replicas differ (rank {rk})")
                break
        states = [first[0]] * world
        got_l, want_l = float(first[1]), float(loss_ref)
        if np.isnan(got_l) != np.isnan(want_l) or (not np.isnan(want_l) and not (got_l == want_l or abs(got_l - want_l) <= 1e-4 * abs(want_l) + 1e-5)):
            why.append(f"step {t}: loss {got_l!r} vs {want_l!r}")
        if not torch.equal(first[0].rng_key, st_ref.rng_key) or int(first[0].optim_state[0]) != int(st_ref.optim_state[0]):
            why.append(f"step {t}: key or step counter differs")
        a, b = first[0].optim_state[1].cpu().numpy(), st_ref.optim_state[1].cpu().numpy()
        if not np.array_equal(np.isnan(a), np.isnan(b)):
            why.append(f"step {t}: parameters: NaN pattern differs ({int(np.isnan(a).sum())} vs {int(np.isnan(b).sum())})")
        else:
            fin = ~np.isnan(b)
            bad = _far(a[fin], b[fin], 0.02 * c["lr"] * (t + 1) + 1e-4 * np.abs(b[fin]))
            if _frac(bad) > 0.01:
                why.append(f"step {t}: {int(bad.sum())} of {int(fin.sum())} parameters out of tolerance")
        if why:
            break
    torch.cuda.synchronize()
    c["ok"], c["why"] = not why, "; ".join(why)
    c["mask_sum"] = int(mask.sum()) if use_mask else B
    return c


# ------------------------------------------------------------------ the batchifiers' get_batch (gathers: bit-exact)
def draw_batches_case(seed):
    r = np.random.default_rng(500_009 * seed + 41)
    c = {"seed": int(seed), "family": "batches"}
    c["N"] = int(r.choice([1, 2, 17, 100, 1000, 4097, 100_000]))
    c["B"] = int(max(1, min(c["N"], int(r.choice([1, 2, 7, 64, 1000, 4096])))))
    c["d"] = int(r.choice([1, 2, 3, 17, 64, 513]))
    c["kind"] = str(r.choice(["without", "with", "split", "poisson"]))
    c["i"] = int(r.choice([0, 1, 5, 1000]))
    c["quantile"] = float(r.choice([0.5, 0.99]))
    c["suppress"] = bool(r.random() < 0.4)
    c["label_dtype"] = str(r.choice(["float32", "int32", "int64"]))
    c["key"] = int(r.integers(0, 2**31))
    return c


def run_batches_case(c, O, dump=False):
    import scipy.stats
    import torch
    import d3p_amd.random as rng
    from d3p_amd.minibatch import poisson_batchify_data, split_batchify_data, subsample_batchify_data
    r = np.random.default_rng(c["seed"] + 19)
    N, B, d = c["N"], c["B"], c["d"]
    X = r.normal(size=(N, d)).astype(np.float32)
    y = (r.integers(0, 1000, size=N)).astype(c["label_dtype"])
    table = (torch.tensor(X).cuda(), torch.tensor(y).cuda())
    key, okey = rng.PRNGKey(c["key"]), O.PRNGKey(c["key"])
    why = []
    if c["kind"] in ("without", "with"):
        init, gb = subsample_batchify_data(table, B, with_replacement=c["kind"] == "with")
        nb, st = init(key)
        if nb != N // B:
            why.append("number of batches")
        bx, by = gb(c["i"], st)
        fk = O.fold_in(okey, c["i"])
        idx = O.feistel_sample(fk, N, B) if c["kind"] == "without" else np.asarray(O.randint(fk, (B,), 0, N)).astype(np.int64)
        want_x, want_y, want_m = X[idx], y[idx], None
    elif c["kind"] == "split":
        init, gb = split_batchify_data(table, B)
        nb, st = init(key)
        i = c["i"] % max(nb, 1)
        bx, by = gb(i, st)
        perm = O.feistel_sample(okey, N, N)
        idx = perm[i * B:(i + 1) * B]
        want_x, want_y, want_m = X[idx], y[idx], None
    else:
        q = min(1.0, B / N)
        maxB = min(max(int(scipy.stats.poisson(N * q).ppf(c["quantile"])), 1), N)
        init, gb = poisson_batchify_data(table, q, maxB, handle_oversized_batch="suppress" if c["suppress"] else "truncate")
        _, st = init(key) if int(q * N) > 0 else (0, key)
        (bx, by), mask = gb(c["i"], st)
        idx, nsel, nvalid = O.poisson_select(O.fold_in(okey, c["i"]), np.float32(q), N, maxB, c["suppress"])
        want_m = np.arange(maxB) < nvalid
        want_x = np.where(want_m[:, None], X[idx], 0).astype(np.float32)       # (mask * taken, minibatch.py:127-129)
        want_y = np.where(want_m, y[idx], 0).astype(y.dtype)
        if not np.array_equal(mask.cpu().numpy(), want_m):
            why.append("poisson mask")
    gx, gy = bx.cpu().numpy(), by.cpu().numpy()
    if gx.shape != want_x.shape or not np.array_equal(gx, want_x):
        why.append("batch rows")
    if gy.dtype != want_y.dtype or gy.shape != want_y.shape or not np.array_equal(gy, want_y):
        why.append(f"batch labels ({gy.dtype} {gy.shape} vs {want_y.dtype} {want_y.shape})")
    c["ok"], c["why"] = not why, "; ".join(why)
    return c


FAMILIES = {"big": (draw_big_case, None), "stepwise": (draw_stepwise_case, None), "batches": (draw_batches_case, run_batches_case), "shards": (draw_shards_case, run_shards_case),
            "posshards": (draw_posshards_case, run_posshards_case), "update": (draw_case, None), "staged": (draw_staged_case, run_staged_case), "gmm": (draw_gmm_case, run_gmm_case),
            "vae": (draw_vae_case, run_vae_case),
            "rng": (draw_rng_case, run_rng_case)}


def main():
    if len(sys.argv) > 1 and sys.argv[1] in FAMILIES:
        fam = sys.argv.pop(1)
    else:
        fam = "update"
    draw, run = FAMILIES[fam]
    run = run or run_case
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    out = open(sys.argv[3], "a") if len(sys.argv) > 3 else None
    from oracle import oracle as O
    O.build()
    bad = 0
    for seed in range(first, first + count):
        c = draw(seed)
        try:
            c = run(c, O)
        except Exception as e:  # noqa: BLE001 -- a sweep reports every case
            c["ok"], c["why"] = False, f"{type(e).__name__}: {e}"
        bad += 0 if c["ok"] else 1
        line = json.dumps(c)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()
    print(json.dumps({"fuzz_vs_oracle": "ok" if bad == 0 else "MISMATCH", "family": fam, "cases": count, "failed": bad, "first_seed": first}), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
