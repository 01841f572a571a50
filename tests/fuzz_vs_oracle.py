#!/usr/bin/env python3
"""Randomised sweep of the update path against the CPU oracle (test infrastructure: the oracle is the checker).

A case = one seed -> a random model family (logistic regression with / without intercept, Gaussian mean), guide (AutoDiagonalNormal,
the exp-parametrised hand-written one, the example's two-site guide), shape (d from 1 to wide rows, B from 1 to 5000), table size,
batch source (Feistel subsampling or Poisson sampling through `run_steps`, or `update` on an explicit batch with a random mask),
clipping threshold, noise scale, step size, number of steps and first batch index.  The HIP trajectory is compared with the oracle's
restatement of d3p/svi.py:395-434 driven by the oracle's own samplers: state key bit-exact, every loss (rtol 1e-4 + 1e-6 (D + N): the loss is a difference of large sums), the NaN pattern of
losses and parameters (empty batches: svi.py:305, :365), final parameters
(rtol 5e-4, atol 5e-5 of the largest) and step counter.

    python tests/fuzz_vs_oracle.py [first_seed=0] [count=40] [out.jsonl]

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
    if gauss:
        model = GaussianMean(d, prior_scale=1.5, obs_scale=0.7)
        guide = AutoDiagonalNormal(model) if c["guide"] == "auto" else DiagonalNormalGuide(model)
        spec = O.gauss_mean_spec(d, prior=1.5, lik_sigma=0.7, lik_scale=N, obs_scale=N, guide_exp=c["guide"] != "auto")
    else:
        model = LogisticRegression(d, prior_scale=1.0, intercept=icpt, intercept_prior_scale=2.0)
        guide = MeanFieldGuide(model) if c["guide"] == "meanfield" else AutoDiagonalNormal(model)
        spec = O.logreg_spec(d, icpt, 1.0, 2.0, lik_scale=N, obs_scale=N, guide_exp=c["guide"] == "meanfield")
    svi = DPSVI(model, guide, Adam(c["lr"]), Trace_ELBO(), c["clip"], c["sigma"], num_obs_total=N, **({"d": d} if gauss else {}))
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
        unc0 = (r.normal(size=D) * c["init_scale"] - 2.0).astype(np.float32)
        ost = O.LogregState(O.PRNGKey(c["key"]), D, loc0, unc0)
        params0 = np.concatenate([loc0, unc0])
    st = DPSVIState(svi.optim.init(torch.tensor(params0).cuda()), rng.PRNGKey(c["key"]), float(N))
    Xd = torch.tensor(X).cuda()
    yd = None if gauss else torch.tensor(y).cuda()
    table = (Xd,) if gauss else (Xd, yd)
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
        st, losses = svi.run_steps(st, gb, rng.PRNGKey(c["bkey"]), c["first"], steps)
        for t in range(steps):
            idx = O.feistel_sample(O.fold_in(O.PRNGKey(c["bkey"]), c["first"] + t), N, B)
            el.append(upd(spec, hy, ost, X[idx], None if gauss else y[idx])[0])
    else:
        q = B / N
        if int(q * N) == 0:
            q = 1.0 / N
        maxB = min(max(int(scipy.stats.poisson(N * q).ppf(c["quantile"])), 1), N)   # (max_batch_size > N is an error, minibatch.py:116)
        _, gb = poisson_batchify_data(table, q, maxB, handle_oversized_batch="suppress" if c["suppress"] else "truncate")
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
    both_nan = np.isnan(got_l) & np.isnan(want_l)
    if not np.array_equal(np.isnan(got_l), np.isnan(want_l)):
        why.append("losses: NaN pattern differs")
    else:
        # (the loss is a difference of sums over D latent terms and N-scaled likelihood terms: an absolute fp32 floor beside the rtol)
        ok_l = both_nan | (np.abs(got_l - want_l) <= LOSS_RTOL * np.abs(want_l) + 1e-6 * (D + N))
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
        bad = np.abs(got_p[fin] - want_p[fin]) > PARAM_RTOL * np.abs(want_p[fin]) + PARAM_ATOL * max(scale, 1e-30)
        if bad.any():
            k = int(np.argmax(np.abs(got_p[fin] - want_p[fin])))
            why.append(f"parameter: {got_p[fin][k]!r} vs {want_p[fin][k]!r} (largest {scale:.3g}); {int(bad.sum())} of {int(fin.sum())} out of tolerance")
    if dump:
        c["got_losses"], c["want_losses"] = [float(v) for v in got_l], [float(v) for v in want_l]
        c["got_nan_params"], c["want_nan_params"] = int(np.isnan(got_p).sum()), int(np.isnan(want_p).sum())
        c["mask_sum"] = int(mask.sum()) if c["source"] != "feistel" else None
    c["ok"], c["why"] = not why, "; ".join(why)
    c["final_loss"] = float(want_l[-1]) if len(want_l) else None
    return c


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    out = open(sys.argv[3], "a") if len(sys.argv) > 3 else None
    from oracle import oracle as O
    O.build()
    bad = 0
    for seed in range(first, first + count):
        c = draw_case(seed)
        try:
            c = run_case(c, O)
        except Exception as e:  # noqa: BLE001 -- a sweep reports every case
            c["ok"], c["why"] = False, f"{type(e).__name__}: {e}"
        bad += 0 if c["ok"] else 1
        line = json.dumps(c)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()
    print(json.dumps({"fuzz_vs_oracle": "ok" if bad == 0 else "MISMATCH", "cases": count, "failed": bad, "first_seed": first}), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
