#!/usr/bin/env python3
"""Logistic regression with DP-VI on MI355X -- the workload of the reference's
examples/logistic_regression.py (model :49-66, training loop :118-204) on the d3p_amd surface.

Differences to the reference script, all forced by the environment: the model is declared
(d3p_amd.models.LogisticRegression) instead of traced from a NumPyro function, and an epoch is one `run_steps` call (the
reference's jit(fori_loop(...)) at :149-160).  `--guide handwritten` is the script's OWN guide (:67-86: sample sites 'w' and
'intercept', four parameter leaves with exp scales, one perturbation key per leaf) through the five-stage composition;
`--guide auto` (default) is AutoDiagonalNormal (README.md:99) on the fused device-resident loop.  As in the reference (:135-137) dp_scale is calibrated for --epsilon, with
d3p_amd.dputil on the restated Fourier accountant; --sigma gives it directly.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import d3p_amd._lib as L  # noqa: E402
import d3p_amd.random as rng_suite  # noqa: E402
from d3p_amd.minibatch import poisson_batchify_data, split_batchify_data  # noqa: E402
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, MeanFieldGuide, Trace_ELBO  # noqa: E402
from d3p_amd.svi import DPSVI  # noqa: E402


def create_toy_data(N, d, seed=123):
    """X ~ N(0, 1), y ~ Bernoulli(sigmoid(X w + b)) generated on the device (reference :88-104)."""
    lib = L.load()
    X = torch.empty((2 * N, d), dtype=torch.float32, device="cuda")
    y = torch.empty(2 * N, dtype=torch.float32, device="cuda")
    L.check(lib.d3p_synth_logreg(L.stream_ptr(), seed, 0, 2 * N, d, L.ptr(X), L.ptr(y)))
    return (X[:N].contiguous(), y[:N].contiguous()), (X[N:].contiguous(), y[N:].contiguous())


def main(args):
    L.require_device()
    train, test = create_toy_data(args.num_samples, args.dimensions)
    N = args.num_samples
    q = args.batch_size / N
    train_init, train_fetch = poisson_batchify_data(train, q, max_batch_size=.99, rng_suite=rng_suite)
    test_init, test_fetch = split_batchify_data(test, batch_size=args.batch_size, rng_suite=rng_suite)

    dpsvi_rng = rng_suite.PRNGKey(0)
    dpsvi_rng, svi_init_rng, data_fetch_rng = rng_suite.split(dpsvi_rng, 3)
    num_iter_per_epoch, batchifier_state = train_init(rng_key=data_fetch_rng)
    sample_batch, _ = train_fetch(0, batchifier_state)

    dp_scale = getattr(args, "sigma", None)
    if dp_scale is None:  # examples/logistic_regression.py:135-137: calibrate the noise for the target epsilon
        from d3p_amd.dputil import approximate_sigma_remove_relation
        dp_scale, eps, _ = approximate_sigma_remove_relation(args.epsilon, delta=1 / N**2, q=q,
                                                             num_iter=num_iter_per_epoch * args.num_epochs)
        print("noise scale {:.4f} for epsilon {:.4f}, delta {:.2e}".format(dp_scale, eps, 1 / N**2))
    model = LogisticRegression(args.dimensions, prior_scale=1.0, intercept=True)
    guide = MeanFieldGuide(model) if getattr(args, "guide", "auto") == "handwritten" else AutoDiagonalNormal(model)
    svi = DPSVI(model, guide, Adam(args.learning_rate), Trace_ELBO(), dp_scale=dp_scale,
                clipping_threshold=1., num_obs_total=N, rng_suite=rng_suite)
    svi_state = svi.init(svi_init_rng, *sample_batch)

    accs, train_losses = [], []
    for i in range(args.num_epochs):
        t0 = time.time()
        dpsvi_rng, data_fetch_rng = rng_suite.split(dpsvi_rng, 2)
        num_batches, batchifier_state = train_init(rng_key=data_fetch_rng)
        svi_state, losses = svi.run_steps(svi_state, train_fetch, batchifier_state, 0, num_batches)
        train_loss = float(losses.sum()) / (N * num_batches)
        torch.cuda.synchronize()
        t1 = time.time()

        dpsvi_rng, test_fetch_rng = rng_suite.split(dpsvi_rng, 2)
        num_test_batches, test_state = test_init(rng_key=test_fetch_rng)
        params = svi.get_params(svi_state)
        if "w_loc" in params:   # the hand-written guide's leaves
            w, b = params["w_loc"], params["intercept_loc"]
        else:
            w, b = params["auto_loc"][:-1], params["auto_loc"][-1]
        test_loss, acc = 0.0, 0.0
        for j in range(num_test_batches):
            bx, by = test_fetch(j, test_state)
            test_loss += float(svi.evaluate(svi_state, bx, by)) / (N * num_test_batches)
            acc += float(((bx @ w + b > 0).float() == by).float().mean()) / num_test_batches
        accs.append(acc)
        train_losses.append(train_loss)
        print("Epoch {}: loss = {:.4f}, acc = {:.4f} (loss on training set: {:.4f}) ({:.3f} s.)".format(
            i, test_loss, acc, train_loss, t1 - t0))
    return accs, train_losses


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="parse args")
    parser.add_argument('-e', '--epsilon', default=.1, type=float, help='privacy epsilon (delta = 1 / N^2)')
    parser.add_argument('--sigma', default=None, type=float, help='dp_scale of the Gaussian mechanism (overrides --epsilon)')
    parser.add_argument('-n', '--num-epochs', default=10, type=int, help='number of training epochs')
    parser.add_argument('-lr', '--learning-rate', default=1.0e-2, type=float, help='learning rate')
    parser.add_argument('-batch-size', default=200, type=int, help='batch size')
    parser.add_argument('-d', '--dimensions', default=4, type=int, help='data dimension')
    parser.add_argument('-N', '--num-samples', default=10000, type=int, help='data samples count')
    parser.add_argument('--guide', choices=["auto", "handwritten"], default="auto",
                        help="auto: AutoDiagonalNormal (README.md:99); handwritten: the reference script's own two-site guide (:67-86)")
    main(parser.parse_args())
