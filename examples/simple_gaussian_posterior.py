#!/usr/bin/env python3
"""Posterior of a Gaussian mean with DP-VI on MI355X -- the workload of the reference's
examples/simple_gaussian_posterior.py (BASELINE configs[0]): mu ~ N(0, 1)^d, obs ~ N(mu, 0.1), the
hand-written guide Normal(mu_loc, exp(mu_std_log)) started at the prior (:67-82), training with the
subsampling batchifier (:116) and evaluation with the split batchifier (:117), and the comparison with
the conjugate posterior at the end (:196-204).

Differences to the reference script, forced by the environment: model and guide are declared
(d3p_amd.models.GaussianMean / DiagonalNormalGuide) instead of traced NumPyro functions, the toy data comes
from torch's generator, and an epoch is one `run_steps` call (the reference's jit(fori_loop(...)) at :145-157).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import d3p_amd._lib as L  # noqa: E402
import d3p_amd.random as rng_suite  # noqa: E402
from d3p_amd.minibatch import split_batchify_data, subsample_batchify_data  # noqa: E402
from d3p_amd.models import Adam, DiagonalNormalGuide, GaussianMean, Trace_ELBO  # noqa: E402
from d3p_amd.svi import DPSVI  # noqa: E402


def create_toy_data(N, d, seed=1234):
    """2N draws of obs ~ N(mu_true = 1, 0.1): N for training, N held out (reference :102-111)."""
    g = torch.Generator().manual_seed(seed)
    X = (1.0 + 0.1 * torch.randn(2 * N, d, generator=g)).cuda()
    return X[:N].contiguous(), X[N:].contiguous(), torch.ones(d, device="cuda")


def main(args):
    L.require_device()
    N = args.num_samples
    X_train, X_test, mu_true = create_toy_data(N, args.dimensions)
    train_init, train_fetch = subsample_batchify_data((X_train,), batch_size=args.batch_size, rng_suite=rng_suite)
    test_init, test_fetch = split_batchify_data((X_test,), batch_size=args.batch_size, rng_suite=rng_suite)

    model = GaussianMean(args.dimensions, prior_scale=1.0, obs_scale=0.1)
    svi = DPSVI(model, DiagonalNormalGuide(model), Adam(args.learning_rate), Trace_ELBO(), dp_scale=args.sigma,
                clipping_threshold=args.clip_threshold, d=args.dimensions, num_obs_total=N, rng_suite=rng_suite)

    dpsvi_rng = rng_suite.PRNGKey(0)
    dpsvi_rng, svi_init_rng, batchifier_rng = rng_suite.split(dpsvi_rng, 3)
    _, batchifier_state = train_init(rng_key=batchifier_rng)
    svi_state = svi.init(svi_init_rng, *train_fetch(0, batchifier_state))

    q = args.batch_size / N
    delta = getattr(args, "delta", None)
    if delta is not None:  # examples/simple_gaussian_posterior.py:136-140
        eps = svi.get_epsilon(delta, q, num_epochs=args.num_epochs)
        print("Privacy epsilon {} (for sigma: {}, delta: {}, C: {}, q: {})".format(eps, args.sigma, delta, args.clip_threshold, q))

    for i in range(args.num_epochs):
        t0 = time.time()
        dpsvi_rng, data_fetch_rng = rng_suite.split(dpsvi_rng, 2)
        num_batches, batchifier_state = train_init(rng_key=data_fetch_rng)
        svi_state, losses = svi.run_steps(svi_state, train_fetch, batchifier_state, 0, num_batches)
        train_loss = float(losses.sum()) / (N * num_batches)
        t1 = time.time()
        if i % max(args.num_epochs // 10, 1) == 0:
            dpsvi_rng, test_fetch_rng = rng_suite.split(dpsvi_rng, 2)
            num_test_batches, test_state = test_init(rng_key=test_fetch_rng)
            test_loss = sum(float(svi.evaluate(svi_state, *test_fetch(j, test_state)))
                            for j in range(num_test_batches)) / (N * num_test_batches)
            print("Epoch {}: loss = {:.4f} (on training set: {:.4f}) ({:.3f} s.)".format(i, test_loss, train_loss, t1 - t0))

    params = svi.get_params(svi_state)
    mu_loc, mu_std = params["mu_loc"], torch.exp(params["mu_std_log"])
    print("### expected: {}".format(mu_true.tolist()))
    print("### svi result\nmu_loc: {}\nerror: {:.5f}\nmu_std: {}".format(
        mu_loc.tolist(), float(torch.linalg.norm(mu_loc - mu_true)), mu_std.tolist()))
    a_loc, a_std = GaussianMean.analytical_solution(X_train, 1.0, 0.1)
    print("### analytical solution\nmu_loc: {}\nerror: {:.5f}\nmu_std: {:.5f}".format(
        a_loc.tolist(), float(torch.linalg.norm(a_loc - mu_true)), a_std))
    return mu_loc, mu_std, a_loc, a_std


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="parse args")
    parser.add_argument('-n', '--num-epochs', default=100, type=int, help='number of training epochs')
    parser.add_argument('-lr', '--learning-rate', default=1.0e-3, type=float, help='learning rate')
    parser.add_argument('-batch-size', default=100, type=int, help='batch size')
    parser.add_argument('-d', '--dimensions', default=4, type=int, help='data dimension')
    parser.add_argument('-N', '--num-samples', default=10000, type=int, help='data samples count')
    parser.add_argument('--sigma', default=1.0, type=float, help='privacy scale')
    parser.add_argument('--delta', default=1e-5, type=float, help='privacy slack parameter delta')
    parser.add_argument('-C', '--clip-threshold', default=1., type=float, help='clipping threshold for gradients')
    main(parser.parse_args())
