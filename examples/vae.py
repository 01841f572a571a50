#!/usr/bin/env python3
"""Variational auto-encoder with DP-VI on MI355X -- the workload of the reference's examples/vae.py (BASELINE
config 5): encoder :65-87 and decoder :90-104 (stax Dense / softplus / exp / sigmoid), model :106-135 and guide
:138-153 wrapped in handlers.scale(1 / num_samples) :194-195, DPSVI with clipping threshold 10 :211-215, subsampled
training batches :172 and held-out evaluation.

Differences to the reference script, forced by the environment: MNIST cannot be downloaded (no network), so the data are
synthetic 28 x 28 binary images (ten random prototype patterns with 5 % pixel flips); encoder and decoder are the
declared d3p_amd.models.VAEGuide / VAEModel instead of stax modules traced by NumPyro; reconstructions are summarised by
their pixel error instead of being written as image files.  As in the reference (:199-207) dp_scale is calibrated for
--epsilon by d3p_amd.dputil.approximate_sigma; --sigma gives it directly.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import d3p_amd._lib as L  # noqa: E402
import d3p_amd.random as rng_suite  # noqa: E402
from d3p_amd.minibatch import subsample_batchify_data  # noqa: E402
from d3p_amd.models import Adam, Trace_ELBO, VAEGuide, VAEModel  # noqa: E402
from d3p_amd.svi import DPSVI  # noqa: E402


def synthetic_images(N, seed=0, classes=10, flip=0.05):
    g = torch.Generator().manual_seed(seed)
    protos = (torch.rand(classes, 28, 28, generator=g) < 0.25).float()
    which = torch.randint(0, classes, (N,), generator=g)
    flips = (torch.rand(N, 28, 28, generator=g) < flip).float()
    return ((protos[which] + flips) % 2).cuda(), which.cuda(), protos.cuda()


def reconstruct(params, x):
    """Mean reconstruction: encode to z_loc, decode (vae.py:262-296 without the sampling)."""
    sp = torch.nn.functional.softplus
    (W1, b1), _, _, ((Wl, bl), _) = params["encoder$params"]
    (V1, c1), _, (V2, c2), _ = params["decoder$params"]
    z = sp(x.reshape(x.shape[0], -1) @ W1 + b1) @ Wl + bl
    return torch.sigmoid(sp(z @ V1 + c1) @ V2 + c2)


def main(args):
    L.require_device()
    N = args.num_samples
    X, _, _ = synthetic_images(2 * N)
    X_train, X_test = X[:N].contiguous(), X[N:].contiguous()
    train_init, train_fetch = subsample_batchify_data((X_train,), batch_size=args.batch_size, rng_suite=rng_suite)

    dp_scale = getattr(args, "sigma", None)
    if dp_scale is None:  # examples/vae.py:199-207 (substitution relation: batches of fixed size without replacement)
        from d3p_amd.dputil import approximate_sigma
        num_iter = (N // args.batch_size) * args.num_epochs
        dp_scale, eps, _ = approximate_sigma(args.epsilon, 1 / N, args.batch_size / N, num_iter, maxeval=20)
        print(f"using noise scale {dp_scale} for epsilon of {eps} (targeted: {args.epsilon})")
    model = VAEModel(scale=1.0 / N)
    svi = DPSVI(model, VAEGuide(model), Adam(args.learning_rate), Trace_ELBO(), dp_scale=dp_scale,
                clipping_threshold=10., num_obs_total=N, z_dim=args.z_dim, hidden_dim=args.hidden_dim,
                rng_suite=rng_suite)
    dpsvi_rng = rng_suite.PRNGKey(0)
    dpsvi_rng, svi_init_rng, batchifier_rng = rng_suite.split(dpsvi_rng, 3)
    _, batchifier_state = train_init(rng_key=batchifier_rng)
    svi_state = svi.init(svi_init_rng, *train_fetch(0, batchifier_state))

    errs = []
    for i in range(args.num_epochs):
        t0 = time.time()
        dpsvi_rng, data_fetch_rng = rng_suite.split(dpsvi_rng, 2)
        num_batches, batchifier_state = train_init(rng_key=data_fetch_rng)
        losses = []
        for j in range(num_batches):
            svi_state, loss = svi.update(svi_state, *train_fetch(j, batchifier_state))
            losses.append(loss)
        train_loss = float(torch.stack(losses).mean())
        t1 = time.time()
        err = float((reconstruct(svi.get_params(svi_state), X_test[:512]) - X_test[:512].reshape(512, -1)).abs().mean())
        errs.append(err)
        print("Epoch {}: loss on training set = {:.2f}, mean abs reconstruction error = {:.4f} ({:.2f} s.)".format(
            i, train_loss, err, t1 - t0))
    return errs


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="parse args")
    parser.add_argument('-n', '--num-epochs', default=20, type=int, help='number of training epochs')
    parser.add_argument('-lr', '--learning-rate', default=1.0e-3, type=float, help='learning rate')
    parser.add_argument('-batch-size', default=128, type=int, help='batch size')
    parser.add_argument('-z-dim', default=50, type=int, help='size of latent')
    parser.add_argument('-hidden-dim', default=400, type=int, help='size of hidden layer in encoder/decoder networks')
    parser.add_argument('-N', '--num-samples', default=8192, type=int, help='training images')
    parser.add_argument('--epsilon', default=1., type=float, help='targeted value for privacy parameter epsilon')
    parser.add_argument('--sigma', default=None, type=float, help='dp_scale of the Gaussian mechanism (overrides --epsilon)')
    main(parser.parse_args())
