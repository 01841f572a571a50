#!/usr/bin/env python3
"""Soak of the mixture-model and VAE run loops (developer tool): a long device-resident run taken twice from the same state must
end in bitwise identical states and finite losses -- the grouped weight-gradient launch, the tile sums, the in-launch Feistel
indices and the fused gather of the VAE step, the two-launch mixture-model step with its key-chain links.

    python tools/soak_models.py [vae_steps=3000] [gmm_steps=20000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import d3p_amd.random as rng
from d3p_amd.minibatch import subsample_batchify_data
from d3p_amd.models import Adam, GaussianMixtureGuide, GaussianMixtureModel, Trace_ELBO, VAEGuide, VAEModel
from d3p_amd.svi import DPSVI

vae_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
gmm_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ok = True


def twice(name, svi, st0, gb, bstate, steps):
    global ok
    outs = []
    for r in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, losses = svi.run_steps(st0, gb, bstate, 0, steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name} run {r}: {steps} steps in {dt:.2f} s = {dt / steps * 1e6:.1f} us/step, last loss {float(losses[-1]):.6g}", flush=True)
        outs.append((st, losses))
    same = all(torch.equal(a, b) for a, b in zip(outs[0][0].optim_state, outs[1][0].optim_state)) and torch.equal(outs[0][1], outs[1][1]) \
        and torch.equal(outs[0][0].rng_key, outs[1][0].rng_key)
    fin = bool(torch.isfinite(outs[0][1]).all()) and bool(torch.isfinite(outs[0][0].optim_state[1]).all())
    print(f"{name}: bitwise identical runs: {same}; finite: {fin}", flush=True)
    ok = ok and same and fin


N, B, D, H, Z = 60000, 4096, 784, 400, 50
X = (torch.rand(N, D, generator=torch.Generator().manual_seed(1)) < 0.3).float().cuda()
for h2 in (0, 200):
    model = VAEModel(scale=1.0 / N)
    svi = DPSVI(model, VAEGuide(model), Adam(1e-3), Trace_ELBO(), 10.0, 1.0, num_obs_total=N, z_dim=Z, hidden_dim=(H, h2) if h2 else H)
    init, gb = subsample_batchify_data((X,), B)
    _, bs = init(rng.PRNGKey(5))
    twice("VAE 784-%s-50, B = 4096" % ("[400, 200]" if h2 else "400"), svi, svi.init(rng.PRNGKey(0), X[:B]), gb, bs, vae_steps)
del X
K, d, Bg, Ng = 16, 64, 8192, 10**6
Xg = torch.randn(Ng, d, generator=torch.Generator().manual_seed(0)).cuda() * 3
model = GaussianMixtureModel()
svi = DPSVI(model, GaussianMixtureGuide(model), Adam(1e-3), Trace_ELBO(), 20.0, 1.0, k=K, d=d, num_obs_total=Ng)
init, gb = subsample_batchify_data((Xg,), Bg)
_, bs = init(rng.PRNGKey(5))
twice("mixture model K = 16, d = 64, B = 8192", svi, svi.init(rng.PRNGKey(0), Xg[:Bg]), gb, bs, gmm_steps)
print("soak ok" if ok else "SOAK FAILED", flush=True)
sys.exit(0 if ok else 1)
