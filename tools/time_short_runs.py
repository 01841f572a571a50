#!/usr/bin/env python3
"""Wall time of short device-resident runs (the driver's `--steps 20 --warmup 5` shape) through the different entry points:
DPSVI.run_steps (status check on / off), d3p_amd.dist.run_steps_native without and with the one-shot exchange.  Developer tool.

    python tools/time_short_runs.py [steps=20] [repeats=30]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import d3p_amd._lib as L
import d3p_amd.random as rng
from d3p_amd import dist as ddist
from d3p_amd.minibatch import subsample_batchify_data
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
L.require_device()
lib = L.load()
N, d, B = 1_000_000, 512, 4096
X = torch.empty((N, d), device="cuda")
y = torch.empty(N, device="cuda")
L.check(lib.d3p_synth_logreg(L.stream_ptr(), 123, 0, N, d, L.ptr(X), L.ptr(y)))
model = LogisticRegression(d, prior_scale=1.0)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, num_obs_total=N)
params = torch.cat([torch.zeros(d, device="cuda"), torch.full((d,), svi.guide.unconstrained_init_scale(), device="cuda")])
st0 = DPSVIState(svi.optim.init(params), rng.PRNGKey(0), float(N))
bkey = rng.PRNGKey(1)
_, gb = subsample_batchify_data((X, y), B)
eng = ddist.FusedHipEngine(svi, X, y, N, 0, N, L.D3P_BATCH_FEISTEL, B)
lo, hi = ddist.shard_rows(8 * N, 0, 8)
comm = ddist.XchgComm(2 * d + 4)
eng8 = ddist.FusedHipEngine(svi, X, y, 8 * N, lo, hi, L.D3P_BATCH_FEISTEL, 8 * B)


def timed(fn):
    st = st0
    for _ in range(20):
        st = fn(st, 2048)          # clocks up
    torch.cuda.synchronize()
    best, tot = 1e9, 0.0
    for r in range(reps):
        st = fn(st, 5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = fn(st, steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best, tot = min(best, dt), tot + dt
    return 1e6 * best, 1e6 * tot / reps


cases = {
    "DPSVI.run_steps (status check)": lambda st, k: svi.run_steps(st, gb, bkey, 0, k)[0],
    "DPSVI.run_steps (check_status=False)": lambda st, k: svi.run_steps(st, gb, bkey, 0, k, check_status=False)[0],
    "dist.run_steps_native, no comm": lambda st, k: ddist.run_steps_native(eng, st, bkey, 0, k, comm=None, collect_losses=False)[0],
    "dist.run_steps_native, exchange (rank 0 of 8 emulated)": lambda st, k: ddist.run_steps_native(eng8, st, bkey, 0, k, comm=comm, collect_losses=False)[0],
}
for timing in (0, 1):
    L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(timing))
    for name, fn in cases.items():
        b, m = timed(fn)
        print(f"[kernel timing {'on ' if timing else 'off'}] {name:58s} best {b:7.1f} us  mean {m:7.1f} us  -> {steps / (m * 1e-6):9.0f} steps/s (mean)", flush=True)
L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(0))
comm.close()
