"""Soak test of the data-parallel chained launch, updater form (developer tool): long identical runs of one rank -- exchanging with
itself, and as rank 1 of 8 against d3p_xchg_simulate_peers -- must be bitwise identical run to run, also while matrix products
on a second stream compete for the CUs, and no bounded wait may hit its bound.
usage: python tools/soak_xchg.py [steps=50000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import d3p_amd._lib as L
import d3p_amd.random as rng
from d3p_amd import dist as ddist
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
N, d, B, world, rank = 400000, 512, 32768, 8, 1
g = torch.Generator().manual_seed(0)
lo, hi = ddist.shard_rows(N, rank, world)
X = torch.randn(hi - lo, d, generator=g).cuda()
y = (torch.rand(hi - lo, generator=g) < 0.5).float().cuda()
model = LogisticRegression(d)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, num_obs_total=N)
st = DPSVIState(svi.optim.init(torch.cat([torch.zeros(d), torch.full((d,), -2.0)]).cuda()), rng.PRNGKey(3), float(N))
eng = ddist.FusedHipEngine(svi, X, y, N, lo, hi, L.D3P_BATCH_FEISTEL, B)
side = ddist.concurrent_streams(1)[0]
mats = [torch.randn(n, n, device="cuda") for n in (512, 1024, 3072)]


def one(comm, sim, disturbed):
    if sim:
        with torch.cuda.stream(side):
            comm.simulate_peers(steps)
    t0 = time.perf_counter()
    s2, losses = ddist.run_steps_native(eng, st, rng.PRNGKey(4), 0, steps, comm=comm)
    k = 0
    if disturbed:
        done = torch.cuda.Event()
        done.record()
        with torch.cuda.stream(side):
            while not done.query():
                a = mats[k % 3]
                (a @ a).sum()
                k += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    code, nonfinite = ddist.native_run_status(eng)
    assert code == 0, L.describe_abort(code)
    return s2.optim_state[1].clone(), losses.clone(), dt, k


outs = []
solo = ddist.XchgComm(2 * d + 4)
for rep, disturbed in enumerate((False, False, True)):
    p, l, dt, k = one(solo, False, disturbed)
    outs.append((p, l))
    print(f"self-exchange run {rep}{' (disturbed by %d matrix products)' % k if disturbed else ''}: {steps} steps in {dt:.2f} s = {dt / steps * 1e6:.2f} us/step, "
          f"last loss {float(l[-1]):.3f}", flush=True)
solo.close()
comms = ddist.XchgComm.local_group(world, 2 * d + 4)
for rep in range(2):
    p, l, dt, _ = one(comms[rank], True, False)
    outs.append((p, l))
    print(f"rank {rank} of {world} against simulated peers, run {rep}: {dt / steps * 1e6:.2f} us/step", flush=True)
for c in comms:
    c.close()
same = all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs[1:])
finite = bool(torch.isfinite(outs[0][0]).all()) and bool(torch.isfinite(outs[0][1]).all())
print(f"bitwise identical runs: {same}; finite: {finite}", flush=True)
assert same and finite
print("soak ok")
