"""Times a Python loop of DPSVI.update() calls (the reference's per-step API) against DPSVI.run_steps (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3p_amd.random as rng
from d3p_amd.minibatch import subsample_batchify_data
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState

N, d, B, steps = 200000, 512, 4096, 500
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g).cuda(); y = (torch.rand(N, generator=g) < 0.5).float().cuda()
model = LogisticRegression(d)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, num_obs_total=N)
st = DPSVIState(svi.optim.init(torch.cat([torch.zeros(d), torch.full((d,), -2.0)]).cuda()), rng.PRNGKey(3), float(N))
init, gb = subsample_batchify_data((X, y), B)
_, bstate = init(rng.PRNGKey(4))
for phase in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = st
    for t in range(steps):
        s, loss = svi.update(s, *gb(t, bstate))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"update() loop incl. get_batch: {dt / steps * 1e6:.1f} us/step", flush=True)
batch = gb(0, bstate)
torch.cuda.synchronize(); t0 = time.perf_counter()
s = st
for t in range(steps):
    s, loss = svi.update(s, *batch)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"update() loop on one resident batch: {dt / steps * 1e6:.1f} us/step", flush=True)
