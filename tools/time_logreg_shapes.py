"""Times DPSVI.run_steps of the logistic-regression path for a few model shapes (developer tool).
usage: python tools/time_logreg_shapes.py  ->  one line per shape: d, intercept, batch, us per step"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3p_amd.random as rng
from d3p_amd.minibatch import subsample_batchify_data
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState

def run(d, icpt, B, N=200000, steps=960):
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, generator=g).cuda()
    y = (torch.rand(N, generator=g) < 0.5).float().cuda()
    model = LogisticRegression(d, intercept=icpt)
    svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, num_obs_total=N)
    D = d + int(icpt)
    st = DPSVIState(svi.optim.init(torch.cat([torch.zeros(D), torch.full((D,), -2.0)]).cuda()), rng.PRNGKey(3), float(N))
    _, gb = subsample_batchify_data((X, y), B)
    st, _ = svi.run_steps(st, gb, rng.PRNGKey(4), 0, 96)
    st, _ = svi.run_steps(st, gb, rng.PRNGKey(4), 96, steps)  # (first call of a new shape / length pays one-off set-up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st, losses = svi.run_steps(st, gb, rng.PRNGKey(4), 96, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"d={d} intercept={icpt} B={B}: {dt / steps * 1e6:.2f} us/step  loss {float(losses[-1]):.3f}", flush=True)

if __name__ == "__main__":
    for d, icpt, B in ((4, True, 4096), (4, True, 256), (16, False, 4096), (512, False, 4096), (512, True, 4096), (256, False, 4096), (256, True, 4096), (64, True, 1024),
                       (1024, False, 4096), (1024, True, 4096), (100, False, 4096), (300, False, 4096), (2048, False, 4096)):
        run(d, icpt, B)
