import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
import d3p_amd._lib as L
import d3p_amd.random as rng
from d3p_amd.minibatch import subsample_batchify_data
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState
import d3p_amd.svi as S
lib = L.load(); L.require_device()
dev = torch.device("cuda:0")
N, d, B = 1_000_000, 512, 4096
X = torch.empty((N, d), device=dev); y = torch.empty(N, device=dev)
L.check(lib.d3p_synth_logreg(L.stream_ptr(), 123, 0, N, d, L.ptr(X), L.ptr(y)))
model = LogisticRegression(d)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, N=N)
st = DPSVIState(svi.optim.init(torch.cat([torch.zeros(d, device=dev), torch.full((d,), -2.25, device=dev)])), rng.PRNGKey(0), float(N))
_, gb = subsample_batchify_data((X, y), B)
bkey = rng.PRNGKey(1)
st, _ = svi.run_steps(st, gb, bkey, 0, 5)
torch.cuda.synchronize()
for steps in (20, 20, 20, 128, 2048):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st2, losses = svi.run_steps(st, gb, bkey, 5, steps, check_status=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    a, nf = svi.last_run_status()
    t3 = time.perf_counter()
    print(f"steps {steps}: host enqueue {1e6*(t1-t0):.0f} us, device drain {1e6*(t2-t1):.0f} us, status {1e6*(t3-t2):.0f} us, total {1e6*(t3-t0):.0f} us = {1e6*(t3-t0)/steps:.2f} us/step")
# python-only part: time the pieces
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    st2, losses = svi.run_steps(st, gb, bkey, 5, 20, check_status=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
