import sys, os, time
sys.path.insert(0, ".")
import torch
import d3p_amd._lib as L, ctypes as C
import d3p_amd.random as rng
from d3p_amd.minibatch import poisson_batchify_data
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState
N, B, d, steps = 1_000_000, 4096, 512, 4096
lib = L.load()
X = torch.empty((N, d), device="cuda"); y = torch.empty(N, device="cuda")
L.check(lib.d3p_synth_logreg(L.stream_ptr(), 123, 0, N, d, L.ptr(X), L.ptr(y)))
model = LogisticRegression(d)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, N=N)
st = DPSVIState(svi.optim.init(torch.cat([torch.zeros(d), torch.full((d,), -2.25)]).cuda()), rng.PRNGKey(0), float(N))
_, gb = poisson_batchify_data((X, y), B / N, 0.99)
st, _ = svi.run_steps(st, gb, rng.PRNGKey(1), 0, 512)
torch.cuda.synchronize()
for rep in range(3):
    L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(1))
    t0 = time.perf_counter()
    st, losses = svi.run_steps(st, gb, rng.PRNGKey(1), 512 + rep * steps, steps)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(0))
    us, n, ks = C.c_double(), C.c_uint32(), C.c_uint32()
    L.check(lib.d3p_dpvi_logreg_kernel_timing_read(C.byref(us), C.byref(n), C.byref(ks)))
    print(f"poisson N=1e6: NW={os.environ.get('D3P_CHAIN_NW','default')} kernel {us.value / max(ks.value,1):.3f} us/step, wall {el / steps * 1e6:.2f} us/step", flush=True)
