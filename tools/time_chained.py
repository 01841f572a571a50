"""Developer tool: in-kernel time per DP-VI step of the single-GPU run loop at the headline configuration (d = 512, B = 4096,
N = 10^6 rows, Feistel batches), HIP events around every chained launch.  Environment switches (D3P_DBG, D3P_NO_*) select
the variant; D3P_DBG=32 prints the phase anatomy of the fourth launch to stderr.
usage: time_chained.py [steps=2048] [batch=4096] [dim=512] [intercept=0]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3p_amd._lib as L
import d3p_amd.random as rng
from d3p_amd.minibatch import subsample_batchify_data
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = int(sys.argv[3]) if len(sys.argv) > 3 else 512
icpt = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
N = 1_000_000
lib = L.load(); L.require_device()
dev = torch.device("cuda:0")
X = torch.empty((N, d), device=dev); y = torch.empty(N, device=dev)
L.check(lib.d3p_synth_logreg(L.stream_ptr(), 123, 0, N, d, L.ptr(X), L.ptr(y)))
model = LogisticRegression(d, intercept=icpt)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, N=N)
D = d + int(icpt)
st = DPSVIState(svi.optim.init(torch.cat([torch.zeros(D, device=dev), torch.full((D,), -2.25, device=dev)])), rng.PRNGKey(0), float(N))
_, gb = subsample_batchify_data((X, y), B)
bkey = rng.PRNGKey(1)
st, _ = svi.run_steps(st, gb, bkey, 0, 512)
torch.cuda.synchronize()
out = []
for rep in range(3):
    L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(1))
    t0 = time.perf_counter()
    st, losses = svi.run_steps(st, gb, bkey, 512 + rep * steps, steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(0))
    us, n, ks = C.c_double(), C.c_uint32(), C.c_uint32()
    L.check(lib.d3p_dpvi_logreg_kernel_timing_read(C.byref(us), C.byref(n), C.byref(ks)))
    out.append((us.value / max(ks.value, 1), 1e6 * el / steps))
k = sorted(o[0] for o in out)[1]
print(json.dumps({"tag": os.environ.get("TAG", ""), "B": B, "d": d, "intercept": icpt, "steps": steps,
                  "kernel_us_per_step_median": round(k, 3), "kernel_us_per_step": [round(o[0], 3) for o in out],
                  "wall_us_per_step": [round(o[1], 3) for o in out], "final_loss": float(losses[-1]),
                  "env": {k: v for k, v in os.environ.items() if k.startswith("D3P_")}}), flush=True)
