# Timeline of one short run_steps call: prints the wall time of N repetitions and leaves a rocprofv3 trace when run under it.
import sys, time, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import d3p_amd.random as rng
from d3p_amd.svi import DPSVI, DPSVIState
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.minibatch import subsample_batchify_data
N, d, B, steps = 1_000_000, 512, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 20
X = torch.randn(N, d, device="cuda"); y = (torch.rand(N, device="cuda") < 0.5).float()
model = LogisticRegression(d, prior_scale=1.0, intercept=False)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, rng_suite=rng, clip_unscaled_observations=True, N=N)
params = torch.tensor(np.concatenate([np.zeros(d, np.float32), np.full(d, -2.0, np.float32)]), device="cuda")
st = DPSVIState(svi.optim.init(params), rng.PRNGKey(0), float(N))
_, gb = subsample_batchify_data((X, y), B)
bk = rng.PRNGKey(1)
for _ in range(5): st2, _l = svi.run_steps(st, gb, bk, 0, steps)
torch.cuda.synchronize()
ts = []
for r in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st2, _l = svi.run_steps(st, gb, bk, 0, steps)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print(f"run_steps({steps}), every call from the same state: wall us median %.1f min %.1f" % (np.median(ts), min(ts)))
ts = []
cur, fb = st, 0
cur, _l = svi.run_steps(cur, gb, bk, fb, 5); fb += 5     # (what bench.py's warm-up is: a 5-step run in front)
for r in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cur, _l = svi.run_steps(cur, gb, bk, fb, steps); fb += steps
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print(f"run_steps({steps}), each call continuing the previous one: first after a 5-step run %.1f, wall us" % ts[0])
print(f"   ... wall us: median %.1f min %.1f" % (np.median(ts), min(ts)))

# ---- where the host time of one call goes: the fast path of DPSVI.run_steps, re-enacted with time marks
import ctypes as C
import d3p_amd._lib as L
from d3p_amd._lib import check, ptr, stream_ptr
from d3p_amd.svi import BatchSource
lib = L.load()
info = gb.source
marks = {}
def run_marked():
    T = [("t0", time.perf_counter())]
    mk = lambda n: T.append((n, time.perf_counter()))
    Xd = info.dataset[0]; yd = info.dataset[1]
    ok = Xd.is_contiguous() and Xd.dtype == torch.float32 and yd.is_contiguous(); mk("checks")
    Nn, dd = Xd.shape; dev = Xd.device
    model_s = svi._model_struct(dd, {}, st.observation_scale); mk("model_struct")
    hyper = svi._hyper(); mk("hyper")
    bkey = bk.contiguous(); mk("bkey.contiguous")
    losses = torch.empty(max(steps, 1), dtype=torch.float32, device=dev); mk("empty losses")
    step0, params0, m0, v0 = st.optim_state
    key0 = st.rng_key.reshape(16); mk("key reshape")
    n = params0.numel()
    ok = (params0.dtype == m0.dtype == v0.dtype == torch.float32 and m0.numel() == n and v0.numel() == n and params0.is_contiguous()
          and m0.is_contiguous() and v0.is_contiguous() and key0.is_contiguous() and key0.dtype == torch.uint32 and step0.dtype == torch.int32); mk("state checks")
    flat = torch.empty(3 * n, dtype=torch.float32, device=dev); mk("empty flat")
    step, params, m, v = torch.empty_like(step0), flat[:n].view_as(params0), flat[n:2 * n].view_as(m0), flat[2 * n:].view_as(v0); mk("empty_like + 3 views")
    keybuf = torch.empty((2, 16), dtype=torch.uint32, device=dev); mk("empty keybuf")
    src = BatchSource(info.kind, info.batch_size, float(info.q), int(info.suppress), bkey.data_ptr(), None, None, Nn, 0, Nn); mk("BatchSource")
    s1 = svi._state_struct(keybuf, 0, (step, params, m, v)); s0 = svi._state_struct(key0, 0, (step0, params0, m0, v0)); mk("state structs")
    ws = svi._workspace(lib.d3p_dpvi_logreg_workspace(C.byref(model_s), C.byref(src)), dev); mk("workspace")
    sp = stream_ptr(); mk("stream_ptr")
    check(lib.d3p_dpvi_logreg_run_from(sp, C.byref(model_s), C.byref(hyper), C.byref(s1), C.byref(s0), C.byref(src),
                                       0, ptr(Xd), ptr(yd), int(steps), ptr(losses), ptr(ws), ws.numel())); mk("run_from (enqueue)")
    aborted, nonfinite = C.c_int32(0), C.c_int32(0)
    check(lib.d3p_dpvi_logreg_run_status(stream_ptr(), C.byref(model_s), C.byref(src), ptr(ws), ws.numel(), C.byref(aborted), C.byref(nonfinite))); mk("run_status (sync)")
    new_key = keybuf[steps & 1].reshape(4, 4); out = DPSVIState((step, params, m, v), new_key, st.observation_scale), losses[:steps]; mk("result views")
    for (a, ta), (b, tb) in zip(T[:-1], T[1:]): marks.setdefault(b, []).append((tb - ta) * 1e6)
    return out
for _ in range(30): run_marked()
print("host time per segment of one run_steps call (median us):")
for k_, v_ in marks.items(): print("  %-24s %6.1f" % (k_, np.median(v_)))
