"""Developer tool: time the fused main kernel for a few geometries (env D3P_MAIN_W / D3P_MAIN_EPW)."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3p_amd._lib as L
import d3p_amd.random as rng
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI
lib = L.load(); L.require_device()
dev = torch.device("cuda:0")
d, B, N = 512, 4096, 1_000_000
X = torch.empty((N, d), device=dev); y = torch.empty(N, device=dev)
L.check(lib.d3p_synth_logreg(L.stream_ptr(), 123, 0, N, d, L.ptr(X), L.ptr(y)))
model = LogisticRegression(d); svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, N=N)
params = torch.cat([torch.zeros(d, device=dev), torch.full((d,), -2.25, device=dev)])
m = torch.zeros_like(params); v = torch.zeros_like(params); step = torch.zeros((), dtype=torch.int32, device=dev)
keybuf = torch.zeros((2, 16), dtype=torch.int32, device=dev); keybuf[0].copy_(rng.PRNGKey(0).reshape(16).view(torch.int32))
bkey = rng.PRNGKey(1); bidx = torch.zeros(1, dtype=torch.int32, device=dev)
for W, EPW, DBG in [(16, 1, int(os.environ.get("D3P_DBG", "0")))]:
    os.environ["D3P_MAIN_W"] = str(W); os.environ["D3P_MAIN_EPW"] = str(EPW); os.environ["D3P_DBG"] = str(DBG)
    st = L.DpsviState(keybuf.data_ptr(), 0, params.data_ptr(), m.data_ptr(), v.data_ptr(), step.data_ptr())
    src = L.BatchSource(L.D3P_BATCH_FEISTEL, B, 0.0, 0, bkey.data_ptr(), bidx.data_ptr(), None, N, 0, N)
    mdl = svi._model_struct(d, {}, float(N)); hyp = svi._hyper()
    ws = torch.empty(lib.d3p_dpvi_logreg_workspace(C.byref(mdl), C.byref(src)), dtype=torch.uint8, device=dev)
    us = C.c_float(); ev = C.c_float()
    for rep in range(2):
        L.check(lib.d3p_dpvi_logreg_time_main_kernel(L.stream_ptr(), C.byref(mdl), C.byref(hyp), C.byref(st), C.byref(src),
                                                     L.ptr(X), L.ptr(y), L.ptr(ws), ws.numel(), 300, C.byref(us), C.byref(ev)))
    print(json.dumps({"W": W, "EPW": EPW, "DBG": DBG, "avg_us": round(us.value, 3), "event_us": round(ev.value, 3), "GBps": round(8433664 / us.value / 1e3, 1)}), flush=True)
