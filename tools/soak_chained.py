"""Soak test of the chained / pipelined step launch (developer tool): long identical runs must be bitwise identical and no
bounded wait may hit its bound.  usage: python tools/soak_chained.py [steps] [d] [intercept 0/1]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3p_amd._lib as L
import d3p_amd.random as rng
from d3p_amd.minibatch import subsample_batchify_data, poisson_batchify_data
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
icpt = len(sys.argv) > 3 and sys.argv[3] == "1"
N, B = 300000, 4096
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g).cuda(); y = (torch.rand(N, generator=g) < 0.5).float().cuda()
model = LogisticRegression(d, intercept=icpt)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, num_obs_total=N)
D = d + int(icpt)
st = DPSVIState(svi.optim.init(torch.cat([torch.zeros(D), torch.full((D,), -2.0)]).cuda()), rng.PRNGKey(3), float(N))
for name, (init, gb) in (("feistel", subsample_batchify_data((X, y), B)), ("poisson", poisson_batchify_data((X, y), B / N, 0.99))):
    outs = []
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s2, losses = svi.run_steps(st, gb, rng.PRNGKey(4), 0, steps)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        outs.append((s2.optim_state[1].clone(), losses.clone()))
        print(f"{name} run {rep}: {steps} steps in {dt:.2f} s = {dt / steps * 1e6:.2f} us/step, last loss {float(losses[-1]):.3f}", flush=True)
    # third run under UNEVEN load: matrix products of changing size on a second stream compete for the CUs while the chained
    # launches run; a hand-off that depended on timing would show up as a different result
    from d3p_amd.dist import concurrent_streams
    side = concurrent_streams(1)[0]
    mats = [torch.randn(n, n, device="cuda") for n in (512, 1024, 3072)]
    done = torch.cuda.Event()
    s3, losses3 = svi.run_steps(st, gb, rng.PRNGKey(4), 0, steps, check_status=False)   # (asynchronous: the products below run beside it)
    done.record()
    k = 0
    with torch.cuda.stream(side):
        while not done.query():
            a = mats[k % 3]
            (a @ a).sum()
            k += 1
    torch.cuda.synchronize()
    print(f"{name}: disturbed run with {k} concurrent matrix products; run status (aborted, nonfinite) = {svi.last_run_status()}", flush=True)
    outs.append((s3.optim_state[1].clone(), losses3.clone()))
    same = (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and
            torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1]))
    finite = bool(torch.isfinite(outs[0][0]).all()) and bool(torch.isfinite(outs[0][1]).all())
    print(f"{name}: bitwise identical runs: {same}; finite: {finite}", flush=True)
    assert same and finite
print("soak ok")
