"""Times DPSVI.update() with the logistic-regression example's OWN two-site guide (MeanFieldGuide; examples/logistic_regression.py:67-86)
against AutoDiagonalNormal over the same model (with intercept), on one resident batch (developer tool).
    python tools/time_example_guide.py [B=4096] [d=512]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3p_amd.random as rng
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, MeanFieldGuide, Trace_ELBO
from d3p_amd.svi import DPSVI

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N, steps = 200000, 300
g = torch.Generator().manual_seed(0)
X = torch.randn(B, d, generator=g).cuda(); y = (torch.rand(B, generator=g) < 0.5).float().cuda()
model = LogisticRegression(d, intercept=True)
for name, guide in (("AutoDiagonalNormal (fused step)", AutoDiagonalNormal(model)), ("MeanFieldGuide (the example's own guide)", MeanFieldGuide(model))):
    svi = DPSVI(model, guide, Adam(1e-3), Trace_ELBO(), 1.0, 1.0, num_obs_total=N)
    st = svi.init(rng.PRNGKey(3), X, y)
    for phase in range(2):
        s = st
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(steps):
            s, loss = svi.update(s, X, y)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {dt / steps * 1e6:.1f} us per update (B = {B}, d = {d} + intercept), loss {float(loss):.4g}", flush=True)
svi_staged = svi        # (the last one: MeanFieldGuide) -- the reference's five-stage composition on the same guide
for phase in range(2):
    s = st
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(steps):
        s, loss = svi_staged._update_staged(s, X, y)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"MeanFieldGuide through the five-stage composition: {dt / steps * 1e6:.1f} us per update, loss {float(loss):.4g}", flush=True)
