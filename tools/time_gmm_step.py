"""Times the stage-wise DP-VI step of the Gaussian-mixture model at BASELINE config 3's shape (K=16, d=64, B=8192)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3p_amd.random as rng
from d3p_amd.models import Adam, GaussianMixtureGuide, GaussianMixtureModel, Trace_ELBO
from d3p_amd.svi import DPSVI

K, d, B, N = 16, 64, 8192, 10**6
g = torch.Generator().manual_seed(0)
X = torch.randn(B, d, generator=g).cuda() * 3
model = GaussianMixtureModel()
svi = DPSVI(model, GaussianMixtureGuide(model), Adam(1e-3), Trace_ELBO(), 20.0, 1.0, k=K, d=d, num_obs_total=N)
st = svi.init(rng.PRNGKey(0), X)
for _ in range(5):
    st, l = svi.update(st, X)
torch.cuda.synchronize()
t0 = time.time()
n = 50
for _ in range(n):
    st, l = svi.update(st, X)
torch.cuda.synchronize()
print("staged update: %.1f us/step, loss %.4g" % ((time.time() - t0) / n * 1e6, float(l)))
from d3p_amd.minibatch import subsample_batchify_data
Xbig = torch.randn(10**6, d, generator=g).cuda() * 3
init, get_batch = subsample_batchify_data((Xbig,), B)
nb, bstate = init(rng.PRNGKey(5))
st2 = svi.init(rng.PRNGKey(0), Xbig[:B])
st2, losses = svi.run_steps(st2, get_batch, bstate, 0, 20)
torch.cuda.synchronize()
t0 = time.time()
st2, losses = svi.run_steps(st2, get_batch, bstate, 20, 200)
torch.cuda.synchronize()
dt = (time.time() - t0) / 200
print("run_steps (device loop, N=1e6 rows, Feistel batches): %.1f us/step = %.0f steps/s, %.3g per-example grads/s" % (dt * 1e6, 1 / dt, B / dt))
key = rng.PRNGKey(1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
svi._compute_per_example_gradients(st, key, X)
e0.record()
for _ in range(20):
    svi._compute_per_example_gradients(st, key, X)
e1.record(); torch.cuda.synchronize()
print("per-example gradient stage: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
