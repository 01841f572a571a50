"""Times DPSVI.update for the VAE at BASELINE config 5's shape (784 -> 400 -> 50, B = 4096)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3p_amd.random as rng
from d3p_amd.models import Adam, Trace_ELBO, VAEGuide, VAEModel
from d3p_amd.svi import DPSVI

N, B = 60000, 4096
X = (torch.rand(B, 28, 28, generator=torch.Generator().manual_seed(0)) < 0.3).float().cuda()
model = VAEModel(scale=1.0 / N)
svi = DPSVI(model, VAEGuide(model), Adam(1e-3), Trace_ELBO(), 10.0, 1.0, num_obs_total=N, z_dim=50, hidden_dim=400)
st = svi.init(rng.PRNGKey(0), X)
for _ in range(5):
    st, l = svi.update(st, X)
torch.cuda.synchronize()
t0 = time.time()
n = 50
for _ in range(n):
    st, l = svi.update(st, X)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
flops = 2 * B * (784 * 400 * 3 + 400 * 784 * 3 + 400 * 50 * 6 + 50 * 400 * 3)
print("VAE update: %.1f us/step = %.0f steps/s, %.3g per-example grads/s, %.1f TFLOP/s (GEMM flops only), loss %.4g"
      % (dt * 1e6, 1 / dt, B / dt, flops / dt / 1e12, float(l)))
