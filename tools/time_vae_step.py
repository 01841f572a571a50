"""Times DPSVI.update for the VAE at BASELINE config 5's shape, B = 4096: 784 -> 400 -> 50 (the reference's network), or with
an argument `200` the literal 784 -> [400, 200] -> 50 of BASELINE.json.
    python tools/time_vae_step.py [hidden2]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3p_amd.random as rng
from d3p_amd.models import Adam, Trace_ELBO, VAEGuide, VAEModel
from d3p_amd.svi import DPSVI

H2 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N, B, D, H, Z = 60000, 4096, 784, 400, 50
X = (torch.rand(B, 28, 28, generator=torch.Generator().manual_seed(0)) < 0.3).float().cuda()
model = VAEModel(scale=1.0 / N)
svi = DPSVI(model, VAEGuide(model), Adam(1e-3), Trace_ELBO(), 10.0, 1.0, num_obs_total=N, z_dim=Z, hidden_dim=(H, H2) if H2 else H)
st = svi.init(rng.PRNGKey(0), X)
for _ in range(60):   # (18 ms: the clocks need that long to settle after an idle phase -- with 5 updates the figure reads 3 % high)
    st, l = svi.update(st, X)
torch.cuda.synchronize()
t0 = time.time()
n = 50
for _ in range(n):
    st, l = svi.update(st, X)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
# every dense layer costs three products of 2 B in out flops (forward, backward-data, weight gradient); the first encoder layer
# has no backward-data product
hs = [H] + ([H2] if H2 else [])
dec, enc = [Z] + hs[::-1] + [D], [D] + hs
layers = list(zip(dec[:-1], dec[1:])) + list(zip(enc[:-1], enc[1:])) + [(hs[-1], 2 * Z)]
flops = 2 * B * (3 * sum(i * o for i, o in layers) - D * H)
print("VAE update (hidden %s, P = %d): %.1f us/step = %.0f steps/s, %.3g per-example grads/s, %.1f TFLOP/s (GEMM flops only), loss %.4g"
      % (hs, st.optim_state[1].numel(), dt * 1e6, 1 / dt, B / dt, flops / dt / 1e12, float(l)))
