"""Rank-local cost of the data-parallel VAE step on ONE GPU (the rank's share of a `world`-rank job; no peers): the Python-driven
loop (local_sums -> torch all_reduce skipped -> apply), the native loop without a collective, and the native loop with a
one-rank RCCL communicator in one bucket (in the stream) and in two buckets (second stream + events).  What it shows is the
LAUNCH / EVENT side of the bucketed overlap; what the overlap gains needs real links (bench.py --gpus N measures both)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import d3p_amd.random as rng  # noqa: E402
from d3p_amd.dist import FMeshComm, NativeComm, VaeHipEngine, vae_run_steps  # noqa: E402
from d3p_amd.models import Adam, Trace_ELBO, VAEGuide, VAEModel  # noqa: E402
from d3p_amd.svi import DPSVI  # noqa: E402


def main():
    N, D, H, Z = 60000, 784, 400, 50
    comm = NativeComm()
    out = []
    for H2 in (0, 200):
        for B_local, world in ((4096, 8), (512, 8)):
            model = VAEModel(scale=1.0 / N)
            svi = DPSVI(model, VAEGuide(model), Adam(1e-3), Trace_ELBO(), 10.0, 1.0, num_obs_total=N, z_dim=Z, hidden_dim=(H, H2) if H2 else H)
            X = (torch.rand(B_local, 28, 28, generator=torch.Generator().manual_seed(1)) < 0.3).float().cuda()
            st = svi.init(rng.PRNGKey(0), X)
            Bg = B_local * world
            rec = {"hidden": [H] + ([H2] if H2 else []), "B_local": B_local, "B_total": Bg}
            mesh = FMeshComm(int(st.optim_state[1].numel()) + 2)      # (one rank: no peers -- the launch-side cost of the full-mesh forms)
            for name, kw in (("python_loop", {}), ("native_no_collective", {"comm": "local"}), ("native_rccl_1_bucket", {"comm": comm, "buckets": 1}),
                             ("native_rccl_2_buckets", {"comm": comm, "buckets": 2}), ("native_full_mesh_fused", {"comm": mesh}),
                             ("native_full_mesh_3_launches", {"comm": mesh, "buckets": 1})):
                eng = VaeHipEngine(svi)
                if name == "native_no_collective":   # (the call asks for a communicator when the rank holds a share: time it as a whole batch)
                    run = lambda k, c=False: vae_run_steps(eng, st, X, B_local, 0, k, collect_losses=c, **kw)  # noqa: E731
                else:
                    # the emulated share: positions 0 .. B_local - 1 of a global batch of Bg (python loop: world 1, no all_reduce)
                    run = lambda k, c=False: vae_run_steps(eng, st, X, Bg, 0, k, collect_losses=c, **kw)  # noqa: E731
                run(60)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    run(100)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / 100)
                rec[name + "_us_per_step"] = round(best * 1e6, 2)
            mesh.close()
            out.append(rec)
            print(json.dumps(rec), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
