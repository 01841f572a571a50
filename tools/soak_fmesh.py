"""Soak of the full-mesh float all-reduce with virtual ranks on one GPU: `python tools/soak_fmesh.py [world=3] [epochs=3000] [n=688886]`.
Every epoch all ranks reduce fresh vectors (a function of epoch and rank); every 50th epoch the result is compared, on every rank,
bit for bit with the rank-order sum.  No bounded wait may run out."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from d3p_amd import dist as ddist  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 688_886
comms = ddist.FMeshComm.local_group(world, n)
streams = ddist.concurrent_streams(world)
base = [torch.randn(n, generator=torch.Generator().manual_seed(11 + r)).cuda() for r in range(world)]
work = [torch.empty(n, device="cuda") for _ in range(world)]
bad = 0
t0 = time.perf_counter()
for e in range(epochs):
    scale = 1.0 + 0.001 * (e % 997)
    for r in range(world):
        with torch.cuda.stream(streams[r]):
            torch.mul(base[r], scale + 0.01 * r, out=work[r])
            comms[r].allreduce(work[r])
    if e % 50 == 49 or e == epochs - 1:
        torch.cuda.synchronize()
        want = base[0] * (scale + 0.0)
        for r in range(1, world):
            want = want + base[r] * (scale + 0.01 * r)
        for r in range(world):
            if comms[r].stopped() or not torch.equal(work[r], want):
                bad += 1
                print(f"epoch {e}, rank {r}: {'stopped' if comms[r].stopped() else 'sum differs'}", flush=True)
        if bad:
            break
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for c in comms:
    c.close()
print(f"soak_fmesh: world {world}, {epochs} epochs of {n} floats in {dt:.1f} s ({dt / epochs * 1e6:.0f} us per epoch incl. the vector's refresh): "
      + ("FAILED" if bad else "every checked epoch bit for bit the rank-order sum, no wait ran out"), flush=True)
sys.exit(1 if bad else 0)
