#!/usr/bin/env python3
"""Two-PROCESS check of the full-mesh float all-reduce (d3p_fmesh_*) and of the data-parallel VAE loop on it: hipIpc-mapped
inboxes, system-scope tagged words across process boundaries.  The parent never touches a GPU; it starts two fresh rank
processes (gloo carries the IPC handles).  One process per GPU when there are two, else both ranks on cuda:0 (the ranks' launches
must then be co-resident: 256 workgroups of 256 threads each).

    python tools/fmesh_two_rank_check.py

Exit code 0 = (a) the bare collective gave, on both ranks and over five epochs, bit for bit the rank-order sum r0 + r1 of random
vectors of 10 007 and 688 886 floats; (b) three data-parallel VAE steps (784-400-50, batch 128 sharded by position, native loop
d3p_dpvi_vae_run_dist with the mesh as its collective) ended with bitwise identical replicas that match the single-process
update-by-update run to fp32 rounding."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    from d3p_amd.models import Adam, Trace_ELBO, VAEGuide, VAEModel
    from d3p_amd.svi import DPSVI
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank if torch.cuda.device_count() >= world else 0)
    ok = True
    stopped = False
    for n in (10_007, 688_886):
        comm = ddist.FMeshComm(n)
        if torch.cuda.device_count() < world:
            comm.set_grid(48)      # (the ranks share one GPU: room for each other's kernels)
        try:
            vecs = [torch.randn(n, generator=torch.Generator().manual_seed(7 * n + r)) for r in range(world)]
            for e in range(5):
                scaled = [v * float(e + 1) for v in vecs]
                want = scaled[0].clone()
                for v in scaled[1:]:
                    want = want + v           # rank order, fp32
                work = scaled[rank].cuda()
                dist.barrier()
                comm.allreduce(work)
                torch.cuda.synchronize()
                good = bool(torch.equal(work.cpu(), want))
                if not good:
                    print(f"rank {rank}: n = {n}, epoch {e}: the sum differs from the rank-order sum", file=sys.stderr, flush=True)
                ok &= good
            stopped |= comm.stopped()
        finally:
            comm.close(collective=sys.exc_info()[0] is None)   # (after an exception on this rank alone the peers are not in their close)
    # (b) the VAE loop
    N, B, D, H, Z, steps = 60000, 128, 784, 400, 50, 3
    X = torch.tensor((np.random.default_rng(31).random((B, D)) < 0.4).astype(np.float32)).cuda()
    model = VAEModel(z_dim=Z, hidden_dim=H, scale=1.0 / N)

    def make():
        return DPSVI(model, VAEGuide(model), Adam(1e-2), Trace_ELBO(), 3.0, 0.8, num_obs_total=N)
    st0 = make().init(rng.PRNGKey(85), X)
    P = int(st0.optim_state[1].numel())
    comm = ddist.FMeshComm(P + 2)
    if torch.cuda.device_count() < world:
        comm.set_grid(48)
    try:
        pos0, b_local = ddist.shard_batch(B, rank, world)
        dist.barrier()
        st, losses = ddist.vae_run_steps(ddist.VaeHipEngine(make()), st0, X[pos0:pos0 + b_local], B, pos0, steps, comm=comm)
        torch.cuda.synchronize()
        stopped |= comm.stopped()
        res = [None] * world
        dist.all_gather_object(res, (st.optim_state[1].cpu().numpy(), losses.cpu().numpy(), st.rng_key.cpu().numpy(), int(st.optim_state[0])))
        for p, l, k, c in res[1:]:
            ok &= np.array_equal(p, res[0][0]) and np.array_equal(l, res[0][1]) and np.array_equal(k, res[0][2]) and c == res[0][3] == steps
        if rank == 0:
            svi, ref, ref_l = make(), st0, []
            for _ in range(steps):
                ref, l = svi.update(ref, X)
                ref_l.append(float(l))
            ok &= np.array_equal(ref.rng_key.cpu().numpy(), res[0][2])
            ok &= np.allclose(np.asarray(ref_l, np.float32), res[0][1], rtol=2e-5)
            mr = ref.optim_state[2].cpu().numpy()
            ok &= np.allclose(st.optim_state[2].cpu().numpy(), mr, rtol=2e-4, atol=2e-5 * np.abs(mr).max())
        flags = [None] * world
        dist.all_gather_object(flags, (ok, stopped))
        ok = all(f[0] for f in flags)
        stopped = any(f[1] for f in flags)
        if stopped:
            print(f"rank {rank}: stopped -- a bounded wait of the full-mesh all-reduce ran out", file=sys.stderr, flush=True)
        if rank == 0:
            print(json.dumps({"fmesh_two_rank_check": "ok" if (ok and not stopped) else ("STOPPED" if stopped else "MISMATCH"),
                              "devices": torch.cuda.device_count(), "final_loss": float(res[0][1][-1])}), flush=True)
        dist.barrier()
    finally:
        comm.close(collective=sys.exc_info()[0] is None)
        dist.destroy_process_group()
    return 0 if (ok and not stopped) else 1


def main():
    if "RANK" in os.environ:
        sys.exit(rank_main())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        try:
            rc = rc or p.wait(timeout=500)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = rc or 124
    sys.exit(rc)


if __name__ == "__main__":
    main()
