#!/usr/bin/env python3
"""Two-PROCESS check of the one-shot exchange (d3p_xchg_*): hipIpc-mapped inboxes, system-scope rows and flags across
process boundaries.  The parent never touches a GPU; it starts two fresh rank processes (gloo carries the IPC handles) and
checks their results against each other; rank 0 also computes the single-rank reference.

    python tools/xchg_two_rank_check.py            # one process per GPU when there are two, else both ranks on cuda:0
    D3P_XCHG_CHECK_WORLD=4 python tools/xchg_two_rank_check.py      # four rank processes

Exit code 0 = the bare collective summed exactly over three epochs and the row-sharded 24-step run ended with bitwise
identical replicas that match the single-rank run to fp32 rounding."""
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
    import d3p_amd._lib as L
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.svi import DPSVI, DPSVIState
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank if torch.cuda.device_count() >= world else 0)
    n, d, B, steps = 6000, 512, 512, int(os.environ.get("D3P_XCHG_CHECK_STEPS", "24"))   # (both ranks may share ONE GPU here: 24 x 19 workgroups per launch leave room for the
    #  other rank's launch beside the first one's; at 40 steps the first launch fills the GPU, see tests/test_dist.py)
    g = torch.Generator().manual_seed(5)
    X = torch.randn(n, d, generator=g)
    y = (torch.rand(n, generator=g) < 0.5).float()
    model = LogisticRegression(d)
    svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-2), Trace_ELBO(), 1.0, 0.7, N=n)
    st0 = DPSVIState(svi.optim.init(torch.cat([torch.zeros(d), torch.full((d,), -2.0)]).cuda()), rng.PRNGKey(3), float(n))
    comm = ddist.XchgComm(2 * d + 4)
    ok = True
    try:
        acc = torch.randint(-2**40, 2**40, (4, 2 * d + 4), generator=torch.Generator().manual_seed(100 + rank), dtype=torch.int64).cuda()
        mine = acc.sum(dim=0).cpu()
        box = [None] * world
        dist.all_gather_object(box, mine)
        for _ in range(3):
            work = acc.clone()
            comm.allreduce(work, 4)
            torch.cuda.synchronize()
            ok &= bool(torch.equal(work[0].cpu(), sum(box))) and not bool(work[1:].any())
        lo, hi = ddist.shard_rows(n, rank, world)
        engine = ddist.FusedHipEngine(svi, X[lo:hi].cuda(), y[lo:hi].cuda(), n, lo, hi, L.D3P_BATCH_FEISTEL, B)
        # D3P_XCHG_CHECK_REPEAT=N: the run N times on the same exchange (its epochs keep counting: both slot parities, tags of
        # earlier runs in the slots), each continuing from the state the last one left; the replicas are compared every time
        repeats = int(os.environ.get("D3P_XCHG_CHECK_REPEAT", "1"))
        st, first = st0, 2
        for rep in range(repeats):
            dist.barrier()   # (the ranks share one GPU here: start together, or the first launch fills it alone)
            st, losses = ddist.run_steps_native(engine, st, rng.PRNGKey(4), first, steps, comm=comm)
            torch.cuda.synchronize()
            code, _ = ddist.native_run_status(engine)
            if code:
                print(f"rank {rank}, run {rep}: stopped -- {L.describe_abort(code)}", file=sys.stderr, flush=True)
            first += steps
            mine = torch.cat([st.optim_state[1], losses]).cpu().numpy().tobytes() if rep + 1 < repeats else b""
            both = [None] * world
            dist.all_gather_object(both, (code, mine))           # (every rank takes the same decision)
            if any(c for c, _ in both):
                ok = False
                break
            if any(b != both[0][1] for _, b in both[1:]):
                print(f"rank {rank}, run {rep}: replicas differ", file=sys.stderr, flush=True)
                ok = False
                break
        res = [None] * world
        dist.all_gather_object(res, (st.optim_state[1].cpu().numpy(), losses.cpu().numpy(), st.rng_key.cpu().numpy()))
        for p, l, k in res[1:]:
            ok &= np.array_equal(p, res[0][0]) and np.array_equal(l, res[0][1]) and np.array_equal(k, res[0][2])
        if rank == 0:
            single = ddist.FusedHipEngine(svi, X.cuda(), y.cuda(), n, 0, n, L.D3P_BATCH_FEISTEL, B)
            ref, ref_l = ddist.run_steps_native(single, st0, rng.PRNGKey(4), 2, steps * repeats, comm=None)
            ref_l = ref_l[-steps:]
            ok &= np.array_equal(ref.rng_key.cpu().numpy(), res[0][2])
            if repeats <= 4:   # (over many runs the two summation orders drift apart by more than rounding; the replicas may not)
                tol = 2e-5 * repeats
                ok &= np.allclose(ref_l.cpu().numpy(), res[0][1], rtol=tol, atol=0)
                ok &= np.allclose(ref.optim_state[1].cpu().numpy(), res[0][0], rtol=tol, atol=2e-6 * repeats)
            print(json.dumps({"xchg_two_rank_check": "ok" if ok else "MISMATCH", "devices": torch.cuda.device_count(),
                              "final_loss": float(res[0][1][-1])}), flush=True)
        dist.barrier()
    finally:
        comm.close(collective=sys.exc_info()[0] is None)
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    if "RANK" in os.environ:
        sys.exit(rank_main())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    world = int(os.environ.get("D3P_XCHG_CHECK_WORLD", "2"))   # (more than two processes: up to the 6 a GPU box allows on its card)
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
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
