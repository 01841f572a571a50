#!/usr/bin/env python3
"""Soak of the hipIpc communicators' teardown (round 6): WORLD processes (default 4) on the GPUs there are (all on cuda:0 on a one-GPU
box) create a full-mesh communicator AND an exchange communicator of a DIFFERENT size every round, run one checked collective on
each and close them -- ROUNDS times (default 40).  Before round 6 a rank that was first out of close() freed its inbox while a peer
still had it mapped, and its next hipIpcGetMemHandle failed ("invalid argument") or a peer wrote through a stale mapping: this loop
is that scenario, over and over.  Exit code 0 = every sum right, nothing stopped, no error.

    python tools/soak_teardown.py [world=4] [rounds=40]"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main():
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from d3p_amd import dist as ddist
    rank, world, rounds = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["D3P_SOAK_ROUNDS"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shared = torch.cuda.device_count() < world
    torch.cuda.set_device(0 if shared else rank)
    ok = True
    for it in range(rounds):
        n = 1000 + 7919 * it % 300_000                 # another vector length every round: another inbox size, another allocation
        try:
            mesh = ddist.FMeshComm(n)
        except Exception as e:  # noqa: BLE001
            print(f"rank {rank}, round {it}: FMeshComm({n}) failed: {e}", file=sys.stderr, flush=True)
            raise
        if shared:
            mesh.set_grid(max(8, 128 // world))
        vecs = [torch.randn(n, generator=torch.Generator().manual_seed(31 * it + r)) for r in range(world)]
        want = vecs[0].clone()
        for v in vecs[1:]:
            want = want + v
        work = vecs[rank].cuda()
        dist.barrier()
        mesh.allreduce(work)
        torch.cuda.synchronize()
        good = bool(torch.equal(work.cpu(), want)) and not mesh.stopped()
        words = 64 + 16 * (it % 40)
        try:
            xc = ddist.XchgComm(words) if not os.environ.get("D3P_SOAK_MESH_ONLY") else None
        except Exception as e:  # noqa: BLE001
            print(f"rank {rank}, round {it}: XchgComm({words}) failed: {e}", file=sys.stderr, flush=True)
            raise
        if xc is None:
            ok &= good
            mesh.close()
            continue
        acc = torch.randint(-2**40, 2**40, (4, words), generator=torch.Generator().manual_seed(1000 * it + rank), dtype=torch.int64)
        mine = acc.sum(dim=0)
        box = [None] * world
        dist.all_gather_object(box, mine)
        dev = acc.cuda()
        dist.barrier()
        xc.allreduce(dev, 4)
        torch.cuda.synchronize()
        good &= bool(torch.equal(dev[0].cpu(), sum(box))) and not bool(dev[1:].any())
        if not good:
            print(f"rank {rank}, round {it}: wrong sum or stopped collective", file=sys.stderr, flush=True)
        ok &= good
        # close with a skew: the slow rank still has its peers' inboxes mapped while the fast ones are already closing.  (The closes are
        # collectives: every rank closes its communicators in the SAME order, as with any sequence of collectives.)
        if rank == it % world:
            import time
            time.sleep(0.01)
        if it % 2:
            mesh.close(); xc.close()
        else:
            xc.close(); mesh.close()
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if rank == 0:
        print(f"soak_teardown: {world} processes x {rounds} rounds (a mesh and an exchange created, used and closed per round): "
              + ("ok" if all(flags) else "FAILED"), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if all(flags) else 1


def main():
    if "RANK" in os.environ:
        sys.exit(rank_main())
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    assert 2 <= world <= 6
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), D3P_SOAK_ROUNDS=str(rounds))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        try:
            rc = rc or p.wait(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = rc or 124
    for p in procs:
        if p.poll() is None:
            p.kill()
    sys.exit(rc)


if __name__ == "__main__":
    main()
