#!/usr/bin/env python3
"""Many-rank check of the full-mesh all-reduce (d3p_fmesh_*) and of the data-parallel VAE step on it (k_vae_fmesh_step, the fused form,
and the three-launch form) at BASELINE configs[4]'s per-rank batch, with the oracle as the anchor.  Test infrastructure (it loads
oracle/ on rank 0); driven by tests/test_dist.py, never imported by the product.

    D3P_FMESH_CHECK_WORLD=4 D3P_FMESH_CHECK_B_LOCAL=512 python tests/fmesh_ranks_check.py        # 4 PROCESSES over hipIpc
    D3P_FMESH_CHECK_VIRTUAL=8 D3P_FMESH_CHECK_B_LOCAL=512 python tests/fmesh_ranks_check.py      # 8 ranks in ONE process (streams)

Process mode: the parent never touches a GPU; it starts WORLD fresh rank processes (gloo carries the IPC handles), one per GPU
when there are enough GPUs, else all on cuda:0 (the pool's GPU boxes allow at most 6 processes on a card: world <= 4 here).
Virtual mode: WORLD ranks in one process on streams that run beside each other -- the parent sets GPU_MAX_HW_QUEUES so that the
process has that many hardware queues -- inboxes wired directly: the many-rank code paths of an 8-GPU job (chunk = ceil(688 886 / 8),
8 inboxes, 7 peers per poll) that no process-per-rank rehearsal on this pool can reach.

Checks (exit code 0 = all hold):
 (a) the bare collective on 688 886 floats (and 10 007): on every rank, over five epochs, bit for bit the RANK-ORDER fp32 sum
     ((r0 + r1) + r2) + ... of the ranks' vectors;
 (b) three data-parallel VAE steps (784-400-50, the batch sharded by position, native loop d3p_dpvi_vae_run_dist) in the fused
     form and the three-launch form: replicas bitwise identical over the ranks AND over the two forms; losses / keys / Adam's m
     against the single-process update-by-update run of the WHOLE batch to fp32 rounding;
 (c) the oracle anchor: ONE data-parallel step with a mask that keeps 12 examples of every rank's shard (the kernels run at the
     full per-rank shape); Adam's first moment after it against  O.vae_step_sums  on exactly those examples (eps keyed by GLOBAL
     position) -> O.perturb (one key per leaf, noise once) -> O.adam."""
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_OBS, D, H, Z, STEPS = 60000, 784, 400, 50, 3
CLIP, SIGMA, LR = 3.0, 0.8, 1e-2


def _problem(B_total):
    import numpy as np
    import torch
    from d3p_amd.models import Adam, Trace_ELBO, VAEGuide, VAEModel
    from d3p_amd.svi import DPSVI
    model = VAEModel(z_dim=Z, hidden_dim=H, scale=1.0 / N_OBS)

    def make():
        return DPSVI(model, VAEGuide(model), Adam(LR), Trace_ELBO(), CLIP, SIGMA, num_obs_total=N_OBS)
    X = torch.tensor((np.random.default_rng(31).random((B_total, D)) < 0.4).astype(np.float32))
    return make, X


def _initial_state(make, P_from, key_seed):
    """Parameters 0.03 N(0, 1) (the size tests/test_gpu_configs.py pins the single-device step to the oracle at), observation scale 1."""
    import numpy as np
    import torch
    import d3p_amd.random as rng
    from d3p_amd.svi import DPSVIState
    params = (0.03 * np.random.default_rng(17).normal(size=P_from)).astype(np.float32)
    svi = make()
    return DPSVIState(svi.optim.init(torch.tensor(params).cuda()), rng.PRNGKey(key_seed), 1.0), params


def _oracle_first_moment(params, X_np, mask_np, key_seed):
    """Adam's m after ONE masked update, from the oracle: eps by global position (svi.py:289-290), explicit per-example gradients
    of the selected examples, clip, sum, perturbation once (svi.py:350-377, one key per leaf :487-491), Adam (svi.py:379-393)."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    B = X_np.shape[0]
    spec = O.vae_spec(D, H, Z, scale=1.0, obs_scale=1.0)
    P = O.vae_num_params(spec)
    ks = O.split(O.PRNGKey(key_seed), 3)
    eps = O.px_eps(O.convert_to_jax_rng_key(ks[1]), B, Z)
    sel = np.flatnonzero(mask_np)
    sums, _, _ = O.vae_step_sums(spec, params, X_np[sel], eps[sel], CLIP, None)
    n = float(len(sel))
    f = B / n
    g = O.perturb(ks[2], sums[:P] / B, O.vae_leaf_sizes(D, H, Z, 0), SIGMA, CLIP, n, 1.0, f)
    _, m, _ = O.adam(params, np.zeros(P), np.zeros(P), g, 0, lr=LR)
    return m, sums[P] / B * f, ks[0]


def _mask(B_total, world):
    import numpy as np
    from d3p_amd.dist import shard_batch
    mask = np.zeros(B_total, bool)
    r = np.random.default_rng(99)
    for rk in range(world):
        pos0, b = shard_batch(B_total, rk, world)
        mask[pos0 + r.choice(b, size=min(12, b), replace=False)] = True
    return mask


def _rank_order_sum(vecs):
    want = vecs[0].clone()
    for v in vecs[1:]:
        want = want + v
    return want


def _compare_with_single(ok, tag, st, losses, ref, ref_l):
    import numpy as np
    ok &= np.array_equal(ref.rng_key.cpu().numpy(), st.rng_key.cpu().numpy())
    good = np.allclose(np.asarray(ref_l, np.float32), losses.cpu().numpy(), rtol=2e-5)
    mr = ref.optim_state[2].cpu().numpy()
    good &= np.allclose(st.optim_state[2].cpu().numpy(), mr, rtol=2e-4, atol=2e-5 * np.abs(mr).max())
    if not good:
        print(f"{tag}: the data-parallel run differs from the single-process run", file=sys.stderr, flush=True)
    return ok and good


def _compare_with_oracle(tag, st, loss, params, X_np, mask_np, key_seed):
    import numpy as np
    m_o, loss_o, key_o = _oracle_first_moment(params, X_np, mask_np, key_seed)
    m = st.optim_state[2].cpu().numpy()
    good = np.array_equal(st.rng_key.cpu().numpy().ravel(), np.asarray(key_o).ravel()) and int(st.optim_state[0]) == 1
    good &= abs(float(loss) - loss_o) <= 5e-5 * abs(loss_o)
    good &= np.allclose(m, m_o, rtol=2e-4, atol=2e-5 * np.abs(m_o).max())
    if not good:
        print(f"{tag}: the masked data-parallel step differs from the oracle (loss {float(loss)} vs {loss_o}, "
              f"max |dm| {np.abs(m - m_o).max():.3e} of {np.abs(m_o).max():.3e})", file=sys.stderr, flush=True)
    return bool(good)


def rank_main():
    """One PROCESS per rank."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    from d3p_amd import dist as ddist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    b_local = int(os.environ.get("D3P_FMESH_CHECK_B_LOCAL", "512"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shared = torch.cuda.device_count() < world
    torch.cuda.set_device(0 if shared else rank)
    grid = max(8, 128 // world)      # (ranks that share a GPU: room for each other's kernels)
    ok, stopped = True, False
    for n in (10_007, 688_886):
        comm = ddist.FMeshComm(n)
        if shared:
            comm.set_grid(grid)
        try:
            vecs = [torch.randn(n, generator=torch.Generator().manual_seed(7 * n + r)) for r in range(world)]
            for e in range(5):
                scaled = [v * float(e + 1) for v in vecs]
                want = _rank_order_sum(scaled)
                work = scaled[rank].cuda()
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                comm.allreduce(work)
                torch.cuda.synchronize()
                if os.environ.get("D3P_FMESH_CHECK_VERBOSE"):
                    print(f"rank {rank}: n = {n}, epoch {e}: {1e3 * (time.perf_counter() - t0):.2f} ms, stopped {comm.stopped()}", file=sys.stderr, flush=True)
                good = bool(torch.equal(work.cpu(), want))
                if not good:
                    print(f"rank {rank}: n = {n}, epoch {e}: the sum differs from the rank-order sum", file=sys.stderr, flush=True)
                ok &= good
            stopped |= comm.stopped()
        finally:
            comm.close(collective=sys.exc_info()[0] is None)   # (after an exception on this rank alone the peers are not in their close)
    B = b_local * world
    make, X = _problem(B)
    Xd = X.cuda()
    import d3p_amd._lib as L
    P = int(L.load().d3p_vae_num_params(ctypes.byref(make()._vae_struct(D, {}, 1.0))))
    st0, params = _initial_state(make, P, 85)
    comm = ddist.FMeshComm(P + 2)
    if shared:
        comm.set_grid(grid)
    try:
        pos0, bl = ddist.shard_batch(B, rank, world)
        runs = {}
        for form, buckets in (("fused", 0), ("three launches", 1)):
            dist.barrier()
            st, losses = ddist.vae_run_steps(ddist.VaeHipEngine(make()), st0, Xd[pos0:pos0 + bl], B, pos0, STEPS, comm=comm, buckets=buckets,
                                             check_status=False)
            torch.cuda.synchronize()
            stopped |= comm.stopped()
            runs[form] = (st, losses)
        (sa, la), (sb, lb) = runs["fused"], runs["three launches"]
        same_forms = torch.equal(la, lb) and torch.equal(sa.rng_key, sb.rng_key) and all(torch.equal(a, b) for a, b in zip(sa.optim_state, sb.optim_state))
        if not same_forms:
            print(f"rank {rank}: the fused step and the three-launch step differ", file=sys.stderr, flush=True)
        ok &= bool(same_forms)
        st, losses = runs["fused"]
        # (c) the masked step
        mask = _mask(B, world)
        dist.barrier()
        stm, lm = ddist.vae_run_steps(ddist.VaeHipEngine(make()), st0, Xd[pos0:pos0 + bl], B, pos0, 1, comm=comm, check_status=False,
                                      mask=torch.tensor(mask[pos0:pos0 + bl]).cuda())
        torch.cuda.synchronize()
        stopped |= comm.stopped()
        res = [None] * world
        dist.all_gather_object(res, tuple(t.cpu().numpy().tobytes() for t in
                                          (st.optim_state[1], st.optim_state[2], losses, st.rng_key, st.optim_state[0],
                                           stm.optim_state[1], stm.optim_state[2], lm, stm.rng_key)))
        bitwise = all(r == res[0] for r in res[1:])
        if not bitwise:
            print(f"rank {rank}: replicas differ", file=sys.stderr, flush=True)
        ok &= bitwise and int(st.optim_state[0]) == STEPS
        if rank == 0 and not stopped:
            svi, ref, ref_l = make(), st0, []
            for _ in range(STEPS):
                ref, l = svi.update(ref, Xd)
                ref_l.append(float(l))
            ok = _compare_with_single(ok, "rank 0", st, losses, ref, ref_l)
            ok &= _compare_with_oracle("rank 0", stm, lm[0], params, X.numpy(), mask, 85)
        flags = [None] * world
        dist.all_gather_object(flags, (bool(ok), bool(stopped)))
        ok = all(f[0] for f in flags)
        stopped = any(f[1] for f in flags)
        if stopped:
            print(f"rank {rank}: stopped -- a bounded wait of the full-mesh all-reduce ran out", file=sys.stderr, flush=True)
        if rank == 0:
            print(json.dumps({"fmesh_ranks_check": "ok" if (ok and not stopped) else ("STOPPED" if stopped else "MISMATCH"),
                              "mode": "processes", "world": world, "b_local": b_local, "devices": torch.cuda.device_count(),
                              "final_loss": float(losses[-1])}), flush=True)
        dist.barrier()
    finally:
        comm.close(collective=sys.exc_info()[0] is None)
        dist.destroy_process_group()
    return 0 if (ok and not stopped) else 1


def virtual_main(world):
    """`world` ranks in THIS process, one stream each."""
    sys.path.insert(0, ROOT)
    import numpy as np  # noqa: F401
    import torch
    import d3p_amd._lib as L
    from d3p_amd import dist as ddist
    b_local = int(os.environ.get("D3P_FMESH_CHECK_B_LOCAL", "512"))
    L.require_device()
    try:
        streams = ddist.concurrent_streams(world)
    except L.D3PError as e:
        print(json.dumps({"fmesh_ranks_check": "SKIPPED", "mode": "virtual", "world": world, "why": str(e)}), flush=True)
        return 77
    grid = max(8, 128 // world)
    ok, stopped = True, False
    n = 688_886
    comms = ddist.FMeshComm.local_group(world, n)
    for c in comms:
        c.set_grid(grid)
    try:
        vecs = [torch.randn(n, generator=torch.Generator().manual_seed(7 * n + r)) for r in range(world)]
        for e in range(5):
            scaled = [v * float(e + 1) for v in vecs]
            want = _rank_order_sum(scaled)
            work = [v.cuda() for v in scaled]
            torch.cuda.synchronize()
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    comms[r].allreduce(work[r])
            torch.cuda.synchronize()
            for r in range(world):
                stopped |= comms[r].stopped()
                good = bool(torch.equal(work[r].cpu(), want))
                if not good:
                    print(f"virtual rank {r}: epoch {e}: the sum differs from the rank-order sum", file=sys.stderr, flush=True)
                ok &= good
    finally:
        for c in comms:
            c.close()
    B = b_local * world
    make, X = _problem(B)
    Xd = X.cuda()
    P = int(L.load().d3p_vae_num_params(ctypes.byref(make()._vae_struct(D, {}, 1.0))))
    st0, params = _initial_state(make, P, 86)
    comms = ddist.FMeshComm.local_group(world, P + 2)
    for c in comms:
        c.set_grid(grid)
    try:
        def run(steps, buckets, mask=None):
            engines = [ddist.VaeHipEngine(make()) for _ in range(world)]
            outs = []
            torch.cuda.synchronize()
            for r in range(world):
                pos0, bl = ddist.shard_batch(B, r, world)
                with torch.cuda.stream(streams[r]):
                    outs.append(ddist.vae_run_steps(engines[r], st0, Xd[pos0:pos0 + bl], B, pos0, steps, comm=comms[r], buckets=buckets,
                                                    check_status=False, mask=None if mask is None else torch.tensor(mask[pos0:pos0 + bl]).cuda()))
            torch.cuda.synchronize()
            halted = any(c.stopped() for c in comms)
            return outs, halted
        fused, h1 = run(STEPS, 0)
        three, h2 = run(STEPS, 1)
        mask = _mask(B, world)
        masked, h3 = run(1, 0, mask)
        stopped |= h1 or h2 or h3

        def same(a, b):
            (sa, la), (sb, lb) = a, b
            return torch.equal(la, lb) and torch.equal(sa.rng_key, sb.rng_key) and all(torch.equal(x, y) for x, y in zip(sa.optim_state, sb.optim_state))
        for r in range(world):
            good = same(fused[r], fused[0]) and same(three[r], fused[0]) and same(masked[r], masked[0])
            if not good:
                print(f"virtual rank {r}: replicas / forms differ", file=sys.stderr, flush=True)
            ok &= bool(good)
        ok &= int(fused[0][0].optim_state[0]) == STEPS
        if not stopped:
            svi, ref, ref_l = make(), st0, []
            for _ in range(STEPS):
                ref, l = svi.update(ref, Xd)
                ref_l.append(float(l))
            ok = _compare_with_single(ok, "virtual ranks", fused[0][0], fused[0][1], ref, ref_l)
            ok &= _compare_with_oracle("virtual ranks", masked[0][0], masked[0][1][0], params, X.numpy(), mask, 86)
    finally:
        for c in comms:
            c.close()
    if stopped:
        print("virtual ranks: stopped -- a bounded wait of the full-mesh all-reduce ran out", file=sys.stderr, flush=True)
    print(json.dumps({"fmesh_ranks_check": "ok" if (ok and not stopped) else ("STOPPED" if stopped else "MISMATCH"),
                      "mode": "virtual", "world": world, "b_local": b_local, "chunk": -(-(P + 2) // world),
                      "hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES")}), flush=True)
    return 0 if (ok and not stopped) else 1


def main():
    if "RANK" in os.environ:
        sys.exit(rank_main())
    if os.environ.get("D3P_FMESH_CHECK_VIRTUAL_CHILD"):
        sys.exit(virtual_main(int(os.environ["D3P_FMESH_CHECK_VIRTUAL_CHILD"])))
    virtual = int(os.environ.get("D3P_FMESH_CHECK_VIRTUAL", "0"))
    if virtual:
        # a fresh process whose HIP runtime starts with enough hardware queues for `virtual` streams beside the default one
        env = dict(os.environ, D3P_FMESH_CHECK_VIRTUAL_CHILD=str(virtual))
        env.setdefault("GPU_MAX_HW_QUEUES", str(virtual + 2))
        sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, timeout=900).returncode)
    world = int(os.environ.get("D3P_FMESH_CHECK_WORLD", "4"))
    assert 2 <= world <= 6, "at most 6 processes may use one GPU on this pool"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        try:
            rc = rc or p.wait(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = rc or 124
    for p in procs:
        if p.poll() is None:
            p.kill()
    sys.exit(rc)


if __name__ == "__main__":
    main()
