#!/usr/bin/env python3
"""Headline benchmark: DP-VI update steps/s and per-example gradients/s for Bayesian logistic
regression (d=512, batch 4096 per GPU, AutoDiagonalNormal) on MI355X -- BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W

N > 1 either under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) or as plain `python bench.py --gpus N`: the parent then starts N rank
processes itself -- fresh interpreters, before anything in the parent has touched a GPU -- and rank 0 prints the line.

A step = one DPSVI.update on one freshly sampled minibatch: key schedule -> Feistel subsampling
-> fused per-example gradient / clip / sum -> Gaussian mechanism (ChaCha20) -> Adam.  Inputs
(the synthetic table) are resident in HBM before the timed region.  Prints ONE JSON line:
`value` is tied to --steps; `steady_state` (a fixed leg: the median of 5 x 1024 steps) and `north_star_N1e7` (the same workload over a
10^7-row table) are measured in the same run whatever --steps is.  Order of the legs: steady_state first, then the
headline leg (its own --warmup steps + exactly --steps timed steps, bracketed by barriers), then the others -- a short
headline run as the very first GPU work of the process would be measured at the GPU's idle clocks (`leg_order` in the line).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (the host driver of this pool only supports dmabuf IPC: without this the exchange's hipIpcGetMemHandle fails; it must be in the
# environment before the HIP runtime starts, also when the ranks come from torch.distributed.run and not from spawn_ranks)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, the fp32 matrix (= vector) peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak; an fp32-accurate product on the bf16 pipe is SIX bf16 MFMAs: / 6 = 417 TFLOP/s
SIMDS, SHADER_GHZ = 1024, 2.4   # 256 CUs x 4 SIMD-32; peak shader clock
# wave64 VALU issue: MI355X_MICROARCH.md gives 2 cycles per plain fp32 instruction per SIMD -- the nominal peak used here; no
# opcode measured reaches it (profiles/r03_valu_opcodes.json, launch-based: VOP2 add / xor / fmac 2.4, v_fma_f32 3.0, VOP3 integer /
# DPP / packed 4.3, transcendental 8.3 cycles), so a `frac` against this peak is conservative
VALU_PEAK_GINSTR = SIMDS * SHADER_GHZ / 2.0


def algorithmic_bytes(B, d, P):
    """gathered feature rows + labels + indices, params read, gradient written+read."""
    return B * (4 * d + 4 + 4) + 3 * 4 * P


VALU_MIX_CYCLES = 3.302        # profiles/r02_valu_probe.json: the step's noise mix (threefry2x32-20 + erf_inv, 460 executed instructions per
                               # example) issues one wave64 VALU instruction per 3.30 cycles and SIMD at 4 waves per SIMD (16-wave workgroups)


def copy_peak(dev, gib=2, repeats=5):
    """the device-to-device copy rate of THIS box -- d3p_hbm_copy (16 B per lane, four loads in flight per thread) over
    `gib` GiB, `repeats` timed copies between HIP events on the launch stream; bytes moved = read + written = 2 x size."""
    import torch
    import d3p_amd._lib as L
    lib = L.load()
    n = int(gib) << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    src.fill_(1)
    for _ in range(2):
        L.check(lib.d3p_hbm_copy(L.stream_ptr(), L.ptr(dst), L.ptr(src), n, 16))
    torch.cuda.synchronize()
    rates = []
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.d3p_hbm_copy(L.stream_ptr(), L.ptr(dst), L.ptr(src), n, 16))
        e1.record()
        torch.cuda.synchronize()
        rates.append(2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    ok = bool(torch.equal(dst[:4096], src[:4096]) and int(dst[-1]) == 1)
    del src, dst
    torch.cuda.empty_cache()
    rates.sort()
    return {"GBps_median": round(rates[len(rates) // 2], 1), "GBps_best": round(rates[-1], 1), "GBps_all": [round(r, 1) for r in rates],
            "bytes_copied": n, "repeats": repeats, "copied_correctly": ok,
            "kernel": "k_hbm_copy<16 B per lane, 4 loads in flight> (d3p_hbm_copy), 8192 workgroups of 256 threads, grid-stride, nontemporal loads and stores; rate = (read + "
                      "written bytes) / HIP-event time"}


def cpu_baseline(d, B, seconds, rows):
    """The oracle's C loop over DPSVI.update (reference dataflow: gather, per-example gradients materialised B x P like
    jax.vmap, clip pass, mean, perturb, Adam; Feistel sampling and the gather in C) on the host cores: a 1-thread figure
    and the best of a few thread counts, each on a bounded sample."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    X, y = O.synth_logreg(123, 0, rows, d)
    spec = O.logreg_spec(d, False, 1.0, 1.0, lik_scale=rows, obs_scale=rows)
    hy = O.Hyper(1.0, 1.0, 1e-3, 0.9, 0.999, 1e-8)
    bkey = O.PRNGKey(1)

    def timed(threads, budget):
        st = O.LogregState(O.PRNGKey(0), d, np.zeros(d, np.float32), np.full(d, -2.25, np.float32))
        O.logreg_run_feistel(spec, hy, st, X, y, bkey, 0, B, 1, threads)          # warm-up step (page-in, thread start)
        done, chunk, t0 = 0, 1, time.perf_counter()
        while True:
            O.logreg_run_feistel(spec, hy, st, X, y, bkey, 1 + done, B, chunk, threads)
            done += chunk
            el = time.perf_counter() - t0
            if el >= budget:
                return done / el, done, el
            chunk = max(1, min(64, int(done / el * (budget - el)) // 2 or 1))

    one, n1, e1 = timed(1, min(3.0, seconds / 4))
    best = (one, 1, n1, e1)
    cands = sorted({t for t in (8, 16, 32, 64, 128, cores) if 1 < t <= cores})
    per = max(1.0, (seconds - e1) / max(1, len(cands)))
    tried = {1: round(one, 2)}
    for t in cands:
        sps, n, el = timed(t, per)
        tried[t] = round(sps, 2)
        if sps > best[0]:
            best = (sps, t, n, el)
    sps, threads, n, el = best
    return {"value": B * sps, "unit": "examples/s", "steps_per_sec": round(sps, 3), "cores": threads, "kind": "port",
            "host_cores": cores, "one_thread_steps_per_sec": round(one, 3), "one_thread_value": round(B * one, 1),
            "steps_per_sec_by_threads": tried,
            "sample": f"{n} update steps (B={B}, d={d}) in {el:.1f} s on {threads} OpenMP threads (best of {sorted(tried)}; "
                      f"1 thread: {n1} steps in {e1:.1f} s) over a {rows}-row synthetic table; oracle/ C loop "
                      "d3po_logreg_run_feistel = the reference dataflow (Feistel sampling, gather, B x P per-example "
                      "gradients materialised like jax.vmap, clip pass, mean, ChaCha20 perturbation, Adam), float32"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes here.  The parent never initialises a
    GPU (counting devices does not), the ranks are fresh interpreters; rank 0 prints the JSON line to the shared stdout."""
    import torch
    have = torch.cuda.device_count()
    if os.environ.get("D3P_BENCH_SHARE_GPU"):  # developer rehearsal of the N > 1 flow on a one-GPU box: every rank on cuda:0
        have = n
    if have < n:
        print(f"[bench] --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:          # one rank failed: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def aux_workloads(dev, table7=None, want=("gmm", "vae", "vae2", "poisson")):
    """The other BASELINE workloads, each a short leg (<= ~3 s) with its own roofline, so that the driver's line witnesses them:
    config 3 (mixture model K=16 d=64 B=8192 N=1e7), config 5 (VAE 784-400-50 and the literal [400, 200] variant, B=4096) and
    the north_star's Poisson sampler at N=1e7.  `table7`: the resident (X, y) of the 1e7-row logistic-regression table."""
    import torch
    import d3p_amd.random as rng
    from d3p_amd.minibatch import poisson_batchify_data, subsample_batchify_data
    from d3p_amd.models import (Adam, AutoDiagonalNormal, GaussianMixtureGuide, GaussianMixtureModel, LogisticRegression, Trace_ELBO,
                                VAEGuide, VAEModel)
    from d3p_amd.svi import DPSVI, DPSVIState
    out = {}

    def timed(fn, warm, steps):
        """`fn(first, k)` enqueues k steps; wall time between two device synchronisations + HIP events on the launch stream."""
        fn(0, warm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        fn(warm, steps)
        e1.record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, e0.elapsed_time(e1) * 1e-3

    if "gmm" in want:
        K, d, B, N = 16, 64, 8192, 10_000_000
        g = torch.Generator(device=dev).manual_seed(7)
        comp = torch.randint(0, 3, (N, 1), generator=g, device=dev).float()            # examples/gaussian_mixture_model.py:91-93,
        X = torch.randn(N, d, generator=g, device=dev) + 4.0 * (comp - 1.0)          # three well-separated components
        del comp
        model = GaussianMixtureModel()
        svi = DPSVI(model, GaussianMixtureGuide(model), Adam(1e-3), Trace_ELBO(), 20.0, 1.0, k=K, d=d, num_obs_total=N)
        _, gb = subsample_batchify_data((X,), B)
        bkey = rng.PRNGKey(5)
        st = [svi.init(rng.PRNGKey(0), X[:B])]

        def run(first, k):
            st[0], losses = svi.run_steps(st[0], gb, bkey, first, k)
            run.loss = losses[-1]
        steps = 256
        wall, ev = timed(run, 256, steps)   # (13 ms of warm-up: see the VAE leg)
        P = K + K * d
        alg = B * (4 * d + 4) + 3 * 4 * P                          # 2 175 168 B per step
        # wave64 VALU instructions per step, measured with SQ_INSTS_VALU (profiles/r04_gmm_pmc.json: k_gmm_px 18.13 M, k_gmm_head 3.05 M;
        # round 3: 18.46 + 3.24)
        valu = 21.2e6
        out["gmm_config3"] = {
            "workload": "BASELINE configs[2]: mixture model K=16 d=64, N=1e7 rows resident, batch 8192 (Feistel), C=20, sigma=1, Adam 1e-3",
            "steps": steps, "warmup": 256, "steps_per_sec": round(steps / wall, 2), "value": round(B * steps / wall, 1),
            "unit": "examples/s", "us_per_step": round(1e6 * ev / steps, 3), "final_loss": float(run.loss),
            "roofline": {"bound": "valu", "achieved": round(valu * steps / ev / 1e9, 2), "peak": VALU_PEAK_GINSTR, "unit": "Ginstr/s (wave64)",
                         "frac": round(valu * steps / ev / 1e9 / VALU_PEAK_GINSTR, 4),
                         # (what the step achieves, for comparison with the per-opcode rates of profiles/r03_valu_opcodes.json:
                         # VOP2 2.4, VOP3 / DPP / packed 4.3, transcendental 8.3 cycles per wave64 instruction and SIMD)
                         "cycles_per_valu_instruction_and_simd": round(ev / steps * SHADER_GHZ * 1e9 * SIMDS / valu, 2),
                         "valu_instructions_per_step": valu, "instruction_count_source": "profiles/r04_gmm_pmc.json (SQ_INSTS_VALU of "
                         "k_gmm_px + k_gmm_head); not re-counted in this run",
                         "hbm": {"algorithmic_bytes_per_step": alg, "achieved_GBps": round(alg * steps / ev / 1e9, 2),
                                 "frac": round(alg * steps / ev / 1e9 / HBM_PEAK_GBPS, 5)},
                         "timing": "HIP events on the launch stream around the whole 256-step device-resident run (2 launches per "
                                   "step + the per-64-step preparation)"}}
        del X, gb, svi, st
        torch.cuda.empty_cache()

    for tag, H2 in (("vae", 0), ("vae2", 200)):
        if tag not in want:
            continue
        N, B, D, H, Z = 60000, 4096, 784, 400, 50
        X = (torch.rand(B, 28, 28, generator=torch.Generator().manual_seed(0)) < 0.3).float().to(dev)
        model = VAEModel(scale=1.0 / N)
        svi = DPSVI(model, VAEGuide(model), Adam(1e-3), Trace_ELBO(), 10.0, 1.0, num_obs_total=N, z_dim=Z,
                    hidden_dim=(H, H2) if H2 else H)
        st = [svi.init(rng.PRNGKey(0), X)]

        def run(first, k):
            for _ in range(k):
                st[0], run.loss = svi.update(st[0], X)
        steps = 40
        # (48 warm-up updates = 15 ms: in the driver's short command this leg follows seconds of little GPU work, and with 8 warm-up
        # updates the first VAE leg was timed while the clocks were still ramping -- 360-382 us per step instead of 313)
        # three timed blocks of `steps` updates back to back, the MEDIAN block reported (a single 13 ms block read 326 .. 348 us per step
        # for the two-layer shape on consecutive runs of one box: clock ramps)
        blocks = [timed(run, 48 if i == 0 else 0, steps) for i in range(3)]
        wall, ev = sorted(blocks, key=lambda b: b[1])[1]
        hs = [H] + ([H2] if H2 else [])
        dec, enc = [Z] + hs[::-1] + [D], [D] + hs
        layers = list(zip(dec[:-1], dec[1:])) + list(zip(enc[:-1], enc[1:])) + [(hs[-1], 2 * Z)]
        # every dense layer: forward, backward-data, weight-gradient product of 2 B in out flops; the first encoder layer has no
        # backward-data product
        flops = 2 * B * (3 * sum(i * o for i, o in layers) - D * H)
        Pn = int(st[0].optim_state[1].numel())
        out["vae_config5" + ("_400_200" if H2 else "")] = {
            "workload": "BASELINE configs[4]: VAE 784 -> %s -> 50 (P = %d), batch 4096, C=10, sigma=1, Adam 1e-3; one DPSVI.update "
                        "per step%s" % (hs, Pn, "" if H2 else " (13 launches)"),
            "steps": steps, "warmup": 48, "steps_per_sec": round(steps / wall, 2), "value": round(B * steps / wall, 1),
            "unit": "examples/s", "us_per_step": round(1e6 * ev / steps, 2), "final_loss": float(run.loss),
            "us_per_step_blocks": [round(1e6 * b[1] / steps, 2) for b in blocks],
            "roofline": {"bound": "mfma", "achieved": round(flops * steps / ev / 1e12, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops * steps / ev / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "gemm_flop_per_step": flops,
                         "frac_of_bf16_peak_over_6": round(flops * steps / ev / 1e12 / (MFMA_BF16_PEAK_TFLOPS / 6.0), 4),
                         "dtype": "f32-accurate: the large products as six v_mfma_f32_32x32x16_bf16 on an EXACT three-way bf16 split of both "
                                  "fp32 operands, fp32 accumulate (dropped terms < 2^-23 relative; DESIGN.md 1c); peak quoted = the fp32 MFMA peak "
                                  "the reference's float32 arithmetic would be priced against",
                         "timing": "HIP events on the launch stream around 40 consecutive updates (all kernels of a step, not only "
                                   "the matrix products)"}}
        del X, svi, st
        torch.cuda.empty_cache()

    if "poisson" in want and table7 is not None:
        X7, y7 = table7
        N, d = X7.shape
        B = 4096
        model = LogisticRegression(d, prior_scale=1.0)
        svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), 1.0, 1.0, num_obs_total=N)
        params = torch.cat([torch.zeros(d, device=dev), torch.full((d,), svi.guide.unconstrained_init_scale(), device=dev)])
        st = [DPSVIState(svi.optim.init(params), rng.PRNGKey(0), float(N))]
        _, gb = poisson_batchify_data((X7, y7), B / N, 0.99)                 # examples/logistic_regression.py:126-127
        bkey = rng.PRNGKey(1)

        def run(first, k):
            st[0], losses = svi.run_steps(st[0], gb, bkey, first, k)
            run.loss = losses[-1]
        steps = 256
        wall, ev = timed(run, 384, steps)
        blocks = (N + 15) // 16
        lane_ops = blocks * 980.0            # ~980 32-bit integer operations per 64-byte ChaCha20 block
        peak = SIMDS * 64 * SHADER_GHZ * 1e9 / 2.0   # lane-operations per second at 2 cycles per wave64 integer instruction
        out["poisson_N1e7"] = {
            "workload": "north_star sampler: logistic regression d=512 over N=1e7 resident rows, Poisson batches q = 4096 / N padded "
                        "to the 0.99 quantile (N / 16 ChaCha20 blocks + an ordered compaction over N per step, fused: the keystream "
                        "never reaches HBM)",
            "steps": steps, "warmup": 384, "steps_per_sec": round(steps / wall, 2), "value": round(B * steps / wall, 1),
            "unit": "examples/s (expected batch 4096)", "us_per_step": round(1e6 * ev / steps, 3), "final_loss": float(run.loss),
            "roofline": {"bound": "valu", "achieved": round(lane_ops * steps / ev / 1e12, 3), "peak": round(peak / 1e12, 2),
                         "unit": "T integer lane-operations/s (ChaCha20 keystream of the Bernoulli mask)",
                         "frac": round(lane_ops * steps / ev / peak, 4), "chacha_blocks_per_step": blocks,
                         # the mix of the block function -- 1/3 VOP3 rotates at 4.3, 2/3 VOP2 add / xor at 2.4 cycles per wave64
                         # instruction (profiles/r03_valu_opcodes.json, launch-based) -- issues at 3.03 cycles on average, not 2
                         "frac_of_measured_issue_rate": round(lane_ops * steps / ev / peak * 3.03 / 2.0, 4),
                         "keystream_bytes_per_step": 4 * N,
                         "note": "whole step (mask + compaction + the DP-VI step) over the mask's integer work alone",
                         "timing": "HIP events on the launch stream around the 256-step device-resident run"}}
    return out


def vae_dist_workload(dev, world, rank, group_barrier, share_gpu=False, emulate=0):
    """BASELINE configs[4] at N > 1 ("VAE ... 1 vs 8 GPU"): the epoch body of examples/vae.py:227-246 data-parallel -- the batch
    sharded by position (4096 examples per GPU: weak scaling like the headline), ONE sum-all-reduce of the P + 2 fp32 sums per step
    (2.76 MB for 784-400-50), the noise added once after it.  Drivers, each timed: the NATIVE loop (d3p_dpvi_vae_run_dist: one C
    call for the run, RCCL on the library's communicator) with the reduce in ONE bucket in the stream and in TWO buckets on a second
    stream (the decoder's sums travel while the encoder's weight-gradient products run), the native loop with the FULL-MESH
    reduce-scatter + all-gather of d3p_fmesh_* as its collective, and the Python-driven loop over torch.distributed.  The leg's figures are the fastest driver's; all are listed.  Before timing, 3 steps with every driver from the
    same state: replicas bitwise equal over the ranks, drivers equal to fp32 rounding.
    emulate = W (one GPU): rank 0's share of a W-rank job, no peers (a one-rank communicator): the rank-local cost, at the weak-scaling
    share (4096 per GPU) and at the strong-scaling one (4096 / W)."""
    import torch
    import torch.distributed as dist
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    from d3p_amd.models import Adam, Trace_ELBO, VAEGuide, VAEModel
    from d3p_amd.svi import DPSVI
    out = {}
    ranks = emulate or world
    comm = None
    if not share_gpu:
        try:
            comm = ddist.NativeComm()
        except Exception as e:  # noqa: BLE001 -- the Python-driven loop remains
            print(f"[bench] rank {rank}: no native communicator for the VAE legs ({e})", file=sys.stderr)
    if world > 1:   # every rank takes the same drivers
        flag = torch.tensor([int(comm is not None)], dtype=torch.int32, device=dev)
        if share_gpu:
            flag = flag.cpu()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not int(flag[0]) and comm is not None:
            comm.close()
            comm = None
    shapes = [("vae_config5", 0, 4096), ("vae_config5_400_200", 200, 4096)]
    if emulate:
        shapes += [("vae_config5_strong_share", 0, 4096 // emulate)]
    try:
        for tag, H2, Bl in shapes:
            N, D, H, Z = 60000, 784, 400, 50
            Bg = Bl * ranks
            pos0 = Bl * rank
            X = (torch.rand(Bl, 28, 28, generator=torch.Generator().manual_seed(1000 + rank)) < 0.3).float().to(dev)
            model = VAEModel(scale=1.0 / N)
            svi = DPSVI(model, VAEGuide(model), Adam(1e-3), Trace_ELBO(), 10.0, 1.0, num_obs_total=N, z_dim=Z,
                        hidden_dim=(H, H2) if H2 else H)
            st0 = svi.init(rng.PRNGKey(0), X)          # (a function of the key and the shapes: identical on every rank)
            drivers = [("torch_loop", {})] if (world > 1 or comm is None) else []
            if comm is not None:
                drivers += [("native_1_bucket", {"comm": comm, "buckets": 1}), ("native_2_buckets", {"comm": comm, "buckets": 2})]
            # the full-mesh reduce-scatter + all-gather over the peers' hipIpc-mapped inboxes (d3p_fmesh_*) as the step's collective
            mesh = None
            if world > 1:
                try:
                    mesh = ddist.FMeshComm(int(st0.optim_state[1].numel()) + 2)
                    if share_gpu:
                        mesh.set_grid(48)      # (the ranks share one GPU here: room for each other's kernels)
                except Exception as e:  # noqa: BLE001
                    print(f"[bench] rank {rank}: no full-mesh collective for the VAE legs ({e})", file=sys.stderr)
                flag = torch.tensor([int(mesh is not None)], dtype=torch.int32, device=dev)
                if share_gpu:
                    flag = flag.cpu()
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if not int(flag[0]) and mesh is not None:
                    mesh.close(collective=False)   # (a peer has none: this rank gives its mesh up alone)
                    mesh = None
                if mesh is not None:   # (fused: tile sums + collective + update in one launch; _3_launches: kept apart)
                    drivers += [("native_full_mesh", {"comm": mesh}), ("native_full_mesh_3_launches", {"comm": mesh, "buckets": 1})]
            # ---- correctness first: 3 steps per driver from the same state
            check, ref = {}, None
            for name, kw in drivers:
                # (a mesh driver's status is read below and AGREED over the ranks: no rank raises alone in front of the all_gather)
                kw3 = dict(kw, check_status=False) if (mesh is not None and kw.get("comm") is mesh) else kw
                s3, l3 = ddist.vae_run_steps(ddist.VaeHipEngine(svi), st0, X, Bg, pos0, 3, **kw3)
                torch.cuda.synchronize()
                sig = torch.cat([s3.optim_state[0].reshape(1).to(torch.int32), s3.rng_key.reshape(16).view(torch.int32),
                                 s3.optim_state[1].view(torch.int32)]).contiguous()
                same = True
                if world > 1:
                    buf = sig.cpu() if share_gpu else sig
                    got = [torch.empty_like(buf) for _ in range(world)]
                    dist.all_gather(got, buf)
                    same = all(torch.equal(got[0], g) for g in got[1:])
                par = s3.optim_state[1]
                if ref is None:
                    ref = par.clone()
                stopped = bool(kw.get("comm") is mesh and mesh is not None and mesh.stopped())
                rel = float((par - ref).abs().max() / ref.abs().max())
                good = bool(same) and not stopped and int(s3.optim_state[0]) == 3 and rel <= 1e-2   # (Adam on near-zero gradients: see tests)
                if world > 1:   # every rank takes the same decision
                    gflag = torch.tensor([int(good)], dtype=torch.int32, device=dev)
                    if share_gpu:
                        gflag = gflag.cpu()
                    dist.all_reduce(gflag, op=dist.ReduceOp.MIN)
                    good = bool(int(gflag[0]))
                check[name] = {"replicas_bitwise": bool(same), "step": int(s3.optim_state[0]), "stopped": stopped,
                               "max_rel_diff_vs_first_driver": rel, "ok": good}
            drivers = [(n_, k_) for n_, k_ in drivers if check[n_]["ok"]] or drivers[:1]   # (a driver that failed its check is not timed)
            # ---- timing
            warm, steps = (8, 10) if share_gpu else (48, 40)
            timed, stopped_legs = {}, []
            for name, kw in drivers:
                on_mesh = mesh is not None and kw.get("comm") is mesh
                if on_mesh:   # (the status is read HERE, after the timed region, and agreed over the ranks: no rank raises alone)
                    kw = dict(kw, check_status=False)
                engine = ddist.VaeHipEngine(svi)
                group_barrier()
                st, _ = ddist.vae_run_steps(engine, st0, X, Bg, pos0, warm, collect_losses=False, **kw)
                group_barrier()
                t0 = time.perf_counter()
                st, losses = ddist.vae_run_steps(engine, st, X, Bg, pos0, steps, **kw)
                group_barrier()
                wall = time.perf_counter() - t0
                halted = int(on_mesh and mesh.stopped())
                if world > 1:
                    t = torch.tensor([wall, float(halted)], dtype=torch.float64, device=dev)
                    if share_gpu:
                        t = t.cpu()
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    wall, halted = float(t[0]), int(t[1])
                del engine
                if halted:    # a bounded wait of the collective ran out on some rank: the state is partly updated -- not a timing
                    stopped_legs.append(name)
                    print(f"[bench] rank {rank}: VAE driver {name} was stopped by a bounded wait during the timed run -- dropped", file=sys.stderr)
                    continue
                timed[name] = (wall, float(losses[-1]))
            if not timed:
                out[tag + f"_dp{world}"] = {"error": "every driver's timed run was stopped by a bounded wait", "stopped_drivers": stopped_legs,
                                            "collective_check": check}
                if mesh is not None:
                    mesh.close()
                continue
            best = min(timed, key=lambda k: timed[k][0])
            wall, final_loss = timed[best]
            hs = [H] + ([H2] if H2 else [])
            dec, enc = [Z] + hs[::-1] + [D], [D] + hs
            layers = list(zip(dec[:-1], dec[1:])) + list(zip(enc[:-1], enc[1:])) + [(hs[-1], 2 * Z)]
            flops = 2 * Bl * (3 * sum(i * o for i, o in layers) - D * H)   # per RANK and step
            Pn = int(st0.optim_state[1].numel())
            key = tag + (f"_rank_local_of_{emulate}" if emulate else f"_dp{world}")
            B_done = Bl if emulate else Bg
            out[key] = {
                "workload": "BASELINE configs[4] data-parallel: VAE 784 -> %s -> 50 (P = %d), batch %d per GPU (global %d, sharded by position), "
                            "C=10, sigma=1, Adam 1e-3; per step local sums -> one sum-all-reduce of %d fp32 -> apply (noise once)" % (hs, Pn, Bl, Bg, Pn + 2),
                "n_gpus": world, "steps": steps, "warmup": warm, "driver": best, "steps_per_sec": round(steps / wall, 2),
                "value": round(B_done * steps / wall, 1), "unit": "examples/s (whole job)" if not emulate else "examples/s (this rank's share)",
                "us_per_step": round(1e6 * wall / steps, 2), "final_loss": final_loss,
                "us_per_step_by_driver": {k: round(1e6 * v[0] / steps, 2) for k, v in timed.items()},
                "collective_check": check, "drivers_stopped_while_timed": stopped_legs,
                "collective": {"bytes_per_step": 4 * (Pn + 2),
                               "backend": ("none (one rank: the communicator has no peers)" if emulate else
                                           "gloo (torch_loop) / full mesh over hipIpc (native_full_mesh) -- shared-GPU rehearsal" if share_gpu else
                                           "RCCL: the library's communicator (native_*_bucket) / torch.distributed nccl (torch_loop) / full-mesh "
                                           "reduce-scatter + all-gather over the peers' hipIpc inboxes (native_full_mesh)")},
                "roofline": {"bound": "mfma", "achieved": round(flops * steps / wall / 1e12, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s per GPU",
                             "frac": round(flops * steps / wall / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "gemm_flop_per_step_and_gpu": flops,
                             "frac_of_bf16_peak_over_6": round(flops * steps / wall / 1e12 / (MFMA_BF16_PEAK_TFLOPS / 6.0), 4),
                             "timing": "wall clock between barriers around the steps (max over ranks): kernels + the collective"}}
            if mesh is not None:
                mesh.close()
            del X, svi, st, st0
            torch.cuda.empty_cache()
    finally:
        if comm is not None:
            comm.close()
    return out


class RunStopped(RuntimeError):
    """A data-parallel run that a bounded wait stopped (d3p_dpvi_logreg_run_status): its numbers are not measurements."""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)   # whole prepared batches of 128 steps
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--rows-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--batch-per-gpu", type=int, default=4096)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-batch", action="store_true",
                    help="skip the extra leg that times the same step kernel at batch 32768 (throughput regime)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the steady_state and north_star_N1e7 legs (profiling runs)")
    ap.add_argument("--no-aux-workloads", action="store_true",
                    help="skip the legs for the other BASELINE workloads (mixture model, VAE, Poisson sampler at N = 1e7)")
    ap.add_argument("--steady-steps", type=int, default=1024, help="steps per repeat of the steady_state leg (whole prepared batches of 128)")
    ap.add_argument("--steady-repeats", type=int, default=5)
    ap.add_argument("--sampler", choices=["feistel", "poisson"], default="feistel",
                    help="feistel = subsample_batchify_data w/o replacement (headline); poisson = poisson_batchify_data "
                         "with q = B/N and the 0.99-quantile padding of examples/logistic_regression.py:126-127")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="developer switch: run rank 0's share of an N-rank weak-scaling job on this GPU, without the "
                         "collective (local cost of the data-parallel step)")
    ap.add_argument("--force-dist-loop", action="store_true",
                    help="developer switch: use the stepwise data-parallel loop (d3p_amd.dist) even with one rank")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    import d3p_amd._lib as L
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    from d3p_amd.minibatch import poisson_batchify_data, subsample_batchify_data
    from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.svi import DPSVI, DPSVIState

    # stdout carries ONE thing, the JSON line: from here on file descriptor 1 is stderr for everybody in this process -- Python prints,
    # libraries that write to stdout on their own (gloo announces its ranks there, RCCL can be made to) -- and the line goes out through
    # the saved descriptor (emit)
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(text):
        sys.stdout.flush()
        os.write(json_fd, (text + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    share_gpu = bool(os.environ.get("D3P_BENCH_SHARE_GPU"))  # rehearsal: all ranks on cuda:0, gloo instead of RCCL (which refuses
    if share_gpu:                                               # two ranks on one device); the one-shot exchange works as usual,
        local_rank = 0                                          # as one launch per step: the in-launch form needs each rank's
        # launch resident beside the others', which one GPU only gives to SMALL launches: D3P_BENCH_SHARE_GPU=inlaunch keeps the in-launch
        # form (the production form of a real N-GPU run) for a rehearsal at a small shape, e.g. `--gpus 2 --batch-per-gpu 512 --steps 24
        # --warmup 24 --no-extra-legs` (24 x 19 sixteen-wave workgroups per launch leave room for the other rank's launch)
        if os.environ["D3P_BENCH_SHARE_GPU"] != "inlaunch":
            os.environ.setdefault("D3P_XCHG_PER_STEP", "1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or (args.force_dist_loop and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    L.require_device()
    lib = L.load()

    emu = args.emulate_world if args.emulate_world > 1 else 0
    ranks = emu or world
    d, Bg = args.dim, args.batch_per_gpu * ranks
    D = d
    P = 2 * D
    single = world == 1 and not args.force_dist_loop and not emu
    if emu:
        args.force_dist_loop = True

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def read_kernel_timing():
        us, launches, steps = C.c_double(), C.c_uint32(), C.c_uint32()
        L.check(lib.d3p_dpvi_logreg_kernel_timing_read(C.byref(us), C.byref(launches), C.byref(steps)))
        return us.value, launches.value, steps.value

    comm = comm_x = comm_r = None
    dist_driver = "none"
    if not single and not os.environ.get("D3P_DIST_TWO_PHASE") and not os.environ.get("D3P_DIST_TORCH_LOOP"):
        # Preferred data-parallel driver: the native loop (one C call for the whole run; per step one launch and the
        # rank's accumulator exchange on the same stream, communicator owned by libd3p_hip.so).  Every rank must take the
        # same decision, so success is agreed on with an all-reduce of a flag.
        # First choice: the one-shot full-mesh exchange (d3p_xchg_*: one kernel per step writes the folded 8 KB row into the
        # peers over xGMI); second: the RCCL ring all-reduce inside the same native loop.
        def agreed(make):
            ok, c = 1, None
            try:
                c = make()
            except Exception as e:  # noqa: BLE001 -- a failure here only selects the next driver
                ok = 0
                print(f"[bench] rank {rank}: {make.__name__} unavailable ({e})", file=sys.stderr)
            if world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag[0])
            if not ok and c is not None:    # (a peer has none: this rank gives its communicator up alone -- no teardown barrier)
                if isinstance(c, (ddist.XchgComm, ddist.FMeshComm)):
                    c.close(collective=False)
                else:
                    c.close()
                c = None
            return c

        def xchg_comm():
            return ddist.XchgComm(2 * D + 4)

        def rccl_comm():
            return ddist.NativeComm()
        comm_x = None if os.environ.get("D3P_DIST_RCCL") else agreed(xchg_comm)
        # (the RCCL communicator is made even when the exchange is available: the first-contact check below runs both)
        comm_r = None if (share_gpu or (emu and comm_x is not None)) else agreed(rccl_comm)
        comm = comm_x if comm_x is not None else comm_r

    # ---------------------------------------------------------------- first contact: a CORRECTNESS run before any timed leg
    # The first time these ranks meet over real links (xGMI, hipIpc-mapped uncached inboxes, RCCL) must not also be the first
    # measurement: 8 data-parallel steps of the benchmark's shape on a small table with every driver -- the one-shot exchange
    # inside the chained launch, the RCCL all-reduce of the native loop, the Python-driven torch.distributed loop -- then the
    # returned states (step counter, key, parameters, Adam moments) are all-gathered and compared: replicas BITWISE equal across the
    # real ranks (int64 sums are exact and the noise is added once from the same key: SURVEY 8e, F6), exchange = RCCL = torch to
    # fp32 rounding (other workgroup partials).  A driver that fails here is dropped (exchange -> RCCL -> torch) instead of timed.
    collective_check = None

    def run_collective_check():
        nonlocal comm
        check_steps, n_rows = 8, 16384 * ranks
        lo, hi = ddist.shard_rows(n_rows, rank, ranks)
        X = torch.empty((hi - lo, d), dtype=torch.float32, device=dev)
        y = torch.empty(hi - lo, dtype=torch.float32, device=dev)
        L.check(lib.d3p_synth_logreg(L.stream_ptr(), 321, lo, hi - lo, d, L.ptr(X), L.ptr(y)))
        model = LogisticRegression(d, prior_scale=1.0)
        svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-2), Trace_ELBO(), clipping_threshold=1.0, dp_scale=1.0, num_obs_total=n_rows)
        params = torch.cat([torch.zeros(D, device=dev), torch.full((D,), svi.guide.unconstrained_init_scale(), device=dev)])
        state0 = DPSVIState(svi.optim.init(params), rng.PRNGKey(5), float(n_rows))
        bkey = rng.PRNGKey(6)
        engine = ddist.FusedHipEngine(svi, X, y, n_rows, lo, hi, L.D3P_BATCH_FEISTEL, Bg)

        def signature(st):   # [step | key (16) | params, m, v as bit patterns]
            step, par, m, v = st.optim_state
            return torch.cat([step.reshape(1).to(torch.int32), st.rng_key.reshape(16).view(torch.int32), par.view(torch.int32),
                              m.view(torch.int32), v.view(torch.int32)]).contiguous()

        def gathered(sig):
            if world == 1:
                return [sig]
            buf = sig.cpu() if share_gpu else sig      # (gloo rehearsal: host tensors)
            out = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(out, buf)
            return [o.to(dev) for o in out]

        def drive(c, native):
            rec = {"ran": False}
            try:
                if native:
                    st, _ = ddist.run_steps_native(engine, state0, bkey, 0, check_steps, comm=c, collect_losses=False)
                else:
                    st, _ = ddist.run_steps(engine, state0, bkey, 0, check_steps, collect_losses=False)
                torch.cuda.synchronize()
                code = ddist.native_run_status(engine)[0] if native else 0
            except Exception as e:  # noqa: BLE001 -- a failing driver is reported and dropped, never timed
                rec["error"] = f"{type(e).__name__}: {e}"
                st, code = None, -1
            sig = signature(st) if st is not None else torch.full((17 + 3 * P,), -1, dtype=torch.int32, device=dev)
            sigs = gathered(sig)
            rec.update({"ran": st is not None, "status": int(code), "step": int(sig[0]),
                        "replicas_bitwise": bool(all(torch.equal(sigs[0], o) for o in sigs[1:])) and st is not None})
            return rec, (st.optim_state[1].clone() if st is not None else None)

        def rel(a, b):
            if a is None or b is None:
                return None
            return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

        t_rec, t_par = drive(None, False) if world > 1 else ({"ran": False, "note": "one process: no torch.distributed loop to compare"}, None)
        x_rec, x_par = drive(comm_x, True) if comm_x is not None else ({"ran": False}, None)
        r_rec, r_par = drive(comm_r, True) if comm_r is not None else ({"ran": False}, None)
        ref = t_par if t_par is not None else (r_par if r_par is not None else x_par)
        tol = 2e-4     # (parameters after 8 Adam steps of 1e-2; tests/test_dist.py holds the same paths to 1e-4 + 2e-6 absolute)

        def good(rec, par):
            ok = bool(rec.get("ran")) and rec["status"] == 0 and rec["step"] == check_steps and rec["replicas_bitwise"]
            if ok and ref is not None and par is not None:
                ok = rel(par, ref) <= tol
            if world > 1:   # every rank takes the same decision
                flag = torch.tensor([int(ok)], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = bool(int(flag[0]))
            return ok
        x_ok = good(x_rec, x_par) if comm_x is not None else False
        r_ok = good(r_rec, r_par) if comm_r is not None else False
        comm = comm_x if x_ok else (comm_r if r_ok else None)
        for c in (comm_x, comm_r):
            if c is not None and c is not comm:
                c.close()
        chosen = "xchg" if (comm is comm_x and comm is not None) else "rccl" if comm is not None else "torch"
        return {"ranks": world, "emulated_ranks": emu or None, "steps": check_steps, "global_batch": Bg, "rows": n_rows,
                "xchg_bitwise": x_rec.get("replicas_bitwise"), "vs_rccl": rel(x_par, r_par), "vs_torch": rel(x_par, t_par),
                "rccl_vs_torch": rel(r_par, t_par), "tolerance": tol, "xchg": x_rec, "rccl": r_rec, "torch": t_rec,
                "xchg_ok": x_ok if comm_x is not None else None, "rccl_ok": r_ok if comm_r is not None else None,
                "driver_chosen": chosen,
                "what": "replicas_bitwise: step counter, key, parameters and Adam moments all-gathered over the ranks, equal bit for bit; "
                        "vs_*: max |difference| of the parameters / max |parameter| after the same 8 steps with the other driver"}

    if not single and not os.environ.get("D3P_DIST_TWO_PHASE") and not os.environ.get("D3P_DIST_TORCH_LOOP") \
            and not os.environ.get("D3P_BENCH_NO_COLLECTIVE_CHECK"):
        collective_check = run_collective_check()
        if rank == 0:
            print("[bench] collective_check: " + json.dumps(collective_check), file=sys.stderr, flush=True)
    if not single:
        dist_driver = ("native loop, one-shot full-mesh exchange (d3p_xchg), " +
                       ("one exchange launch behind every step launch (D3P_XCHG_PER_STEP)" if os.environ.get("D3P_XCHG_PER_STEP") else
                        "inside the chained launch (updater form) where the shape has k_logreg_chain") if isinstance(comm, ddist.XchgComm) else
                       "native loop, RCCL all-reduce" if comm is not None else "torch")

    def make_workload(n_rows_total, native=True):
        """Table shard + model + state + a `run(state, first, k)` closure for a table of n_rows_total rows.
        native (data-parallel runs only): the one-C-call loop of libd3p_hip.so, else the Python-driven loop."""
        lo, hi = ddist.shard_rows(n_rows_total, rank, ranks)
        X = torch.empty((hi - lo, d), dtype=torch.float32, device=dev)
        y = torch.empty(hi - lo, dtype=torch.float32, device=dev)
        L.check(lib.d3p_synth_logreg(L.stream_ptr(), 123, lo, hi - lo, d, L.ptr(X), L.ptr(y)))
        model = LogisticRegression(d, prior_scale=1.0)
        svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-3), Trace_ELBO(), clipping_threshold=1.0, dp_scale=1.0,
                    num_obs_total=n_rows_total)
        params = torch.cat([torch.zeros(D, device=dev), torch.full((D,), svi.guide.unconstrained_init_scale(), device=dev)])
        state = DPSVIState(svi.optim.init(params), rng.PRNGKey(0), float(n_rows_total))
        bkey = rng.PRNGKey(1)
        if single:
            if args.sampler == "poisson":
                _, get_batch = poisson_batchify_data((X, y), Bg / n_rows_total, 0.99)
            else:
                _, get_batch = subsample_batchify_data((X, y), Bg)

            def run(st, first, k):
                # (the run's status is read AFTER the timed region, like the data-parallel legs do: timed_leg; should a chained launch
                # ever be stopped there, the leg is taken again with run_steps' own check, which re-runs a stopped run launch by launch)
                return svi.run_steps(st, get_batch, bkey, first, k, check_status=run.safe)
            run.svi, run.safe = svi, False
        else:
            engine_cls = ddist.HipEngine if os.environ.get("D3P_DIST_TWO_PHASE") else ddist.FusedHipEngine
            if args.sampler == "poisson":   # q = B / N, padded to the 0.99 quantile of Poisson(B) (examples/logistic_regression.py:126-127)
                from scipy.stats import poisson as _poisson
                engine = engine_cls(svi, X, y, n_rows_total, lo, hi, L.D3P_BATCH_POISSON, int(_poisson.ppf(0.99, Bg)), q=Bg / n_rows_total)
            else:
                engine = engine_cls(svi, X, y, n_rows_total, lo, hi, L.D3P_BATCH_FEISTEL, Bg)

            def run(st, first, k):
                if comm is not None and native:
                    return ddist.run_steps_native(engine, st, bkey, first, k, comm=comm, collect_losses=False)
                return ddist.run_steps(engine, st, bkey, first, k, collect_losses=False)
            run.native_engine = engine if (comm is not None and native) else None
        return svi, state, run, (X, y), bkey

    def timed_leg(run, state, first, warm, steps):
        """warm-up, barrier, `steps` timed steps bracketed by barriers (max over ranks), HIP-event kernel timing."""
        barrier()   # (the ranks enter the warm-up together: its exchanges wait for every peer, with a bound)
        state0_of_leg = state
        state, _ = run(state, first, warm)
        barrier()
        L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(1))
        t0 = time.perf_counter()
        state, losses = run(state, first + warm, steps)
        barrier()
        elapsed = time.perf_counter() - t0
        L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(0))
        kt = read_kernel_timing()
        one = getattr(run, "svi", None)
        if one is not None and not run.safe:   # single rank: a stopped run must not become a number
            aborted, _ = one.last_run_status()
            if aborted:
                print("[bench] the chained launch was stopped by a bounded wait (" + one.last_abort_code() + "); taking the leg again "
                      "with DPSVI.run_steps' status check and launch-by-launch fallback", file=sys.stderr, flush=True)
                run.safe = True
                return timed_leg(run, state0_of_leg, first, warm, steps)
        eng = getattr(run, "native_engine", None)
        if eng is not None:  # a stopped run must not become a number
            code, _ = ddist.native_run_status(eng)
            if code:
                raise RunStopped(f"rank {rank}: the data-parallel run was stopped by a bounded wait -- {L.describe_abort(code)}")
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0])
        return state, losses, elapsed, kt

    def kernel_record(kt, B_rank):
        us, launches, ksteps = kt
        if not launches or not ksteps:
            return None
        alg_step = algorithmic_bytes(B_rank, d, P)
        ach = alg_step * ksteps / (us * 1e-6) / 1e9
        return {"achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBPS, 4), "kernel_us_per_step": round(us / ksteps, 3),
                "avg_launch_us": round(us / launches, 3), "launches": launches, "steps_per_launch": round(ksteps / launches, 3),
                "algorithmic_bytes_per_step": alg_step}

    # the copy rate of THIS box beside the vendor's 8 TB/s (rank 0; < 0.1 s, before any leg)
    box_copy = None
    if rank == 0 and not os.environ.get("D3P_BENCH_NO_COPY_PEAK"):
        try:
            box_copy = copy_peak(dev)
        except Exception as e:  # noqa: BLE001 -- a measurement aid must not cost the line
            print(f"[bench] copy peak not measured ({type(e).__name__}: {e})", file=sys.stderr)

    aux = {}   # the other BASELINE workloads (single-GPU runs): filled by measure()
    progress = {}   # rank 0: the line's fields as soon as the headline leg of a measure() call is done (the legs behind it fill in)

    def measure(native, extra_legs):
        """All GPU legs with one driver; returns the fields of the JSON line (rank 0) or None."""
        n_rows = args.rows_per_gpu * ranks
        svi, state0, run, table, bkey = make_workload(n_rows, native)
        # ---------------------------------------------------------------- steady state: a fixed leg, whatever --steps is.
        # It runs FIRST, on the same workload and from the same initial state (batches warmup + steps ...): the short headline leg
        # that follows then finds the GPU at its working clocks -- measured: the 20 steps of the driver's run take 9.1 us of kernel
        # time per step when they are the first GPU work after start-up, 8.3 us after a few thousand steps (--no-extra-legs
        # shows the cold figure).  Every leg does its own warm-up steps and is bracketed by barriers.
        steady = None
        if extra_legs:
            # "5 repeats, report median" -- five timed blocks of --steady-steps steps back to back (2048 warm-up steps in front
            # of the first, one prepared batch of 128 in front of the others), each bracketed like the headline leg; the MEDIAN block is
            # the leg's figure, all five are listed
            reps, st_s, first_s = [], state0, args.warmup + args.steps
            for i in range(args.steady_repeats):
                warm_s = 2048 if i == 0 else 128
                st_s, _, el_s, kt_s = timed_leg(run, st_s, first_s, warm_s, args.steady_steps)
                first_s += warm_s + args.steady_steps
                reps.append((el_s, kt_s))
            del st_s
            if rank == 0:
                el_s, kt_s = sorted(reps, key=lambda r: r[0])[len(reps) // 2]
                sps = args.steady_steps / el_s
                steady = {"steps": args.steady_steps, "repeats": len(reps), "warmup": 2048, "statistic": "median of the repeats",
                          "steps_per_sec": round(sps, 2), "value": round(Bg * sps, 1),
                          "ms_per_step": round(1000.0 * el_s / args.steady_steps, 6), "kernel": kernel_record(kt_s, Bg // ranks),
                          "steps_per_sec_repeats": [round(args.steady_steps / r[0], 2) for r in reps]}
        # ---------------------------------------------------------------- headline leg (value is tied to --steps)
        state, losses, elapsed, kt = timed_leg(run, state0, 0, args.warmup, args.steps)
        steps_per_s = args.steps / elapsed
        final_loss = float(losses[-1]) if losses is not None else None

        # ---- dominant kernel: the step kernel of the timed region itself, HIP events on the launch stream
        roofline = None
        krec = kernel_record(kt, Bg // ranks) if rank == 0 else None
        if krec is not None:
            chained = single and not os.environ.get("D3P_NO_CHAINED_STEPS")
            traffic, traffic_src = None, None
            import glob
            tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
            tfile = tfiles[-1] if tfiles else ""
            if chained and tfile:
                # NOT measured in this run: PMC counters need rocprofv3's own passes (tools/profile_round.sh); the figure is the
                # committed profile's bytes per step x this run's steps per launch, with its provenance beside it
                try:
                    rec = json.load(open(tfile))
                    per_step = rec.get("hbm_bytes_per_step")
                    traffic = per_step * krec["steps_per_launch"] if per_step is not None else None
                    traffic_src = {"file": "profiles/" + os.path.basename(tfile), "commit": rec.get("commit"), "kernel": rec.get("kernel"),
                                   "profiled_steps": rec.get("steps"), "hbm_bytes_per_step": per_step,
                                   "ratio_to_algorithmic": (round(per_step / krec["algorithmic_bytes_per_step"], 3)
                                                            if per_step is not None else None),
                                   "split": rec.get("split"),
                                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at the named commit "
                                           "(FETCH doubled per the gfx950 correction -- an UPPER bound: `split` says which part the correction "
                                           "is calibrated for); not re-measured in this run"}
                except Exception:  # noqa: BLE001
                    traffic, traffic_src = None, None
            if chained:
                kname = ("k_logreg_chain<PLIST, STAMPS=0, ICPT=0, XCHG=0, W=16> (chained launch: the <= 128 DP-VI steps of a prepared batch "
                         "per launch, 128 sixteen-wave workgroups per step; k_logreg_main<MODE 3> for shapes other than d = 512)")
            elif isinstance(comm, ddist.XchgComm) and native and not os.environ.get("D3P_XCHG_PER_STEP"):
                kname = ("k_logreg_chain<PLIST=1, STAMPS=0, ICPT=0, XCHG=1, W=8> (data-parallel chained launch, round-2 form D3P_XCHG_W8=1: per step 256 "
                         "compute workgroups + 1 key-chain + 2 exchange workgroups that carry the one-shot full-mesh sum-exchange over xGMI)"
                         if os.environ.get("D3P_XCHG_W8") else
                         "k_logreg_chain<PLIST=1, STAMPS=0, ICPT=0, XCHG=1, W=16> (data-parallel chained launch, updater form: 128 sixteen-wave "
                         "workgroups per step; the last arrivers of the 8 arrival groups fold the rank's sums, exchange them full-mesh over xGMI as "
                         "tagged 8-byte words, apply noise + Adam once and publish the parameters as tagged words the next step polls)")
            elif comm is not None and native:
                kname = "k_logreg_main<MODE 2> (one launch per DP-VI step) + the step's collective (k_xchg or ncclAllReduce) on the same stream"
            else:
                kname = "k_logreg_main<MODE 2> (one launch per DP-VI step; torch.distributed.all_reduce between the launches)"
            # ---- the kernel's OTHER bound, from counters: wave64 VALU instructions per step (SQ_INSTS_VALU of the step kernel, a
            # rocprofv3 --pmc pass of this command kept under profiles/) over this run's kernel time per step
            valu = None
            vfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_logreg_valu_pmc.json")))
            if chained and vfiles:
                try:
                    vrec = json.load(open(vfiles[-1]))
                    ips = float(vrec["valu_instructions_per_step"])
                    rate = ips / (krec["kernel_us_per_step"] * 1e-6) / 1e9      # G wave64 instructions / s, chip-wide
                    valu = {"instructions_per_step": ips, "achieved_Ginstr_per_s": round(rate, 1),
                            "peak_nominal_Ginstr_per_s": VALU_PEAK_GINSTR, "frac_of_nominal_issue": round(rate / VALU_PEAK_GINSTR, 4),
                            "measured_mix_cycles_per_instruction_and_simd": VALU_MIX_CYCLES,
                            "frac_of_measured_mix_rate": round(rate / (SIMDS * SHADER_GHZ / VALU_MIX_CYCLES), 4),
                            "valu_floor_us_per_step_at_mix_rate": round(ips / (SIMDS * SHADER_GHZ / VALU_MIX_CYCLES) * 1e-3, 3),
                            "active_valu_cycles_per_step": vrec.get("active_valu_cycles_per_step"),
                            "source": {"file": "profiles/" + os.path.basename(vfiles[-1]), "commit": vrec.get("commit"),
                                       "kernel": vrec.get("kernel"), "note": "SQ_INSTS_VALU per launch / steps per launch from a rocprofv3 --pmc "
                                       "pass of this command; not re-counted in this run.  nominal = 2 cycles per wave64 instruction and SIMD; "
                                       "mix rate = profiles/r02_valu_probe.json (the noise mix at 4 waves per SIMD)"}}
                except Exception:  # noqa: BLE001
                    valu = None
            roofline = {"bound": "hbm",
                        "kernel": kname,
                        "achieved": krec["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": krec["frac"],
                        "peak_measured_copy": box_copy["GBps_median"] if box_copy else None,
                        "frac_of_measured_copy": round(krec["achieved"] / box_copy["GBps_median"], 4) if box_copy else None,
                        "measured_copy": box_copy, "valu": valu,
                        "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": round(krec["algorithmic_bytes_per_step"] * krec["steps_per_launch"], 1),
                        "algorithmic_bytes_per_step": krec["algorithmic_bytes_per_step"],
                        "avg_launch_us": krec["avg_launch_us"], "launches": krec["launches"],
                        "steps_per_launch": krec["steps_per_launch"], "kernel_us_per_step": krec["kernel_us_per_step"],
                        "co_bound": "between steps the cross-workgroup exchange (accumulator atomics -> arrival -> release -> "
                                    "update prologue); inside a step VALU issue (eps generation: threefry2x32-20 + erf_inv); "
                                    "DESIGN.md section 6",
                        "timing": "HIP start/stop events (hipExtLaunchKernel) around " + ("EVERY" if single else "every 16th") +
                                  " step-kernel launch of the timed region, on the launch stream "
                                  "(d3p_dpvi_logreg_kernel_timing_*); achieved = algorithmic bytes of the steps covered / "
                                  "summed kernel time"}

        del table, run, state, state0, svi
        torch.cuda.empty_cache()
        B_done = Bg // emu if emu else Bg   # (--emulate-world runs ONE rank's share: the examples this GPU processed, not the job's)
        result = None
        if rank == 0:
            result = {"value": round(B_done * steps_per_s, 1), "steps_per_sec": round(steps_per_s, 2),
                      "ms_per_step": round(1000.0 * elapsed / args.steps, 6), "final_loss": final_loss, "steady_state": steady,
                      "north_star_N1e7": None, "roofline": roofline, "rows": n_rows,
                      "leg_order": ("steady_state, headline, north_star_N1e7 (+ poisson_N1e7 on the same table), large_batch, gmm_config3, "
                                    "vae_config5, vae_config5_400_200" if extra_legs else "headline only (cold GPU)"),
                      "driver": "single-GPU chained launch" if single else (dist_driver if (native and comm is not None) else "torch")}
            if native:
                progress["m"] = result   # (what the watchdog prints should a LATER leg of the native loop stall)

        if native and os.environ.get("D3P_BENCH_INJECT_STOP"):   # developer switch: rehearse the "a later leg was stopped" exit
            raise RunStopped(f"rank {rank}: injected stop behind the headline leg")
        # ---------------------------------------------------------------- north_star: the same workload over N = 10^7 rows
        north = None
        if extra_legs and args.sampler == "feistel":
            n7 = 10_000_000
            svi7, st7, run7, table7, _ = make_workload(n7, native)
            st7, _, el7, kt7 = timed_leg(run7, st7, 0, 256, 2048)
            if rank == 0:
                sps = 2048 / el7
                north = {"rows": n7, "table_GB": round(n7 * (d + 1) * 4 / 1e9, 2), "steps": 2048, "warmup": 256,
                         "steps_per_sec": round(sps, 2), "value": round(Bg * sps, 1), "ms_per_step": round(1000.0 * el7 / 2048, 6),
                         "kernel": kernel_record(kt7, Bg // ranks),
                         "note": "north_star's table size (N = 10M rows, row-sharded over the ranks); same batch per GPU"}
            if single and rank == 0 and not args.no_aux_workloads:
                aux.update(aux_workloads(dev, table7, want=("poisson",)))
            del table7, run7, st7, svi7
            torch.cuda.empty_cache()

        # ---------------------------------------------------------------- the same kernel in the throughput regime
        # (context for `frac`: at batch 4096 a step is latency-bound by the cross-workgroup exchange; at batch 32768 = 8
        # examples per wave the kernel is bound by its own instruction issue)
        if roofline is not None and single and not args.no_large_batch and args.sampler == "feistel":
            Bl = 32768
            svi_l, st_l, _, (Xl, yl), bkey_l = make_workload(args.rows_per_gpu)
            _, gb_l = subsample_batchify_data((Xl, yl), Bl)
            st_l, _ = svi_l.run_steps(st_l, gb_l, bkey_l, 0, 384)   # (13 ms of warm-up: this leg follows table generation, see the VAE leg)
            torch.cuda.synchronize()
            L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(1))
            svi_l.run_steps(st_l, gb_l, bkey_l, 384, 320)
            torch.cuda.synchronize()
            L.check(lib.d3p_dpvi_logreg_kernel_timing_enable(0))
            us_l, n_l, steps_l = read_kernel_timing()
            if steps_l:
                ach = algorithmic_bytes(Bl, d, P) * steps_l / (us_l * 1e-6) / 1e9
                roofline["large_batch"] = {"batch": Bl, "achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBPS, 4),
                                           "kernel_us_per_step": round(us_l / steps_l, 3),
                                           "note": "same kernel and method, 320 steps at batch 32768 on one GPU (8 examples per "
                                                   "wave): the instruction-issue regime of the fused step with JAX-faithful noise"}
            del Xl, yl, gb_l, st_l, svi_l
        if single and rank == 0 and extra_legs and not args.no_aux_workloads:
            aux.update(aux_workloads(dev, None, want=("gmm", "vae", "vae2")))
        if (world > 1 or emu) and extra_legs and not args.no_aux_workloads and not aux:   # configs[4] at N > 1 (every rank takes part)
            try:
                aux.update(vae_dist_workload(dev, world, rank, barrier, share_gpu, emulate=emu))
            except Exception as e:  # noqa: BLE001 -- an auxiliary leg must not cost the headline line
                aux["vae_config5_dp_error"] = f"{type(e).__name__}: {e}"
        if rank != 0:
            return None
        result["north_star_N1e7"] = north
        return result

    def line(m, cpu=None, extra=None):
        out = {
            "metric": "DP-VI per-example grads/sec (logreg d=512 B=4096/GPU, AutoDiagonalNormal); steps/sec in steps_per_sec",
            "value": m["value"], "unit": "examples/s", "steps_per_sec": m["steps_per_sec"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: logistic regression d=512, 1e6 rows per GPU (fp32, HBM-resident), "
                                   "batch 4096 per GPU by " + ("Feistel subsampling w/o replacement" if args.sampler == "feistel" else
                                                          "Poisson sampling q = B/N padded to the 0.99 quantile") + ", AutoDiagonalNormal, "
                                   "C=1, sigma=1, Adam 1e-3",
                       "rows": m["rows"], "dim": d, "global_batch": Bg, "parallelism": f"dp{world}",
                       "collective": "none" if world == 1 else "1 sum-exchange per step of the rank's int64 fixed-point accumulator "
                                                               "(d3p_amd.dist); driver: " + m["driver"]},
            "final_loss": m["final_loss"], "steady_state": m["steady_state"], "north_star_N1e7": m["north_star_N1e7"],
            "leg_order": m.get("leg_order"),
            "roofline": m["roofline"], "cpu_baseline": cpu,
            "workloads": aux or None,
        }
        if collective_check is not None:
            out["collective_check"] = collective_check
            out["native_drivers"] = {"xchg_ran": bool(collective_check["xchg"].get("ran")), "xchg_ok": collective_check["xchg_ok"],
                                     "rccl_ran": bool(collective_check["rccl"].get("ran")), "rccl_ok": collective_check["rccl_ok"],
                                     "chosen": collective_check["driver_chosen"],
                                     "all_dropped": collective_check["driver_chosen"] == "torch"}
        if emu:
            out["note_emulate_world"] = (f"developer run: rank 0's share of an emulated {emu}-rank weak-scaling job on ONE GPU (exchange with "
                                         "itself); `value` counts the examples this GPU processed")
        if extra:
            out.update(extra)
        return json.dumps(out)

    extra = None
    # N > 1 with EVERY native driver dropped (the in-launch exchange and the native loop over RCCL both failed their first-contact
    # check, or could not be created): the only loop left is the Python-driven torch.distributed one.  Its number is printed -- marked --
    # and the process exits NON-ZERO, so that a torch-loop figure is never mistaken for the design's (D3P_BENCH_ALLOW_TORCH_ONLY=1, or
    # one of the developer switches that ask for that loop, turns the exit code off).
    torch_only = (world > 1 and comm is None and not os.environ.get("D3P_DIST_TWO_PHASE") and not os.environ.get("D3P_DIST_TORCH_LOOP")
                  and not os.environ.get("D3P_BENCH_ALLOW_TORCH_ONLY"))
    if torch_only and rank == 0:
        print("[bench] ERROR: every native data-parallel driver was dropped (collective_check above); the line below is the Python-driven "
              "torch.distributed loop and the exit code will be 3", file=sys.stderr, flush=True)
    if single or comm is None:
        m = measure(False, not args.no_extra_legs)
        if torch_only and rank == 0:
            extra = {"error": "every native data-parallel driver (d3p_xchg in-launch exchange, native loop over RCCL) was dropped by the "
                              "first-contact check or could not be created; `value` is the Python-driven torch.distributed loop, NOT the "
                              "design's data-parallel path; exit code 3"}
    else:
        # Data-parallel run with the native loop available.  The Python-driven loop (torch.distributed's own RCCL) is
        # measured FIRST and kept as the fallback line: should the native loop stall on some rank, a watchdog prints that
        # line and ends the process instead of leaving the driver without a number.
        fb = measure(False, False)
        import threading

        def give_up():
            if rank == 0:
                if progress.get("m") is not None:   # the native headline leg is done: a leg BEHIND it stalled
                    emit(line(progress["m"], extra={"note": "a leg behind the headline leg did not finish within the watchdog limit and was "
                                                             "cut; the headline figures are the native data-parallel loop's",
                                                     "torch_loop": {"steps_per_sec": fb["steps_per_sec"], "value": fb["value"]}}))
                else:
                    emit(line(fb, extra={"note": "native data-parallel loop did not finish within the watchdog limit; this is "
                                                  "the Python-driven torch.distributed loop"}))
            os._exit(0)
        dog = threading.Timer(float(os.environ.get("D3P_BENCH_WATCHDOG_S", "240")), give_up)
        dog.daemon = True
        dog.start()
        try:
            m = measure(True, not args.no_extra_legs)
        except RunStopped as e:   # every rank ends up here (a stopped rank sends no rows: its peers' waits run out too)
            print(f"[bench] {e}", file=sys.stderr, flush=True)
            if rank == 0:
                if progress.get("m") is not None:
                    emit(line(progress["m"], extra={"note": f"a leg behind the headline leg was stopped ({e}) and cut; the headline figures "
                                                             "are the native data-parallel loop's",
                                                     "torch_loop": {"steps_per_sec": fb["steps_per_sec"], "value": fb["value"]}}))
                else:
                    emit(line(fb, extra={"note": f"native data-parallel loop stopped ({e}); this is the Python-driven "
                                                  "torch.distributed loop"}))
            os._exit(0)
        barrier()
        dog.cancel()
        if rank == 0:
            extra = {"torch_loop": {"steps_per_sec": fb["steps_per_sec"], "value": fb["value"],
                                    "note": "same steps driven from Python through torch.distributed.all_reduce"}}

    if dist.is_initialized():
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
    elif comm is not None:
        comm.close()
    # the CPU baseline: rank 0 only, after every GPU leg and after the ranks have left the process group (the other ranks are
    # gone or idle: the host cores are rank 0's); the same bounded sample at every N -- per-GPU work is fixed (weak scaling)
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(d, args.batch_per_gpu, args.cpu_seconds, rows=200_000)
    if rank == 0:
        emit(line(m, cpu, extra))
    if torch_only and rank == 0:   # (rank 0 alone, and after its line: a launcher ends the other ranks when one exits non-zero)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)


if __name__ == "__main__":
    main()
