"""Data-parallel DP-VI over the GPUs of one node (one process per GPU, torch.distributed / RCCL).

The reference is single-device (no collectives anywhere); this module is the
build's own multi-GPU strategy:

  * the table is row-sharded contiguously, rank r holds rows [row_lo, row_hi);
  * every rank evaluates the SAME minibatch sampler with the SAME keys (Feistel indices / Poisson
    mask are functions of (key, i) only) and processes the examples whose rows it holds;
  * one sum-all-reduce per step of [clipped-gradient sum (P) | loss sum | example count];
  * the Gaussian noise is added ONCE, after the reduce, with the same perturbation key on every
    rank, so parameters stay replicated without a broadcast.

The compute is behind a small engine interface so that the orchestration (sharding arithmetic,
what is reduced, when noise is added) can be exercised on CPU with the gloo backend.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import BatchSource, check, ptr, stream_ptr


def _device_shard(t, what):
    """A rank's shard of the table as the kernels take it: a contiguous CUDA tensor (a host pointer must never reach a kernel)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.D3PError(f"{what}: the rank's shard must be a CUDA tensor")
    if t.is_floating_point() and t.dtype != torch.float32:
        raise _lib.D3PError(f"{what}: the rank's shard must be float32 (found {t.dtype}): the kernels read it in place")
    return t.contiguous()


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous row range [lo, hi) of `rank`; the first n_rows % world ranks hold one extra row."""
    base, rem = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _runs_beside(a, b, spin_cycles=20_000_000):
    """True if work on stream `b` completes while a spin kernel is still running on stream `a` (None = the current stream)."""
    cur = torch.cuda.current_stream()
    a = cur if a is None else a
    b = cur if b is None else b
    torch.cuda.synchronize()
    x = torch.zeros(8, device="cuda")
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(spin_cycles)
        ea.record()
    with torch.cuda.stream(b):
        x.add_(1.0)
        eb.record()
    eb.synchronize()
    beside = not ea.query()
    torch.cuda.synchronize()
    return beside


def concurrent_streams(n):
    """`n` streams whose kernels run BESIDE each other's and beside the current stream's (n <= 3 unless the process was started with
    GPU_MAX_HW_QUEUES > 4 in its environment: then up to that many - 1).  A HIP process has 4 hardware
    queues by default (GPU_MAX_HW_QUEUES); further streams share them, and two streams that share a queue are serialised -- every
    fourth stream of torch's pool shares the default stream's (tools/probes/stream_queue_probe.py).  Ranks that live in ONE process
    on separate streams (XchgComm.local_group; d3p_xchg_simulate_peers on a side stream) wait for each other INSIDE their launches:
    on two streams of one queue the second launch never starts and the bounded waits stop the run.  Streams are probed, not
    assumed: a spin kernel on one, a small kernel on the other."""
    import os
    try:
        queues = max(4, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
    except ValueError:
        queues = 4
    if n > queues - 1:
        raise _lib.D3PError(f"concurrent_streams: a process has {queues} hardware queues -- the current stream and at most {queues - 1} beside it")
    _runs_beside(None, None, 1000)   # (first use of the spin kernel)
    chosen = []
    for _ in range(16 * queues):
        if len(chosen) == n:
            return chosen
        s = torch.cuda.Stream()
        if any(s.cuda_stream == c.cuda_stream for c in chosen):
            continue
        if _runs_beside(None, s) and all(_runs_beside(c, s) for c in chosen):
            chosen.append(s)
    if len(chosen) == n:
        return chosen
    raise _lib.D3PError(f"concurrent_streams: found only {len(chosen)} of {n} streams that run beside the current stream")


class HipEngine:
    """local_sums / finalize through libd3p_hip.so for one rank's shard."""

    def __init__(self, svi, X_local, y_local, n_rows_global, row_lo, row_hi, kind, batch_size, q=0.0,
                 suppress=False, **model_kwargs):
        _lib.require_device()
        self.svi, self.X, self.y = svi, _device_shard(X_local, "HipEngine: X"), _device_shard(y_local, "HipEngine: y")
        assert self.X.shape[0] == row_hi - row_lo
        self.n, self.lo, self.hi = int(n_rows_global), int(row_lo), int(row_hi)
        self.kind, self.B, self.q, self.suppress = kind, int(batch_size), float(q), bool(suppress)
        self.model_kwargs = model_kwargs
        self.dev = self.X.device
        self.d = int(self.X.shape[1])
        self.P = 2 * svi.model.latent_dim(self.d)
        self.sums = torch.empty(self.P + 2, dtype=torch.float32, device=self.dev)

    def begin(self, state, batch_key, first_batch):
        self._setup(state, batch_key, first_batch)
        check(_lib.load().d3p_dpvi_logreg_begin(stream_ptr(), *self._args, ptr(self.ws), self.ws.numel()))

    def _setup(self, state, batch_key, first_batch, copy=True):
        """copy=False (run_steps_native): the new state's arrays are left EMPTY -- the native run copies the old state inside
        its first kernel (d3p_dpvi_logreg_run_dist_from) and takes the batch index by value: no launches here."""
        self.model = self.svi._model_struct(self.d, self.model_kwargs, state.observation_scale)
        self.hyper = self.svi._hyper()
        from .svi import _fresh_optim_state
        step0, params0, m0, v0 = state.optim_state
        self.svi._require_sizes(params0.numel(), self.P)
        self.svi._require_device_state(state)
        key0 = state.rng_key.reshape(16)
        n = params0.numel()
        self.keybuf = torch.empty((2, 16), dtype=torch.uint32, device=self.dev)
        from . import random as _rng
        self.bkey = _rng._key(batch_key)                  # (a 16-word CUDA key: anything else is a TypeError, not an address)
        self.frm = None
        if (not copy and params0.dtype == m0.dtype == v0.dtype == torch.float32 and m0.numel() == n and v0.numel() == n
                and params0.is_contiguous() and m0.is_contiguous() and v0.is_contiguous() and key0.is_contiguous()
                and key0.dtype == torch.uint32 and step0.dtype == torch.int32):
            flat = torch.empty(3 * n, dtype=torch.float32, device=self.dev)
            self.step, self.params, self.m, self.v = (torch.empty_like(step0), flat[:n].view_as(params0), flat[n:2 * n].view_as(m0),
                                                      flat[2 * n:].view_as(v0))
            self._from_keep = (step0, params0, m0, v0, key0)
            self.frm = self.svi._state_struct(key0, 0, (step0, params0, m0, v0))
            self.bidx = None
        else:
            self.step, self.params, self.m, self.v = _fresh_optim_state(state.optim_state)   # (one copy kernel for params, m, v)
            self.keybuf[0].copy_(key0)
            self.bidx = torch.full((1,), int(first_batch), dtype=torch.int32, device=self.dev)
        self.src = BatchSource(self.kind, self.B, self.q, int(self.suppress), self.bkey.data_ptr(),
                               self.bidx.data_ptr() if self.bidx is not None else None, None, self.n, self.lo, self.hi)
        lib = _lib.load()
        nbytes = lib.d3p_dpvi_logreg_workspace(C.byref(self.model), C.byref(self.src))
        if getattr(self, "ws", None) is None or self.ws.numel() < nbytes:   # kept across runs: per-run host time matters for short runs
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        self.loss = torch.empty(1, dtype=torch.float32, device=self.dev)
        self.observation_scale = state.observation_scale

        self.st = self.svi._state_struct(self.keybuf, 0, (self.step, self.params, self.m, self.v))
        self._args = (C.byref(self.model), C.byref(self.hyper), C.byref(self.st), C.byref(self.src))
        self.t = 0            # step inside the prepared batch
        self.prepared = 0     # steps the current batch holds
        self.done = 0
        self.remaining = None

    def _setup_lean(self, state, batch_key, first_batch, nl):
        """run_steps_native's set-up for a state the native run can copy by itself (contiguous fp32 arrays): ONE allocation for
        the new state, the losses, the step counter and the key slots -- [params n | m n | v n | losses nl | step | pad | 2 x 16
        key words] in 4-byte words --, raw addresses in the structs, no tensor views yet (_lean_views).  False: not applicable."""
        from ._lib import DpsviState
        step0, params0, m0, v0 = state.optim_state
        self.svi._require_sizes(params0.numel(), self.P)
        self.svi._require_device_state(state)
        key0 = state.rng_key.reshape(16)
        n = params0.numel()
        if not (params0.dtype == m0.dtype == v0.dtype == torch.float32 and m0.numel() == n and v0.numel() == n
                and params0.is_contiguous() and m0.is_contiguous() and v0.is_contiguous() and key0.is_contiguous()
                and key0.dtype == torch.uint32 and step0.dtype == torch.int32):
            return False
        self.model = self.svi._model_struct(self.d, self.model_kwargs, state.observation_scale)
        self.hyper = self.svi._hyper()
        from . import random as _rng
        self.bkey = _rng._key(batch_key)                  # (a 16-word CUDA key: anything else is a TypeError, not an address)
        koff = (3 * n + nl + 1 + 3) & ~3
        buf = torch.empty(koff + 32, dtype=torch.float32, device=self.dev)
        base = buf.data_ptr()
        self._lean = (buf, n, nl, koff, params0, m0, v0)
        self._losses_ptr = base + 12 * n
        self._from_keep = (step0, params0, m0, v0, key0)
        self.frm = DpsviState(key0.data_ptr(), 0, params0.data_ptr(), m0.data_ptr(), v0.data_ptr(), step0.data_ptr())
        self.st = DpsviState(base + 4 * koff, 0, base, base + 4 * n, base + 8 * n, base + 4 * (3 * n + nl))
        self.bidx = None
        self.src = BatchSource(self.kind, self.B, self.q, int(self.suppress), self.bkey.data_ptr(), None, None, self.n, self.lo, self.hi)
        nbytes = _lib.load().d3p_dpvi_logreg_workspace(C.byref(self.model), C.byref(self.src))
        if getattr(self, "ws", None) is None or self.ws.numel() < nbytes:   # kept across runs
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        self.observation_scale = state.observation_scale
        self._args = (C.byref(self.model), C.byref(self.hyper), C.byref(self.st), C.byref(self.src))
        self.t = self.prepared = self.done = 0
        self.remaining = None
        return True

    def _lean_views(self):
        """The tensors of _setup_lean's allocation (made after the run has been enqueued); returns the losses."""
        buf, n, nl, koff, params0, m0, v0 = self._lean
        params, m, v, losses, tail = buf.split((n, n, n, nl, koff + 32 - 3 * n - nl))
        self.params, self.m, self.v = params.view_as(params0), m.view_as(m0), v.view_as(v0)
        self.step = tail[0].view(torch.int32)
        self.keybuf = tail[koff - 3 * n - nl:].view(torch.uint32).view(2, 16)
        self.loss = None
        return losses

    def chain_grid(self, data_parallel):
        """(workgroups per step, waves per workgroup) of the chained launch the native run loop takes for the shape of the last
        run (d3p_dpvi_logreg_chain_grid; (0, 0): one launch per step).  For tests and measurements: nothing is launched."""
        nw, waves = C.c_uint32(), C.c_int32()
        check(_lib.load().d3p_dpvi_logreg_chain_grid(C.byref(self.model), C.byref(self.src), int(bool(data_parallel)), C.byref(nw),
                                                     C.byref(waves)))
        return int(nw.value), int(waves.value)

    STEP_BATCH = 32

    def plan(self, num_steps):
        """Tell the engine how many steps will follow so that it prepares keys/indices in batches."""
        self.remaining = int(num_steps)

    def _ensure_prepared(self):
        if self.t >= self.prepared:
            k = self.STEP_BATCH if self.remaining is None else max(1, min(self.STEP_BATCH, self.remaining))
            check(_lib.load().d3p_dpvi_logreg_prepare(stream_ptr(), *self._args, k, ptr(self.ws), self.ws.numel()))
            self.prepared, self.t = k, 0

    def local_sums(self):
        self._ensure_prepared()
        check(_lib.load().d3p_dpvi_logreg_step_sums(stream_ptr(), *self._args, self.t, ptr(self.X), ptr(self.y), None,
                                                    ptr(self.sums), ptr(self.ws), self.ws.numel()))
        return self.sums

    def finalize(self, sums):
        check(_lib.load().d3p_dpvi_logreg_step_finalize(stream_ptr(), *self._args, self.t, ptr(sums), ptr(self.loss),
                                                        None, ptr(self.ws), self.ws.numel()))
        self.t += 1
        self.done += 1
        if self.remaining is not None:
            self.remaining -= 1
        return self.loss

    def end(self):
        from .svi import DPSVIState
        # NOTE: the key chain ran ahead to the end of the prepared batch; `end` is only exact when every
        # prepared step was consumed (run_steps plans the batches so that this holds).
        assert self.t == self.prepared or self.done == 0, "unconsumed prepared steps"
        check(_lib.load().d3p_dpvi_logreg_end(stream_ptr(), *self._args, self.done, ptr(self.ws), self.ws.numel()))
        return DPSVIState((self.step, self.params, self.m, self.v), self.keybuf[self.done & 1].reshape(4, 4).clone(),
                          self.observation_scale)


class FusedHipEngine(HipEngine):
    """One kernel launch + one int64 all-reduce per step: the rank's clipped sums go into a fixed-point
    accumulator (exact integer sums: every rank obtains bitwise identical totals under any reduction
    order) and the update of step g is applied in the prologue of launch g+1."""

    def begin(self, state, batch_key, first_batch):
        super().begin(state, batch_key, first_batch)
        lib = _lib.load()
        # rows too wide for the register-tiled kernel have no one-launch step: the engine then IS the two-kernel engine (float sums
        # instead of the int64 accumulator; callers sum whatever local_sums hands them)
        self.two_kernel = not lib.d3p_dpvi_logreg_fused_step_supported(C.byref(self.model), C.byref(self.src))
        self.losses = None
        if self.two_kernel:
            self._loss_log = []
            return
        off, words = C.c_size_t(), C.c_size_t()
        check(lib.d3p_dpvi_logreg_acc_layout(C.byref(self.model), C.byref(self.src), C.byref(off), C.byref(words)))
        self.acc = self.ws[off.value: off.value + 3 * words.value * 8].view(torch.int64).reshape(3, words.value)
        check(lib.d3p_dpvi_logreg_acc_reset(stream_ptr(), *self._args, ptr(self.ws), self.ws.numel()))
        self.g = 0
        self.buf = 1            # flipped before the first prepare
        self.prev = None        # (t, buf) of the previous step
        self.losses = None

    def plan(self, num_steps):
        super().plan(num_steps)
        self.losses = torch.zeros(max(int(num_steps), 1), dtype=torch.float32, device=self.dev)

    def _ensure_prepared(self):
        if self.two_kernel:
            return super()._ensure_prepared()
        if self.t >= self.prepared:
            k = self.STEP_BATCH if self.remaining is None else max(1, min(self.STEP_BATCH, self.remaining))
            self.buf ^= 1
            check(_lib.load().d3p_dpvi_logreg_prepare_buf(stream_ptr(), *self._args, k, self.buf, ptr(self.ws),
                                                          self.ws.numel()))
            self.prepared, self.t = k, 0

    def _launch(self, flush):
        have_prev = self.prev is not None
        pt, pb = self.prev if have_prev else (0, 0)
        loss_ptr = None
        if have_prev and self.losses is not None and self.g - 1 < self.losses.numel():
            loss_ptr = C.c_void_p(self.losses.data_ptr() + 4 * (self.g - 1))
        check(_lib.load().d3p_dpvi_logreg_fused_step(stream_ptr(), *self._args, self.g, self.t if not flush else 0,
                                                     self.buf, int(have_prev), pt, pb, ptr(self.X), ptr(self.y), loss_ptr,
                                                     int(flush), ptr(self.ws), self.ws.numel()))

    def local_sums(self):
        if self.two_kernel:
            return super().local_sums()
        self._ensure_prepared()
        self._launch(False)
        return self.acc[self.g % 3]

    def finalize(self, sums):
        if self.two_kernel:
            loss = super().finalize(sums)
            if self.losses is not None and self.done - 1 < self.losses.numel():
                self.losses[self.done - 1].copy_(loss.reshape(()))
            return None
        self.prev = (self.t, self.buf)
        self.g += 1
        self.t += 1
        self.done += 1
        if self.remaining is not None:
            self.remaining -= 1
        return None            # the loss of this step is written by the next launch

    def end(self):
        if not self.two_kernel and self.prev is not None:
            self._launch(True)  # apply the update of the last step
        return super().end()


class NativeComm:
    """RCCL communicator owned by libd3p_hip.so (d3p_comm_*): rank 0 draws the id, torch.distributed carries it to the
    other ranks, every rank joins.  Used by the native data-parallel loop, which then needs no Python per step."""

    def __init__(self, group=None):
        import torch.distributed as dist
        _lib.require_device()
        lib = _lib.load()
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        buf = (C.c_uint8 * 128)()
        err = None
        if self.rank == 0 and lib.d3p_comm_unique_id(buf, 128) != 0:
            err = (lib.d3p_last_error() or b"").decode(errors="replace") or "d3p_comm_unique_id failed"
        box = [(bytes(buf), err)]      # (rank 0's failure travels with the id: the other ranks are not left waiting in the broadcast)
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        if box[0][1]:
            raise _lib.D3PError("RCCL communicator not created: rank 0: " + box[0][1])
        ident = (C.c_uint8 * 128).from_buffer_copy(box[0][0])
        handle = C.c_void_p()
        check(lib.d3p_comm_init(ident, 128, self.world, self.rank, C.byref(handle)))
        self.handle = handle

    def close(self):
        if self.handle:
            check(_lib.load().d3p_comm_destroy(self.handle))
            self.handle = None


IPC_HANDLE_BYTES = 80   # what a communicator hands to its peers: the 64-byte hipIpc handle of the process's arena (or of the inbox's own
                        # allocation) + the inbox's offset + a marker (csrc/d3p_ipc_arena.h)


def _create_and_gather(comm, create, destroy, group):
    """First half of XchgComm / FMeshComm.__init__: create the rank's side (its inbox and the 80-byte handle of it) and gather
    the handles of all ranks -- with the ranks' SUCCESS gathered beside them, so that a rank whose creation failed does not leave its
    peers waiting in the gather: every rank raises D3PError together (and gives back what it made), and a caller that falls back to
    another driver (bench.py) does so on every rank.  Returns the handles in rank order, or None for a single-process group."""
    import torch.distributed as dist
    handle, buf = C.c_void_p(), (C.c_uint8 * IPC_HANDLE_BYTES)()
    rc = create(handle, buf)
    err = None if rc == 0 else ((_lib.load().d3p_last_error() or b"").decode(errors="replace") or f"error {rc}")
    if comm.local:
        if err:
            raise _lib.D3PError("libd3p_hip: " + err)
        comm.handle = handle
        return None
    mine = (bytes(buf) if err is None else None, err)
    if comm.world > 1:
        box = [None] * comm.world
        dist.all_gather_object(box, mine, group=group)
    else:
        box = [mine]
    failed = [(r, e) for r, (_, e) in enumerate(box) if e]
    if failed:
        if err is None:
            destroy(handle)
        raise _lib.D3PError("communicator not created: " + "; ".join(f"rank {r}: {e}" for r, e in failed))
    comm.handle = handle
    return [h for h, _ in box]


def _connect_and_agree(comm, rc, destroy, group):
    """Second half of XchgComm / FMeshComm.__init__: the ranks meet after mapping their peers' inboxes -- nobody writes into an inbox
    that its owner has not finished setting up -- and they meet with the OUTCOME of the mapping: a rank that could not map a peer
    does not leave the others behind a barrier, every rank raises D3PError together and gives its side back."""
    import torch.distributed as dist
    err = None if rc == 0 else ((_lib.load().d3p_last_error() or b"").decode(errors="replace") or f"error {rc}")
    if comm.world > 1:
        box = [None] * comm.world
        dist.all_gather_object(box, err, group=group)
    else:
        box = [err]
    failed = [(r, e) for r, e in enumerate(box) if e]
    if failed:
        destroy(comm.handle)
        comm.handle = None
        raise _lib.D3PError("communicator not connected: " + "; ".join(f"rank {r}: {e}" for r, e in failed))


def _teardown_barrier(comm, collective, disconnect):
    """First half of XchgComm / FMeshComm.close: every rank unmaps its peers' inboxes, then the ranks meet, and only then does anybody
    free.  The barrier is skipped for single-process groups, one-rank jobs, and on request."""
    import torch.distributed as dist
    check(disconnect(comm.handle))
    if collective is None:
        collective = not comm.local and comm.world > 1
    if collective and dist.is_available() and dist.is_initialized():
        dist.barrier(group=getattr(comm, "group", None))


class XchgComm:
    """One-shot full-mesh exchange owned by libd3p_hip.so (d3p_xchg_*): every rank's inbox is mapped into its peers with
    hipIpc handles (gathered here through torch.distributed, any backend), and the step's collective becomes ONE kernel
    that writes the rank's folded 8 KB accumulator row into the 7 peers over xGMI and adds the 8 rows locally -- one hop
    instead of a ring's 14.  `words` = int64 words of the message: 2 D + 4 for a model with D latents."""

    def __init__(self, words, group=None, _local=None):
        import torch.distributed as dist
        _lib.require_device()
        lib = _lib.load()
        if _local is not None:                      # (rank, world) of a single-process group: see local_group
            self.rank, self.world = _local
        else:
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.words = int(words)
        self.group = group
        self.handle = None
        self.local = _local is not None   # (ranks of ONE process: a rank's run may only be waited for once every rank's is enqueued)
        box = _create_and_gather(self, lambda h, buf: lib.d3p_xchg_create(self.world, self.rank, self.words, C.byref(h), buf, IPC_HANDLE_BYTES),
                                 lib.d3p_xchg_destroy, group)
        if box is None:
            return
        allh = (C.c_uint8 * (IPC_HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(box))
        _connect_and_agree(self, lib.d3p_xchg_connect(self.handle, allh, IPC_HANDLE_BYTES), lib.d3p_xchg_destroy, group)

    @classmethod
    def local_group(cls, world, words):
        """`world` ranks inside ONE process (each drives its own stream): the inboxes are wired to each other directly
        (d3p_xchg_connect_local) instead of through hipIpc handles; kernels and protocol are those of the multi-process
        path.  For tests on a single GPU and for single-process multi-stream use.  The ranks wait for each other inside their
        launches, so their streams must not share a hardware queue: take them from `concurrent_streams`."""
        comms = [cls(words, _local=(r, world)) for r in range(world)]
        arr = (C.c_void_p * world)(*[c.handle for c in comms])
        for c in comms:
            check(_lib.load().d3p_xchg_connect_local(c.handle, arr, world))
        return comms

    def allreduce(self, acc, replicas):
        """acc: int64 tensor [replicas, words] (this rank's accumulator replicas) -> row 0 = global sums, other rows 0."""
        check(_lib.load().d3p_xchg_allreduce(stream_ptr(), self.handle, ptr(acc), int(replicas)))

    def simulate_peers(self, num_exchanges):
        """Test / rehearsal helper (d3p_xchg_simulate_peers): enqueue, on the CURRENT stream, one workgroup that plays the other
        world - 1 ranks for the next `num_exchanges` exchanges (all-zero rows, protocol-faithful).  Enqueue it on another stream
        than the run, before the run."""
        check(_lib.load().d3p_xchg_simulate_peers(stream_ptr(), self.handle, int(num_exchanges)))

    def close(self, collective=None):
        """Teardown.  With the ranks in separate processes it is a COLLECTIVE (every rank calls it): unmap the peers' inboxes, meet in a
        barrier, then free the own inbox -- a rank that frees its inbox while a peer still has it mapped breaks its own next
        hipIpcGetMemHandle (dmabuf IPC: "invalid argument" on the next communicator).  As with any sequence of collectives, the ranks
        close their communicators in the SAME order (the barrier is anonymous: two ranks closing two communicators in opposite orders
        would pair the wrong halves -- tools/soak_teardown.py's first version did, and met the old failure again).
        `collective=False`: no barrier (a rank that must give its communicator up ALONE, e.g. because a peer failed to create one)."""
        if self.handle:
            torch.cuda.synchronize()
            _teardown_barrier(self, collective, _lib.load().d3p_xchg_disconnect)
            check(_lib.load().d3p_xchg_destroy(self.handle))
            self.handle = None


class FMeshComm:
    """Full-mesh sum-all-reduce of a float vector owned by libd3p_hip.so (d3p_fmesh_*, csrc/d3p_fmesh.hip): reduce-scatter + all-gather
    over the peers' hipIpc-mapped inboxes, two hops with every xGMI link carrying 1 / world of the vector, rank-order sums (bitwise
    identical results on every rank).  The collective of the data-parallel VAE step (`vae_run_steps(..., comm=FMeshComm(P + 2))`).
    `n_floats`: the length of the vector (P + 2 for a model with P parameters).  Handles travel through torch.distributed, any
    backend; `local_group` wires ranks that live in one process (tests: their streams from `concurrent_streams`)."""

    def __init__(self, n_floats, group=None, _local=None):
        import torch.distributed as dist
        _lib.require_device()
        lib = _lib.load()
        if _local is not None:
            self.rank, self.world = _local
        else:
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n = int(n_floats)
        self.group = group
        self.handle = None
        self.local = _local is not None   # (ranks of ONE process: a rank's run may only be waited for once every rank's is enqueued)
        box = _create_and_gather(self, lambda h, buf: lib.d3p_fmesh_create(self.world, self.rank, self.n, C.byref(h), buf, IPC_HANDLE_BYTES),
                                 lib.d3p_fmesh_destroy, group)
        if box is None:
            return
        allh = (C.c_uint8 * (IPC_HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(box))
        _connect_and_agree(self, lib.d3p_fmesh_connect(self.handle, allh, IPC_HANDLE_BYTES), lib.d3p_fmesh_destroy, group)

    @classmethod
    def local_group(cls, world, n_floats):
        comms = [cls(n_floats, _local=(r, world)) for r in range(world)]
        arr = (C.c_void_p * world)(*[c.handle for c in comms])
        for c in comms:
            check(_lib.load().d3p_fmesh_connect_local(c.handle, arr, world))
        return comms

    def set_grid(self, workgroups):
        """Workgroups per launch (default one per CU).  Ranks that SHARE a GPU must leave room for each other's kernels: 48."""
        check(_lib.load().d3p_fmesh_set_grid(self.handle, int(workgroups)))

    def allreduce(self, vec):
        """In-place sum over the ranks of a contiguous float32 tensor of n_floats elements, on the current stream."""
        assert vec.dtype == torch.float32 and vec.is_contiguous() and vec.numel() == self.n
        check(_lib.load().d3p_fmesh_allreduce(stream_ptr(), self.handle, ptr(vec), self.n))

    def stopped(self):
        """True if a bounded wait of an all-reduce on this mesh ran out (after synchronising the current stream)."""
        w = C.c_int32(0)
        check(_lib.load().d3p_fmesh_status(stream_ptr(), self.handle, C.byref(w)))
        return bool(w.value)

    def close(self, collective=None):
        """Teardown; a COLLECTIVE when the ranks are separate processes (see XchgComm.close): unmap the peers, barrier, free."""
        if self.handle:
            torch.cuda.synchronize()
            _teardown_barrier(self, collective, _lib.load().d3p_fmesh_disconnect)
            check(_lib.load().d3p_fmesh_destroy(self.handle))
            self.handle = None


def run_steps_native(engine, state, batch_key, first_batch, num_steps, comm=None, collect_losses=True):
    """The whole data-parallel run in one C call: per step one kernel launch and ONE collective of the int64 accumulator
    on the same stream, no host work between steps.  `comm`: a NativeComm (RCCL ring all-reduce, d3p_dpvi_logreg_run_dist),
    an XchgComm (one-shot full-mesh exchange, d3p_dpvi_logreg_run_xchg) or None (single rank).
    `engine` is a FusedHipEngine (it supplies the shard and the model)."""
    from .svi import DPSVIState
    if engine._setup_lean(state, batch_key, first_batch, max(int(num_steps), 1)):
        # (everything the run writes is one allocation handed over as raw addresses; the views are made after the enqueue,
        # while the device is already running -- what the host does in front of the first launch is wall time of a short run)
        is_x = isinstance(comm, XchgComm)
        check(_lib.load().d3p_dpvi_logreg_run_dist_from(
            stream_ptr(), comm.handle if (comm is not None and not is_x) else None, comm.handle if is_x else None,
            engine._args[0], engine._args[1], engine._args[2], C.byref(engine.frm), engine._args[3], int(first_batch),
            ptr(engine.X), ptr(engine.y), int(num_steps), engine._losses_ptr if collect_losses else None, ptr(engine.ws),
            engine.ws.numel()))
        losses = engine._lean_views()
        new_state = DPSVIState((engine.step, engine.params, engine.m, engine.v),
                               engine.keybuf[int(num_steps) & 1].reshape(4, 4), engine.observation_scale)
        return new_state, (losses[:int(num_steps)] if collect_losses else None)
    engine._setup(state, batch_key, first_batch, copy=False)
    losses = torch.zeros(max(int(num_steps), 1), dtype=torch.float32, device=engine.dev) if collect_losses else None
    if engine.frm is not None:
        is_x = isinstance(comm, XchgComm)
        check(_lib.load().d3p_dpvi_logreg_run_dist_from(
            stream_ptr(), comm.handle if (comm is not None and not is_x) else None, comm.handle if is_x else None,
            engine._args[0], engine._args[1], engine._args[2], C.byref(engine.frm), engine._args[3], int(first_batch),
            ptr(engine.X), ptr(engine.y), int(num_steps), ptr(losses), ptr(engine.ws), engine.ws.numel()))
    elif isinstance(comm, XchgComm):
        check(_lib.load().d3p_dpvi_logreg_run_xchg(stream_ptr(), comm.handle, *engine._args, ptr(engine.X), ptr(engine.y),
                                                   int(num_steps), ptr(losses), ptr(engine.ws), engine.ws.numel()))
    else:
        check(_lib.load().d3p_dpvi_logreg_run_dist(stream_ptr(), comm.handle if comm is not None else None, *engine._args,
                                                   ptr(engine.X), ptr(engine.y), int(num_steps), ptr(losses), ptr(engine.ws),
                                                   engine.ws.numel()))
    new_state = DPSVIState((engine.step, engine.params, engine.m, engine.v),
                           engine.keybuf[int(num_steps) & 1].reshape(4, 4), engine.observation_scale)   # (keybuf is this run's own)
    return new_state, (losses[:int(num_steps)] if collect_losses else None)


def native_run_status(engine):
    """(abort code, nonfinite) of the engine's last ``run_steps_native`` after synchronising the stream; the code is 0 or names
    the bounded wait that stopped the run (``_lib.describe_abort``)."""
    aborted, nonfinite = C.c_int32(0), C.c_int32(0)
    check(_lib.load().d3p_dpvi_logreg_run_status(stream_ptr(), engine._args[0], engine._args[3], ptr(engine.ws), engine.ws.numel(),
                                                 C.byref(aborted), C.byref(nonfinite)))
    return int(aborted.value) & 0xffffffff, bool(nonfinite.value)


def run_steps(engine, state, batch_key, first_batch, num_steps, group=None, collect_losses=True):
    """num_steps x [local_sums -> all_reduce(SUM) -> finalize] on every rank of `group`.

    With world size 1 the all-reduce is skipped.  Returns (new_state, losses)."""
    import torch.distributed as dist
    import os
    initialised = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if initialised else 1
    force = initialised and os.environ.get("D3P_FORCE_ALLREDUCE") is not None  # plumbing test on one rank
    engine.begin(state, batch_key, first_batch)
    if hasattr(engine, "plan"):
        engine.plan(num_steps)
    losses = []
    for _ in range(int(num_steps)):
        sums = engine.local_sums()
        if world > 1 or force:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)   # the ONLY data-path collective
        loss = engine.finalize(sums)                                   # noise added once, after the reduce
        if collect_losses and loss is not None:
            losses.append(loss.clone())
    new_state = engine.end()
    if getattr(engine, "losses", None) is not None and collect_losses:
        return new_state, engine.losses[:int(num_steps)]
    return new_state, (torch.stack(losses).reshape(-1) if losses else None)


# ------------------------------------------------------------------------------------------------------------------
# VAE (BASELINE config 5, "1 vs 8 GPU"): the dataset (MNIST-sized) is replicated, the BATCH is sharded by position
# ------------------------------------------------------------------------------------------------------------------
def shard_batch(batch_size: int, rank: int, world: int):
    """(pos0, B_local): the contiguous batch positions of `rank` (the first batch_size % world ranks hold one more)."""
    lo, hi = shard_rows(batch_size, rank, world)
    return lo, hi - lo


class VaeHipEngine:
    """local_sums / apply of one rank through libd3p_hip.so (d3p_dpvi_vae_local_sums / d3p_dpvi_vae_apply)."""

    def __init__(self, svi, **model_kwargs):
        _lib.require_device()
        if not svi._is_vae():
            raise _lib.D3PError("VaeHipEngine: the DPSVI object must hold a VAEModel / VAEGuide pair")
        self.svi, self.model_kwargs = svi, model_kwargs

    def begin(self, state, X_local, batch_size_total, pos0, mask=None, eps=None):
        svi, lib = self.svi, _lib.load()
        self.X = svi._vae_flat(_device_shard(X_local, "VaeHipEngine: X"))
        self.B_local, D = self.X.shape
        self.B_total, self.pos0 = int(batch_size_total), int(pos0)
        dev = self.X.device
        self.vm = svi._vae_struct(D, self.model_kwargs, state.observation_scale)
        self.hyper = svi._hyper()
        svi._require_device_state(state)
        svi._require_sizes(state.optim_state[1].numel(), lib.d3p_vae_num_params(C.byref(self.vm)))
        self.step, self.params, self.m, self.v = (t.clone() for t in state.optim_state)
        self.keybuf = torch.empty((2, 16), dtype=torch.uint32, device=dev)
        self.keybuf[0].copy_(state.rng_key.reshape(16))
        self.st = svi._state_struct(self.keybuf, 0, (self.step, self.params, self.m, self.v))
        if mask is not None and mask.numel() != self.B_local:
            raise ValueError(f"mask: {mask.numel()} entries for a shard of {self.B_local} examples")
        self.mask = None if mask is None else _device_shard(mask, "mask").to(torch.uint8).contiguous()
        self.eps = eps
        self.ws = svi._workspace(lib.d3p_dpvi_vae_workspace(C.byref(self.vm), self.B_local), dev, "vae_step")
        P = int(lib.d3p_vae_num_params(C.byref(self.vm)))
        self.sums = torch.empty(P + 2, dtype=torch.float32, device=dev)
        self.loss = torch.empty(1, dtype=torch.float32, device=dev)
        self.observation_scale = state.observation_scale

    def local_sums(self):
        check(_lib.load().d3p_dpvi_vae_local_sums(
            stream_ptr(), C.byref(self.vm), C.byref(self.hyper), C.byref(self.st), ptr(self.X), ptr(self.mask),
            self.B_local, self.B_total, self.pos0, ptr(self.eps), ptr(self.sums), ptr(self.ws), self.ws.numel()))
        return self.sums

    def apply(self, sums, grad_out=None):
        from .svi import DPSVIState
        check(_lib.load().d3p_dpvi_vae_apply(
            stream_ptr(), C.byref(self.vm), C.byref(self.hyper), C.byref(self.st), ptr(sums), self.B_total, self.B_local,
            ptr(self.loss), ptr(grad_out), ptr(self.ws), self.ws.numel()))
        new_state = DPSVIState((self.step, self.params, self.m, self.v), self.keybuf[1].reshape(4, 4),
                               self.observation_scale)
        return new_state, self.loss[0]


def vae_run_steps(engine, state, X_local, batch_size_total, pos0, num_steps, group=None, collect_losses=True, comm=None, buckets=0,
                  check_status=None, mask=None):
    """`num_steps` data-parallel VAE updates on the SAME resident batch shard (the epoch body of examples/vae.py:227-246 with the
    batch sharded by position): per step local sums -> ONE sum-all-reduce of the P + 2 sums -> apply (noise once, after the
    reduce, identical on every rank), the state advancing in the engine's own buffers.
    `comm`: a NativeComm -- the whole run is ONE C call (d3p_dpvi_vae_run_dist): nothing happens on the host between steps, the
    reduce is RCCL's on the library's communicator, in two buckets on a second stream (`buckets` = 2; 1: one all-reduce in the
    stream; 0: the library's choice) so that the decoder's sums travel while the encoder's weight-gradient products run; a
    FMeshComm: the same call with the full-mesh reduce-scatter + all-gather of d3p_fmesh.hip as the step's collective; with
    `comm="local"` the same call without a collective (one rank).  comm=None: the Python-driven loop over `group`
    (torch.distributed.all_reduce; any backend).  Returns (new_state, losses[num_steps] or None).
    `check_status` (default: on for a FMeshComm whose ranks are processes; off for `FMeshComm.local_group` -- ranks of one process enqueue
    one after the other, so a rank's run can only be waited for once every rank's is enqueued: check `comm.stopped()` then): after
    the run read the mesh's status word (a device synchronisation) and raise D3PError when a bounded wait ran out -- the state is then
    partly updated in place and must not be used or timed.
    `mask`: validity of the shard's examples (`update(..., mask=)`, svi.py:395; masked examples contribute nothing, the noise carries
    B / n)."""
    import torch.distributed as dist
    from .svi import DPSVIState
    engine.begin(state, X_local, batch_size_total, pos0, mask=mask)
    losses = torch.empty(int(num_steps), dtype=torch.float32, device=engine.X.device) if collect_losses else None
    if comm is not None:
        rccl = comm.handle if isinstance(comm, NativeComm) else None
        mesh = comm.handle if isinstance(comm, FMeshComm) else None
        check(_lib.load().d3p_dpvi_vae_run_dist(
            stream_ptr(), rccl, mesh, C.byref(engine.vm), C.byref(engine.hyper), C.byref(engine.st), ptr(engine.X), ptr(engine.mask),
            engine.B_local, engine.B_total, engine.pos0, int(num_steps), ptr(losses), int(buckets), ptr(engine.ws), engine.ws.numel()))
        if mesh is not None and (check_status if check_status is not None else not comm.local):
            if comm.stopped():
                raise _lib.D3PError("vae_run_steps: the full-mesh collective was stopped by a bounded wait (d3p_fmesh_status); "
                                    "parameters and moments are partly updated")
        new_state = DPSVIState((engine.step, engine.params, engine.m, engine.v), engine.keybuf[int(num_steps) & 1].reshape(4, 4).clone(),
                               engine.observation_scale)
        return new_state, losses
    initialised = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if initialised else 1
    new_state = state
    for t in range(int(num_steps)):
        sums = engine.local_sums()
        if world > 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)   # the ONLY data-path collective (2.76 MB for 784-400-50)
        new_state, loss = engine.apply(sums)
        if losses is not None:
            losses[t].copy_(loss)
        engine.keybuf[0].copy_(engine.keybuf[1])   # the key after this step is the key before the next one
    return new_state, losses


class GmmHipEngine:
    """local_sums / apply of one rank for the mixture model (d3p_dpvi_gmm_local_sums / d3p_dpvi_gmm_apply)."""

    def __init__(self, svi, **model_kwargs):
        _lib.require_device()
        if not svi._is_gmm():
            raise _lib.D3PError("GmmHipEngine: the DPSVI object must hold a GaussianMixtureModel / GaussianMixtureGuide pair")
        self.svi, self.model_kwargs = svi, model_kwargs

    def begin(self, state, X_local, batch_size_total, pos0, mask=None, eps=None):
        svi, lib = self.svi, _lib.load()
        if eps is not None:
            raise _lib.D3PError("GmmHipEngine: the mixture model has no external-noise mode")
        self.X = _device_shard(X_local, "GmmHipEngine: X")
        self.B_local, d = self.X.shape
        self.B_total, self.pos0 = int(batch_size_total), int(pos0)
        dev = self.X.device
        self.gm = svi._gmm_struct(d, self.model_kwargs, state.observation_scale)
        self.hyper = svi._hyper()
        svi._require_device_state(state)
        svi._require_sizes(state.optim_state[1].numel(), self.gm.K + self.gm.K * d)
        self.step, self.params, self.m, self.v = (t.clone() for t in state.optim_state)
        self.keybuf = torch.empty((2, 16), dtype=torch.uint32, device=dev)
        self.keybuf[0].copy_(state.rng_key.reshape(16))
        self.st = svi._state_struct(self.keybuf, 0, (self.step, self.params, self.m, self.v))
        if mask is not None and mask.numel() != self.B_local:
            raise ValueError(f"mask: {mask.numel()} entries for a shard of {self.B_local} examples")
        self.mask = None if mask is None else _device_shard(mask, "mask").to(torch.uint8).contiguous()
        self.ws = svi._workspace(lib.d3p_dpvi_gmm_workspace(C.byref(self.gm), self.B_local), dev, "gmm_step")
        self.sums = torch.empty(self.params.numel() + 2, dtype=torch.float32, device=dev)
        self.loss = torch.empty(1, dtype=torch.float32, device=dev)
        self.observation_scale = state.observation_scale

    def local_sums(self):
        check(_lib.load().d3p_dpvi_gmm_local_sums(
            stream_ptr(), C.byref(self.gm), C.byref(self.hyper), C.byref(self.st), ptr(self.X), ptr(self.mask),
            self.B_local, self.B_total, self.pos0, ptr(self.sums), ptr(self.ws), self.ws.numel()))
        return self.sums

    def apply(self, sums, grad_out=None):
        from .svi import DPSVIState
        check(_lib.load().d3p_dpvi_gmm_apply(
            stream_ptr(), C.byref(self.gm), C.byref(self.hyper), C.byref(self.st), ptr(sums), self.B_total, self.B_local,
            ptr(self.loss), ptr(grad_out), ptr(self.ws), self.ws.numel()))
        new_state = DPSVIState((self.step, self.params, self.m, self.v), self.keybuf[1].reshape(4, 4),
                               self.observation_scale)
        return new_state, self.loss[0]


def sharded_batch_update(engine, state, X_local, batch_size_total, pos0, group=None, mask=None, _eps=None, _grad_out=None):
    """One data-parallel DPSVI.update (svi.py:395-434) with the BATCH sharded by position over the ranks of `group` (the VAE
    and mixture-model engines above; the dataset is replicated or the rank holds the rows of its positions)."""
    return vae_update(engine, state, X_local, batch_size_total, pos0, group=group, mask=mask, _eps=_eps, _grad_out=_grad_out)


def vae_update(engine, state, X_local, batch_size_total, pos0, group=None, mask=None, _eps=None, _grad_out=None):
    """One data-parallel DPSVI.update (svi.py:395-434) of the VAE on every rank of `group`: the rank's examples are the
    positions pos0 .. pos0 + len(X_local) - 1 of the global batch.  local sums -> ONE all_reduce(SUM) of the P + 2 sums
    (skipped with world size 1) -> apply (noise once, after the reduce, identical on every rank).  Returns (state, loss)."""
    import torch.distributed as dist
    initialised = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if initialised else 1
    engine.begin(state, X_local, batch_size_total, pos0, mask=mask, eps=_eps)
    sums = engine.local_sums()
    if world > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)   # the ONLY data-path collective
    return engine.apply(sums, _grad_out) if _grad_out is not None else engine.apply(sums)
