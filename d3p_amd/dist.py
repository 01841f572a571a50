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


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous row range [lo, hi) of `rank`; the first n_rows % world ranks hold one extra row."""
    base, rem = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HipEngine:
    """local_sums / finalize through libd3p_hip.so for one rank's shard."""

    def __init__(self, svi, X_local, y_local, n_rows_global, row_lo, row_hi, kind, batch_size, q=0.0,
                 suppress=False, **model_kwargs):
        _lib.require_device()
        self.svi, self.X, self.y = svi, X_local.contiguous(), y_local.contiguous()
        assert self.X.shape[0] == row_hi - row_lo
        self.n, self.lo, self.hi = int(n_rows_global), int(row_lo), int(row_hi)
        self.kind, self.B, self.q, self.suppress = kind, int(batch_size), float(q), bool(suppress)
        self.model_kwargs = model_kwargs
        self.dev = self.X.device
        self.d = int(self.X.shape[1])
        self.P = 2 * svi.model.latent_dim(self.d)
        self.sums = torch.empty(self.P + 2, dtype=torch.float32, device=self.dev)

    def begin(self, state, batch_key, first_batch):
        self.model = self.svi._model_struct(self.d, self.model_kwargs, state.observation_scale)
        self.hyper = self.svi._hyper()
        self.step, self.params, self.m, self.v = (t.clone() for t in state.optim_state)
        self.keybuf = torch.empty((2, 16), dtype=torch.uint32, device=self.dev)
        self.keybuf[0].copy_(state.rng_key.reshape(16))
        self.slot = 0
        self.bkey = batch_key.contiguous()
        self.bidx = torch.tensor([int(first_batch)], dtype=torch.int32, device=self.dev)
        self.src = BatchSource(self.kind, self.B, self.q, int(self.suppress), self.bkey.data_ptr(),
                               self.bidx.data_ptr(), None, self.n, self.lo, self.hi)
        lib = _lib.load()
        nbytes = lib.d3p_dpvi_logreg_workspace(C.byref(self.model), C.byref(self.src))
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        self.loss = torch.empty(1, dtype=torch.float32, device=self.dev)
        self.observation_scale = state.observation_scale

    def _state(self):
        return self.svi._state_struct(self.keybuf, self.slot, (self.step, self.params, self.m, self.v))

    def local_sums(self):
        st = self._state()
        check(_lib.load().d3p_dpvi_logreg_local_sums(stream_ptr(), C.byref(self.model), C.byref(self.hyper),
                                                     C.byref(st), C.byref(self.src), ptr(self.X), ptr(self.y), None,
                                                     ptr(self.sums), ptr(self.ws), self.ws.numel()))
        return self.sums

    def finalize(self, sums):
        st = self._state()
        check(_lib.load().d3p_dpvi_logreg_finalize(stream_ptr(), C.byref(self.model), C.byref(self.hyper),
                                                   C.byref(st), C.byref(self.src), ptr(sums), ptr(self.loss), None,
                                                   ptr(self.ws), self.ws.numel()))
        self.slot ^= 1
        return self.loss

    def end(self):
        from .svi import DPSVIState
        return DPSVIState((self.step, self.params, self.m, self.v), self.keybuf[self.slot].reshape(4, 4).clone(),
                          self.observation_scale)


def run_steps(engine, state, batch_key, first_batch, num_steps, group=None, collect_losses=True):
    """num_steps x [local_sums -> all_reduce(SUM) -> finalize] on every rank of `group`.

    With world size 1 the all-reduce is skipped.  Returns (new_state, losses)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    engine.begin(state, batch_key, first_batch)
    losses = []
    for _ in range(int(num_steps)):
        sums = engine.local_sums()
        if world > 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)   # the ONLY data-path collective
        loss = engine.finalize(sums)                                   # noise added once, after the reduce
        if collect_losses:
            losses.append(loss.clone())
    new_state = engine.end()
    return new_state, (torch.stack(losses).reshape(-1) if losses else None)
