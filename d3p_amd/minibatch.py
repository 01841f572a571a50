"""Minibatch samplers: mirror of d3p.minibatch (reference d3p/minibatch.py:23-322).

Each factory returns ``(init, get_batch)`` like the reference.  The dataset is a tuple of CUDA
tensors kept resident in HBM; ``get_batch`` draws indices on the device (Feistel permutation /
Poisson selection) and gathers the rows with a HIP kernel.  The returned ``get_batch`` also carries
``get_batch.source`` (a description of the sampler) so that ``DPSVI.run_steps`` can fuse the
sampler into the update step without materialising the batch.
"""
import scipy.stats
import torch

from . import _lib
from . import random as strong_rng
from ._lib import check, ptr, stream_ptr
from .util import example_count, feistel_indices, take_rows

__all__ = ["subsample_batchify_data", "split_batchify_data", "poisson_batchify_data",
           "q_to_batch_size", "batch_size_to_q"]


class BatchSourceInfo:
    """What DPSVI.run_steps needs to know to fuse a sampler into the update kernel."""

    def __init__(self, kind, dataset, batch_size, q=0.0, suppress=False, rng_suite=strong_rng):
        self.kind = kind
        self.dataset = dataset
        self.batch_size = batch_size
        self.q = q
        self.suppress = suppress
        self.rng_suite = rng_suite


def _check_dataset(dataset):
    if not dataset:
        raise ValueError("The data set must not be empty")
    num_records = example_count(dataset[0])
    for arr in dataset:
        if num_records != example_count(arr):
            raise ValueError("All arrays constituting the data set must have the same number of records")
    return num_records


def poisson_sample_idxs(rng_key, q, N, rng_suite, cutoff_size=None):
    """d3p/minibatch.py:29-39 -> (idxs[cutoff], counts) with counts = [num_selected, valid]."""
    if cutoff_size is None or cutoff_size > N:
        cutoff_size = N
    lib = _lib.load()
    ws_bytes = lib.d3p_poisson_select_workspace(N)
    dev = rng_key.device
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    idxs = torch.empty(max(cutoff_size, 1), dtype=torch.uint32, device=dev)
    counts = torch.empty(2, dtype=torch.uint32, device=dev)
    kind = getattr(rng_suite, "RNG_KIND", None)
    if kind is None:
        raise _lib.D3PError("poisson sampling needs d3p_amd.random or d3p_amd.random.debug as rng_suite")
    return idxs, counts, ws, kind, cutoff_size


def poisson_batchify_data(dataset, q, max_batch_size, handle_oversized_batch="truncate", rng_suite=strong_rng):
    """d3p/minibatch.py:42-133."""
    if not dataset:
        raise ValueError("The data set must not be empty")
    if not isinstance(dataset, tuple):
        raise ValueError("Parameter dataset must be a tuple containing arrays of equal length.")
    if q < 0 or q > 1:
        raise ValueError("Parameter q must be >=0 and <=1.")
    num_records = _check_dataset(dataset)
    if max_batch_size < 0:
        raise ValueError("max_batch_size must be a positive integer denoting the maximum batch size,"
                         " or a float between 0 and 1 denoting the maximum batch size in terms of Poisson probability mass.")
    if not isinstance(max_batch_size, int):
        max_batch_size = int(scipy.stats.poisson(num_records * q).ppf(max_batch_size))
    if handle_oversized_batch not in ("truncate", "suppress"):
        raise ValueError("handle_oversized_batch must be 'truncate' or 'suppress'")
    suppress = handle_oversized_batch == "suppress"

    def init(rng_key):
        """d3p/minibatch.py:94-101 (ZeroDivisionError if int(q * N) == 0, as the reference)."""
        return num_records // int(q * num_records), rng_key

    def get_batch(i, batchifier_state):
        """d3p/minibatch.py:104-131 -> (batch_tuple, mask)."""
        _lib.require_device()
        if max_batch_size > num_records:
            # d3p/minibatch.py:116: poisson_sample_idxs clamps its cutoff to N, then `assert len(idxs) == max_batch_size` fails
            raise AssertionError("poisson_batchify_data: max_batch_size exceeds the number of records")
        rng_key = rng_suite.fold_in(batchifier_state, i)
        idxs, counts, ws, kind, cutoff = poisson_sample_idxs(rng_key, q, num_records, rng_suite, max_batch_size)
        check(_lib.load().d3p_poisson_select_rng(stream_ptr(), kind, ptr(rng_key.contiguous()), float(q),
                                                 num_records, cutoff, int(suppress), ptr(idxs), ptr(counts),
                                                 ptr(ws), ws.numel()))
        valid = counts[1:2]
        mask = torch.arange(max_batch_size, device=idxs.device) < valid.view(torch.int32)
        idxs = idxs[:max_batch_size]
        return tuple(take_rows(a, idxs, valid) for a in dataset), mask

    get_batch.source = BatchSourceInfo(_lib.D3P_BATCH_POISSON, dataset, max_batch_size, q, suppress, rng_suite)
    return init, get_batch


def subsample_batchify_data(dataset, batch_size=None, q=None, with_replacement=False, rng_suite=strong_rng,
                            return_mask=False):
    """d3p/minibatch.py:136-239."""
    if batch_size is None and q is None:
        raise ValueError("Either batch_size or batch ratio q must be given")
    if batch_size is not None and q is not None:
        raise ValueError("Only one of batch_size and batch ratio q must be given")
    num_records = _check_dataset(dataset)
    if batch_size is None:
        batch_size = q_to_batch_size(q, num_records)

    def init(rng_key):
        """d3p/minibatch.py:185-192."""
        return num_records // batch_size, rng_key

    def _finish(ret_idx):
        batch = tuple(take_rows(a, ret_idx) for a in dataset)
        if return_mask:
            return batch, torch.ones(batch_size, dtype=torch.bool, device=ret_idx.device)
        return batch

    def get_batch_with_replacement(i, batchifier_state):
        """d3p/minibatch.py:195-214."""
        _lib.require_device()
        batch_rng_key = rng_suite.fold_in(batchifier_state, i)
        ret_idx = rng_suite.randint(batch_rng_key, (batch_size,), 0, num_records)
        return _finish(ret_idx.view(torch.uint32))

    def get_batch_without_replacement(i, batchifier_state):
        """d3p/minibatch.py:218-237."""
        _lib.require_device()
        batch_rng_key = rng_suite.fold_in(batchifier_state, i)
        ret_idx = feistel_indices(batch_rng_key, num_records, batch_size, rng_suite)
        return _finish(ret_idx)

    if with_replacement:
        return init, get_batch_with_replacement
    get_batch_without_replacement.source = BatchSourceInfo(_lib.D3P_BATCH_FEISTEL, dataset, batch_size,
                                                           rng_suite=rng_suite)
    return init, get_batch_without_replacement


def split_batchify_data(dataset, batch_size=None, q=None, rng_suite=strong_rng, return_mask=False):
    """d3p/minibatch.py:242-312."""
    if batch_size is None and q is None:
        raise ValueError("Either batch_size or batch ratio q must be given")
    if batch_size is not None and q is not None:
        raise ValueError("Only one of batch_size and batch ratio q must be given")
    num_records = _check_dataset(dataset)
    if batch_size is None:
        batch_size = q_to_batch_size(q, num_records)

    def init(rng_key):
        """d3p/minibatch.py:281-290: a full Feistel shuffle of range(N)."""
        _lib.require_device()
        return num_records // batch_size, feistel_indices(rng_key, num_records, num_records, rng_suite)

    def get_batch(i, idxs):
        """d3p/minibatch.py:293-310."""
        ret_idx = idxs[i * batch_size:(i + 1) * batch_size]
        batch = tuple(take_rows(a, ret_idx) for a in dataset)
        if return_mask:
            return batch, torch.ones(batch_size, dtype=torch.bool, device=ret_idx.device)
        return batch

    return init, get_batch


def q_to_batch_size(q, N):
    """d3p/minibatch.py:315-317."""
    return int(N * q)


def batch_size_to_q(batch_size, N):
    """d3p/minibatch.py:320-322."""
    return batch_size / N
