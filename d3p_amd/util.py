"""Mirror of the hot-path helpers of d3p.util (reference d3p/util.py:68-77, :216-301)."""
import torch

from . import _lib
from . import random as strong_rng
from ._lib import check, ptr, stream_ptr

__all__ = ["example_count", "sample_from_array", "take_rows"]


def example_count(a):
    """d3p/util.py:68-77."""
    try:
        return tuple(a.shape)[0]
    except (IndexError, AttributeError):
        return 1


def take_rows(a: torch.Tensor, idx: torch.Tensor, valid_count: torch.Tensor = None) -> torch.Tensor:
    """``jnp.take(a, idx, axis=0)`` on the device (d3p/minibatch.py:126-129, :210, :233, :306).

    With ``valid_count`` (device uint32 scalar) rows at positions >= valid_count are zero-filled,
    which is the mask multiply of d3p/minibatch.py:127-129.
    """
    if not a.is_cuda:
        raise _lib.D3PError("take_rows: the dataset must live on the GPU")
    a = a.contiguous()
    n_rows = a.shape[0]
    row_elems = 1
    for s in a.shape[1:]:
        row_elems *= int(s)
    row_bytes = row_elems * a.element_size()
    n = int(idx.numel())
    if row_bytes % 4 != 0:
        raise _lib.D3PError("take_rows: row size must be a multiple of 4 bytes")
    out = torch.empty((n,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
    idx = idx.contiguous()
    check(_lib.load().d3p_take_rows(stream_ptr(), ptr(a), n_rows, row_bytes, ptr(idx), n,
                                    ptr(valid_count), ptr(out)))
    return out


def feistel_indices(rng_key, capacity: int, n: int, rng_suite=strong_rng) -> torch.Tensor:
    """Indices ``permute(0..n-1)`` of the keyed Feistel permutation over ``range(capacity)``."""
    if n > capacity:
        raise ValueError("cannot sample more elements than the array holds without replacement")
    if hasattr(rng_suite, "_feistel_sample"):
        return rng_suite._feistel_sample(rng_key, capacity, n)
    # generic rng_suite: round constants from its random_bits (d3p/util.py:240-242)
    rc = rng_suite.random_bits(rng_key, 32, (10, 3)).contiguous()
    out = torch.empty(max(n, 1), dtype=torch.uint32, device=rc.device)
    check(_lib.load().d3p_feistel_from_constants(stream_ptr(), ptr(rc), int(capacity), int(n), ptr(out)))
    return out[:n]


def sample_from_array(rng_key, x: torch.Tensor, n: int, axis: int = 0, rng_suite=strong_rng) -> torch.Tensor:
    """Samples `n` elements of `x` along `axis` without replacement (d3p/util.py:216-301)."""
    _lib.require_device()
    axis = axis % x.dim()
    capacity = x.shape[axis]
    idx = feistel_indices(rng_key, capacity, n, rng_suite)
    if axis == 0:
        return take_rows(x, idx)
    moved = x.movedim(axis, 0).contiguous()
    return take_rows(moved, idx).movedim(0, axis)
