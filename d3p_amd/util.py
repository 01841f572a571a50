"""Mirror of d3p.util: the hot-path helpers (reference d3p/util.py:68-77, :216-301) and, host-side only, its shape / type
predicates (d3p/util.py:29-65, :80-213) over torch tensors, numpy arrays and Python scalars."""
import numbers

import numpy as np
import torch

from . import _lib
from . import random as strong_rng
from ._lib import check, ptr, stream_ptr

__all__ = ["example_count", "sample_from_array", "take_rows", "map_over_secondary_dims", "has_shape", "is_array", "is_scalar",
           "is_integer", "is_int_scalar", "normalize", "unvectorize_shape", "unvectorize_shape_1d", "unvectorize_shape_2d",
           "unvectorize_shape_3d"]


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
    if not (isinstance(idx, torch.Tensor) and idx.is_cuda and idx.element_size() == 4 and not idx.is_floating_point()):
        raise _lib.D3PError("take_rows: the indices must be a CUDA tensor of 32-bit integers")
    if valid_count is not None and not (isinstance(valid_count, torch.Tensor) and valid_count.is_cuda and valid_count.element_size() == 4):
        raise _lib.D3PError("take_rows: valid_count must be a CUDA tensor holding one 32-bit integer")
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


# ---- shape / type predicates (not on the hot path; kept so that code written against d3p.util imports unchanged)

def _shape_of(a):
    return tuple(a.shape) if hasattr(a, "shape") else tuple(np.shape(a))


def map_over_secondary_dims(f):
    """For T of shape (a, b_1 .. b_k): the array of f(T[:, i_1 .. i_k]), shape (b_1 .. b_k) (d3p/util.py:29-65)."""
    def mapped(T):
        T = torch.as_tensor(T)
        if T.dim() < 1:
            raise ValueError("map_over_secondary_dims: the input needs at least one dimension")
        columns = T.reshape(T.shape[0], -1)
        values = [torch.as_tensor(f(columns[:, j])) for j in range(columns.shape[1])]
        out = torch.stack(values) if values else torch.empty(0, dtype=T.dtype, device=T.device)
        return out.reshape(T.shape[1:])
    mapped.__name__ = getattr(f, "__name__", "mapped")
    mapped.__doc__ = getattr(f, "__doc__", None)
    return mapped


def has_shape(a):
    """True for anything with a ``shape`` attribute (d3p/util.py:80-90)."""
    return hasattr(a, "shape")


def is_array(a):
    """True for an array type with at least one dimension (d3p/util.py:93-101)."""
    return has_shape(a) and len(_shape_of(a)) > 0


def is_scalar(x):
    """True for a scalar and for an array of exactly one element, whatever its rank (d3p/util.py:104-117)."""
    if isinstance(x, (numbers.Number, np.generic)):
        return True
    if not has_shape(x):
        return False
    count = 1
    for extent in _shape_of(x):
        count *= int(extent)
    return count == 1


def is_integer(x):
    """True for integer-typed scalars and arrays (d3p/util.py:120-127)."""
    if isinstance(x, torch.Tensor):
        return not (x.dtype.is_floating_point or x.dtype.is_complex or x.dtype == torch.bool)
    if has_shape(x) and hasattr(x, "dtype"):
        return np.issubdtype(x.dtype, np.integer)
    return isinstance(x, numbers.Integral) and not isinstance(x, bool)


def is_int_scalar(x):
    """d3p/util.py:130-137."""
    return is_scalar(x) and is_integer(x)


def normalize(x):
    """x / ||x||_2 (d3p/util.py:140-146)."""
    if isinstance(x, torch.Tensor):
        xf = x if x.dtype.is_floating_point else x.to(torch.float32)
        return xf / torch.linalg.vector_norm(xf)
    xa = np.asarray(x, dtype=np.result_type(np.asarray(x).dtype, np.float32))
    return xa / np.linalg.norm(xa)


def unvectorize_shape(a, d):
    """Shape of `a`, padded FROM THE FRONT with ones up to `d` dimensions (d3p/util.py:149-171)."""
    shape = _shape_of(a)
    return (1,) * max(d - len(shape), 0) + shape


def unvectorize_shape_1d(a):
    return unvectorize_shape(a, 1)


def unvectorize_shape_2d(a):
    return unvectorize_shape(a, 2)


def unvectorize_shape_3d(a):
    return unvectorize_shape(a, 3)
