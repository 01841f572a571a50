"""Cryptographically secure rng_suite (ChaCha20 counter mode) on the GPU.

Mirror of ``d3p.random`` (reference d3p/random/__init__.py:28-155): the module-level protocol
``PRNGKey, split, fold_in, random_bits, uniform, normal, randint, convert_to_jax_rng_key,
PRNGState``.  Keys are (4, 4) ``torch.uint32`` CUDA tensors holding an RFC 8439 ChaCha20 state;
every function launches a kernel of libd3p_hip.so on the current stream.  The key/nonce/counter
layout is this build's own: jax-chacha-prng, which defines it in the
reference, is not vendored there.
"""
import secrets
from typing import Optional, Sequence, Union

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr

PRNGState = torch.Tensor
ChaChaKeySizeInBytes = 32
RNG_KIND = 0  # ChaCha keystream (see d3p_poisson_select_rng)

_CONSTANTS = (0x61707865, 0x3320646E, 0x79622D32, 0x6B206574)
_UINT = {8: torch.uint8, 16: torch.uint16, 32: torch.uint32, 64: torch.uint64}
_FLOATS = (torch.float16, torch.bfloat16, torch.float32, torch.float64)


def _seed_to_bytes(seed) -> bytes:
    if isinstance(seed, (int, np.integer)):
        return (int(seed) % (1 << 256)).to_bytes(32, "big")
    if isinstance(seed, (bytes, bytearray)):
        if len(seed) > ChaChaKeySizeInBytes:
            raise ValueError("seed must be at most 256 bit long")
        return bytes(seed).ljust(32, b"\0")
    if isinstance(seed, torch.Tensor):
        seed = seed.detach().cpu().numpy()
    a = np.asarray(seed).astype(np.uint32).ravel()
    if a.size > 8:
        raise ValueError("seed must be at most 256 bit long")
    return np.concatenate([a, np.zeros(8 - a.size, np.uint32)]).astype("<u4").tobytes()


def _state_words(seed) -> np.ndarray:
    """Host-side packing of a seed into the 16-word ChaCha state (no arithmetic on the stream)."""
    words = np.zeros(16, np.uint32)
    words[:4] = _CONSTANTS
    words[4:12] = np.frombuffer(_seed_to_bytes(seed), dtype="<u4")
    return words


def PRNGKey(seed: Optional[Union[int, bytes, Sequence[int]]] = None) -> PRNGState:
    """d3p/random/__init__.py:35-47.  ``None`` draws 32 bytes from ``secrets``."""
    _lib.require_device()
    if seed is None:
        seed = secrets.token_bytes(ChaChaKeySizeInBytes)
    words = _state_words(seed)
    return torch.from_numpy(words.view(np.int32).copy()).cuda().view(torch.uint32).reshape(4, 4)


def _key(key) -> torch.Tensor:
    if not isinstance(key, torch.Tensor) or not key.is_cuda or key.numel() != 16 or key.element_size() != 4:
        raise TypeError("rng key must be a CUDA tensor of 16 32-bit words (d3p_amd.random.PRNGKey)")
    return key.contiguous()


def _numel(shape) -> int:
    shape = tuple(shape) if not isinstance(shape, int) else (shape,)
    n = 1
    for s in shape:
        if s < 0:
            raise ValueError("shape must be non-negative")
        n *= int(s)
    return n


def split(key: PRNGState, num: int = 2) -> torch.Tensor:
    """d3p/random/__init__.py:29 -> (num, 4, 4)."""
    key = _key(key)
    out = torch.empty((num, 4, 4), dtype=torch.uint32, device=key.device)
    check(_lib.load().d3p_rng_split(stream_ptr(), ptr(key), int(num), ptr(out)))
    return out


def fold_in(key: PRNGState, data: int) -> PRNGState:
    """d3p/random/__init__.py:30."""
    key = _key(key)
    out = torch.empty((4, 4), dtype=torch.uint32, device=key.device)
    check(_lib.load().d3p_rng_fold_in(stream_ptr(), ptr(key), int(data) & 0xFFFFFFFF, ptr(out)))
    return out


def random_bits(key: PRNGState, bit_width: int, shape: Sequence[int]) -> torch.Tensor:
    """d3p/random/__init__.py:31: uniformly random unsigned integers of `bit_width` bits."""
    if bit_width not in _UINT:
        raise ValueError("requires bit_width in {8, 16, 32, 64}")
    key = _key(key)
    shape = tuple(shape)
    n = _numel(shape)
    n_bytes = (n * bit_width + 511) // 512 * 64
    buf = torch.empty(max(n_bytes, 64), dtype=torch.uint8, device=key.device)
    check(_lib.load().d3p_rng_random_bits(stream_ptr(), ptr(key), bit_width, n, ptr(buf)))
    return buf[: n * bit_width // 8].view(_UINT[bit_width]).reshape(shape)


def _float_dtype(dtype, what):
    if dtype is None or dtype is float:
        return torch.float32
    if dtype not in _FLOATS:
        raise ValueError(f"dtype argument to `{what}` must be a float dtype, got {dtype}")
    if dtype != torch.float32:
        raise _lib.D3PError(f"`{what}`: only float32 is implemented on the device path (the reference "
                            "runs without jax_enable_x64, i.e. float32)")
    return dtype


def uniform(key: PRNGState, shape: Sequence[int] = (), dtype=torch.float32, minval=0.0, maxval=1.0):
    """d3p/random/__init__.py:32."""
    _float_dtype(dtype, "uniform")
    key = _key(key)
    shape = tuple(shape)
    n = _numel(shape)
    out = torch.empty(max(n, 1), dtype=torch.float32, device=key.device)
    check(_lib.load().d3p_rng_uniform(stream_ptr(), ptr(key), n, float(minval), float(maxval), ptr(out)))
    return out[:n].reshape(shape)


def normal(key: PRNGState, shape: Sequence[int] = (), dtype=torch.float32) -> torch.Tensor:
    """d3p/random/__init__.py:50-81."""
    _float_dtype(dtype, "normal")
    key = _key(key)
    shape = tuple(shape)
    n = _numel(shape)
    out = torch.empty(max(n, 1), dtype=torch.float32, device=key.device)
    check(_lib.load().d3p_rng_normal(stream_ptr(), ptr(key), n, ptr(out)))
    return out[:n].reshape(shape)


_INT_BITS = {torch.int8: 8, torch.int16: 16, torch.int32: 32, torch.int64: 64}


def _int_dtype(dtype):
    """torch / numpy / python integer dtypes -> torch dtype (d3p/random/__init__.py:101-103: anything else is a TypeError)."""
    if dtype is int:
        return torch.int32                     # jax.dtypes.canonicalize_dtype without x64
    if isinstance(dtype, torch.dtype):
        t = dtype
    else:
        import numpy as np
        try:
            t = {"int8": torch.int8, "int16": torch.int16, "int32": torch.int32, "int64": torch.int64}.get(np.dtype(dtype).name)
        except TypeError:
            t = None
    if t not in _INT_BITS:
        raise TypeError(f"dtype argument to `randint` must be an integer dtype, got {dtype}")
    return t


def randint(key: PRNGState, shape: Sequence[int], minval: int, maxval: int, dtype=torch.int32):
    """d3p/random/__init__.py:84-146: masked rejection sampling in the 8-, 16-, 32- or 64-bit integer dtype."""
    t = _int_dtype(dtype)
    key = _key(key)
    shape = tuple(shape)
    n = _numel(shape)
    out = torch.empty(max(n, 1), dtype=t, device=key.device)
    check(_lib.load().d3p_rng_randint_bits(stream_ptr(), ptr(key), n, _INT_BITS[t], int(minval), int(maxval), ptr(out)))
    return out[:n].reshape(shape)


def convert_to_jax_rng_key(rng_key: PRNGState) -> torch.Tensor:
    """d3p/random/__init__.py:149-155: two 32-bit words = a threefry ("JAX") key."""
    return random_bits(rng_key, 32, (2,))


# hooks used by d3p_amd.util / d3p_amd.minibatch to stay on fused kernels for this suite
def _feistel_sample(key, capacity, n):
    key = _key(key)
    out = torch.empty(max(n, 1), dtype=torch.uint32, device=key.device)
    check(_lib.load().d3p_feistel_sample(stream_ptr(), ptr(key), int(capacity), int(n), ptr(out)))
    return out[:n]
