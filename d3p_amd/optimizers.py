"""Mirror of ``d3p.optimizers`` (reference d3p/optimizers.py): the ADADP adaptive-learning-rate optimiser of
Koskela & Honkela (arXiv:1809.03832) with numpyro's optimiser protocol --
``init(params) -> (0, (x, lr, x_stepped, x_prev))``, ``update(g, state) -> (i + 1, (...))``, ``get_params(state)``.

The arithmetic runs in ``d3p_adadp_step`` (include/d3p_hip.h) on the concatenation of all leaves: the error estimate
is one norm over every site (optimizers.py:75-87), so the tree structure only matters for packing and unpacking.
Reference quirks kept on purpose: the learning-rate factor is clamped to the literals [0.9, 1.1] whatever
``alpha_min`` / ``alpha_max`` say (optimizers.py:89-91), and the error is relative to ``max(1, x)``, not ``max(1, |x|)``.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _flatten(tree):
    from .svi import _tree_flatten
    return _tree_flatten(tree)


def _unflatten(treedef, leaves):
    from .svi import _tree_unflatten
    return _tree_unflatten(treedef, leaves)


def _dev_f32(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32)
    return torch.as_tensor(np.asarray(x, dtype=np.float32)).to(device)


class ADADP:
    """d3p.optimizers.ADADP(step_size=1e-3, tol=1.0, stability_check=True, alpha_min=0.9, alpha_max=1.1)."""

    def __init__(self, step_size=1e-3, tol=1.0, stability_check=True, alpha_min=0.9, alpha_max=1.1):
        self.step_size = float(step_size)
        self.tol = float(tol)
        self.stability_check = bool(stability_check)
        self.alpha_min, self.alpha_max = float(alpha_min), float(alpha_max)   # accepted and unused, as in the reference

    def init(self, params):
        """(0, (x0, lr, zeros_like(x0), x0)) (optimizers.py:53-56)."""
        leaves, treedef = _flatten(params)
        zeros = [torch.zeros_like(l) if isinstance(l, torch.Tensor) else np.zeros_like(l) for l in leaves]
        return 0, (params, self.step_size, _unflatten(treedef, zeros), params)

    def get_params(self, opt_state):
        return opt_state[1][0]

    def update(self, g, opt_state):
        """One ADADP step on the device; every leaf of the returned state is a new CUDA tensor."""
        _lib.require_device()
        lib = _lib.load()
        i, (x, lr, x_stepped, x_prev) = opt_state
        dev = torch.device("cuda", torch.cuda.current_device())
        xl, treedef = _flatten(x)
        shapes = [tuple(np.shape(l)) for l in xl]

        def pack(tree):
            leaves, _ = _flatten(tree)
            if len(leaves) != len(shapes):
                raise ValueError("ADADP: gradient / state trees do not match the parameter tree")
            return torch.cat([_dev_f32(l, dev).reshape(-1) for l in leaves]).contiguous() if leaves else \
                torch.zeros(0, dtype=torch.float32, device=dev)

        xf, gf, sf, pf = pack(x).clone(), pack(g), pack(x_stepped).clone(), pack(x_prev).clone()
        if not (gf.numel() == sf.numel() == pf.numel() == xf.numel()):
            raise ValueError("ADADP: gradient / state leaves do not have the parameters' sizes")
        lr_t = _dev_f32(lr, dev).reshape(1).clone()
        step = (i.to(device=dev, dtype=torch.int32).reshape(1).clone() if isinstance(i, torch.Tensor)
                else torch.tensor([int(i)], dtype=torch.int32, device=dev))
        ws = torch.empty(int(lib.d3p_adadp_workspace()), dtype=torch.uint8, device=dev)
        check(lib.d3p_adadp_step(stream_ptr(), ptr(xf), ptr(lr_t), ptr(sf), ptr(pf), ptr(step), ptr(gf), xf.numel(),
                                 self.tol, int(self.stability_check), ptr(ws), ws.numel()))

        def unpack(flat):
            out, pos = [], 0
            for shp in shapes:
                n = int(np.prod(shp)) if shp else 1
                out.append(flat[pos:pos + n].reshape(shp))
                pos += n
            return _unflatten(treedef, out)

        new_i = step[0] if isinstance(i, torch.Tensor) else int(i) + 1
        return new_i, (unpack(xf), lr_t[0], unpack(sf), unpack(pf))


def adadp(step_size=1e-3, tol=1.0, stability_check=True, alpha_min=0.9, alpha_max=1.1):
    """d3p.optimizers.adadp (optimizers.py:29-112): the ``(init_fun, update_fun, get_params)`` triple in the style of
    ``jax.example_libraries.optimizers`` that the reference's ``ADADP`` class wraps -- ``init(x0) -> (x0, lr, zeros_like(x0), x0)``,
    ``update(i, g, state) -> state``, ``get_params(state) -> x`` -- over the same device step (``d3p_adadp_step``)."""
    opt = ADADP(step_size, tol, stability_check, alpha_min, alpha_max)

    def init(x0):
        return opt.init(x0)[1]

    def update(i, g, state):
        return opt.update(g, (i, state))[1]

    def get_params(state):
        return state[0]

    return init, update, get_params
