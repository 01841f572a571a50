"""d3p_amd -- MI355X-native drop-in for the DP-VI update path of DPBayes/d3p.

Host Python mirrors d3p's public surface for that path (``d3p_amd.svi.DPSVI``, the ``rng_suite``
modules ``d3p_amd.random`` / ``d3p_amd.random.debug``, the batchifiers in ``d3p_amd.minibatch``)
and drives hand-written gfx950 kernels through the C-ABI of ``libd3p_hip.so`` (include/d3p_hip.h).
PyTorch is used for device memory, streams and torch.distributed only.
"""
from .version import __version__  # noqa: F401
