"""ctypes binding of libd3p_hip.so (C-ABI declared in include/d3p_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``d3p_amd._lib.build()`` with
``hipcc --offload-arch=gfx950``.  There is no CPU fallback: if the shared object is missing, or no
HIP device is visible, every compute entry point raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libd3p_hip.so")
_SRC = [os.path.join(_HERE, "csrc", f) for f in ("d3p_rng.hip", "d3p_dpvi.hip", "d3p_stages.hip", "d3p_gmm.hip", "d3p_vae.hip", "d3p_fmesh.hip")]
_DEPS = _SRC + [os.path.join(_HERE, "csrc", f) for f in ("d3p_device.h", "d3p_host.h", "d3p_logreg_kernel.h", "d3p_logreg_chain.h", "d3p_logreg_persist.h", "d3p_logreg_wide.h", "d3p_fmesh.h", "d3p_ipc_arena.h")] + [
    os.path.join(os.path.dirname(_HERE), "include", "d3p_hip.h")]

D3P_BATCH_EXPLICIT, D3P_BATCH_FEISTEL, D3P_BATCH_POISSON = 0, 1, 2
D3P_FAMILY_LOGREG, D3P_FAMILY_GAUSS_MEAN = 0, 1
D3P_GUIDE_SOFTPLUS, D3P_GUIDE_EXP = 0, 1


class D3PError(RuntimeError):
    pass


class LogregModel(C.Structure):
    _fields_ = [("d", C.c_int32), ("intercept", C.c_int32), ("prior_w", C.c_float),
                ("prior_b", C.c_float), ("lik_scale", C.c_float), ("inv_obs", C.c_float),
                ("family", C.c_int32), ("guide_transform", C.c_int32), ("lik_sigma", C.c_float)]


class GmmModel(C.Structure):
    _fields_ = [("K", C.c_int32), ("d", C.c_int32), ("prior_mu_scale", C.c_float), ("lik_scale", C.c_float),
                ("inv_obs", C.c_float)]


class VaeModel(C.Structure):
    _fields_ = [("D", C.c_int32), ("H", C.c_int32), ("Z", C.c_int32), ("scale", C.c_float), ("inv_obs", C.c_float),
                ("H2", C.c_int32)]   # H2 > 0: a second hidden layer on each side (BASELINE config 5's [400, 200] variant)


class DpsviHyper(C.Structure):
    _fields_ = [("clip", C.c_float), ("dp_scale", C.c_float), ("lr", C.c_float),
                ("b1", C.c_float), ("b2", C.c_float), ("adam_eps", C.c_float)]


class DpsviState(C.Structure):
    _fields_ = [("rng_key", C.c_void_p), ("key_slot", C.c_int32), ("params", C.c_void_p),
                ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("step", C.c_void_p)]


class BatchSource(C.Structure):
    _fields_ = [("kind", C.c_int32), ("B", C.c_uint32), ("q", C.c_float), ("suppress", C.c_int32),
                ("batch_key", C.c_void_p), ("batch_index", C.c_void_p), ("mask", C.c_void_p),
                ("n_rows", C.c_uint64), ("row_lo", C.c_uint64), ("row_hi", C.c_uint64)]


_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _digest(paths):
    """sha256 over the compile flags and the CONTENTS of `paths` (in order)."""
    import hashlib
    h = hashlib.sha256(" ".join(_FLAGS).encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def build(force=False, verbose=False):
    """Compile libd3p_hip.so for gfx950 (hipcc cross-compiles without a GPU): one object per source under build/, then one
    link.  Staleness is decided by CONTENT, not by modification times (a snapshot pushed to another machine need not preserve
    them): the library carries `libd3p_hip.so.srchash` = sha256 of flags + every source and header, every object a digest of
    its source + the headers; whatever does not match is rebuilt (the compiles run in parallel)."""
    want = _digest(_DEPS)
    stamp = _SO + ".srchash"
    if not force and os.path.exists(_SO) and _read(stamp) == want:
        return _SO
    objdir = os.path.join(os.path.dirname(_HERE), "build", "d3p_hip")
    os.makedirs(objdir, exist_ok=True)
    headers = [p for p in _DEPS if p not in _SRC]
    jobs, objs = [], []
    for src in _SRC:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        obj_want = _digest([src] + headers)
        if force or not os.path.exists(obj) or _read(obj + ".srchash") != obj_want:
            cmd = ["hipcc"] + _FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd), obj, obj_want))
    for cmd, proc, obj, obj_want in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        with open(obj + ".srchash", "w") as f:
            f.write(obj_want + "\n")
    cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", _SO] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return _SO


def is_stale():
    """True when libd3p_hip.so is missing or was not built from the sources in this tree (content digest)."""
    return not os.path.exists(_SO) or _read(_SO + ".srchash") != _digest(_DEPS)


_V, _U64, _U32, _I32, _F, _SZ = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_float, C.c_size_t
_PM, _PH, _PS, _PB = C.POINTER(LogregModel), C.POINTER(DpsviHyper), C.POINTER(DpsviState), C.POINTER(BatchSource)

# name -> (restype, argtypes); every symbol include/d3p_hip.h declares
SIGNATURES = {
    "d3p_abi_version": (C.c_int, []),
    "d3p_last_error": (C.c_char_p, []),
    "d3p_device_count": (C.c_int, []),
    "d3p_rng_split": (C.c_int, [_V, _V, C.c_int, _V]),
    "d3p_rng_fold_in": (C.c_int, [_V, _V, _U32, _V]),
    "d3p_rng_random_bits": (C.c_int, [_V, _V, C.c_int, _U64, _V]),
    "d3p_rng_uniform": (C.c_int, [_V, _V, _U64, _F, _F, _V]),
    "d3p_rng_normal": (C.c_int, [_V, _V, _U64, _V]),
    "d3p_rng_randint": (C.c_int, [_V, _V, _U64, _I32, _I32, _V]),
    "d3p_rng_randint_bits": (C.c_int, [_V, _V, _U64, C.c_int, C.c_int64, C.c_int64, _V]),
    "d3p_tf_split": (C.c_int, [_V, _V, C.c_int, _V]),
    "d3p_tf_fold_in": (C.c_int, [_V, _V, _U32, _V]),
    "d3p_tf_random_bits": (C.c_int, [_V, _V, _U64, _V]),
    "d3p_tf_uniform": (C.c_int, [_V, _V, _U64, _F, _F, _V]),
    "d3p_tf_normal": (C.c_int, [_V, _V, _U64, _V]),
    "d3p_tf_randint": (C.c_int, [_V, _V, _U64, _I32, _I32, _V]),
    "d3p_logreg_evaluate_workspace": (_SZ, [_PM, _U32]),
    "d3p_logreg_evaluate": (C.c_int, [_V, _PM, _V, _V, _V, _U32, _V, _V, _V, _SZ]),
    "d3p_logreg_evaluate_sites": (C.c_int, [_V, _PM, _V, _V, _V, _U32, _V, C.POINTER(C.c_int32), _I32, _V, _V, _SZ]),
    "d3p_gmm_log_prob": (C.c_int, [_V, _V, _U32, _I32, _V, _V, _V, _I32, _V]),
    "d3p_feistel_sample": (C.c_int, [_V, _V, _U32, _U32, _V]),
    "d3p_feistel_from_constants": (C.c_int, [_V, _V, _U32, _U32, _V]),
    "d3p_poisson_select_rng": (C.c_int, [_V, C.c_int, _V, _F, _U32, _U32, C.c_int, _V, _V, _V, _SZ]),
    "d3p_poisson_select_batch": (C.c_int, [_V, C.c_int, _V, _SZ, _F, _U32, _U32, C.c_int, _V, _SZ, _V, _SZ, _U32, _V, _SZ]),
    "d3p_poisson_shard_flags": (C.c_int, [_V, C.c_int, _V, _SZ, _F, _U32, _U32, _U32, _U32, _V, _V, _SZ]),
    "d3p_poisson_shard_write": (C.c_int, [_V, _U32, _U32, _U32, _U32, _V, _SZ, _V, _V, _V, _SZ, _U32, _V, _SZ]),
    "d3p_xchg_poisson_counts": (C.c_int, [_V, _V, _V, _U32, _U32, C.c_int, _V, _SZ, _V, _V, _SZ]),
    "d3p_perturb_apply": (C.c_int, [_V, _V, _V, _U64, _F, _F, _V, _F, _V]),
    "d3p_poisson_select_workspace": (_SZ, [_U32]),
    "d3p_poisson_select": (C.c_int, [_V, _V, _F, _U32, _U32, C.c_int, _V, _V, _V, _SZ]),
    "d3p_take_rows": (C.c_int, [_V, _V, _U64, _U32, _V, _U32, _V, _V]),
    "d3p_logreg_px_grads_workspace": (_SZ, [_PM, _U32]),
    "d3p_logreg_px_grads": (C.c_int, [_V, _PM, _V, _V, _V, _V, _U32, _V, _V, _V, _V, _V, _V, _SZ]),
    "d3p_clip_rows": (C.c_int, [_V, _V, _U32, _U32, _F]),
    "d3p_full_norm": (C.c_int, [_V, _V, _U64, _V, _V, _SZ]),
    "d3p_full_norm_ord": (C.c_int, [_V, _V, _U64, C.c_double, _V]),
    "d3p_combine": (C.c_int, [_V, _V, _V, _U32, _U32, _V, _V]),
    "d3p_perturb": (C.c_int, [_V, _V, _V, C.POINTER(C.c_int32), C.c_int, _F, _F, _V, _F, _V, _V]),
    "d3p_adam_step": (C.c_int, [_V, _V, _V, _V, _V, _V, _U32, _F, _F, _F, _F]),
    "d3p_sgd_step": (C.c_int, [_V, _V, _V, _V, _U32, _F]),
    "d3p_gmm_px_grads_workspace": (C.c_size_t, [_I32, _U32]),
    "d3p_gmm_evaluate_workspace": (C.c_size_t, [_V, _U32]),
    "d3p_gmm_evaluate": (C.c_int, [_V, _V, _V, _V, _U32, _V, _V, _V, C.c_size_t]),
    "d3p_dpvi_gmm_workspace": (C.c_size_t, [_V, _U32]),
    "d3p_dpvi_gmm_update": (C.c_int, [_V, _V, _V, _V, _V, _V, _U32, _V, _V, _V, C.c_size_t]),
    "d3p_dpvi_gmm_run": (C.c_int, [_V, _V, _V, _V, _V, _U32, _V, _U32, _U32, _U32, _V, _V, C.c_size_t]),
    "d3p_gmm_px_grads": (C.c_int, [_V, _V, _V, _V, _V, _U32, _V, _V, _V, _V, _V, _V, C.c_size_t]),
    "d3p_vae_num_params": (C.c_int64, [_V]),
    "d3p_dpvi_vae_workspace": (C.c_size_t, [_V, _U32]),
    "d3p_gemm_f32": (C.c_int, [_V, _V, C.c_int64, C.c_int64, _V, C.c_int64, C.c_int64, _V, _I32, _I32, _I32, _I32, _V, _F, _I32]),
    "d3p_vae_step_sums": (C.c_int, [_V, _V, _V, _V, _V, _U32, _V, _V, _F, _V, _V, _V, _V, C.c_size_t]),
    "d3p_vae_evaluate": (C.c_int, [_V, _V, _V, _V, _U32, _V, _V, _V, _V, C.c_size_t]),
    "d3p_dpvi_vae_update": (C.c_int, [_V, _V, _V, _V, _V, _V, _U32, _V, _V, _V, _V, C.c_size_t]),
    "d3p_dpvi_vae_update_from": (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _U32, _V, _V, _V, _V, C.c_size_t]),
    "d3p_dpvi_vae_run": (C.c_int, [_V, _V, _V, _V, _V, _U32, _V, _U32, _U32, _U32, _V, _V, _V, _V, C.c_size_t]),
    "d3p_dpvi_gmm_local_sums": (C.c_int, [_V, _V, _V, _V, _V, _V, _U32, _U32, _U32, _V, _V, C.c_size_t]),
    "d3p_dpvi_gmm_apply": (C.c_int, [_V, _V, _V, _V, _V, _U32, _U32, _V, _V, _V, C.c_size_t]),
    "d3p_dpvi_vae_local_sums": (C.c_int, [_V, _V, _V, _V, _V, _V, _U32, _U32, _U32, _V, _V, _V, C.c_size_t]),
    "d3p_dpvi_vae_apply": (C.c_int, [_V, _V, _V, _V, _V, _U32, _U32, _V, _V, _V, C.c_size_t]),
    "d3p_dpvi_vae_run_dist": (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _V, _U32, _U32, _U32, _U32, _V, C.c_int32, _V, C.c_size_t]),
    "d3p_fmesh_create": (C.c_int, [C.c_int32, C.c_int32, C.c_uint64, _V, _V, C.c_size_t]),
    "d3p_fmesh_connect": (C.c_int, [_V, _V, C.c_size_t]),
    "d3p_fmesh_connect_local": (C.c_int, [_V, _V, C.c_int32]),
    "d3p_fmesh_set_grid": (C.c_int, [_V, C.c_int32]),
    "d3p_fmesh_allreduce": (C.c_int, [_V, _V, _V, C.c_uint64]),
    "d3p_fmesh_status": (C.c_int, [_V, _V, _V]),
    "d3p_fmesh_destroy": (C.c_int, [_V]),
    "d3p_fmesh_disconnect": (C.c_int, [_V]),
    "d3p_dpvi_logreg_chain_status": (C.c_int, [_V, _V, _V, _V, C.c_size_t, _V]),
    "d3p_dpvi_logreg_run_status": (C.c_int, [_V, _V, _V, _V, C.c_size_t, _V, _V]),
    "d3p_xchg_create": (C.c_int, [_I32, _I32, _U32, _V, _V, _SZ]),
    "d3p_xchg_connect": (C.c_int, [_V, _V, _SZ]),
    "d3p_xchg_connect_local": (C.c_int, [_V, _V, _I32]),
    "d3p_xchg_destroy": (C.c_int, [_V]),
    "d3p_xchg_disconnect": (C.c_int, [_V]),
    "d3p_xchg_allreduce": (C.c_int, [_V, _V, _V, _I32]),
    "d3p_xchg_simulate_peers": (C.c_int, [_V, _V, _U32]),
    "d3p_dpvi_logreg_run_xchg": (C.c_int, [_V, _V, _PM, _PH, _PS, _PB, _V, _V, _U32, _V, _V, _SZ]),
    "d3p_dpvi_logreg_set_run_form": (C.c_int, [C.c_int]),
    "d3p_dpvi_logreg_chain_grid": (C.c_int, [_V, _V, C.c_int, _V, _V]),
    "d3p_dpvi_logreg_kernel_timing_enable": (C.c_int, [C.c_int]),
    "d3p_dpvi_logreg_kernel_timing_read": (C.c_int, [_V, _V, _V]),
    "d3p_comm_unique_id": (C.c_int, [_V, C.c_size_t]),
    "d3p_comm_init": (C.c_int, [_V, C.c_size_t, _I32, _I32, _V]),
    "d3p_comm_destroy": (C.c_int, [_V]),
    "d3p_dpvi_logreg_run_dist": (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _V, _U32, _V, _V, C.c_size_t]),
    "d3p_adadp_workspace": (C.c_size_t, []),
    "d3p_adadp_step": (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _U32, _F, C.c_int, _V, C.c_size_t]),
    "d3p_dpvi_logreg_workspace": (_SZ, [_PM, _PB]),
    "d3p_dpvi_logreg_local_sums": (C.c_int, [_V, _PM, _PH, _PS, _PB, _V, _V, _V, _V, _V, _SZ]),
    "d3p_dpvi_logreg_finalize": (C.c_int, [_V, _PM, _PH, _PS, _PB, _V, _V, _V, _V, _SZ]),
    "d3p_dpvi_logreg_begin": (C.c_int, [_V, _PM, _PH, _PS, _PB, _V, _SZ]),
    "d3p_dpvi_logreg_prepare": (C.c_int, [_V, _PM, _PH, _PS, _PB, _U32, _V, _SZ]),
    "d3p_dpvi_logreg_step_sums": (C.c_int, [_V, _PM, _PH, _PS, _PB, _U32, _V, _V, _V, _V, _V, _SZ]),
    "d3p_dpvi_logreg_step_finalize": (C.c_int, [_V, _PM, _PH, _PS, _PB, _U32, _V, _V, _V, _V, _SZ]),
    "d3p_dpvi_logreg_end": (C.c_int, [_V, _PM, _PH, _PS, _PB, _U32, _V, _SZ]),
    "d3p_dpvi_logreg_prepare_buf": (C.c_int, [_V, _PM, _PH, _PS, _PB, _U32, C.c_int, _V, _SZ]),
    "d3p_dpvi_logreg_acc_layout": (C.c_int, [_PM, _PB, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "d3p_dpvi_logreg_acc_reset": (C.c_int, [_V, _PM, _PH, _PS, _PB, _V, _SZ]),
    "d3p_dpvi_logreg_fused_step_supported": (C.c_int, [_PM, _PB]),
    "d3p_dpvi_logreg_fused_step": (C.c_int, [_V, _PM, _PH, _PS, _PB, _U32, _U32, C.c_int, C.c_int, _U32, C.c_int, _V, _V,
                                             _V, C.c_int, _V, _SZ]),
    "d3p_dpvi_logreg_run": (C.c_int, [_V, _PM, _PH, _PS, _PB, _V, _V, _U32, _V, _V, _SZ]),
    "d3p_dpvi_logreg_run_from": (C.c_int, [_V, _PM, _PH, _PS, _PS, _PB, _U32, _V, _V, _U32, _V, _V, _SZ]),
    "d3p_dpvi_logreg_run_dist_from": (C.c_int, [_V, _V, _V, _PM, _PH, _PS, _PS, _PB, _U32, _V, _V, _U32, _V, _V, _SZ]),
    "d3p_dpvi_logreg_time_main_kernel": (C.c_int, [_V, _PM, _PH, _PS, _PB, _V, _V, _V, _SZ, C.c_int,
                                                   C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "d3p_selftest_wave_sums": (C.c_int, [_V, _V, _U32, _V]),
    "d3p_synth_logreg": (C.c_int, [_V, _U32, _U64, _U64, _I32, _V, _V]),
    "d3p_hbm_copy": (C.c_int, [_V, _V, _V, _U64, _I32]),
    "d3p_dpvi_leaves_begin": (C.c_int, [_V, _V, _I32, _V, _V, _U32, _V, _V, _V, _V]),
    "d3p_dpvi_leaves_finalize": (C.c_int, [_V, _PH, _V, _V, _V, C.POINTER(C.c_int32), _I32, _U32, C.c_float,
                                           _V, _V, _V, _V, _V, _V, _V, _V, _V, _V]),
    "d3p_px_eps_sites": (C.c_int, [_V, _V, _U32, _U32, _U32, C.POINTER(C.c_int32), _I32, _V]),
}

_lib = None


def load():
    """dlopen libd3p_hip.so and attach signatures (no GPU required for this step)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise D3PError(
                f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  d3p_amd has no CPU fallback.")
        # torch bundles its own libamdhip64; import it first so that the loader binds this
        # library to the SAME HIP runtime instance (two runtimes in one process do not share a device)
        import torch  # noqa: F401
        # D3P_HIP_LIBRARY: a developer switch -- load another build of the library (diagnostic instantiations: tools/gemm_diag.sh)
        lib = C.CDLL(os.environ.get("D3P_HIP_LIBRARY") or _SO)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.d3p_abi_version() != 9:
            raise D3PError("libd3p_hip.so ABI version mismatch")
        _lib = lib
    return _lib


_device_checked = False


def require_device():
    """Fail loudly unless torch sees a ROCm device and the library can reach it."""
    global _device_checked
    if _device_checked:
        return
    import torch
    if not torch.cuda.is_available():
        raise D3PError("d3p_amd needs a ROCm GPU (MI355X / gfx950): torch.cuda.is_available() is "
                       "False and there is no CPU fallback.")
    if load().d3p_device_count() <= 0:
        raise D3PError("libd3p_hip.so sees no HIP device")
    _device_checked = True


def check(rc):
    if rc != 0:
        msg = load().d3p_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        raise D3PError(f"libd3p_hip error {rc}: {msg}")


def stream_ptr():
    """The current torch stream of the current device as a raw hipStream_t (the fast accessor when this torch has it:
    torch.cuda.current_stream() costs ~9 us per call, which matters for per-step calls)."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device address of a torch tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


_ABORT_KINDS = {1: "a bounded wait", 2: "exchange workgroup: the step's compute workgroups did not all arrive",
                3: "exchange workgroup: the row of a rank did not come (detail = that rank)",
                4: "compute workgroup: the previous step was not released (detail 1: by the exchange)",
                5: "key-chain workgroup: the previous link did not come", 6: "exchange kernel: the row of a rank did not come"}


def describe_abort(code: int) -> str:
    """Text of a run's abort code (d3p_logreg_kernel.h: kind | step << 8 | detail << 20); '' for 0."""
    code = int(code) & 0xffffffff
    if code == 0:
        return ""
    kind, step, detail = code & 0xff, (code >> 8) & 0xfff, (code >> 20) & 0xfff
    return f"{_ABORT_KINDS.get(kind, 'unknown wait')} [kind {kind}, step {step} of its launch, detail {detail}]"
