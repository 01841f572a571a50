"""numpy-facing wrappers around oracle/libd3p_oracle.so (the CPU restatement in d3p_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under d3p_amd/ imports this module.  Parity status (what is pinned
against published vectors and what is this build's own layout) is stated in d3p_oracle.c's header.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libd3p_oracle.so")


def _src_digest():
    import hashlib
    h = hashlib.sha256()
    for name in ("d3p_oracle.c", "Makefile"):
        with open(os.path.join(_HERE, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False):
    """(staleness by content, like d3p_amd._lib.build: the stamp file beside the library holds the digest of its sources)"""
    want, stamp = _src_digest(), _SO + ".srchash"
    have = ""
    if os.path.exists(stamp):
        with open(stamp) as f:
            have = f.read().strip()
    if force or not os.path.exists(_SO) or have != want:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libd3p_oracle.so"],
                              stdout=subprocess.DEVNULL)
        with open(stamp, "w") as f:
            f.write(want + "\n")
    return _SO


_lib = None


class LogregSpec(C.Structure):
    _fields_ = [("d", C.c_int32), ("intercept", C.c_int32), ("prior_w", C.c_float),
                ("prior_b", C.c_float), ("lik_scale", C.c_float), ("inv_obs", C.c_float),
                ("family", C.c_int32), ("guide_exp", C.c_int32), ("lik_sigma", C.c_float)]


class GmmSpec(C.Structure):
    _fields_ = [("K", C.c_int32), ("d", C.c_int32), ("prior_mu_scale", C.c_float), ("lik_scale", C.c_float),
                ("inv_obs", C.c_float)]


class VaeSpec(C.Structure):
    _fields_ = [("D", C.c_int32), ("H", C.c_int32), ("Z", C.c_int32), ("scale", C.c_float), ("inv_obs", C.c_float),
                ("H2", C.c_int32)]


class Hyper(C.Structure):
    _fields_ = [("clip", C.c_float), ("dp_scale", C.c_float), ("lr", C.c_float),
                ("b1", C.c_float), ("b2", C.c_float), ("adam_eps", C.c_float)]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.d3po_erfinv_f32.restype = C.c_float
        _lib.d3po_erfinv_f32.argtypes = [C.c_float]
        _lib.d3po_feistel_permute.restype = C.c_uint32
        _lib.d3po_full_norm.restype = C.c_float
        _lib.d3po_combine.restype = C.c_float
        _lib.d3po_logreg_px_loss_grad.restype = C.c_float
        _lib.d3po_logreg_update.restype = C.c_float
        _lib.d3po_logreg_evaluate.restype = C.c_float
        _lib.d3po_logreg_evaluate_sites.restype = C.c_float
        _lib.d3po_logreg_run_feistel.restype = C.c_float
        _lib.d3po_digamma.restype = C.c_double
        _lib.d3po_digamma.argtypes = [C.c_double]
        _lib.d3po_gamma_grad.restype = C.c_double
        _lib.d3po_gamma_grad.argtypes = [C.c_double, C.c_double]
        _lib.d3po_gamma_sample.restype = C.c_double
        _lib.d3po_gmm_px_loss_grad_given.restype = C.c_float
        _lib.d3po_gmm_evaluate.restype = C.c_float
        _lib.d3po_vae_num_params.restype = C.c_int64
        _lib.d3po_vae_evaluate.restype = C.c_float
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


# ------------------------------------------------------------------ ChaCha20 suite
def chacha20_block(state):
    s = _u32(state).reshape(16)
    out = np.empty(16, np.uint32)
    lib().d3po_chacha20_block(_p(s), _p(out))
    return out


def seed_to_bytes(seed):
    """PRNGKey seed handling (d3p/random/__init__.py:35-47): int / bytes / uint32 array -> 32 bytes."""
    if isinstance(seed, (int, np.integer)):
        return (int(seed) % (1 << 256)).to_bytes(32, "big")
    if isinstance(seed, (bytes, bytearray)):
        if len(seed) > 32:
            raise ValueError("seed must be at most 256 bit")
        return bytes(seed).ljust(32, b"\0")
    a = np.asarray(seed, dtype=np.uint32).ravel()
    if a.size > 8:
        raise ValueError("seed must be at most 256 bit")
    return np.concatenate([a, np.zeros(8 - a.size, np.uint32)]).astype("<u4").tobytes()


def PRNGKey(seed):
    b = np.frombuffer(seed_to_bytes(seed), dtype=np.uint8).copy()
    out = np.empty(16, np.uint32)
    lib().d3po_key_from_bytes(_p(b), _p(out))
    return out.reshape(4, 4)


def split(key, num=2):
    out = np.empty((num, 16), np.uint32)
    lib().d3po_split(_p(_u32(key).reshape(16)), C.c_int(num), _p(out))
    return out.reshape(num, 4, 4)


def fold_in(key, data):
    out = np.empty(16, np.uint32)
    lib().d3po_fold_in(_p(_u32(key).reshape(16)), C.c_uint32(int(data) & 0xFFFFFFFF), _p(out))
    return out.reshape(4, 4)


_UINT = {8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}


def random_bits(key, bit_width, shape):
    n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
    out = np.empty(max(n, 1), _UINT[bit_width])
    rc = lib().d3po_random_bits(_p(_u32(key).reshape(16)), C.c_int(bit_width), C.c_uint64(n), _p(out))
    if rc:
        raise ValueError("bad bit width")
    return out[:n].reshape(shape)


def uniform(key, shape=(), minval=0.0, maxval=1.0):
    n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
    out = np.empty(max(n, 1), np.float32)
    lib().d3po_uniform(_p(_u32(key).reshape(16)), C.c_uint64(n), C.c_float(minval), C.c_float(maxval),
                       _p(out))
    return out[:n].reshape(shape)


def normal(key, shape=()):
    n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
    out = np.empty(max(n, 1), np.float32)
    lib().d3po_normal(_p(_u32(key).reshape(16)), C.c_uint64(n), _p(out))
    return out[:n].reshape(shape)


def randint(key, shape, minval, maxval, dtype=np.int32):
    """d3p/random/__init__.py:84-146 for int8 / int16 / int32 / int64."""
    dt = np.dtype(dtype)
    if dt.kind != "i":
        raise TypeError(f"dtype argument to `randint` must be an integer dtype, got {dtype}")
    n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
    out = np.empty(max(n, 1), dt)
    rc = lib().d3po_randint_bits(_p(_u32(key).reshape(16)), C.c_uint64(n), C.c_int(8 * dt.itemsize), C.c_int64(int(minval)),
                                 C.c_int64(int(maxval)), _p(out))
    assert rc == 0
    return out[:n].reshape(shape)


def randint32_legacy(key, shape, minval, maxval):
    """The int32-only restatement (d3po_randint32), kept to pin the generic one against."""
    n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
    out = np.empty(max(n, 1), np.int32)
    lib().d3po_randint32(_p(_u32(key).reshape(16)), C.c_uint64(n), C.c_int32(minval),
                         C.c_int32(maxval), _p(out))
    return out[:n].reshape(shape)


def convert_to_jax_rng_key(key):
    return random_bits(key, 32, (2,))


def erfinv_f32(x):
    x = _f32(x)
    return np.array([lib().d3po_erfinv_f32(C.c_float(float(v))) for v in x.ravel()],
                    np.float32).reshape(x.shape)


# ------------------------------------------------------------------ threefry (jax.random layouts)
def threefry2x32(k0, k1, c0, c1):
    out = np.empty(2, np.uint32)
    lib().d3po_threefry2x32(C.c_uint32(k0), C.c_uint32(k1), C.c_uint32(c0), C.c_uint32(c1), _p(out))
    return out


def tf_random_words(key, n):
    out = np.empty(max(n, 1), np.uint32)
    lib().d3po_tf_random_words(_p(_u32(key)), C.c_uint64(n), _p(out))
    return out[:n]


def tf_split(key, num=2):
    out = np.empty((num, 2), np.uint32)
    lib().d3po_tf_split(_p(_u32(key)), C.c_int(num), _p(out))
    return out


def tf_fold_in(key, data):
    out = np.empty(2, np.uint32)
    lib().d3po_tf_fold_in(_p(_u32(key)), C.c_uint32(int(data) & 0xFFFFFFFF), _p(out))
    return out


def tf_uniform(key, n, lo=0.0, hi=1.0):
    out = np.empty(max(n, 1), np.float32)
    lib().d3po_tf_uniform(_p(_u32(key)), C.c_uint64(n), C.c_float(lo), C.c_float(hi), _p(out))
    return out[:n]


def tf_normal(key, n):
    out = np.empty(max(n, 1), np.float32)
    lib().d3po_tf_normal(_p(_u32(key)), C.c_uint64(n), _p(out))
    return out[:n]


def px_sample_key(jax_key, B, p):
    out = np.empty(2, np.uint32)
    lib().d3po_px_sample_key(_p(_u32(jax_key)), C.c_uint32(B), C.c_uint32(p), _p(out))
    return out


def px_eps(jax_key, B, D):
    """eps (B x D): the guide's per-example standard-normal draws (svi.py:289-290 + numpyro plumbing)."""
    return np.stack([tf_normal(px_sample_key(jax_key, B, p), D) for p in range(B)])


def px_site_keys(jax_key, B, p, n_sites):
    """Per-site sample keys of example p (numpyro's seed handler: rng, site_key = split(rng) at every sample statement). UNPINNED."""
    out = np.empty(2 * n_sites, np.uint32)
    lib().d3po_px_site_keys(_p(_u32(jax_key)), C.c_uint32(B), C.c_uint32(p), C.c_int(n_sites), _p(out))
    return out.reshape(n_sites, 2)


def px_eps_sites(jax_key, B_total, site_sizes, pos0=0, B_local=None):
    """eps (B_local x sum(site_sizes)): every site's standard-normal draw from its OWN key, sites in the guide's program order."""
    B_local = B_total - pos0 if B_local is None else B_local
    sizes = np.ascontiguousarray(np.asarray(site_sizes, np.int32))
    eps = np.empty((B_local, int(sizes.sum())), np.float32)
    lib().d3po_px_eps_sites(_p(_u32(jax_key)), C.c_uint32(B_total), C.c_uint32(pos0), C.c_uint32(B_local), _p(sizes),
                            C.c_int(sizes.size), _p(eps))
    return eps


# ------------------------------------------------------------------ samplers
def feistel_constants(key):
    rc = np.empty(30, np.uint32)
    lib().d3po_feistel_constants(_p(_u32(key).reshape(16)), _p(rc))
    return rc


def feistel_sample(key, capacity, n):
    out = np.empty(max(n, 1), np.uint32)
    lib().d3po_feistel_sample(_p(_u32(key).reshape(16)), C.c_uint32(capacity), C.c_uint32(n), _p(out))
    return out[:n]


def poisson_select(key, q, N, cutoff, suppress=False):
    idx = np.empty(max(cutoff, 1), np.uint32)
    counts = np.empty(2, np.uint32)
    lib().d3po_poisson_select(_p(_u32(key).reshape(16)), C.c_float(q), C.c_uint32(N),
                              C.c_uint32(cutoff), C.c_int(int(suppress)), _p(idx), _p(counts))
    return idx[:cutoff], int(counts[0]), int(counts[1])


# ------------------------------------------------------------------ DP-VI stages (logistic regression)
def logreg_spec(d, intercept=False, prior_w=1.0, prior_b=1.0, lik_scale=1.0, obs_scale=1.0, guide_exp=False):
    return LogregSpec(d, int(intercept), prior_w, prior_b, lik_scale, 1.0 / obs_scale, 0, int(guide_exp), 0.0)


def gauss_mean_spec(d, prior=1.0, lik_sigma=0.1, lik_scale=1.0, obs_scale=1.0, guide_exp=True):
    """examples/simple_gaussian_posterior.py:51-81 (family 1); labels are unused (pass None)."""
    return LogregSpec(d, 0, prior, prior, lik_scale, 1.0 / obs_scale, 1, int(guide_exp), lik_sigma)


def logreg_px_grads(spec, loc, unc, Xb, yb, eps, mask=None):
    B = Xb.shape[0]
    D = spec.d + spec.intercept
    Xb, eps, loc, unc = _f32(Xb), _f32(eps), _f32(loc), _f32(unc)
    yb = np.zeros(B, np.float32) if yb is None else _f32(yb)   # labels are unused by the Gaussian family
    m = None if mask is None else _f32(mask)
    px_loss = np.empty(B, np.float32)
    px_grads = np.empty((B, 2 * D), np.float32)
    factor = C.c_float()
    n = lib().d3po_logreg_px_grads(C.byref(spec), _p(loc), _p(unc), _p(Xb), _p(yb), _p(eps),
                                   None if m is None else _p(m), C.c_int(B), _p(px_loss),
                                   _p(px_grads), C.byref(factor))
    return px_loss, px_grads, n, factor.value


def clip_rows(px_grads, c):
    g = _f32(px_grads).copy()
    rc = lib().d3po_clip_rows(_p(g), C.c_int(g.shape[0]), C.c_int(g.shape[1]), C.c_float(c))
    if rc:
        raise ValueError("The clipping threshold must be greater than 0.")
    return g


def full_norm(parts):
    if parts is None:
        return 0.0
    if isinstance(parts, np.ndarray):
        parts = [parts]
    flat = [np.asarray(a, np.float32).ravel() for a in _leaves(parts)]
    if not flat:
        return 0.0
    v = np.ascontiguousarray(np.concatenate(flat))
    return float(lib().d3po_full_norm(_p(v), C.c_uint64(v.size)))


def _leaves(t):
    if t is None:
        return []
    if isinstance(t, dict):
        return [l for k in sorted(t) for l in _leaves(t[k])]
    if isinstance(t, (list, tuple)):
        return [l for x in t for l in _leaves(x)]
    return [t]


def combine(px_grads, px_loss):
    g, l = _f32(px_grads), _f32(px_loss)
    avg = np.empty(g.shape[1], np.float32)
    loss = lib().d3po_combine(_p(g), _p(l), C.c_int(g.shape[0]), C.c_int(g.shape[1]), _p(avg))
    return float(loss), avg


def perturb(key, avg, site_sizes, dp_scale, clip, num_elements, obs_scale, factor):
    avg = _f32(avg)
    sizes = np.ascontiguousarray(np.asarray(site_sizes, np.int32))
    out = np.empty_like(avg)
    lib().d3po_perturb(_p(_u32(key).reshape(16)), _p(avg), _p(sizes), C.c_int(sizes.size),
                       C.c_float(dp_scale), C.c_float(clip), C.c_float(num_elements),
                       C.c_float(obs_scale), C.c_float(factor), _p(out))
    return out


def adam(x, m, v, g, i, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    x, m, v, g = _f32(x).copy(), _f32(m).copy(), _f32(v).copy(), _f32(g)
    lib().d3po_adam(_p(x), _p(m), _p(v), _p(g), C.c_int(x.size), C.c_int(i), C.c_float(lr),
                    C.c_float(b1), C.c_float(b2), C.c_float(eps))
    return x, m, v


def adadp(x, lr, x_stepped, x_prev, g, i, tol=1.0, stability_check=True):
    """One ADADP update (d3p/optimizers.py:29-112) on flat vectors; returns (x, lr, x_stepped, x_prev)."""
    x, xs, xp, g = _f32(x).copy(), _f32(x_stepped).copy(), _f32(x_prev).copy(), _f32(g)
    l = C.c_float(lr)
    lib().d3po_adadp(_p(x), C.byref(l), _p(xs), _p(xp), _p(g), C.c_int(x.size), C.c_int(i), C.c_float(tol),
                     C.c_int(int(stability_check)))
    return x, l.value, xs, xp


class LogregState:
    """Mutable mirror of DPSVIState for the oracle's full update (svi.py:37-40)."""

    def __init__(self, key, D, loc0=None, unc0=None):
        self.key = _u32(key).reshape(16).copy()
        self.params = np.zeros(2 * D, np.float32)
        if loc0 is not None:
            self.params[:D] = loc0
        if unc0 is not None:
            self.params[D:] = unc0
        self.m = np.zeros(2 * D, np.float32)
        self.v = np.zeros(2 * D, np.float32)
        self.step = C.c_int32(0)


def logreg_update(spec, hyper, st, Xb, yb, mask=None, eps=None):
    """One DPSVI.update (svi.py:395-434); returns (loss, perturbed_grads)."""
    B = Xb.shape[0]
    D = spec.d + spec.intercept
    Xb = _f32(Xb)
    yb = np.zeros(B, np.float32) if yb is None else _f32(yb)
    m = None if mask is None else _f32(mask)
    e = None if eps is None else _f32(eps)
    scratch = np.empty(B * 2 * D + B + B * D + 4 * D, np.float32)
    grad = np.empty(2 * D, np.float32)
    loss = lib().d3po_logreg_update(C.byref(spec), C.byref(hyper), _p(st.key), _p(st.params), _p(st.m),
                                    _p(st.v), C.byref(st.step), _p(Xb), _p(yb),
                                    None if m is None else _p(m), C.c_int(B),
                                    None if e is None else _p(e), _p(scratch), _p(grad))
    return float(loss), grad


class MeanFieldLogregState:
    """State of DPSVI with the example's OWN guide (examples/logistic_regression.py:67-86): four parameter leaves, flat in
    tree_flatten order of the parameter dict (sorted names): intercept_loc (1), intercept_std_log (1), w_loc (d), w_std_log (d);
    all initialised to zeros (`param("w_loc", zeros(d))`, `param("intercept_std_log", 0.)`)."""

    def __init__(self, key, d):
        self.key = _u32(key).reshape(16).copy()
        self.d = d
        self.params = np.zeros(2 * d + 2, np.float32)
        self.m = np.zeros(2 * d + 2, np.float32)
        self.v = np.zeros(2 * d + 2, np.float32)
        self.step = 0

    @staticmethod
    def leaf_sizes(d):
        return [1, 1, d, d]

    @staticmethod
    def tree_from_kernel(d):
        """Index array: tree-ordered flat vector = kernel-ordered vector [w_loc, intercept_loc | w_std_log, intercept_std_log][perm]."""
        D = d + 1
        return np.concatenate([[d], [D + d], np.arange(d), D + np.arange(d)]).astype(np.int64)


def meanfield_logreg_update(spec, hyper, st, Xb, yb, mask=None):
    """One DPSVI.update (svi.py:395-434) of the logistic regression with intercept under the example's hand-written two-site
    guide.  spec: logreg_spec(d, intercept=True, guide_exp=True, ...).  Stage by stage: split(key, 3) (svi.py:208-211); per-example,
    per-SITE guide noise (px_eps_sites: 'w' then 'intercept'); per-example gradients (the joint density is the same as the
    one-site guide's over [w, intercept]); the leaves in tree order; joint clip (svi.py:68-124); mean (:327-348); perturbation with ONE
    KEY PER LEAF, split(key, 4) (:487-491); Adam.  Returns (loss, perturbed gradient in tree order)."""
    d = spec.d
    D = d + 1
    assert spec.intercept and spec.guide_exp == 1
    ks = split(st.key, 3)
    jax_key = convert_to_jax_rng_key(ks[1])
    B = Xb.shape[0]
    eps = px_eps_sites(jax_key, B, [d, 1])
    perm = MeanFieldLogregState.tree_from_kernel(d)
    kern = np.empty(2 * D, np.float32)
    kern[perm] = st.params
    px_loss, px_grads, n, factor = logreg_px_grads(spec, kern[:D], kern[D:], Xb, yb, eps, mask)
    px_tree = np.ascontiguousarray(px_grads[:, perm])
    clipped = clip_rows(px_tree, hyper.clip)
    loss, avg = combine(clipped, px_loss)
    g = perturb(ks[2], avg, MeanFieldLogregState.leaf_sizes(d), hyper.dp_scale, hyper.clip, float(n), 1.0 / spec.inv_obs, factor)
    st.params, st.m, st.v = adam(st.params, st.m, st.v, g, st.step, lr=hyper.lr, b1=hyper.b1, b2=hyper.b2, eps=hyper.adam_eps)
    st.step += 1
    st.key = _u32(ks[0]).reshape(16).copy()
    return loss, g


def logreg_run_feistel(spec, hyper, st, X, y, batch_key, first_batch, B, steps, threads=1):
    """`steps` DPSVI.update calls on Feistel minibatches of the resident table (X, y), all inside one C call on `threads`
    OpenMP threads (bench.py's cpu_baseline).  Advances `st` in place; returns the last loss."""
    X, y = _f32(X), _f32(y)
    d = spec.d
    scratch = np.empty(B * d + B + B * 2 * d + B + B * d + 8 * d, np.float32)
    idx = np.empty(B, np.uint32)
    return float(lib().d3po_logreg_run_feistel(C.byref(spec), C.byref(hyper), _p(st.key), _p(st.params), _p(st.m), _p(st.v),
                                               C.byref(st.step), _p(X), _p(y), C.c_uint32(X.shape[0]),
                                               _p(_u32(batch_key).reshape(16)), C.c_uint32(first_batch), C.c_int(B),
                                               C.c_int(steps), C.c_int(threads), _p(scratch), _p(idx)))


def logreg_evaluate(spec, loc, unc, Xb, yb, jax_key):
    """DPSVI.evaluate (svi.py:436-449): -ELBO of a batch with one guide draw; jax_key = convert(split(key, 1)[0])."""
    Xb, loc, unc = _f32(Xb), _f32(loc), _f32(unc)
    yb = np.zeros(Xb.shape[0], np.float32) if yb is None else _f32(yb)
    return float(lib().d3po_logreg_evaluate(C.byref(spec), _p(loc), _p(unc), _p(Xb), _p(yb), C.c_int(Xb.shape[0]),
                                            _p(_u32(jax_key))))


def meanfield_logreg_evaluate(spec, params_tree, Xb, yb, jax_key):
    """DPSVI.evaluate under the example's two-site guide; params_tree in MeanFieldLogregState's order."""
    d = spec.d
    D = d + 1
    kern = np.empty(2 * D, np.float32)
    kern[MeanFieldLogregState.tree_from_kernel(d)] = _f32(params_tree)
    Xb = _f32(Xb)
    sizes = np.asarray([d, 1], np.int32)
    return float(lib().d3po_logreg_evaluate_sites(C.byref(spec), _p(kern[:D].copy()), _p(kern[D:].copy()), _p(Xb), _p(_f32(yb)),
                                                  C.c_int(Xb.shape[0]), _p(_u32(jax_key)), _p(sizes), C.c_int(2)))


def tf_randint(key, n, minval, maxval):
    out = np.empty(max(n, 1), np.int32)
    lib().d3po_tf_randint32(_p(_u32(key)), C.c_uint64(n), C.c_int32(minval), C.c_int32(maxval), _p(out))
    return out[:n]


def gmm_log_prob(x, locs, scales, pis):
    """GaussianMixture(locs, scales, pis).log_prob(x) for rows of x (d3p/gmm.py:71-86)."""
    x, locs, scales, pis = _f32(x), _f32(locs), _f32(scales), _f32(pis)
    out = np.empty(x.shape[0], np.float32)
    lib().d3po_gmm_log_prob(_p(x), C.c_int(x.shape[0]), C.c_int(x.shape[1]), _p(locs), _p(scales), _p(pis),
                            C.c_int(locs.shape[0]), _p(out))
    return out


def gmm_sample_with_intermediates(key, locs, scales, pis, sample_shape=()):
    """d3p/gmm.py:91-95 on the oracle's threefry functions: component_key, samples_key = split(key); zs = sum(cumsum(pi) < u) with
    u = uniform(component_key, sample_shape + (1,)) (numpyro's CategoricalProbs.sample for probabilities: UNPINNED);
    xs = locs[zs] + scales[zs] * normal(samples_key, sample_shape + event_shape).  float32 throughout."""
    locs, scales, pis = _f32(locs), _f32(scales), _f32(pis)
    shape = tuple(int(s) for s in sample_shape)
    n = int(np.prod(shape)) if shape else 1
    ks = tf_split(key, 2)
    u = tf_uniform(ks[0], n).reshape(shape + (1,))
    cum = np.cumsum(pis, dtype=np.float32)
    zs = np.minimum((cum < u).sum(axis=-1), len(pis) - 1)
    ev = locs.shape[1:]
    eps = tf_normal(ks[1], n * int(np.prod(ev))).reshape(shape + ev)
    return (locs[zs] + scales[zs] * eps).astype(np.float32), zs


def synth_logreg(seed, row0, nrows, d):
    X = np.empty((nrows, d), np.float32)
    y = np.empty(nrows, np.float32)
    lib().d3po_synth_logreg(C.c_uint32(seed), C.c_uint64(row0), C.c_uint64(nrows), C.c_int(d), _p(X), _p(y))
    return X, y


def synth_wtrue(seed, d):
    w = np.empty(d + 1, np.float32)
    lib().d3po_synth_wtrue(C.c_uint32(seed), C.c_int(d), _p(w))
    return w


# ------------------------------------------------------------------ Gaussian-mixture model step (config 3)
def gmm_spec(K, d, prior_mu_scale=10.0, lik_scale=1.0, obs_scale=1.0):
    return GmmSpec(K, d, prior_mu_scale, lik_scale, 1.0 / obs_scale)


def digamma(x):
    return lib().d3po_digamma(float(x))


def gamma_grad(alpha, x):
    """d/dalpha of the Gamma(alpha, 1) quantile at fixed CDF value (implicit reparametrisation)."""
    return lib().d3po_gamma_grad(float(alpha), float(x))


def gamma_sample(key, comp, alpha):
    return lib().d3po_gamma_sample(_p(_u32(key)), C.c_uint32(comp), C.c_double(alpha))


def gmm_site_keys(jax_key, B, p):
    out = np.empty(6, np.uint32)
    lib().d3po_gmm_site_keys(_p(_u32(jax_key)), C.c_uint32(B), C.c_uint32(p), _p(out))
    return out.reshape(3, 2)


def gmm_px_latents(spec, alpha_log, jax_key, B, p):
    g = np.empty(spec.K, np.float64)
    eps = np.empty(spec.K * spec.d, np.float32)
    sigs = np.empty(spec.K * spec.d, np.float32)
    lib().d3po_gmm_px_latents(C.byref(spec), _p(_f32(alpha_log)), _p(_u32(jax_key)), C.c_uint32(B), C.c_uint32(p),
                              _p(g), _p(eps), _p(sigs))
    return g, eps.reshape(spec.K, spec.d), sigs.reshape(spec.K, spec.d)


def gmm_px_loss_grad_given(spec, alpha_log, mus_loc, x, g, eps, sigs, mask=1.0):
    P = spec.K + spec.K * spec.d
    grad = np.empty(P, np.float32)
    g = np.ascontiguousarray(g, np.float64)
    L = lib().d3po_gmm_px_loss_grad_given(C.byref(spec), _p(_f32(alpha_log)), _p(_f32(mus_loc)), _p(_f32(x)), _p(g),
                                          _p(_f32(eps)), _p(_f32(sigs)), C.c_float(mask), _p(grad))
    return float(L), grad


def gmm_px_grads(spec, params, Xb, jax_key, mask=None):
    B = Xb.shape[0]
    P = spec.K + spec.K * spec.d
    px_loss = np.empty(B, np.float32)
    px_grads = np.empty((B, P), np.float32)
    factor = C.c_float()
    m = None if mask is None else _f32(mask)
    n = lib().d3po_gmm_px_grads(C.byref(spec), _p(_f32(params)), _p(_f32(Xb)), None if m is None else _p(m), C.c_int(B),
                                _p(_u32(jax_key)), _p(px_loss), _p(px_grads), C.byref(factor))
    return px_loss, px_grads, n, factor.value


def gmm_evaluate(spec, params, Xb, jax_key):
    """DPSVI.evaluate for the mixture model; jax_key = convert(split(state.rng_key, 1)[0])."""
    Xb = _f32(Xb)
    return float(lib().d3po_gmm_evaluate(C.byref(spec), _p(_f32(params)), _p(Xb), C.c_int(Xb.shape[0]), _p(_u32(jax_key))))


# ------------------------------------------------------------------ VAE step (config 5)
def vae_spec(D, H, Z, scale=1.0, obs_scale=1.0, H2=0):
    """H2 > 0: a second hidden layer on each side (BASELINE config 5's 784 -> [400, 200] -> 50 variant)."""
    return VaeSpec(D, H, Z, scale, 1.0 / obs_scale, H2)


def vae_leaf_sizes(D, H, Z, H2=0):
    """Sizes of the parameter leaves in tree_flatten order (decoder layers, encoder layers, Wl, bl, Ws, bs)."""
    hs = [H] + ([H2] if H2 else [])
    dims = [Z] + hs[::-1] + [D]
    sizes = []
    for i, o in zip(dims[:-1], dims[1:]):          # decoder z -> .. -> D
        sizes += [i * o, o]
    dims = [D] + hs
    for i, o in zip(dims[:-1], dims[1:]):          # encoder x -> .. -> last hidden
        sizes += [i * o, o]
    return sizes + [hs[-1] * Z, Z, hs[-1] * Z, Z]


def vae_num_params(spec):
    return int(lib().d3po_vae_num_params(C.byref(spec)))


def vae_step_sums(spec, params, X, eps, clip, mask=None):
    """(sums[P + 2], norms[B], px_loss[B]) with explicit per-example gradients (examples/vae.py:65-153)."""
    B = X.shape[0]
    P = vae_num_params(spec)
    sums = np.empty(P + 2, np.float32)
    norms = np.empty(B, np.float32)
    px_loss = np.empty(B, np.float32)
    m = None if mask is None else _f32(mask)
    lib().d3po_vae_step_sums(C.byref(spec), _p(_f32(params)), _p(_f32(X)), None if m is None else _p(m), C.c_int(B),
                             _p(_f32(eps)), C.c_float(clip), _p(sums), _p(norms), _p(px_loss))
    return sums, norms, px_loss


def vae_evaluate(spec, params, X, jax_key):
    """DPSVI.evaluate for the VAE; spec.scale = handlers.scale x N / B; jax_key = convert(split(state.rng_key, 1)[0])."""
    X = _f32(X)
    return float(lib().d3po_vae_evaluate(C.byref(spec), _p(_f32(params)), _p(X), C.c_int(X.shape[0]), _p(_u32(jax_key))))
