"""The kernels the benchmark times, DIRECTLY against the CPU oracle at their production shape.

`k_logreg_chain` only exists for d = 512 (with or without the intercept column); the small-shape tests in
test_gpu_dpsvi.py therefore exercise the generic template.  Here `DPSVI.run_steps` runs BASELINE config 2's shape
(d = 512, B = 4096, N >= 1e5) for >= 140 steps -- i.e. across the 128-step boundary between two chained launches -- and
every loss, the final key and the final parameters are compared with the oracle's restatement of
`d3p/svi.py:395-434` driven by `examples/logistic_regression.py:149-160`'s loop:

  (i)   plain                       k_logreg_chain<PLIST=0, ICPT=0, XCHG=0>   vs  O.logreg_run_feistel
  (ii)  intercept                   k_logreg_chain<ICPT=1>                     vs  per-step O.logreg_update
  (iii) Poisson batches             k_logreg_chain<PLIST=1>                    vs  per-step O.logreg_update (masked)
  (iv)  data-parallel exchange      k_logreg_chain<XCHG=1> (world 1: exchange with itself, 140 steps; two virtual ranks on two
                                    streams in a child process, launches of 2 steps)   vs  the single-process oracle trajectory
  (v)   B = 32768 x 5 steps         the throughput-regime kernel                vs  O.logreg_run_feistel

Tolerances (fp32 path, stated once): losses rtol 5e-5, final key bit-exact, parameters rtol 2e-4 (atol 2e-5: Adam steps of
1e-2 on parameters that cross zero).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOSS_RTOL, PARAM_RTOL, PARAM_ATOL = 5e-5, 2e-4, 2e-5
THREADS = max(1, min(16, os.cpu_count() or 1))


def np_(t):
    return t.detach().cpu().numpy()


def _table(N, d, seed):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, d, generator=g)
    y = (torch.rand(N, generator=g) < 0.5).float()
    return X, y


def _svi(d, icpt, N, sigma=0.7, lr=1e-2):
    from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.svi import DPSVI
    model = LogisticRegression(d, prior_scale=1.0, intercept=icpt, intercept_prior_scale=2.0)
    return DPSVI(model, AutoDiagonalNormal(model), Adam(lr), Trace_ELBO(), 1.0, sigma, N=N)


def _state(svi, key, D, N):
    from d3p_amd.svi import DPSVIState
    p = torch.cat([torch.zeros(D), torch.full((D,), -2.0)]).cuda()
    return DPSVIState(svi.optim.init(p), key, float(N))


def _oracle_state(O, seed, D):
    return O.LogregState(O.PRNGKey(seed), D, np.zeros(D, np.float32), np.full(D, -2.0, np.float32))


def _compare(new_st, losses, ost, elosses, steps):
    assert bool(torch.isfinite(losses).all())
    np.testing.assert_allclose(np_(losses), np.asarray(elosses, np.float32), rtol=LOSS_RTOL)
    assert np.array_equal(np_(new_st.rng_key).ravel(), ost.key)
    assert int(new_st.optim_state[0]) == steps
    np.testing.assert_allclose(np_(new_st.optim_state[1]), ost.params, rtol=PARAM_RTOL, atol=PARAM_ATOL)


def _oracle_losses_feistel(O, spec, hy, ost, X, y, bkey, first, B, steps):
    """Every step's loss from the oracle's C run loop (one call per step, so each loss is seen)."""
    out = []
    for t in range(steps):
        out.append(O.logreg_run_feistel(spec, hy, ost, X, y, bkey, first + t, B, 1, threads=THREADS))
    return out


def test_chain_kernel_plain_vs_oracle_across_a_launch_boundary(gpu, O):
    """(i) d = 512, B = 4096, N = 1e5, 150 steps = launches of 128 + 22 steps."""
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, B, steps, first = 100_000, 512, 4096, 150, 3
    X, y = _table(N, d, 11)
    svi = _svi(d, False, N)
    st = _state(svi, rng.PRNGKey(3), d, N)
    _, gb = subsample_batchify_data((X.cuda(), y.cuda()), B)
    new_st, losses = svi.run_steps(st, gb, rng.PRNGKey(4), first, steps)
    assert svi.last_run_status() == (False, False)

    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 3, d)
    el = _oracle_losses_feistel(O, spec, hy, ost, X.numpy(), y.numpy(), O.PRNGKey(4), first, B, steps)
    _compare(new_st, losses, ost, el, steps)


def test_chain_kernel_intercept_vs_oracle_across_a_launch_boundary(gpu, O):
    """(ii) examples/logistic_regression.py:49-66's shape: d = 512 + intercept (D = 513), B = 4096, 140 steps."""
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, B, steps, first = 100_000, 512, 4096, 140, 0
    X, y = _table(N, d, 12)
    svi = _svi(d, True, N)
    st = _state(svi, rng.PRNGKey(5), d + 1, N)
    _, gb = subsample_batchify_data((X.cuda(), y.cuda()), B)
    new_st, losses = svi.run_steps(st, gb, rng.PRNGKey(6), first, steps)
    assert svi.last_run_status() == (False, False)

    Xn, yn = X.numpy(), y.numpy()
    spec = O.logreg_spec(d, True, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 5, d + 1)
    el = []
    for t in range(steps):
        idx = O.feistel_sample(O.fold_in(O.PRNGKey(6), first + t), N, B)
        el.append(O.logreg_update(spec, hy, ost, Xn[idx], yn[idx])[0])
    _compare(new_st, losses, ost, el, steps)


def test_chain_kernel_poisson_batches_vs_oracle_across_a_launch_boundary(gpu, O):
    """(iii) Poisson batches (q = 4096 / N, max_batch_size = the 0.99 quantile; d3p/minibatch.py:42-133) through the dense
    position lists of the PLIST instantiation, 140 steps."""
    import scipy.stats
    import d3p_amd.random as rng
    from d3p_amd.minibatch import poisson_batchify_data
    N, d, steps, first = 100_000, 512, 140, 1
    q = 4096 / N
    maxB = int(scipy.stats.poisson(N * q).ppf(0.99))
    X, y = _table(N, d, 13)
    svi = _svi(d, False, N)
    st = _state(svi, rng.PRNGKey(7), d, N)
    _, gb = poisson_batchify_data((X.cuda(), y.cuda()), q, 0.99)
    new_st, losses = svi.run_steps(st, gb, rng.PRNGKey(8), first, steps)
    assert svi.last_run_status() == (False, False)

    Xn, yn = X.numpy(), y.numpy()
    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 7, d)
    el, counts = [], []
    for t in range(steps):
        idx, nsel, nvalid = O.poisson_select(O.fold_in(O.PRNGKey(8), first + t), np.float32(q), N, maxB)
        counts.append(nvalid)
        mask = (np.arange(maxB) < nvalid).astype(np.float32)
        el.append(O.logreg_update(spec, hy, ost, Xn[idx], yn[idx], mask)[0])
    assert min(counts) < maxB            # the masks are not all-true: the lists really are ragged
    _compare(new_st, losses, ost, el, steps)


@pytest.mark.parametrize("icpt", [False, True])
def test_chain_kernel_exchange_on_one_rank_vs_oracle_across_a_launch_boundary(gpu, O, icpt):
    """(iv, first half) the XCHG instantiation -- since round 4 the 16-wave UPDATER form: the last arrivers of the 8 arrival groups
    fold the sums, run the exchange, apply noise + Adam once and publish the state as tagged words the next step polls -- with a
    world of one rank, 140 steps at B = 4096 (across the launch boundary: nothing is pending between launches in this form), without
    and with the intercept column (D = 513: 1030 accumulator columns, the odd latent layout)."""
    import d3p_amd._lib as L
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    N, d, B, steps, first = 100_000, 512, 4096, 140, 2
    D = d + int(icpt)
    X, y = _table(N, d, 14)
    svi = _svi(d, icpt, N)
    st = _state(svi, rng.PRNGKey(9), D, N)
    eng = ddist.FusedHipEngine(svi, X.cuda(), y.cuda(), N, 0, N, L.D3P_BATCH_FEISTEL, B)
    comm = ddist.XchgComm(2 * D + 4)
    try:
        new_st, losses = ddist.run_steps_native(eng, st, rng.PRNGKey(10), first, steps, comm=comm)
        torch.cuda.synchronize()
        code, _ = ddist.native_run_status(eng)
    finally:
        comm.close()
    assert code == 0, L.describe_abort(code)

    spec = O.logreg_spec(d, icpt, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 9, D)
    if icpt:
        el, Xn, yn = [], X.numpy(), y.numpy()
        for t in range(steps):
            idx = O.feistel_sample(O.fold_in(O.PRNGKey(10), first + t), N, B)
            el.append(O.logreg_update(spec, hy, ost, Xn[idx], yn[idx])[0])
    else:
        el = _oracle_losses_feistel(O, spec, hy, ost, X.numpy(), y.numpy(), O.PRNGKey(10), first, B, steps)
    _compare(new_st, losses, ost, el, steps)


_TWO_RANK_CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import d3p_amd._lib as L
import d3p_amd.random as rng
from d3p_amd import dist as ddist
from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
from d3p_amd.svi import DPSVI, DPSVIState
N, d, B, steps, first, world = 100_000, 512, 4096, 12, 2, 2
g = torch.Generator().manual_seed(15)
X = torch.randn(N, d, generator=g).cuda(); y = (torch.rand(N, generator=g) < 0.5).float().cuda()
model = LogisticRegression(d)
svi = DPSVI(model, AutoDiagonalNormal(model), Adam(1e-2), Trace_ELBO(), 1.0, 0.7, N=N)
st0 = DPSVIState(svi.optim.init(torch.cat([torch.zeros(d), torch.full((d,), -2.0)]).cuda()), rng.PRNGKey(21), float(N))
comms = ddist.XchgComm.local_group(world, 2 * d + 4)
streams = ddist.concurrent_streams(world)
engines, results = [], []
for r in range(world):
    lo, hi = ddist.shard_rows(N, r, world)
    engines.append(ddist.FusedHipEngine(svi, X[lo:hi], y[lo:hi], N, lo, hi, L.D3P_BATCH_FEISTEL, B))
torch.cuda.synchronize()
# 6 launches of 2 steps per rank, every one enqueued without a host synchronisation: call k of a rank continues from the state
# call k - 1 of the same rank returned (same stream => ordered); the ranks only meet inside the exchange
states = [st0] * world
losses = [[] for _ in range(world)]
for k in range(steps // 2):
    for r in range(world):
        with torch.cuda.stream(streams[r]):
            states[r], l = ddist.run_steps_native(engines[r], states[r], rng.PRNGKey(22), first + 2 * k, 2, comm=comms[r])
            losses[r].append(l)
torch.cuda.synchronize()
results = [(states[r], torch.cat(losses[r])) for r in range(world)]
codes = [ddist.native_run_status(e)[0] for e in engines]
for c in comms:
    c.close()
assert codes == [0] * world, [L.describe_abort(c) for c in codes]
for st, losses in results[1:]:
    assert torch.equal(st.optim_state[1], results[0][0].optim_state[1]) and torch.equal(losses, results[0][1])
    assert torch.equal(st.rng_key, results[0][0].rng_key)
st, losses = results[0]
np.savez(sys.argv[1], losses=losses.cpu().numpy(), params=st.optim_state[1].cpu().numpy(),
         key=st.rng_key.cpu().numpy().ravel(), step=int(st.optim_state[0]))
''' % (ROOT,)


@pytest.mark.coresident
def test_chain_kernel_exchange_between_two_virtual_ranks_vs_oracle(gpu, O):
    """(iv, second half) two row-sharded ranks on two streams of ONE GPU, exchange inside the chained launch, at B = 4096
    (about 2048 positions per rank): the ranks' launches must be co-resident on the one GPU (tests/test_dist.py explains), so
    the child process runs 6 launches of 2 steps per rank, all enqueued without a host synchronisation -- the same kernel
    instantiation as a production rank's 128-step launches.  Replicas bitwise identical; trajectory = the oracle's single-process one (per-example
    noise is keyed by the GLOBAL batch position, the noise is added once after the exchange)."""
    N, d, B, steps, first = 100_000, 512, 4096, 12, 2
    with tempfile.NamedTemporaryFile(suffix=".npz") as f:
        r = subprocess.run([sys.executable, "-c", _TWO_RANK_CODE, f.name], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        got = np.load(f.name)
        losses, params, key, step = got["losses"], got["params"], got["key"], int(got["step"])
    g = torch.Generator().manual_seed(15)
    X = torch.randn(N, d, generator=g).numpy()
    y = (torch.rand(N, generator=g) < 0.5).float().numpy()
    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 21, d)
    el = _oracle_losses_feistel(O, spec, hy, ost, X, y, O.PRNGKey(22), first, B, steps)
    np.testing.assert_allclose(losses, np.asarray(el, np.float32), rtol=LOSS_RTOL)
    assert np.array_equal(key, ost.key) and step == steps
    np.testing.assert_allclose(params, ost.params, rtol=PARAM_RTOL, atol=PARAM_ATOL)


def test_throughput_regime_kernel_vs_oracle(gpu, O):
    """(v) B = 32768 (the per-GPU batch of `roofline.large_batch`; BASELINE configs[3]'s global batch), d = 512, 5 steps."""
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, B, steps, first = 100_000, 512, 32768, 5, 0
    X, y = _table(N, d, 16)
    svi = _svi(d, False, N)
    st = _state(svi, rng.PRNGKey(31), d, N)
    _, gb = subsample_batchify_data((X.cuda(), y.cuda()), B)
    new_st, losses = svi.run_steps(st, gb, rng.PRNGKey(32), first, steps)
    assert svi.last_run_status() == (False, False)

    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 31, d)
    el = _oracle_losses_feistel(O, spec, hy, ost, X.numpy(), y.numpy(), O.PRNGKey(32), first, B, steps)
    _compare(new_st, losses, ost, el, steps)


@pytest.mark.parametrize("icpt,B", [(False, 32768), (True, 32768), (False, 4096)])
def test_aligned_gradients_column_sums_beyond_2048_C_vs_oracle(gpu, O, icpt, B):
    """A column whose clipped per-example gradients all point the same way: its sum over the batch is only bounded by B C
    (svi.py:343-346 before the division), i.e. B 2^40 in the accumulator's fixed point -- 2^55 at B = 32768.  Round 3's 16-wave
    prologue converted that int64 with a bit-pattern trick that is exact below 2^51 (|sum| < 2048 C) and silently distorted larger
    sums (advisor finding, round 3).  Here: all labels 1, one constant feature of 100 (no intercept) or features of 0.01 and the
    intercept column (the shape of examples/logistic_regression.py:49-66 on small-scale data) -- the column's sum is ~ 0.97 B C
    in both.  Adam normalises a gradient's magnitude away (the first update is lr sign(g)), so the check is on the MOMENTS
    the gradient leaves behind, against the oracle: m = 0.1 g, v = 0.001 g^2 after one step."""
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, steps = 40_000, 512, 2
    g = torch.Generator().manual_seed(5)
    if icpt:
        X = 0.01 * torch.randn(N, d, generator=g)
    else:
        X = torch.randn(N, d, generator=g)
        X[:, 0] = 100.0
    y = torch.ones(N)
    D = d + int(icpt)
    svi = _svi(d, icpt, N, sigma=0.0)    # no Gaussian-mechanism noise: the moments are the clipped mean gradient's alone
    st = _state(svi, rng.PRNGKey(91), D, N)
    _, gb = subsample_batchify_data((X.cuda(), y.cuda()), B)
    new_st, losses = svi.run_steps(st, gb, rng.PRNGKey(92), 0, steps)
    assert svi.last_run_status() == (False, False)
    spec = O.logreg_spec(d, icpt, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.0, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 91, D)
    el = _oracle_losses_feistel(O, spec, hy, ost, X.numpy(), y.numpy(), O.PRNGKey(92), 0, B, steps) if not icpt else None
    if icpt:
        el, Xn, yn = [], X.numpy(), y.numpy()
        for t in range(steps):
            idx = O.feistel_sample(O.fold_in(O.PRNGKey(92), t), N, B)
            el.append(O.logreg_update(spec, hy, ost, Xn[idx], yn[idx])[0])
    col = D - 1 if icpt else 0
    m_dev, v_dev = np_(new_st.optim_state[2]), np_(new_st.optim_state[3])
    # the column this test is about: after two steps m = 0.19 g with g = obs_scale x (column sum / B), so the sum in units of C is
    assert abs(ost.m[col]) / 0.19 / N * B > 1.1 * 2048
    np.testing.assert_allclose(m_dev[col], ost.m[col], rtol=1e-4)
    np.testing.assert_allclose(v_dev[col], ost.v[col], rtol=2e-4)
    np.testing.assert_allclose(m_dev, ost.m, rtol=2e-3, atol=2e-4 * float(np.abs(ost.m).max()))
    _compare(new_st, losses, ost, el, steps)


@pytest.mark.parametrize("d,icpt,B", [(64, False, 3000), (192, False, 2500), (256, False, 4096), (384, False, 1800), (256, True, 2000),
                                      (4, True, 200)])
def test_generic_kernel_widths_chained_run_vs_oracle(gpu, O, d, icpt, B):
    """Widths the lean kernel does not take, through the generic kernel's chained form (k_logreg_main<V = 1, NK, MODE 3>) with the
    launch geometry of round 4 -- one / two / four columns per lane and half as 16- / 16- / 8-wave workgroups, the scalar-load form
    for every row that is not a full 16-byte-load tile; (4, True, 200) is examples/logistic_regression.py's default shape --:
    10 steps each against per-step O.logreg_update."""
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    N, steps, first = 20_000, 10, 3
    X, y = _table(N, d, 300 + d)
    svi = _svi(d, icpt, N)
    D = d + int(icpt)
    st = _state(svi, rng.PRNGKey(63), D, N)
    _, gb = subsample_batchify_data((X.cuda(), y.cuda()), B)
    new_st, losses = svi.run_steps(st, gb, rng.PRNGKey(64), first, steps)
    assert svi.last_run_status() == (False, False)
    Xn, yn = X.numpy(), y.numpy()
    spec = O.logreg_spec(d, icpt, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 63, D)
    el = []
    for t in range(steps):
        idx = O.feistel_sample(O.fold_in(O.PRNGKey(64), first + t), N, B)
        el.append(O.logreg_update(spec, hy, ost, Xn[idx], yn[idx], None)[0])
    _compare(new_st, losses, ost, el, steps)


@pytest.mark.parametrize("B,icpt,sampler", [(20, False, "feistel"), (100, False, "feistel"), (33, True, "feistel"),
                                            (700, True, "poisson"), (1500, False, "poisson")])
def test_chain_kernel_small_and_ragged_grids_vs_oracle(gpu, O, B, icpt, sampler):
    """The 16-wave form away from its benchmark shape: one, four and a few dozen workgroups per step (fewer arrival groups than
    8; waves without an item), the intercept's extra columns together with Poisson position lists, a padded Poisson batch whose
    item count changes from step to step -- 12 steps each against per-step O.logreg_update."""
    import scipy.stats
    import d3p_amd.random as rng
    from d3p_amd.minibatch import poisson_batchify_data, subsample_batchify_data
    N, d, steps, first = 20_000, 512, 12, 4
    X, y = _table(N, d, 100 + B)
    svi = _svi(d, icpt, N)
    D = d + int(icpt)
    st = _state(svi, rng.PRNGKey(61), D, N)
    if sampler == "feistel":
        _, gb = subsample_batchify_data((X.cuda(), y.cuda()), B)
        maxB = B
    else:
        q = B / N
        maxB = int(scipy.stats.poisson(N * q).ppf(0.99))
        _, gb = poisson_batchify_data((X.cuda(), y.cuda()), q, 0.99)
    new_st, losses = svi.run_steps(st, gb, rng.PRNGKey(62), first, steps)
    assert svi.last_run_status() == (False, False)
    Xn, yn = X.numpy(), y.numpy()
    spec = O.logreg_spec(d, icpt, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 61, D)
    el = []
    for t in range(steps):
        bk = O.fold_in(O.PRNGKey(62), first + t)
        if sampler == "feistel":
            idx, mask = O.feistel_sample(bk, N, B), None
        else:
            idx, _, nvalid = O.poisson_select(bk, np.float32(B / N), N, maxB)
            mask = (np.arange(maxB) < nvalid).astype(np.float32)
        el.append(O.logreg_update(spec, hy, ost, Xn[idx], yn[idx], mask)[0])
    _compare(new_st, losses, ost, el, steps)


def _shard_oracle_trajectory(O, spec, hy, ost, Xn, yn, bkey, first, N, B, steps, lo, hi):
    """The oracle's trajectory of ONE shard of a row-sharded job whose other ranks add all-zero rows to the exchange (what
    d3p_xchg_simulate_peers plays): every step samples the GLOBAL batch (util.py:216-301); all B examples count (a Feistel batch
    has no padding: n = B, factor 1, svi.py:305), the examples whose rows lie outside [lo, hi) contribute zero loss and zero
    gradient; per-example noise stays keyed by the global batch position.  d3po_logreg_update's composition of the checker's stages
    (svi.py:395-434), with the other ranks' rows zeroed between the per-example gradients and the clip."""
    D = spec.d + spec.intercept
    el, counts = [], []
    for t in range(steps):
        idx = O.feistel_sample(O.fold_in(bkey, first + t), N, B)
        own = (idx >= lo) & (idx < hi)
        counts.append(int(own.sum()))
        ks = np.asarray(O.split(ost.key, 3)).reshape(3, 16)
        jax_key = O.convert_to_jax_rng_key(ks[1])
        eps = np.zeros((B, D), np.float32)                      # (only the shard's rows are looked at)
        for p in np.flatnonzero(own):
            eps[p] = O.tf_normal(O.px_sample_key(jax_key, B, int(p)), D)
        px_loss, px_grads, n, factor = O.logreg_px_grads(spec, ost.params[:D], ost.params[D:], Xn[idx], yn[idx], eps)
        assert n == B and factor == 1.0
        px_loss[~own] = 0.0
        px_grads[~own] = 0.0
        loss, avg = O.combine(O.clip_rows(px_grads, hy.clip), px_loss)
        g = O.perturb(ks[2], avg, [D, D], hy.dp_scale, hy.clip, n, 1.0 / spec.inv_obs, factor)
        ost.params, ost.m, ost.v = O.adam(ost.params, ost.m, ost.v, g, ost.step.value, hy.lr, hy.b1, hy.b2, hy.adam_eps)
        ost.step.value += 1
        ost.key = ks[0].copy()
        el.append(loss)
    return el, counts


@pytest.mark.parametrize("B,nw", [(171, 6), (280, 9), (531, 17), (1366, 43)])
def test_updater_form_alone_on_grids_off_the_xcd_multiple_vs_oracle(gpu, O, B, nw):
    """The data-parallel UPDATER form on grids that are NOT a multiple of the 8 XCDs -- 6, 9, 17, 43 workgroups per step: the
    item counts of a rank of 512 / 3, 560 / 2, 4245 / 8 (a Poisson batch padded to its 0.99 quantile over 8 ranks) and 4096 / 3 --
    with a world of one rank (the exchange with itself: no second stream, nothing that depends on the box), 140 steps across the
    launch boundary, against the ORACLE trajectory: every loss, the final key, the parameters and the optimiser's step counter
    (svi.py:379-393, :432-434).  Round 4's driver run returned step 12 of 16 at nw = 9: the counter was a per-step plain store from
    a workgroup on a different XCD every step; it is k_flush's now."""
    import d3p_amd._lib as L
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    N, d, steps, first = 20_000, 512, 140, 1
    X, y = _table(N, d, 300 + nw)
    svi = _svi(d, False, N)
    st = _state(svi, rng.PRNGKey(31), d, N)
    eng = ddist.FusedHipEngine(svi, X.cuda(), y.cuda(), N, 0, N, L.D3P_BATCH_FEISTEL, B)
    comm = ddist.XchgComm(2 * d + 4)
    try:
        new_st, losses = ddist.run_steps_native(eng, st, rng.PRNGKey(32), first, steps, comm=comm)
        torch.cuda.synchronize()
        code, _ = ddist.native_run_status(eng)
    finally:
        comm.close()
    assert code == 0, L.describe_abort(code)
    assert eng.chain_grid(True) == (nw, 16)
    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 31, d)
    el = _oracle_losses_feistel(O, spec, hy, ost, X.numpy(), y.numpy(), O.PRNGKey(32), first, B, steps)
    _compare(new_st, losses, ost, el, steps)


@pytest.mark.coresident
@pytest.mark.parametrize("world,B,nw", [(3, 512, 6), (2, 560, 9), (8, 4245, 17), (3, 4096, 43)])
def test_updater_form_as_a_rank_of_many_on_grids_off_the_xcd_multiple_vs_oracle(gpu, O, world, B, nw):
    """The same grids as rank 1 of a 3 / 2 / 8 / 3-rank job (d3p_xchg_simulate_peers plays the others: all-zero rows, one
    workgroup on a second stream): the updaters' sends and their waits for 2 / 1 / 7 / 2 rows, row-sharded batches with position
    lists whose length changes from step to step, 140 steps -- against the oracle's trajectory of that shard (the other ranks'
    examples masked), not only against another HIP run."""
    import d3p_amd._lib as L
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    N, d, steps, first, rank = 20_000, 512, 140, 1, 1
    X, y = _table(N, d, 400 + nw)
    svi = _svi(d, False, N)
    st = _state(svi, rng.PRNGKey(33), d, N)
    lo, hi = ddist.shard_rows(N, rank, world)
    comms = ddist.XchgComm.local_group(world, 2 * d + 4)
    side = ddist.concurrent_streams(1)[0]
    try:
        eng = ddist.FusedHipEngine(svi, X[lo:hi].cuda(), y[lo:hi].cuda(), N, lo, hi, L.D3P_BATCH_FEISTEL, B)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            comms[rank].simulate_peers(steps)
        new_st, losses = ddist.run_steps_native(eng, st, rng.PRNGKey(34), first, steps, comm=comms[rank])
        torch.cuda.synchronize()
        code, _ = ddist.native_run_status(eng)
    finally:
        for c in comms:
            c.close()
    assert code == 0, L.describe_abort(code)
    assert eng.chain_grid(True) == (nw, 16)
    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 33, d)
    el, counts = _shard_oracle_trajectory(O, spec, hy, ost, X.numpy(), y.numpy(), O.PRNGKey(34), first, N, B, steps, lo, hi)
    assert min(counts) < max(counts)
    _compare(new_st, losses, ost, el, steps)


@pytest.mark.parametrize("B,nw", [(1100, 35), (2900, 91)])
def test_chain_kernel_single_rank_grids_off_the_xcd_multiple_vs_oracle(gpu, O, B, nw):
    """The single-rank 16-wave form at 35 and 91 workgroups per step (arrival group = blockIdx.x % 8 with a group size that
    changes from step to step), 140 steps against the oracle."""
    import d3p_amd._lib as L
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    N, d, steps, first = 50_000, 512, 140, 0
    X, y = _table(N, d, 500 + nw)
    svi = _svi(d, False, N)
    st = _state(svi, rng.PRNGKey(35), d, N)
    eng = ddist.FusedHipEngine(svi, X.cuda(), y.cuda(), N, 0, N, L.D3P_BATCH_FEISTEL, B)
    new_st, losses = ddist.run_steps_native(eng, st, rng.PRNGKey(36), first, steps, comm=None)
    torch.cuda.synchronize()
    assert ddist.native_run_status(eng) == (0, False)
    assert eng.chain_grid(False) == (nw, 16)
    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 35, d)
    el = _oracle_losses_feistel(O, spec, hy, ost, X.numpy(), y.numpy(), O.PRNGKey(36), first, B, steps)
    _compare(new_st, losses, ost, el, steps)


@pytest.mark.parametrize("B", [64, 4096])
def test_runs_that_continue_each_other_walk_the_trajectory_of_one_run(gpu, O, B):
    """`run_steps` is a function of the state (svi.py:395-434 returns a NEW state; the epoch loop of
    examples/logistic_regression.py:186-191 calls one jit(fori_loop) per epoch on the state the last one returned): a sequence of
    runs that continue each other -- 5, 20, 3, 140, 1, 30, 129 steps: runs that end inside and across the 128-step launches --
    must walk the trajectory of ONE run of the total length: every loss, the parameters, the step counter, and the state key of
    sum(steps) successive split(key, 3)[0] (svi.py:208-211) bit for bit against the oracle.  Then a run from ANOTHER key on the
    same workspace and a run that re-starts from the first state: nothing a run leaves in the workspace may leak into the next
    (written for the key chain's look-ahead, tools/experiments/r05_key_chain_lookahead.patch; kept as the test of the property)."""
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    N, d = 20_000, 512
    X, y = _table(N, d, 72)
    svi = _svi(d, False, N)
    st0 = _state(svi, rng.PRNGKey(91), d, N)
    _, gb = subsample_batchify_data((X.cuda(), y.cuda()), B)
    lens = [5, 20, 3, 140, 1, 30, 129]
    total = sum(lens)
    one_state, one_losses = svi.run_steps(st0, gb, rng.PRNGKey(92), 0, total)
    one_losses = one_losses.clone()
    one_params = one_state.optim_state[1].clone()
    one_key = one_state.rng_key.clone()
    st, first, parts = st0, 0, []
    for n in lens:
        st, l = svi.run_steps(st, gb, rng.PRNGKey(92), first, n)
        assert svi.last_run_status() == (False, False)
        parts.append(l.clone())
        first += n
        key = O.PRNGKey(91)
        for _ in range(first):
            key = O.split(key, 3)[0]
        assert np.array_equal(np_(st.rng_key).ravel(), np.asarray(key).ravel()), f"key after {first} steps"
        assert int(st.optim_state[0]) == first
    assert torch.equal(st.rng_key, one_key)
    np.testing.assert_allclose(np_(torch.cat(parts)), np_(one_losses), rtol=1e-6)
    np.testing.assert_allclose(np_(st.optim_state[1]), np_(one_params), rtol=1e-5, atol=1e-7)
    # the oracle's trajectory over the first three runs (28 steps: every one of them started from a look-ahead record but the first)
    spec = O.logreg_spec(d, False, 1.0, 2.0, lik_scale=N, obs_scale=N)
    hy = O.Hyper(1.0, 0.7, 1e-2, 0.9, 0.999, 1e-8)
    ost = _oracle_state(O, 91, d)
    el = _oracle_losses_feistel(O, spec, hy, ost, X.numpy(), y.numpy(), O.PRNGKey(92), 0, B, 28)
    np.testing.assert_allclose(np_(torch.cat(parts))[:28], np.asarray(el, np.float32), rtol=LOSS_RTOL)
    # another key on the same workspace, then the first state again: neither run may take the record that is there
    other = _state(svi, rng.PRNGKey(93), d, N)
    o_state, _ = svi.run_steps(other, gb, rng.PRNGKey(92), 0, 7)
    key = O.PRNGKey(93)
    for _ in range(7):
        key = O.split(key, 3)[0]
    assert np.array_equal(np_(o_state.rng_key).ravel(), np.asarray(key).ravel())
    again, l_again = svi.run_steps(st0, gb, rng.PRNGKey(92), 0, 25)
    assert torch.equal(l_again, torch.cat(parts)[:25]) or np.allclose(np_(l_again), np_(torch.cat(parts))[:25], rtol=1e-6)
    key = O.PRNGKey(91)
    for _ in range(25):
        key = O.split(key, 3)[0]
    assert np.array_equal(np_(again.rng_key).ravel(), np.asarray(key).ravel())


@pytest.mark.parametrize("steps", [129, 255, 259, 385])
def test_key_chain_links_across_launch_boundaries(gpu, O, steps):
    """The key chain of every batch of 128 steps but the first is made link by link in the tail of workgroup 0 of the launch
    before it, the running key travelling as self-validating tagged words (chain_step_ll): runs whose later batches have 1, 127,
    3 + 128 and 1 + 2 x 128 steps end with the state key of `steps` successive split(key, 3)[0] (svi.py:208-211), bit for bit,
    and with the step counter at `steps` -- at a small batch (two workgroups per step), where a link has the least time."""
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    N, d, B = 5000, 512, 64
    X, y = _table(N, d, 71)
    svi = _svi(d, False, N)
    st = _state(svi, rng.PRNGKey(81), d, N)
    _, gb = subsample_batchify_data((X.cuda(), y.cuda()), B)
    new_st, losses = svi.run_steps(st, gb, rng.PRNGKey(82), 0, steps)
    assert svi.last_run_status() == (False, False)
    assert bool(torch.isfinite(losses).all()) and int(new_st.optim_state[0]) == steps
    key = O.PRNGKey(81)
    for _ in range(steps):
        key = O.split(key, 3)[0]
    assert np.array_equal(np_(new_st.rng_key).ravel(), np.asarray(key).ravel())
