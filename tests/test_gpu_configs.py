"""BASELINE.json `configs` at their FULL sizes on one MI355X (the small-shape parity tests live beside each kernel's
own test file; these close the gap between those and the benchmark configurations):

  configs[3]  logistic regression d = 512, global batch 32768 row-sharded 8 ways -- 8 virtual ranks on one GPU;
  configs[2]  mixture model K = 16, d = 64, batch 8192 -- oracle on a slice + full-size properties;
  configs[4]  VAE 784 / 400 / 50, batch 4096 -- oracle on a slice + full-size properties;

plus the regression tests of the round-1 advisor finding: the one-launch-per-step kernel (MODE 2) with FEWER threads than
accumulator words (few examples per rank, wide rows) must clear every word of the next accumulator.

Tolerances are those of the per-kernel test files (stated there): per-example quantities rtol 1e-4 (GMM) / 5e-5 (VAE norms),
batch sums rtol 2e-4 + 2e-5 max, parameters after Adam rtol 2e-5."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def np_(t):
    return t.detach().cpu().numpy()


def _logreg_svi(d, n, lr=1e-2, sigma=0.6):
    from d3p_amd.models import Adam, AutoDiagonalNormal, LogisticRegression, Trace_ELBO
    from d3p_amd.svi import DPSVI
    model = LogisticRegression(d)
    return DPSVI(model, AutoDiagonalNormal(model), Adam(lr), Trace_ELBO(), 1.0, sigma, N=n)


def _run_virtual_ranks(svi, X, y, n, Bg, world, st0, bkey, first, steps, ddist, L):
    """`world` FusedHipEngines over disjoint row shards of ONE resident table (views, no copies); the int64 all-reduce is
    a sum by hand.  Returns (final states, engines, per-step example counts)."""
    engines = []
    for rk in range(world):
        lo, hi = ddist.shard_rows(n, rk, world)
        engines.append(ddist.FusedHipEngine(svi, X[lo:hi], y[lo:hi], n, lo, hi, L.D3P_BATCH_FEISTEL, Bg))
    for e in engines:
        e.begin(st0, bkey, first)
        e.plan(steps)
    counts = []
    P = engines[0].P
    for _ in range(steps):
        bufs = [e.local_sums() for e in engines]
        total = torch.stack(bufs).sum(dim=0)                  # what the int64 all-reduce computes
        counts.append(int(total.reshape(4, -1)[:, P + 1].sum()))     # 4 replicas x [P gradient columns | loss | count | ...]
        for e, b in zip(engines, bufs):
            b.copy_(total)
            e.finalize(b)
    return [e.end() for e in engines], engines, counts


# ------------------------------------------------------------------------------------------------ advisor regression
@pytest.mark.parametrize("n,d,Bg,world,steps", [(3000, 512, 16, 1, 10), (3000, 512, 64, 1, 10), (4000, 1024, 512, 4, 8),
                                               (2000, 2048, 256, 2, 7)])
def test_one_launch_step_with_fewer_threads_than_accumulator_words(gpu, n, d, Bg, world, steps):
    """MODE 2 zeroes the NEXT step's accumulator (4 x (P + 2) int64 words) with the threads of its compute workgroups; at
    d = 512 / B <= 64, d = 1024 / 128 examples per rank or d = 2048 / B = 256 there are fewer threads than words.  Words
    left uncleared keep the sums of step g - 3 -- an inflated count would silently shrink the noise scale C / n.  Checked
    against the two-kernel form (partial rows + finalize: no accumulators) over >= 7 steps (two accumulator rotations)."""
    import d3p_amd._lib as L
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    from d3p_amd.svi import DPSVIState
    r = np.random.default_rng(d + Bg)
    X = torch.tensor(r.normal(size=(n, d)).astype(np.float32)).cuda()
    y = torch.tensor((r.random(n) < 0.5).astype(np.float32)).cuda()
    svi = _logreg_svi(d, n)
    params = torch.cat([torch.zeros(d), torch.full((d,), -2.0)]).cuda()
    st0 = DPSVIState(svi.optim.init(params), rng.PRNGKey(10), float(n))
    bkey = rng.PRNGKey(20)
    single = ddist.HipEngine(svi, X, y, n, 0, n, L.D3P_BATCH_FEISTEL, Bg)
    ref_state, ref_losses = ddist.run_steps(single, st0, bkey, 3, steps)
    finals, engines, counts = _run_virtual_ranks(svi, X, y, n, Bg, world, st0, bkey, 3, steps, ddist, L)
    assert counts == [Bg] * steps                              # the count column holds THIS step's examples only
    for f, e in zip(finals, engines):
        assert torch.equal(f.rng_key, ref_state.rng_key) and int(f.optim_state[0]) == steps
        assert torch.equal(f.optim_state[1], finals[0].optim_state[1])
        np.testing.assert_allclose(np_(e.losses), np_(ref_losses), rtol=2e-5)
        np.testing.assert_allclose(np_(f.optim_state[1]), np_(ref_state.optim_state[1]), rtol=2e-5, atol=2e-6)


def test_nonfinite_partials_turn_the_run_into_nan_and_raise_the_status_word(gpu):
    """A diverged model (here: a feature column of Inf) must surface as NaN parameters and losses -- what the reference's
    float sums give -- not as finite garbage from a saturated float -> fixed-point conversion; the run's status word says so."""
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    from d3p_amd.svi import DPSVIState
    n, d, B = 4096, 512, 4096
    g = torch.Generator().manual_seed(0)
    X = torch.randn(n, d, generator=g)
    X[:, 7] = float("inf")
    X = X.cuda()
    y = (torch.rand(n, generator=g) < 0.5).float().cuda()
    svi = _logreg_svi(d, n)
    st0 = DPSVIState(svi.optim.init(torch.cat([torch.zeros(d), torch.full((d,), -2.0)]).cuda()), rng.PRNGKey(1), float(n))
    _, get_batch = subsample_batchify_data((X, y), B)
    st, losses = svi.run_steps(st0, get_batch, rng.PRNGKey(2), 0, 6)
    aborted, nonfinite = svi.last_run_status()
    assert nonfinite and not aborted
    assert torch.isnan(losses).all()
    assert torch.isnan(st.optim_state[1]).all()


def test_constructor_rejects_nonpositive_clip_and_bad_noise_scale(gpu):
    from d3p_amd.svi import DPSVI
    for bad in (-1.0, float("nan"), float("inf")):
        with pytest.raises(ValueError):
            DPSVI(None, None, None, None, bad, 1.0)
    for bad in (-0.5, float("nan"), float("inf")):
        with pytest.raises(ValueError):
            DPSVI(None, None, None, None, 1.0, bad)
    # C == 0 stays a ValueError at the first clip, as in the reference (svi.py:119-120)
    svi = _logreg_svi(8, 100)
    svi._clipping_threshold = 0.0
    with pytest.raises(ValueError):
        svi._clip_gradients(None, {"a": torch.ones(2, 3, device="cuda")})


# ------------------------------------------------------------------------------------------------ configs[3]
def test_config3_eight_virtual_ranks_d512_global_batch_32768(gpu, O):
    """BASELINE configs[3]: d = 512, global batch 32768, table row-sharded over 8 ranks (Feistel capacity 10^8 where the
    GPU holds the 205 GB table, 8 x 10^6 rows otherwise).  Every rank evaluates the same sampler and processes the
    positions whose rows it holds; the summed int64 accumulators give every rank the same update.  Against the
    single-rank run over the whole table: every position owned exactly once, replicas bitwise identical, parameters and
    losses equal to fp32 rounding (a different sharding groups the examples into different fp32 workgroup partials)."""
    import d3p_amd._lib as L
    import d3p_amd.random as rng
    from d3p_amd import dist as ddist
    from d3p_amd.svi import DPSVIState
    d, Bg, world, steps = 512, 32768, 8, 5
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    n = 100_000_000 if free > 230 * 2**30 else 8_000_000
    lib = L.load()
    X = torch.empty((n, d), dtype=torch.float32, device="cuda")
    y = torch.empty(n, dtype=torch.float32, device="cuda")
    L.check(lib.d3p_synth_logreg(L.stream_ptr(), 123, 0, n, d, L.ptr(X), L.ptr(y)))
    svi = _logreg_svi(d, n, lr=1e-3, sigma=1.0)
    params = torch.cat([torch.zeros(d), torch.full((d,), -2.0)]).cuda()
    st0 = DPSVIState(svi.optim.init(params), rng.PRNGKey(0), float(n))
    bkey = rng.PRNGKey(1)
    single = ddist.FusedHipEngine(svi, X, y, n, 0, n, L.D3P_BATCH_FEISTEL, Bg)
    ref_state, ref_losses = ddist.run_steps(single, st0, bkey, 0, steps)
    ref_losses = ref_losses.clone()
    finals, engines, counts = _run_virtual_ranks(svi, X, y, n, Bg, world, st0, bkey, 0, steps, ddist, L)
    assert counts == [Bg] * steps                              # every position is owned by exactly one rank
    for f, e in zip(finals, engines):
        assert torch.equal(f.rng_key, ref_state.rng_key) and int(f.optim_state[0]) == steps
        assert torch.equal(f.optim_state[1], finals[0].optim_state[1])          # bitwise identical replicas
        assert torch.equal(f.optim_state[2], finals[0].optim_state[2])
        assert torch.equal(e.losses, engines[0].losses)
    np.testing.assert_allclose(np_(engines[0].losses), np_(ref_losses), rtol=2e-5)
    np.testing.assert_allclose(np_(finals[0].optim_state[1]), np_(ref_state.optim_state[1]), rtol=2e-5, atol=2e-6)
    assert torch.isfinite(ref_losses).all()
    # the oracle's anchor at this size: step 0 of the run (Feistel indices over the full capacity, the 32768 gathered rows, the
    # reference dataflow with all B x P per-example gradients materialised) gives the loss the eight ranks computed
    spec = O.logreg_spec(d, False, 1.0, 1.0, lik_scale=n, obs_scale=n)
    ost = O.LogregState(O.PRNGKey(0), d, np.zeros(d, np.float32), np.full(d, -2.0, np.float32))
    idx = O.feistel_sample(O.fold_in(O.PRNGKey(1), 0), n, Bg)
    rows = torch.from_numpy(idx.astype(np.int64)).cuda()
    eloss, _ = O.logreg_update(spec, O.Hyper(1.0, 1.0, 1e-3, 0.9, 0.999, 1e-8), ost, np_(X[rows]), np_(y[rows]))
    assert abs(float(engines[0].losses[0]) - eloss) <= 5e-5 * abs(eloss)
    import warnings
    warnings.warn(f"test_config3: table of {n} rows x {d} columns ({n * (d + 1) * 4 / 1e9:.1f} GB; {free / 2**30:.0f} GiB were free): "
                  f"{'BASELINE configs[3] at full size' if n == 100_000_000 else 'reduced table (the GPU did not have 230 GiB free)'}",
                  UserWarning)
    del X, y
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ configs[2]
def test_config2_gmm_k16_d64_batch_8192(gpu, O):
    """BASELINE configs[2] at its batch size: per-example gradients of 64 spread-out positions of the B = 8192 batch
    against the oracle (the site keys of an example are functions of (B, position)); the fused update against the stage
    composition on the device; ||clipped mean|| <= C without noise; bitwise reproducible."""
    import d3p_amd.random as rng
    from tests.test_gpu_gmm_model import make_svi, problem, state_with
    B, K, d, N, Cc = 8192, 16, 64, 10**7, 20.0
    X, params = problem(B, K, d, 4)
    svi = make_svi(K, d, N, C=Cc, sigma=0.0, lr=1e-2)
    key = rng.PRNGKey(5)
    st = state_with(svi, key, params, N)
    Xt = torch.tensor(X).cuda()
    gradient_key = rng.split(key, 3)[1]                       # what update() hands to stage 1 (svi.py:413-416)
    _, px_loss, grads, n, f = svi._compute_per_example_gradients(st, gradient_key, Xt)
    assert float(n) == B and float(f) == 1.0
    G = np.concatenate([np_(grads["alpha_log"]), np_(grads["mus_loc"]).reshape(B, -1)], axis=1)
    spec = O.gmm_spec(K, d, 10.0, lik_scale=N, obs_scale=N)
    jax_key = O.convert_to_jax_rng_key(O.split(O.PRNGKey(5), 3)[1])
    pl = np_(px_loss)
    for p in list(range(0, B, 131)) + [B - 1]:
        g, eps, sigs = O.gmm_px_latents(spec, params[:K], jax_key, B, p)
        eL, eG = O.gmm_px_loss_grad_given(spec, params[:K], params[K:], X[p], g, eps, sigs)
        np.testing.assert_allclose(G[p], eG, rtol=1e-4, atol=1e-5 * np.abs(eG).max())
        assert abs(pl[p] - eL * N) <= 2e-5 * abs(eL * N)      # px_losses *= obs_scale * factor (svi.py:306)
    # fused update (no B x P tensor) == the five stages on the device; sigma = 0: the gradient is the clipped mean
    gout = torch.empty(K + K * d, device="cuda")
    s_f, l_f = svi._update_gmm_fused(st, Xt, _grad_out=gout)
    s_s, l_s = svi._update_staged(st, Xt)
    assert abs(float(l_f) - float(l_s)) <= 2e-5 * abs(float(l_s))
    np.testing.assert_allclose(np_(s_f.optim_state[1]), np_(s_s.optim_state[1]), rtol=1e-5, atol=1e-6)
    rows = torch.tensor(G).cuda()
    nrm = rows.norm(dim=1)
    clipped_mean = (rows * torch.clamp(Cc / nrm, max=1.0)[:, None]).mean(dim=0) * N    # x obs_scale, factor = 1
    np.testing.assert_allclose(np_(gout), np_(clipped_mean), rtol=2e-4, atol=2e-5 * float(clipped_mean.abs().max()))
    assert float(gout.norm()) <= Cc * N * (1 + 1e-5)
    assert float((nrm > Cc).float().mean()) > 0.0            # the threshold actually clips at this configuration
    s_f2, l_f2 = svi._update_gmm_fused(st, Xt)
    assert torch.equal(s_f2.optim_state[1], s_f.optim_state[1]) and float(l_f2) == float(l_f)


# ------------------------------------------------------------------------------------------------ configs[4]
@pytest.mark.parametrize("H2", [0, 200])
def test_config4_vae_784_400_50_batch_4096(gpu, O, H2):
    """BASELINE configs[4] at B = 4096, in the reference's shape (one hidden layer of 400, examples/vae.py:80-103; H2 = 0) and
    in the literal 784 -> [400, 200] -> 50 shape BASELINE.json names (H2 = 200, P = 819 284): norms and losses of a 64-example
    slice against the oracle's explicit per-example gradients; the clipped sums of that slice (selected by the mask, the
    other 4032 rows still go through every GEMM) against the oracle's; linearity of the sums over a split of the batch;
    ||clipped mean|| <= C; bitwise reproducible."""
    import d3p_amd._lib as lib
    from tests.test_gpu_vae import vae_problem
    B, D, H, Z = 4096, 784, 400, 50
    spec, P, params, X, eps = vae_problem(B, D, H, Z, 21, 0.03, H2)
    assert P == (819284 if H2 else 688884)
    L = lib.load()
    model = lib.VaeModel(D, H, Z, 1.0, 1.0, H2)
    ws = torch.empty(int(L.d3p_dpvi_vae_workspace(C.byref(model), B)), dtype=torch.uint8, device="cuda")
    pt, Xt, et = torch.tensor(params).cuda(), torch.tensor(X).cuda(), torch.tensor(eps).cuda()

    def run(mask, clip):
        sums = torch.empty(P + 2, device="cuda")
        norms = torch.empty(B, device="cuda")
        pxl = torch.empty(B, device="cuda")
        mt = None if mask is None else torch.tensor(mask).to(torch.uint8).cuda()
        lib.check(L.d3p_vae_step_sums(lib.stream_ptr(), C.byref(model), lib.ptr(pt), lib.ptr(Xt), lib.ptr(mt), B, lib.ptr(et),
                                      None, clip, lib.ptr(sums), lib.ptr(norms), lib.ptr(pxl), lib.ptr(ws), ws.numel()))
        torch.cuda.synchronize()
        return sums.clone(), norms.clone(), pxl.clone()

    _, norms0, _ = run(None, 1e30)
    clip = float(norms0.median())
    sel = np.arange(0, B, 64)                                  # 64 spread-out examples
    esums, enorms, eloss = O.vae_step_sums(spec, params, X[sel], eps[sel], clip, None)
    full, norms, pxl = run(None, clip)
    np.testing.assert_allclose(np_(norms)[sel], enorms, rtol=5e-5)
    np.testing.assert_allclose(np_(pxl)[sel], eloss, rtol=2e-5, atol=1e-5)
    mask = np.zeros(B, bool)
    mask[sel] = True
    part, _, pxl_m = run(mask, clip)
    got = np_(part)
    assert got[P + 1] == len(sel)
    assert abs(got[P] - esums[P]) <= 2e-5 * abs(esums[P])
    np.testing.assert_allclose(got[:P], esums[:P], rtol=2e-4, atol=2e-5 * np.abs(esums[:P]).max())
    assert torch.all(pxl_m[torch.tensor(~mask).cuda()] == 0)
    # linearity: sums(selected) + sums(the rest) == sums(all)
    rest, _, _ = run(~mask, clip)
    np.testing.assert_allclose(got + np_(rest), np_(full), rtol=2e-4, atol=2e-5 * float(full[:P].abs().max()))
    assert float(full[P + 1]) == B
    assert float((full[:P] / B).norm()) <= clip * (1 + 1e-5)
    assert float((norms > clip).float().mean()) > 0.2 and float((norms < clip).float().mean()) > 0.2
    again, _, _ = run(None, clip)
    assert torch.equal(again, full)


@pytest.mark.parametrize("H2", [0, 200])
def test_config4_update_is_the_composition_of_its_stages_at_batch_4096(gpu, O, H2):
    """DPSVI.update at configs[4]'s full size takes launches no small shape takes: the weight-gradient products as ONE grouped
    launch with a common K range (k_gemm_bf16x3_group), the clip factors applied while those products stage their deltas, the
    exactness pass and the packing of the latent heads inside the key launch, the noise beside the latent kernel.  Its gradient
    must be what the stages give one by one: eps from the gradient key (oracle, svi.py:289-290), the clipped sums of
    d3p_vae_step_sums on that eps (products launched and reduced one by one: the path the test above pins to the oracle),
    the oracle's perturbation (one key per leaf, svi.py:487-491) and Adam."""
    import d3p_amd._lib as lib
    import d3p_amd.random as rng
    from d3p_amd.svi import DPSVIState
    from tests.test_gpu_vae import make_svi, vae_problem
    B, D, H, Z, N = 4096, 784, 400, 50, 60000
    spec, P, params, X, _ = vae_problem(B, D, H, Z, 31, 0.03, H2)
    mask = np.random.default_rng(6).random(B) < 0.9
    svi = make_svi(Z, H, N, C=3.0, sigma=0.8, lr=1e-2, H2=H2)
    st = DPSVIState(svi.optim.init(torch.tensor(params).cuda()), rng.PRNGKey(78), 1.0)
    Xt = torch.tensor(X).cuda()
    gout = torch.empty(P, device="cuda")
    new_st, loss = svi.update(st, Xt, mask=torch.tensor(mask).cuda(), _grad_out=gout)

    ks = O.split(O.PRNGKey(78), 3)
    eps = O.px_eps(O.convert_to_jax_rng_key(ks[1]), B, Z)
    L = lib.load()
    model = lib.VaeModel(D, H, Z, 1.0, 1.0, H2)   # (plate(N) x handlers.scale(1 / N) = 1; observation_scale 1)
    ws = torch.empty(int(L.d3p_dpvi_vae_workspace(C.byref(model), B)), dtype=torch.uint8, device="cuda")
    sums, norms, pxl = torch.empty(P + 2, device="cuda"), torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    pt, et, mt = torch.tensor(params).cuda(), torch.tensor(eps).cuda(), torch.tensor(mask).to(torch.uint8).cuda()
    lib.check(L.d3p_vae_step_sums(lib.stream_ptr(), C.byref(model), lib.ptr(pt), lib.ptr(Xt), lib.ptr(mt), B, lib.ptr(et), None,
                                  3.0, lib.ptr(sums), lib.ptr(norms), lib.ptr(pxl), lib.ptr(ws), ws.numel()))
    sums = np_(sums)
    n = sums[P + 1]
    assert n == mask.sum() and float((norms > 3.0).float().mean()) > 0.05
    f = B / n
    g = O.perturb(ks[2], sums[:P] / B, O.vae_leaf_sizes(D, H, Z, H2), 0.8, 3.0, n, 1.0, f)
    x, m, v = O.adam(params, np.zeros(P), np.zeros(P), g, 0, lr=1e-2)
    eloss = sums[P] / B * f
    assert abs(float(loss) - eloss) <= 5e-5 * abs(eloss)
    np.testing.assert_allclose(np_(gout), g, rtol=2e-4, atol=2e-5 * np.abs(g).max())
    # Adam's first step is lr g / (|g| + 1e-8): where a gradient component is ~ 0 (a handful of the 689 k) its relative error is
    # the step's -- compare the parameters where the gradient is not negligible, and bound the step elsewhere
    big = np.abs(g) > 1e-4 * np.abs(g).max()
    assert big.mean() > 0.99
    got_x = np_(new_st.optim_state[1])
    np.testing.assert_allclose(got_x[big], x[big], rtol=1e-4, atol=2e-5)
    assert np.all(np.abs(got_x - params) <= 1e-2 * (1 + 1e-5))
    assert np.array_equal(np_(new_st.rng_key), ks[0]) and int(new_st.optim_state[0]) == 1
