"""VAE step (BASELINE config 5, examples/vae.py:65-153): the fp32 MFMA GEMM against torch.matmul, the fused
norm / clipped-sum path (no per-example gradient tensor) against the oracle's explicit per-example gradients,
and DPSVI.update against the oracle's stage composition.

Tolerances: GEMM rtol 2e-5 of the row scale (fp32 accumulation order); clipped sums rtol 2e-4 + 2e-5 * max|g|
(sums of B products of fp32 activations; the oracle works in float64); norms rtol 5e-5."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def np_(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def lib(gpu):
    import d3p_amd._lib as L
    return L


# (260, 132, 200) and (388, 68, 96): 16-byte loadable in every form, i.e. the 128 x 64 eight-wave kernel with ragged M / N tiles, a
# partial last K slice (200 = 6.25 slices of 32) and an odd number of slices (96 = 3: one padded)
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (5, 7, 3), (100, 130, 70), (257, 65, 1000), (260, 132, 200), (388, 68, 96), (4096, 400, 784)])
@pytest.mark.parametrize("form", ["nn", "nt", "tn", "tt"])
def test_mfma_gemm_vs_torch(lib, M, N, K, form):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    Bm = torch.randn(K, N, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Cinit = torch.randn(M, N, generator=g).cuda()
    if form == "nn":
        a_t, a_sm, a_sk, b_t, b_sk, b_sn = A.contiguous(), K, 1, Bm.contiguous(), N, 1
    elif form == "nt":      # B stored as N x K
        a_t, a_sm, a_sk, b_t, b_sk, b_sn = A.contiguous(), K, 1, Bm.t().contiguous(), 1, K
    elif form == "tn":      # A stored as K x M
        a_t, a_sm, a_sk, b_t, b_sk, b_sn = A.t().contiguous(), 1, M, Bm.contiguous(), N, 1
    else:                   # both transposed
        a_t, a_sm, a_sk, b_t, b_sk, b_sn = A.t().contiguous(), 1, M, Bm.t().contiguous(), 1, K
    L = lib.load()
    out = Cinit.clone()
    lib.check(L.d3p_gemm_f32(lib.stream_ptr(), lib.ptr(a_t), a_sm, a_sk, lib.ptr(b_t), b_sk, b_sn, lib.ptr(out), N, M, N, K,
                             lib.ptr(bias), 0.5, 1))
    ref = 0.5 * (A.double() @ Bm.double()) + bias.double() + Cinit.double()
    scale = float((A.abs().double() @ Bm.abs().double()).max())
    assert float((out.double() - ref).abs().max()) <= 2e-5 * scale
    out2 = torch.empty(M, N + 3, device="cuda").fill_(7.0)      # leading dimension > N, no bias, overwrite
    lib.check(L.d3p_gemm_f32(lib.stream_ptr(), lib.ptr(a_t), a_sm, a_sk, lib.ptr(b_t), b_sk, b_sn, lib.ptr(out2), N + 3, M, N, K,
                             None, 1.0, 0))
    assert float((out2[:, :N].double() - A.double() @ Bm.double()).abs().max()) <= 2e-5 * scale
    assert torch.all(out2[:, N:] == 7.0)


def test_mfma_gemm_short_deep_product_with_padded_row_stride(lib):
    """A^T B with A stored K x ld, ld > M and M % 4 != 0 (the V1 weight gradient's shape: the [z | z_std] array is B x 100, 50
    columns are z): the eight-wave kernel's last float4 of rows runs past M inside the row stride and is masked row by row."""
    M, N, K, ld = 51, 132, 2048, 100
    g = torch.Generator().manual_seed(7)
    At = torch.randn(K, ld, generator=g).cuda()          # columns >= M are junk that must not leak into the product
    Bm = torch.randn(K, N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda")
    L = lib.load()
    lib.check(L.d3p_gemm_f32(lib.stream_ptr(), lib.ptr(At), 1, ld, lib.ptr(Bm), N, 1, lib.ptr(out), N, M, N, K, None, 1.0, 0))
    ref = At[:, :M].double().t() @ Bm.double()
    scale = float((At[:, :M].abs().double().t() @ Bm.abs().double()).max())
    assert float((out.double() - ref).abs().max()) <= 2e-5 * scale


def test_mfma_gemm_k_not_a_multiple_of_four_with_padded_rows(lib):
    """A B with A stored M x ld, ld > K and K % 4 != 0 (h2 = z V1: z is the first 50 columns of a B x 100 array): the eight-wave
    kernel's last float4 along k runs past K inside the row and is masked element by element."""
    M, N, K, ld = 300, 132, 50, 100
    g = torch.Generator().manual_seed(9)
    A = torch.randn(M, ld, generator=g).cuda()           # columns >= K are junk that must not leak into the product
    Bm = torch.randn(K + 3, N, generator=g).cuda()       # (rows >= K exist in memory, as the next parameter leaf does, and are junk too)
    out = torch.empty(M, N, device="cuda")
    L = lib.load()
    lib.check(L.d3p_gemm_f32(lib.stream_ptr(), lib.ptr(A), ld, 1, lib.ptr(Bm), N, 1, lib.ptr(out), N, M, N, K, None, 1.0, 0))
    ref = A[:, :K].double() @ Bm[:K].double()
    scale = float((A[:, :K].abs().double() @ Bm[:K].abs().double()).max())
    assert float((out.double() - ref).abs().max()) <= 2e-5 * scale


def test_split_product_is_fp32_accurate_on_hostile_operands(lib):
    """The large products run as six bf16 MFMAs on an exact three-way bf16 split of both operands (k_gemm_bf16x3): every
    partial product is exact and the dropped terms are below 2^-23 of |a||b|, so the result must be as close to the float64
    product as an fp32 product is -- also where a bf16 shortcut would show: operands whose magnitudes span 2^+-20 inside one
    row, values with all 24 significand bits set, and a sum that cancels to 2^-12 of its terms.  Bound: 4 x 2^-24 x K^(1/2)-ish
    fp32 accumulation noise, stated as 3e-6 of sum_k |a||b| (a 16-bit product would miss it by a factor of 100)."""
    M, N, K = 512, 256, 1024
    g = torch.Generator().manual_seed(77)
    A = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-20, 21, (M, K), generator=g).float())
    Bm = torch.randn(K, N, generator=g) * torch.exp2(torch.randint(-20, 21, (K, N), generator=g).float())
    A[:, ::7] = torch.nextafter(torch.tensor(2.0), torch.tensor(0.0))          # 0x3fffffff: every significand bit set
    Bm[::5, :] = torch.nextafter(torch.tensor(-1.0), torch.tensor(0.0))
    A[:, 1::2] = -A[:, 0::2] * (1.0 + 2.0 ** -12)                              # pairs that cancel to 2^-12 of their size
    Bm[1::2, :] = Bm[0::2, :]
    A, Bm = A.cuda(), Bm.cuda()
    out = torch.empty(M, N, device="cuda")
    L = lib.load()
    lib.check(L.d3p_gemm_f32(lib.stream_ptr(), lib.ptr(A), K, 1, lib.ptr(Bm), N, 1, lib.ptr(out), N, M, N, K, None, 1.0, 0))
    ref = A.double() @ Bm.double()
    scale = A.abs().double() @ Bm.abs().double()
    err = ((out.double() - ref).abs() / scale).max()
    assert float(err) <= 3e-6, float(err)
    assert bool(torch.isfinite(out).all())
    # non-finite operands propagate like in an fp32 product (a part of Inf is Inf, its remainder NaN)
    A2 = A.clone()
    A2[3, 10] = float("inf")
    lib.check(L.d3p_gemm_f32(lib.stream_ptr(), lib.ptr(A2), K, 1, lib.ptr(Bm), N, 1, lib.ptr(out), N, M, N, K, None, 1.0, 0))
    assert not bool(torch.isfinite(out[3]).any()) and bool(torch.isfinite(out[4]).all())


def vae_problem(B, D, H, Z, seed, pscale=0.2, H2=0):
    import oracle.oracle as O
    r = np.random.default_rng(seed)
    spec = O.vae_spec(D, H, Z, scale=1.0, obs_scale=1.0, H2=H2)
    P = O.vae_num_params(spec)
    params = (r.normal(size=P) * pscale).astype(np.float32)
    X = (r.random((B, D)) < 0.3).astype(np.float32)
    eps = r.normal(size=(B, Z)).astype(np.float32)
    return spec, P, params, X, eps


# (H2 > 0: the two-hidden-layer variant of BASELINE config 5, 784 -> [400, 200] -> 50; the reference has one hidden layer)
@pytest.mark.parametrize("B,D,H,Z,pscale,H2", [(5, 12, 7, 3, 0.3, 0), (70, 100, 33, 9, 0.2, 0), (48, 784, 400, 50, 0.03, 0),
                                               (5, 12, 7, 3, 0.3, 5), (70, 100, 33, 9, 0.2, 21), (48, 784, 400, 50, 0.03, 200)])
@pytest.mark.parametrize("masked", [False, True])
def test_step_sums_vs_explicit_per_example_gradients(lib, O, B, D, H, Z, pscale, H2, masked):
    spec, P, params, X, eps = vae_problem(B, D, H, Z, B + D, pscale, H2)
    mask = (np.random.default_rng(3).random(B) < 0.7) if masked else None
    # a clip threshold in the middle of the norm distribution, so that some rows are clipped and some are not
    _, norms0, _ = O.vae_step_sums(spec, params, X, eps, 1e30, None)
    clip = float(np.median(norms0))
    esums, enorms, eloss = O.vae_step_sums(spec, params, X, eps, clip, None if mask is None else mask.astype(np.float32))
    L = lib.load()
    model = lib.VaeModel(D, H, Z, 1.0, 1.0, H2)
    assert L.d3p_vae_num_params(C.byref(model)) == P == sum(O.vae_leaf_sizes(D, H, Z, H2))
    ws = torch.empty(int(L.d3p_dpvi_vae_workspace(C.byref(model), B)), dtype=torch.uint8, device="cuda")
    sums = torch.empty(P + 2, device="cuda")
    norms = torch.empty(B, device="cuda")
    pxl = torch.empty(B, device="cuda")
    mt = None if mask is None else torch.tensor(mask).to(torch.uint8).cuda()
    pt, Xt, et = torch.tensor(params).cuda(), torch.tensor(X).cuda(), torch.tensor(eps).cuda()   # keep the inputs alive
    lib.check(L.d3p_vae_step_sums(lib.stream_ptr(), C.byref(model), lib.ptr(pt), lib.ptr(Xt), lib.ptr(mt), B, lib.ptr(et), None,
                                  clip, lib.ptr(sums), lib.ptr(norms), lib.ptr(pxl), lib.ptr(ws), ws.numel()))
    np.testing.assert_allclose(np_(norms), enorms, rtol=5e-5)
    np.testing.assert_allclose(np_(pxl), eloss, rtol=2e-5, atol=1e-5)
    got = np_(sums)
    assert got[P + 1] == esums[P + 1]
    assert abs(got[P] - esums[P]) <= 2e-5 * abs(esums[P])
    np.testing.assert_allclose(got[:P], esums[:P], rtol=2e-4, atol=2e-5 * np.abs(esums[:P]).max())
    assert (enorms > clip).any() and (enorms[enorms > 0] < clip).any()


@pytest.mark.parametrize("grey", [False, True])
def test_step_sums_on_the_bf16_pipe_with_binarised_and_grey_batches(lib, O, grey):
    """B = 136 puts the first encoder product and its weight gradient on k_gemm_bf16x3 (M > 96).  A BINARISED batch
    (examples/vae.py:157-168) is exactly bf16: the one-plane path of those two products (k_exact16_flag -> GemmArgs::a_exact16:
    plane 0 of A alone, three of the six bf16 products).  A GREY batch (pixel intensities in [0, 1], what the data holds before
    binarize) is not: the flag kernel must say so and the general three-plane path runs.  Both against the oracle's explicit
    per-example gradients, same tolerances as the small shapes."""
    B, D, H, Z = 136, 784, 400, 50
    spec, P, params, X, eps = vae_problem(B, D, H, Z, 99, 0.03, 0)
    if grey:
        X = np.random.default_rng(5).random((B, D)).astype(np.float32)
        assert (X.view(np.uint32) & 0xffff).any()
    _, norms0, _ = O.vae_step_sums(spec, params, X, eps, 1e30, None)
    clip = float(np.median(norms0))
    esums, enorms, eloss = O.vae_step_sums(spec, params, X, eps, clip, None)
    L = lib.load()
    model = lib.VaeModel(D, H, Z, 1.0, 1.0, 0)
    ws = torch.empty(int(L.d3p_dpvi_vae_workspace(C.byref(model), B)), dtype=torch.uint8, device="cuda")
    sums, norms, pxl = torch.empty(P + 2, device="cuda"), torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    pt, Xt, et = torch.tensor(params).cuda(), torch.tensor(X).cuda(), torch.tensor(eps).cuda()
    lib.check(L.d3p_vae_step_sums(lib.stream_ptr(), C.byref(model), lib.ptr(pt), lib.ptr(Xt), None, B, lib.ptr(et), None,
                                  clip, lib.ptr(sums), lib.ptr(norms), lib.ptr(pxl), lib.ptr(ws), ws.numel()))
    np.testing.assert_allclose(np_(norms), enorms, rtol=5e-5)
    np.testing.assert_allclose(np_(pxl), eloss, rtol=2e-5, atol=1e-5)
    got = np_(sums)
    assert got[P + 1] == esums[P + 1] and abs(got[P] - esums[P]) <= 2e-5 * abs(esums[P])
    np.testing.assert_allclose(got[:P], esums[:P], rtol=2e-4, atol=2e-5 * np.abs(esums[:P]).max())


@pytest.mark.parametrize("seed", [1, 2])
def test_step_sums_random_shapes_vs_oracle(lib, O, seed):
    """Forty random shapes per seed -- batch 1 .. 260, rows of 8 .. 784 features, hidden widths 4 .. 400 (one or two layers), 1 .. 50
    latents, binarised or grey batches, with and without masked rows -- against the oracle's explicit per-example gradients: which
    kernel a product takes (bf16 tiles, grouped or alone; the 64 x 64 fp32 kernel; packed or displaced heads), whether the clip factors
    are applied inside the products or written back, how many tiles a product leaves, all follow from the shape."""
    rs = np.random.default_rng(seed)
    L = lib.load()
    for _ in range(40):
        B = int(rs.choice([1, 3, 17, 64, 97, 130, 200, 260]))
        D = int(rs.choice([8, 12, 33, 64, 100, 200, 784]))
        H = int(rs.choice([4, 7, 16, 40, 100, 400]))
        Z = int(rs.choice([1, 2, 3, 8, 10, 50]))
        H2 = int(rs.choice([0, 0, 5, 12, 200]))
        if D * H > 200000 and B > 140:
            B = 136   # (the oracle materialises B x P gradients)
        masked = bool(rs.integers(2))
        spec, P, params, X, eps = vae_problem(B, D, H, Z, B + D + H, 0.03 if (D > 100 or H > 50 or H2 > 50) else 0.3, H2)
        if rs.integers(2):
            X = rs.random((B, D)).astype(np.float32)
        mask = (rs.random(B) < 0.7) if masked else None
        _, norms0, _ = O.vae_step_sums(spec, params, X, eps, 1e30, None)
        clip = float(np.median(norms0))
        esums, enorms, eloss = O.vae_step_sums(spec, params, X, eps, clip, None if mask is None else mask.astype(np.float32))
        model = lib.VaeModel(D, H, Z, 1.0, 1.0, H2)
        ws = torch.empty(int(L.d3p_dpvi_vae_workspace(C.byref(model), B)), dtype=torch.uint8, device="cuda")
        sums, norms, pxl = torch.empty(P + 2, device="cuda"), torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
        mt = None if mask is None else torch.tensor(mask).to(torch.uint8).cuda()
        pt, Xt, et = torch.tensor(params).cuda(), torch.tensor(X).cuda(), torch.tensor(eps).cuda()
        lib.check(L.d3p_vae_step_sums(lib.stream_ptr(), C.byref(model), lib.ptr(pt), lib.ptr(Xt), lib.ptr(mt), B, lib.ptr(et), None,
                                      clip, lib.ptr(sums), lib.ptr(norms), lib.ptr(pxl), lib.ptr(ws), ws.numel()))
        got, what = np_(sums), f"B={B} D={D} H={H} Z={Z} H2={H2} masked={masked}"
        np.testing.assert_allclose(np_(norms), enorms, rtol=5e-5, err_msg=what)
        np.testing.assert_allclose(np_(pxl), eloss, rtol=2e-5, atol=1e-5, err_msg=what)
        assert got[P + 1] == esums[P + 1], what
        assert abs(got[P] - esums[P]) <= 2e-5 * abs(esums[P]) + 1e-6, what
        np.testing.assert_allclose(got[:P], esums[:P], rtol=2e-4, atol=2e-5 * np.abs(esums[:P]).max(), err_msg=what)


def make_svi(Z, H, N, C=10.0, sigma=1.0, lr=1e-3, H2=0):
    from d3p_amd.models import Adam, Trace_ELBO, VAEGuide, VAEModel
    from d3p_amd.svi import DPSVI
    model = VAEModel(scale=1.0 / N)                       # handlers.scale(model, 1 / num_samples), vae.py:194-195
    return DPSVI(model, VAEGuide(model), Adam(lr), Trace_ELBO(), C, sigma, num_obs_total=N, z_dim=Z,
                 hidden_dim=(H, H2) if H2 else H)


# (B = 136: three of the four weight-gradient products take the bf16 kernel and go out as one grouped launch, the fourth -- 51 rows --
# by itself; the clip factors come back through the norm kernel's scaled rows)
@pytest.mark.parametrize("B,D,H,Z,masked,H2", [(6, 12, 7, 3, False, 0), (40, 784, 400, 50, True, 0), (6, 12, 7, 3, True, 5),
                                               (40, 784, 400, 50, False, 200), (136, 784, 400, 50, True, 0)])
def test_update_vs_oracle_stage_composition(gpu, O, B, D, H, Z, masked, H2):
    """DPSVI.update for the VAE: split(key, 3); per-example eps from the gradient key (svi.py:289-290); clipped sums;
    one perturbation key per parameter leaf in tree_flatten order (svi.py:487-491); numpyro Adam."""
    import d3p_amd.random as rng
    from d3p_amd.svi import DPSVIState
    N = 60000
    spec, P, params, X, _ = vae_problem(B, D, H, Z, 11, 0.03 if D > 100 else 0.3, H2)
    mask = (np.random.default_rng(4).random(B) < 0.7) if masked else None
    svi = make_svi(Z, H, N, C=3.0, sigma=0.8, lr=1e-2, H2=H2)
    st = DPSVIState(svi.optim.init(torch.tensor(params).cuda()), rng.PRNGKey(77), 1.0)
    Xt = torch.tensor(X.reshape(B, -1)).cuda()
    gout = torch.empty(P, device="cuda")
    new_st, loss = svi.update(st, Xt, mask=True if mask is None else torch.tensor(mask).cuda(), _grad_out=gout)

    ks = O.split(O.PRNGKey(77), 3)
    eps = O.px_eps(O.convert_to_jax_rng_key(ks[1]), B, Z)
    sums, _, _ = O.vae_step_sums(spec, params, X, eps, 3.0, None if mask is None else mask.astype(np.float32))
    n = sums[P + 1]
    f = B / n
    sizes = O.vae_leaf_sizes(D, H, Z, H2)                  # 10 leaves, or 14 with the second hidden layer
    assert sizes == [Z * H, H, H * D, D, D * H, H, H * Z, Z, H * Z, Z] or H2
    g = O.perturb(ks[2], sums[:P] / B, sizes, 0.8, 3.0, n, 1.0, f)
    x, m, v = O.adam(params, np.zeros(P), np.zeros(P), g, 0, lr=1e-2)
    eloss = sums[P] / B * f
    assert abs(float(loss) - eloss) <= 5e-5 * abs(eloss)
    np.testing.assert_allclose(np_(gout), g, rtol=2e-4, atol=2e-5 * np.abs(g).max())
    # (Adam's first step is lr g / (|g| + 1e-8): where a gradient component is ~ 0 -- one in a million at P = 689 k -- the step has the
    # relative error of that component; the parameters are compared where the gradient is not negligible, the step is bounded elsewhere)
    big = np.abs(g) > 1e-4 * np.abs(g).max()
    assert big.mean() > 0.99
    got_x = np_(new_st.optim_state[1])
    np.testing.assert_allclose(got_x[big], x[big], rtol=1e-4, atol=2e-5)
    assert np.all(np.abs(got_x - params) <= 1e-2 * (1 + 1e-5))
    assert np.array_equal(np_(new_st.rng_key), ks[0]) and int(new_st.optim_state[0]) == 1
    tree = svi.get_params(new_st)
    HE = H2 if H2 else H
    assert tuple(tree["decoder$params"][0][0].shape) == (Z, HE) and tuple(tree["encoder$params"][-1][1][0][0].shape) == (HE, Z)
    assert tree["decoder$params"][1] == () and tuple(tree["encoder$params"][0][0].shape) == (D, H)
    if H2:
        assert tuple(tree["decoder$params"][2][0].shape) == (H2, H) and tuple(tree["decoder$params"][4][0].shape) == (H, D)
        assert tuple(tree["encoder$params"][2][0].shape) == (H, H2) and len(tree["decoder$params"]) == 6
    else:
        assert tuple(tree["encoder$params"][3][1][0][0].shape) == (H, Z)


def test_update_leaves_the_input_state_alone(gpu):
    """The VAE update is out of place (d3p_dpvi_vae_update_from): the state passed in is bit for bit unchanged, and updating
    from it twice gives identical results (DPSVI.update returns a NEW state, svi.py:395-434)."""
    import d3p_amd.random as rng
    from d3p_amd.svi import DPSVIState
    B, D, H, Z, N = 24, 12, 7, 3, 500
    spec, P, params, X, _ = vae_problem(B, D, H, Z, 5, 0.3)
    svi = make_svi(Z, H, N, C=3.0, sigma=0.8, lr=1e-2)
    Xt = torch.tensor(X.reshape(B, -1)).cuda()
    st = DPSVIState(svi.optim.init(torch.tensor(params).cuda()), rng.PRNGKey(3), 1.0)
    st, _ = svi.update(st, Xt)                           # (a state with non-trivial Adam moments)
    before = [t.clone() for t in st.optim_state] + [st.rng_key.clone()]
    a, la = svi.update(st, Xt)
    b, lb = svi.update(st, Xt)
    for x, y0 in zip(list(st.optim_state) + [st.rng_key], before):
        assert torch.equal(x, y0)
    assert float(la) == float(lb) and torch.equal(a.rng_key, b.rng_key) and int(a.optim_state[0]) == 2
    for x, y0 in zip(a.optim_state, b.optim_state):
        assert torch.equal(x, y0)
    assert not torch.equal(a.optim_state[1], st.optim_state[1])


def test_training_reduces_the_loss_on_structured_binary_images(gpu):
    """A few hundred non-private-ish steps on synthetic 'images' (two prototype patterns + flips): the ELBO improves."""
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    N, Z, H, B = 2048, 8, 64, 128
    g = torch.Generator().manual_seed(0)
    protos = (torch.rand(2, 28, 28, generator=g) < 0.3).float()
    which = torch.randint(0, 2, (N,), generator=g)
    flips = (torch.rand(N, 28, 28, generator=g) < 0.05).float()
    X = ((protos[which] + flips) % 2).cuda()
    svi = make_svi(Z, H, N, C=50.0, sigma=0.05, lr=3e-3)
    init, get_batch = subsample_batchify_data((X,), B)
    nb, bstate = init(rng.PRNGKey(1))
    st = svi.init(rng.PRNGKey(0), *get_batch(0, bstate))
    assert float(st.observation_scale) == 1.0            # plate scale N x handlers.scale(1 / N)
    losses = []
    for i in range(300):
        st, l = svi.update(st, *get_batch(i % nb, bstate))
        losses.append(float(l))
    assert np.isfinite(losses).all()
    assert np.mean(losses[-20:]) < 0.6 * np.mean(losses[:20])


@pytest.mark.parametrize("B,D,H,Z,H2", [(6, 12, 7, 3, 0), (64, 784, 400, 50, 0), (6, 12, 7, 3, 4), (64, 784, 400, 50, 200)])
def test_evaluate_vs_oracle(gpu, O, B, D, H, Z, H2):
    import d3p_amd.random as rng
    from d3p_amd.svi import DPSVIState
    N = 60000
    _, P, params, X, _ = vae_problem(B, D, H, Z, 5, 0.03 if D > 100 else 0.3, H2)
    svi = make_svi(Z, H, N, H2=H2)
    st = DPSVIState(svi.optim.init(torch.tensor(params).cuda()), rng.PRNGKey(31), 1.0)
    got = float(svi.evaluate(st, torch.tensor(X).cuda()))
    spec = O.vae_spec(D, H, Z, scale=1.0 / B, obs_scale=1.0, H2=H2)   # (1 / N) x (N / B)
    jax_key = O.convert_to_jax_rng_key(O.split(O.PRNGKey(31), 1)[0])
    exp = O.vae_evaluate(spec, params, X, jax_key)
    assert abs(got - exp) <= 3e-5 * abs(exp)


# (the second shape is BASELINE configs[4]'s: the step's indices come out of its key launch, one sweep gathers the rows and checks their
# exactness, the weight-gradient products go out grouped -- the launches of the bench leg; D = 35 takes the unfused run loop)
@pytest.mark.parametrize("N,B,D,H,Z,steps,first", [(600, 64, 36, 20, 5, 7, 3), (9000, 4096, 784, 400, 50, 3, 1), (300, 40, 35, 12, 4, 4, 0)])
def test_vae_run_steps_native_loop_matches_stepwise_updates(gpu, N, B, D, H, Z, steps, first):
    """DPSVI.run_steps for the VAE (d3p_dpvi_vae_run: the epoch body of examples/vae.py:227-246 as one native call -- per step
    fold_in, Feistel indices, row gather, update) walks the trajectory of get_batch + update, step by step, bit for bit (the
    same kernels on the same batches), twice the same, and leaves its input state untouched."""
    import d3p_amd.random as rng
    from d3p_amd.minibatch import subsample_batchify_data
    from d3p_amd.models import Adam, Trace_ELBO, VAEGuide, VAEModel
    from d3p_amd.svi import DPSVI
    X = (torch.rand(N, D, generator=torch.Generator().manual_seed(1)) < 0.3).float().cuda()
    model = VAEModel(scale=1.0 / N)
    svi = DPSVI(model, VAEGuide(model), Adam(1e-2), Trace_ELBO(), 5.0, 0.8, num_obs_total=N, z_dim=Z, hidden_dim=H)
    st0 = svi.init(rng.PRNGKey(0), X[:B])
    before = [t.clone() for t in st0.optim_state] + [st0.rng_key.clone()]
    init, get_batch = subsample_batchify_data((X,), B)
    _, bstate = init(rng.PRNGKey(9))
    new_st, losses = svi.run_steps(st0, get_batch, bstate, first, steps)
    ref, ref_losses = st0, []
    for t in range(steps):
        ref, l = svi.update(ref, *get_batch(first + t, bstate))
        ref_losses.append(l.reshape(()))
    assert torch.equal(losses, torch.stack(ref_losses)) and bool(torch.isfinite(losses).all())
    assert torch.equal(new_st.optim_state[1], ref.optim_state[1]) and torch.equal(new_st.rng_key, ref.rng_key)
    assert int(new_st.optim_state[0]) == int(ref.optim_state[0]) == steps
    for a, b in zip([*st0.optim_state, st0.rng_key], before):
        assert torch.equal(a, b)
    again, losses2 = svi.run_steps(st0, get_batch, bstate, first, steps)
    assert torch.equal(again.optim_state[1], new_st.optim_state[1]) and torch.equal(losses2, losses)


@pytest.mark.parametrize("mult", [1.0, 2.0])
def test_step_sums_vs_float32_autograd_where_float32_overflows(lib, mult):
    """The oracle works in float64 inside; with weights large enough for per-example gradient norms of 1e20 .. 1e29 the FLOAT32 norm of an
    example overflows (|g|^2 > 3.4e38): its clip factor is 1 / max(1, inf / C) = 0 and the example drops out -- in the reference (float32
    jax) as on the device, not in the oracle.  A float32 torch.autograd restatement of the same ELBO (784-free small shape, one hidden
    layer) is the comparator there: the same examples overflow, and the clipped sums agree; at ordinary weights all three agree."""
    D, H, Z, B, clip = 100, 40, 50, 64, 0.5
    r = np.random.default_rng(91)
    sizes = [Z * H, H, H * D, D, D * H, H, H * Z, Z, H * Z, Z]        # decoder (V1, c1, V2, c2), encoder (W1, b1, Wl, bl, Ws, bs)
    P = sum(sizes)
    params = (r.normal(size=P) * 0.3 * mult).astype(np.float32)
    X = (r.random((B, D)) < 0.3).astype(np.float32)
    eps = r.normal(size=(B, Z)).astype(np.float32)

    def autograd(dtype):
        p = torch.tensor(params, dtype=dtype, device="cuda", requires_grad=True)
        Xt, et = torch.tensor(X, dtype=dtype, device="cuda"), torch.tensor(eps, dtype=dtype, device="cuda")
        one = torch.tensor(1.0, dtype=dtype, device="cuda")
        sums, norms, pos, leaves = torch.zeros(P, dtype=dtype, device="cuda"), [], 0, []
        for n in sizes:
            leaves.append((pos, n))
            pos += n
        sp = torch.nn.functional.softplus
        for i in range(B):
            V1, c1, V2, c2, W1, b1, Wl, bl, Ws, bs = (p[o:o + n] for o, n in leaves)
            h1 = sp(Xt[i] @ W1.reshape(D, H) + b1)
            zl, u = h1 @ Wl.reshape(H, Z) + bl, h1 @ Ws.reshape(H, Z) + bs
            z = zl + torch.exp(u) * et[i]
            a = sp(z @ V1.reshape(Z, H) + c1) @ V2.reshape(H, D) + c2
            loss = ((-0.5 * et[i] ** 2 - u) + 0.5 * z * z).sum() - (Xt[i] * a - sp(a)).sum()
            g, = torch.autograd.grad(loss, p)
            nrm = g.norm()
            sums = sums + g * (1.0 / torch.maximum(one, nrm / clip))
            norms.append(float(nrm.detach()))
        return sums.detach().cpu().numpy(), np.array(norms)
    s32, n32 = autograd(torch.float32)
    s64, n64 = autograd(torch.float64)
    L = lib.load()
    model = lib.VaeModel(D, H, Z, 1.0, 1.0, 0)
    ws = torch.empty(int(L.d3p_dpvi_vae_workspace(C.byref(model), B)), dtype=torch.uint8, device="cuda")
    sums, norms, pxl = torch.empty(P + 2, device="cuda"), torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    pt, Xt, et = torch.tensor(params).cuda(), torch.tensor(X).cuda(), torch.tensor(eps).cuda()
    lib.check(L.d3p_vae_step_sums(lib.stream_ptr(), C.byref(model), lib.ptr(pt), lib.ptr(Xt), None, B, lib.ptr(et), None, clip,
                                  lib.ptr(sums), lib.ptr(norms), lib.ptr(pxl), lib.ptr(ws), ws.numel()))
    hs, hn = np_(sums)[:P], np_(norms)
    assert np.isfinite(s64).all() and np.isfinite(n64).all()
    assert np.array_equal(np.isfinite(hn), np.isfinite(n32))                 # the same examples overflow in float32
    assert np.isfinite(hs).all() and np.isfinite(s32).all()
    np.testing.assert_allclose(hs, s32, rtol=0, atol=2e-3 * np.abs(s32).max())
    if mult == 1.0:
        assert np.isfinite(hn).all()
        np.testing.assert_allclose(hs, s64, rtol=0, atol=2e-4 * np.abs(s64).max())
        np.testing.assert_allclose(hn, n64, rtol=1e-4)
    else:
        assert 0 < (~np.isfinite(hn)).sum() < B                              # (the case is what it says: some, not all)
