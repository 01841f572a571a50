"""Feistel / Poisson samplers and the batchifier factories vs the oracle (bit-exact indices and
masks) and the properties the reference's tests/test_minibatch.py and tests/test_util.py check."""
import numpy as np
import pytest
import scipy.stats
import torch

pytestmark = pytest.mark.gpu


def np_(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def rng(gpu):
    import d3p_amd.random as r
    return r


@pytest.mark.parametrize("capacity,n", [(1, 1), (2, 2), (3, 3), (100, 100), (100, 99), (100, 1), (105, 50),
                                        (1000, 1000), (10**6, 978), (10**6, 4096), (10**7, 8192),
                                        (10**8, 32768), (2**32 - 1, 1000), (65536, 65536), (65537, 4000)])
def test_feistel_indices_bit_exact(rng, O, capacity, n):
    from d3p_amd.util import feistel_indices
    key = rng.PRNGKey(capacity * 31 + n)
    got = np_(feistel_indices(key, capacity, n))
    exp = O.feistel_sample(O.PRNGKey(capacity * 31 + n), capacity, n)
    assert np.array_equal(got, exp)
    assert got.max() < capacity
    assert np.unique(got).size == n  # a permutation restricted to n positions has no duplicates


def test_feistel_full_size_is_permutation(rng):
    from d3p_amd.util import feistel_indices
    n = 10**6
    got = np_(feistel_indices(rng.PRNGKey(5), n, n))
    assert np.array_equal(np.sort(got), np.arange(n, dtype=np.uint32))  # size-independent property


def test_feistel_generic_rng_suite_path(gpu, O):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import d3p_amd.random.debug as dbg
    from d3p_amd.util import feistel_indices
    key = dbg.PRNGKey(11)
    got = np_(feistel_indices(key, 1000, 1000, rng_suite=dbg))
    assert np.array_equal(np.sort(got), np.arange(1000))
    rc = O.tf_random_words([0, 11], 30)
    rc[::3] |= 1
    exp = [O.lib().d3po_feistel_permute(rc.ctypes.data_as(__import__("ctypes").c_void_p), 1000, p) for p in range(1000)]
    assert np.array_equal(got, np.array(exp, np.uint32))


def test_sample_from_array_reference_properties(rng):
    """reference tests/test_util.py:331-373"""
    from d3p_amd.util import sample_from_array
    x = torch.arange(10**6, device="cuda", dtype=torch.int32)
    s = np_(sample_from_array(rng.PRNGKey(0), x, 978, 0))
    assert np.unique(s).size == 978
    x2 = torch.arange(30, device="cuda", dtype=torch.float32).reshape(10, 3)
    assert tuple(sample_from_array(rng.PRNGKey(0), x2, 5, 0).shape) == (5, 3)
    x3 = torch.arange(40, device="cuda", dtype=torch.float32).reshape(4, 10)
    s3 = sample_from_array(rng.PRNGKey(0), x3, 6, 1)
    assert tuple(s3.shape) == (4, 6)
    assert np.array_equal(np_(s3)[1] - np_(s3)[0], np.full(6, 10.0))
    a = torch.arange(100, device="cuda", dtype=torch.int32)
    for n in (100, 99, 1):
        assert np.unique(np_(sample_from_array(rng.PRNGKey(3), a, n, 0))).size == n
    with pytest.raises(ValueError):
        sample_from_array(rng.PRNGKey(3), a, 101, 0)


@pytest.mark.parametrize("N,q,cutoff,suppress", [(100, 0.1, 100, False), (105, 0.3, 39, False), (105, 0.3, 20, False),
                                                 (105, 0.3, 20, True), (4097, 0.5, 4097, False),
                                                 (10**6, 0.004096, 4245, False), (10**6, 0.004096, 4000, True),
                                                 (1, 1.0, 1, False), (17, 0.0, 5, False), (1000, 1.0, 1000, False),
                                                 # thresholds at and between the 2^-23 steps of the uniform (the kernel compares integers:
                                                 # word >> 9 <= floor(q 2^23)): q = 2^-23, 3.5 x 2^-23, 2^-24, the largest float below 1, 1/3
                                                 (2**20, 2.0**-23, 64, False), (2**20, 3.5 * 2.0**-23, 64, False), (2**20, 2.0**-24, 64, False),
                                                 (4096, float(np.nextafter(np.float32(1.0), np.float32(0.0))), 4096, False),
                                                 (70001, 1.0 / 3.0, 30000, False),
                                                 # the size the benchmark's Poisson leg runs (north_star N = 10^7, q = 4096 / N): 625 k ChaCha blocks, 2442
                                                 # workgroup counts through the scan -- at the 0.99 quantile of Poisson(4096) (scipy: 4245), truncating, suppressing
                                                 (10**7, 4096 / 10**7, 4245, False), (10**7, 4096 / 10**7, 4000, False),
                                                 (10**7, 4096 / 10**7, 4000, True), (10**7 + 13, 4096 / 10**7, 4245, False)])
def test_poisson_select_bit_exact(rng, O, N, q, cutoff, suppress):
    import d3p_amd._lib as L
    from d3p_amd._lib import check, ptr, stream_ptr
    lib = L.load()
    key = rng.PRNGKey(N + cutoff)
    ws = torch.empty(lib.d3p_poisson_select_workspace(N), dtype=torch.uint8, device="cuda")
    idx = torch.empty(cutoff, dtype=torch.uint32, device="cuda")
    counts = torch.empty(2, dtype=torch.uint32, device="cuda")
    check(lib.d3p_poisson_select(stream_ptr(), ptr(key), q, N, cutoff, int(suppress), ptr(idx), ptr(counts), ptr(ws),
                                 ws.numel()))
    eidx, nsel, nvalid = O.poisson_select(O.PRNGKey(N + cutoff), q, N, cutoff, suppress)
    assert np_(counts).tolist() == [nsel, nvalid]
    assert np.array_equal(np_(idx), eidx)


@pytest.mark.parametrize("N,q,cutoff,suppress,world", [
    (10**7, 4096 / 10**7, 4245, False, 8), (10**7, 4096 / 10**7, 4000, False, 8), (10**7, 4096 / 10**7, 4000, True, 8),
    (10**7 + 13, 4096 / 10**7, 4245, False, 3),      # ragged shards, boundaries off the 16-element chunks
    (105, 0.3, 39, False, 4), (105, 0.3, 20, False, 4), (105, 0.3, 20, True, 4), (40, 0.5, 40, False, 8),   # shards smaller than a chunk
    (1000, 1.0, 1000, False, 8), (1000, 1.0, 300, False, 8), (17, 0.0, 5, False, 2), (5, 0.9, 5, False, 8)])   # (more ranks than rows)
def test_poisson_selection_sharded_over_ranks_is_the_single_gpu_selection(rng, O, N, q, cutoff, suppress, world):
    """SURVEY 8(e) / minibatch.py:29-39, :119-124: rank r of a row-sharded job makes the Bernoulli mask of ITS rows only
    (d3p_poisson_shard_flags: the keystream blocks of its rows), learns how many rows the HIGHER shards selected (here summed on the
    host from the shards' counts -- in a run d3p_xchg_poisson_counts does it through the exchange) and writes its valid selected
    rows at their GLOBAL batch positions (d3p_poisson_shard_write).  The union over the `world` virtual ranks must be the oracle's
    selection of the whole table bit for bit: rows, order (descending), truncation to the highest rows, suppression -- and every
    rank's dense list of owned positions must be its contiguous range of positions."""
    import ctypes as Ct
    import d3p_amd._lib as L
    from d3p_amd._lib import check, ptr, stream_ptr
    from d3p_amd.dist import shard_rows
    lib = L.load()
    key = rng.PRNGKey(N + cutoff)
    eidx, nsel, nvalid = O.poisson_select(O.PRNGKey(N + cutoff), q, N, cutoff, suppress)
    shards = [shard_rows(N, r, world) for r in range(world)]
    wss, locals_ = [], []
    for lo, hi in shards:
        ws = torch.empty(max(256, lib.d3p_poisson_select_workspace(max(16, 16 * ((hi + 15) // 16 - lo // 16)))), dtype=torch.uint8, device="cuda")
        cnt = torch.full((1,), 0xFFFFFFFF, dtype=torch.uint32, device="cuda")
        check(lib.d3p_poisson_shard_flags(stream_ptr(), 0, ptr(key), 0, q, N, lo, hi, 1, ptr(cnt), ptr(ws), ws.numel()))
        wss.append(ws)
        locals_.append(int(np_(cnt)[0]))
    assert sum(locals_) == nsel
    valid = (nsel if nsel <= cutoff else 0) if suppress else min(nsel, cutoff)
    assert valid == nvalid
    union = np.full(max(cutoff, 1), 0xFFFFFFFF, np.uint32)
    seen = np.zeros(max(cutoff, 1), np.int32)
    for r, (lo, hi) in enumerate(shards):
        above = sum(locals_[r + 1:])
        counts = torch.from_numpy(np.array([nsel, valid], np.uint32)).cuda()
        ab = torch.from_numpy(np.array([above], np.uint32)).cuda()
        mine = torch.full((max(cutoff, 1),), 0xFFFFFFFF, dtype=torch.uint32, device="cuda")
        plist = torch.full((max(cutoff, 1),), 0xFFFFFFFF, dtype=torch.uint32, device="cuda")
        check(lib.d3p_poisson_shard_write(stream_ptr(), N, lo, hi, cutoff, ptr(counts), 2, ptr(ab), ptr(mine), ptr(plist), cutoff, 1,
                                          ptr(wss[r]), wss[r].numel()))
        m = np_(mine)
        wrote = np.nonzero(m != 0xFFFFFFFF)[0]
        n_owned = min(max(valid - above, 0), locals_[r])
        assert wrote.size == n_owned
        if n_owned:
            assert wrote[0] == above and wrote[-1] == above + n_owned - 1          # a contiguous range of positions
            assert np.all((m[wrote] >= lo) & (m[wrote] < hi))                       # its own rows
            assert np.array_equal(np_(plist)[:n_owned], np.arange(above, above + n_owned, dtype=np.uint32))
        seen[wrote] += 1
        union[wrote] = m[wrote]
    assert np.all(seen[:valid] == 1) and not seen[valid:].any()
    assert np.array_equal(union[:valid], eidx[:valid])


def test_subsample_batchifier(rng):
    """reference tests/test_minibatch.py:117-225"""
    from d3p_amd.minibatch import subsample_batchify_data
    data = torch.arange(105, device="cuda", dtype=torch.float32).reshape(105, 1) + torch.zeros(105, 3, device="cuda")
    label = torch.arange(105, device="cuda", dtype=torch.float32)
    init, get_batch = subsample_batchify_data((data, label), 10)
    key = rng.PRNGKey(0)
    num_batches, state = init(key)
    assert num_batches == 10 and state is key
    b0 = get_batch(0, state)
    assert tuple(b0[0].shape) == (10, 3) and tuple(b0[1].shape) == (10,)
    assert np.unique(np_(b0[1])).size == 10
    assert np.array_equal(np_(b0[0])[:, 0], np_(b0[1]))
    b1 = get_batch(1, state)
    assert not np.array_equal(np_(b0[1]), np_(b1[1]))
    init, get_batch = subsample_batchify_data((data,), q=0.1, return_mask=True)
    (b,), mask = get_batch(3, rng.PRNGKey(1))
    assert tuple(b.shape) == (10, 3) and bool(mask.all())
    init, get_batch = subsample_batchify_data((data, label), 10, with_replacement=True)
    bw = get_batch(0, rng.PRNGKey(2))
    assert tuple(bw[0].shape) == (10, 3) and float(bw[1].max()) < 105


def test_split_batchifier(rng):
    """reference tests/test_minibatch.py:30-71"""
    from d3p_amd.minibatch import split_batchify_data
    data = torch.arange(105, device="cuda", dtype=torch.float32).reshape(105, 1)
    init, get_batch = split_batchify_data((data,), 10)
    num_batches, idxs = init(rng.PRNGKey(0))
    assert num_batches == 10
    assert np.array_equal(np.sort(np_(idxs)), np.arange(105))
    seen = np.concatenate([np_(get_batch(i, idxs)[0]).ravel() for i in range(num_batches)])
    assert np.unique(seen).size == 100


def test_poisson_batchifier(rng, O):
    """reference tests/test_minibatch.py:242-351"""
    from d3p_amd.minibatch import poisson_batchify_data
    N, q = 100, 0.1
    data = torch.arange(N, device="cuda", dtype=torch.float32).reshape(N, 1) + torch.zeros(N, 2, device="cuda")
    label = torch.arange(N, device="cuda", dtype=torch.float32)
    init, get_batch = poisson_batchify_data((data, label), q, N)
    key = rng.PRNGKey(0)
    num_batches, state = init(key)
    assert num_batches == 10
    sizes = []
    for i in range(300):
        (bx, by), mask = get_batch(i, state)
        assert tuple(bx.shape) == (N, 2) and tuple(by.shape) == (N,) and tuple(mask.shape) == (N,)
        m = np_(mask)
        n = int(m.sum())
        sizes.append(n)
        vals = np_(by)[m]
        assert np.unique(vals).size == n                      # no duplicates
        assert np.all(np_(bx)[~m] == 0) and np.all(np_(by)[~m] == 0)   # zero padding
        # oracle: same selection, descending order
        eidx, nsel, nvalid = O.poisson_select(O.fold_in(O.PRNGKey(0), i), q, N, N)
        assert n == nvalid and np.array_equal(vals, eidx[:n].astype(np.float32))
    # batch sizes ~ Binomial(N, q) ~ Poisson(qN): chi-square against the binomial pmf
    sizes = np.array(sizes)
    assert abs(sizes.mean() - q * N) < 4 * np.sqrt(q * N * (1 - q) / len(sizes))
    # truncate / suppress (tests/test_minibatch.py:315-339)
    init, gb = poisson_batchify_data((data,), 0.9, 10, handle_oversized_batch="truncate")
    assert int(gb(0, key)[1].sum()) == 10
    init, gb = poisson_batchify_data((data,), 0.9, 10, handle_oversized_batch="suppress")
    assert int(gb(0, key)[1].sum()) == 0
    # float max_batch_size -> Poisson quantile (tests/test_minibatch.py:341-351)
    d105 = torch.zeros(105, 1, device="cuda")
    init, gb = poisson_batchify_data((d105,), 0.3, 0.9)
    assert tuple(gb(0, key)[1].shape) == (39,)


def test_take_rows_generic(rng):
    from d3p_amd.util import take_rows
    for shape, dtype in [((50, 7), torch.float32), ((50, 4), torch.float32), ((50,), torch.int32),
                         ((50, 3, 5), torch.float32), ((50, 2), torch.float64)]:
        a = (torch.rand(shape, device="cuda") * 100).to(dtype)
        idx = torch.tensor([3, 0, 49, 3, 17], dtype=torch.int32, device="cuda").view(torch.uint32)
        assert torch.equal(take_rows(a, idx), a[np_(idx.view(torch.int32)).tolist()])


def test_take_rows_never_dereferences_an_index_past_the_table(gpu):
    """The samplers only produce valid rows; a caller's own index array may not: an index past the table is clamped to the last row
    (jnp.take's "clip"), indices that are not a device tensor of 32-bit integers are refused -- neither becomes an address."""
    import d3p_amd._lib as L
    from d3p_amd.util import take_rows
    a = torch.arange(40, dtype=torch.float32, device="cuda").reshape(10, 4)
    idx = torch.tensor([0, 9, 10, 4_000_000_000 - 2**32, 3], dtype=torch.int32, device="cuda")      # 10 and 0xEE6B2800 are past the table
    out = take_rows(a, idx.view(torch.uint32))
    assert torch.equal(out, a[torch.tensor([0, 9, 9, 9, 3], device="cuda")])
    with pytest.raises(L.D3PError):
        take_rows(a, idx.cpu())
    with pytest.raises(L.D3PError):
        take_rows(a, idx.to(torch.int64))
    with pytest.raises(L.D3PError):
        take_rows(a.cpu(), idx)
