"""Property tests (hypothesis) of the host-side logic that runs without a GPU: the data-parallel split arithmetic, the jax-like tree
flattening the optimiser vectors rely on, the example guide's leaf permutation, the batch-size helpers of d3p/minibatch.py:315-322."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st


@settings(max_examples=200, deadline=None)
@given(n=st.integers(0, 10**9), world=st.integers(1, 64))
def test_shard_rows_is_a_partition_into_contiguous_near_equal_ranges(n, world):
    from d3p_amd.dist import shard_batch, shard_rows
    bounds = [shard_rows(n, r, world) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n
    sizes = []
    for (lo, hi), (lo2, _) in zip(bounds, bounds[1:] + [(n, n)]):
        assert lo <= hi == lo2
        sizes.append(hi - lo)
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)       # the first n % world ranks hold one more
    for r in range(world):
        assert shard_batch(n, r, world) == (bounds[r][0], bounds[r][1] - bounds[r][0])


_leaf = st.builds(lambda shape, seed: torch.arange(int(np.prod(shape)) if shape else 1, dtype=torch.float32).reshape(shape) + seed,
                  st.lists(st.integers(1, 3), max_size=2).map(tuple), st.integers(0, 5))
_tree = st.recursive(_leaf, lambda kids: st.one_of(st.lists(kids, max_size=3), st.lists(kids, max_size=3).map(tuple),
                                                     st.dictionaries(st.text("abcxyz", min_size=1, max_size=3), kids, max_size=3)),
                     max_leaves=8)


@settings(max_examples=150, deadline=None)
@given(tree=_tree)
def test_tree_flatten_roundtrips_and_sorts_dict_keys_like_jax(tree):
    from d3p_amd.svi import _tree_flatten, _tree_unflatten
    leaves, treedef = _tree_flatten(tree)
    back = _tree_unflatten(treedef, leaves)

    def same(a, b):
        if isinstance(a, torch.Tensor):
            return isinstance(b, torch.Tensor) and torch.equal(a, b)
        if isinstance(a, dict):
            return isinstance(b, dict) and sorted(a) == sorted(b) and all(same(a[k], b[k]) for k in a)
        return type(a) is type(b) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    assert same(tree, back)

    def expected(t):       # jax.tree_util.tree_leaves: depth first, dict keys sorted, () and [] hold nothing
        if isinstance(t, torch.Tensor):
            return [t]
        if isinstance(t, dict):
            return [l for k in sorted(t) for l in expected(t[k])]
        return [l for x in t for l in expected(x)]
    want = expected(tree)
    assert len(leaves) == len(want) and all(torch.equal(a, b) for a, b in zip(leaves, want))


@settings(max_examples=100, deadline=None)
@given(d=st.integers(1, 5000))
def test_example_guide_leaf_permutation_is_a_bijection_onto_the_kernel_columns(d):
    """MeanFieldGuide: tree order (intercept_loc, intercept_std_log, w_loc (d), w_std_log (d)) <-> kernel columns [w, intercept | scales]."""
    from d3p_amd.models import MeanFieldGuide
    perm = MeanFieldGuide.tree_from_kernel(d).numpy()
    D = d + 1
    assert sorted(perm.tolist()) == list(range(2 * D))
    kernel = np.arange(2 * D)
    tree = kernel[perm]
    assert tree[0] == d and tree[1] == D + d                                 # the two intercept leaves come first
    assert np.array_equal(tree[2:2 + d], np.arange(d)) and np.array_equal(tree[2 + d:], D + np.arange(d))
    assert MeanFieldGuide(type("M", (), {"intercept": True})()).leaf_sizes(d) == [1, 1, d, d]


@settings(max_examples=200, deadline=None)
@given(N=st.integers(1, 10**8), B=st.integers(1, 10**8))
def test_batch_size_and_sampling_ratio_helpers(N, B):
    from d3p_amd.minibatch import batch_size_to_q, q_to_batch_size
    B = min(B, N)
    q = batch_size_to_q(B, N)
    assert q == B / N and 0 < q <= 1
    assert q_to_batch_size(q, N) == int(N * q)                              # d3p/minibatch.py:315-317: truncation, as the reference
    assert abs(q_to_batch_size(q, N) - B) <= 1
