"""Integration smoke in the spirit of the reference's tests/test_examples.py: the example script runs end to end
on the d3p_amd surface (Poisson batchifier -> init -> run_steps -> evaluate -> get_params) and learns."""
import argparse
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("guide", ["auto", "handwritten"])
def test_logistic_regression_example_learns(gpu, guide):
    """`handwritten`: the reference script's OWN guide (examples/logistic_regression.py:67-86: two sample sites, four leaves)."""
    spec = importlib.util.spec_from_file_location("ex_logreg", os.path.join(ROOT, "examples", "logistic_regression.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    args = argparse.Namespace(sigma=0.5, num_epochs=12, learning_rate=5e-2, batch_size=200, dimensions=4, num_samples=10000, guide=guide)
    accs, train_losses = mod.main(args)
    assert len(accs) == 12
    # learns: held-out accuracy well above chance (the labels are noisy: ~0.77 is what the true weights reach)
    # and the training loss falls
    assert accs[-1] > 0.7 and train_losses[-1] < 0.8 * train_losses[0]


def test_simple_gaussian_posterior_example_recovers_the_mean(gpu):
    """BASELINE configs[0] with the example's own defaults (N = 1000, d = 4, batch 10 -> q = 0.01, sigma = 1, C = 1)."""
    spec = importlib.util.spec_from_file_location("ex_gauss", os.path.join(ROOT, "examples", "simple_gaussian_posterior.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    args = argparse.Namespace(sigma=1.0, num_epochs=30, learning_rate=1e-2, batch_size=10, dimensions=4,
                              num_samples=1000, clip_threshold=1.0)
    mu_loc, mu_std, a_loc, a_std = mod.main(args)
    # 3000 noisy, clipped steps from mu_loc = 0: the mean has moved most of the way to 1 and the scale shrank
    assert float((mu_loc - a_loc).abs().max()) < 0.35
    assert float(mu_std.max()) < 1.0


def test_gaussian_mixture_example_separates_the_clusters(gpu):
    """BASELINE config 3's example end to end (Poisson batches + mask, update, evaluate, get_params)."""
    spec = importlib.util.spec_from_file_location("ex_gmm", os.path.join(ROOT, "examples", "gaussian_mixture_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    args = argparse.Namespace(num_epochs=12, learning_rate=5e-2, batch_size=64, dimensions=2, num_samples=2048,
                              num_components=3, sigma=0.3)
    acc, pis, modes = mod.main(args)
    assert acc > 0.9
    assert abs(float(pis.sum()) - 1.0) < 1e-5


def test_vae_example_learns_to_reconstruct(gpu):
    """BASELINE config 5's example (784 -> 400 -> 50) end to end on synthetic binary images."""
    spec = importlib.util.spec_from_file_location("ex_vae", os.path.join(ROOT, "examples", "vae.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # little noise and a larger step size than the script's defaults: the test runs only 320 steps
    args = argparse.Namespace(num_epochs=10, learning_rate=3e-3, batch_size=128, z_dim=50, hidden_dim=400, num_samples=4096,
                              sigma=0.01)
    errs = mod.main(args)
    assert errs[-1] < 0.75 * errs[0]
