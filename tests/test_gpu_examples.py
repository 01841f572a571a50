"""Integration smoke in the spirit of the reference's tests/test_examples.py: the example script runs end to end
on the d3p_amd surface (Poisson batchifier -> init -> run_steps -> evaluate -> get_params) and learns."""
import argparse
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_logistic_regression_example_learns(gpu):
    spec = importlib.util.spec_from_file_location("ex_logreg", os.path.join(ROOT, "examples", "logistic_regression.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    args = argparse.Namespace(sigma=0.5, num_epochs=12, learning_rate=5e-2, batch_size=200, dimensions=4, num_samples=10000)
    accs, train_losses = mod.main(args)
    assert len(accs) == 12
    # learns: held-out accuracy well above chance (the labels are noisy: ~0.77 is what the true weights reach)
    # and the training loss falls
    assert accs[-1] > 0.7 and train_losses[-1] < 0.8 * train_losses[0]
