import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "coresident: needs the launches of two streams / processes resident on ONE GPU together "
                                       "(bounded spin waits, subprocesses over hipIpc): collected LAST")


# Collection order of the suite (the driver runs `pytest -x`): the hot path's parity with the oracle and the golden fixtures FIRST,
# the model families next, host logic after that, and every test whose outcome also depends on the box -- two streams' launches
# co-resident on one GPU, spin waits across processes -- LAST, so that such a test going red can never hide the parity evidence.
_FILE_ORDER = [
    "test_golden.py", "test_gpu_production_kernels.py", "test_gpu_dpsvi.py", "test_gpu_rng.py", "test_gpu_minibatch.py",
    "test_gpu_configs.py", "test_gpu_gauss.py", "test_gpu_gmm.py", "test_gpu_gmm_model.py", "test_gpu_vae.py", "test_gpu_adadp.py",
    "test_gpu_examples.py", "test_oracle_pins.py", "test_reference_vectors.py", "test_host_logic.py", "test_accountant.py", "test_dist.py",
]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_FILE_ORDER)}

    def key(pair):
        pos, item = pair
        fname = os.path.basename(str(item.fspath))
        late = 1 if item.get_closest_marker("coresident") is not None else 0
        return (late, rank.get(fname, len(_FILE_ORDER)), pos)

    items[:] = [it for _, it in sorted(enumerate(items), key=key)]


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (oracle/): the checker, never the thing under test in -m gpu tests."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import d3p_amd._lib as L
    L.load()
    L.require_device()
    return torch.device("cuda:0")
