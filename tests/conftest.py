import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
GOLDEN = REPO / "tests" / "golden"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds")


@pytest.fixture(autouse=True)
def _oracle_threads():
    """The CPU oracle's last float bits depend on the intra-op thread count; the goldens were generated with 8."""
    import torch
    torch.set_num_threads(8)
    yield
    torch.set_num_threads(8)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(GOLDEN / name, allow_pickle=False)
    return load


def fingerprint(net):
    """Same recipe as oracle/gen_goldens.py:weight_fingerprint."""
    sd = net.state_dict()
    tot = sum(float(v.double().sum()) for v in sd.values() if v.dtype.is_floating_point)
    sq = sum(float((v.double() ** 2).sum()) for v in sd.values() if v.dtype.is_floating_point)
    return np.array([tot, sq, float(sd["segmentation_head.0.weight"].flatten()[3]),
                     float(sd["encoder.layer3.2.conv1.weight"].flatten()[1234])], dtype=np.float64)
