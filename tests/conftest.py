import os
import sys

import pytest

# the tests run on the deterministic synthetic weights (synth.py): opt out of the "no ImageNet checkpoint" warning
# (tests/test_boundary_cpu.py::test_missing_pretrained_weights_are_loud checks the warning itself)
os.environ.setdefault("QTCNN_RESNET18_WEIGHTS", "none")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_eval():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "eval_b2.npz"))


@pytest.fixture(scope="session")
def golden_train():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "train_b4.npz"))


@pytest.fixture(scope="session")
def golden_attn():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "attn_b2.npz"))


@pytest.fixture(scope="session")
def golden_cnn_lstm():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "cnn_lstm_b2t3.npz"))


@pytest.fixture(scope="session")
def golden_clip3d():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "clip3d.npz"))
