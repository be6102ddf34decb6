import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the tests ask for the deterministic stand-in VGG weights explicitly (nerf_qa_amd/vgg_weights.py refuses to
# fall back to them silently); a test that needs another set passes vgg16_path=
os.environ.setdefault("NQA_VGG16_WEIGHTS", "synth:1234")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def np_convs():
    from nerf_qa_amd import synth
    return synth.vgg16_weights(1234)


@pytest.fixture(scope="session")
def oracle_convs(np_convs):
    from oracle import dists_oracle
    return dists_oracle.convs_from_numpy(np_convs)


@pytest.fixture(scope="session")
def alpha_beta():
    import numpy as np
    import torch
    d = np.load(os.path.join(ROOT, "nerf_qa_amd", "data", "dists_alpha_beta.npz"))
    return torch.from_numpy(d["alpha"]).view(1, -1, 1, 1), torch.from_numpy(d["beta"]).view(1, -1, 1, 1)


GOLDEN = os.path.join(ROOT, "tests", "golden")
