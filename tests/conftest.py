import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def smpl():
    from avatar_amd import synth
    return synth.load_model(0)


@pytest.fixture(scope="session")
def omodel(smpl):
    from oracle import oracle as orc
    return orc.OracleModel(smpl)


@pytest.fixture(scope="session")
def gmodel(smpl):
    from avatar_amd import api
    return api.AvatarModel(smpl)


@pytest.fixture(scope="session")
def frame0(smpl):
    from avatar_amd import synth
    return synth.make_frame(smpl, 0)
