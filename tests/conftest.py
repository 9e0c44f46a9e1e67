"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks, gloo DP tests.
`-m gpu`      : HIP path vs oracle / golden vectors, through the C-ABI, on a real MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_generate_tests(metafunc):
    """GPU tests that take `conv_mode` run twice: dense conv and token-product formulation (textcnn_prod.hip)."""
    if "conv_mode" in metafunc.fixturenames and metafunc.definition.get_closest_marker("gpu") is not None:
        metafunc.parametrize("conv_mode", ["dense", "product"], indirect=True)


@pytest.fixture
def conv_mode(request):
    from review_based_recommender_amd import _lib
    mode = getattr(request, "param", "auto")
    _lib.lib().rbr_set_conv_mode({"auto": 0, "dense": 1, "product": 2}[mode])
    yield mode
    _lib.lib().rbr_set_conv_mode(0)
