import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def fray():
    import fray_amd
    return fray_amd


@pytest.fixture(scope="session")
def abi():
    from fray_amd import abi
    return abi


@pytest.fixture(scope="session")
def oracle(abi):
    """The CPU oracle: test infrastructure, the checker for the HIP path."""
    from oracle.oracle import Oracle
    return Oracle(abi)


def open_scene(fray, name, W=None, H=None, **over):
    s = fray.Scene.parseScene(os.path.join(SCENES, name))
    if W:
        s.settings.frameWidth, s.settings.frameHeight = W, H
    for k, v in over.items():
        if hasattr(s.settings, k):
            setattr(s.settings, k, v)
        elif hasattr(s.camera, k):
            setattr(s.camera, k, v)
        else:
            raise AttributeError(k)
    return s


@pytest.fixture(scope="session")
def gpu(fray):
    rc = fray.lib.frayhip_init(0)
    if rc != 0:
        pytest.fail("frayhip_init(0) failed on a GPU-marked test: %s" % fray.lib.frayhip_last_error())
    return 0
