import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import rta
    p = rta.load()
    from importlib import import_module
    b = import_module("ray_tracer_archive_amd.build")
    b.build()          # no-op when lib/librt_hip.so is newer than its sources
    p.lib()
    return p


@pytest.fixture(scope="session")
def orc():
    from oracle import binding
    binding.build()
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def gpu(pkg):
    """A context on GPU 0. No fallback: if the library or the device is missing this raises."""
    ctx = pkg.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def earth():
    import numpy as np
    from PIL import Image
    return np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "earthmap_rgb.png")).convert("RGB"))
