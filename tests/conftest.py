import os
import sys

import pytest

# In a process that uses both PyTorch and librt_hip.so, torch must come first: torch bundles its own HIP/HSA runtime, and a second
# runtime initialised after ours finds "no HIP GPUs". Loaded first, torch's libamdhip64 is the one copy both share.
try:
    import torch  # noqa: F401
except Exception:      # torch is plumbing here; the CPU tests of the library do not need it
    torch = None

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


METRICS = os.path.join(ROOT, "gpurun_out", "parity_metrics.jsonl")


def record_metric(**kw):
    """Append one observed parity figure (GPU runs write them under gpurun_out/; the tolerances in the tests are ~2x these)."""
    import json
    try:
        os.makedirs(os.path.dirname(METRICS), exist_ok=True)
        with open(METRICS, "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass
