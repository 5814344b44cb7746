import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# The golden vectors inject the sampler's per-step churn noise as torch tensors (monkey-patched torch.randn): the drop-in modules
# built by the tests therefore materialise it the reference's way.  The product default -- draws generated inside the sampler's
# kernels (PlMcedm.noise_source = "device", mcedm_heun_sample_rng) -- has its own tests, which set the attribute explicitly.
os.environ.setdefault("MCEDM_NOISE_SOURCE", "torch")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name)))
        return cache[name]

    return load
