import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Lazy view of one tests/golden/*.npz file; values come back as torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def __getitem__(self, key):
        return torch.from_numpy(self._z[key])

    def keys(self, prefix=""):
        return [k for k in self._z.files if k.startswith(prefix)]

    def state_dict(self, prefix):
        return {k[len(prefix):]: torch.from_numpy(self._z[k]) for k in self._z.files if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def rel_err(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1.0)).item()
