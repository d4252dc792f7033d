import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "glue-factory_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    # `-m gpu` on a box without a GPU is a hard error, not a silent skip (a green run that tested nothing);
    # an unfiltered run on a CPU box skips the GPU tests
    expr = (config.getoption("markexpr", "") or "").replace(" ", "")
    if "gpu" in expr and "notgpu" not in expr:
        pytest.exit("`-m gpu` was selected but no HIP device is visible: the GPU parity tests cannot run here", returncode=3)
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def __getitem__(self, k):
        a = self._z[k]
        if a.dtype.kind in "US":
            return a
        return torch.from_numpy(a)

    def __contains__(self, k):
        return k in self._z.files

    def keys(self):
        return self._z.files


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get
