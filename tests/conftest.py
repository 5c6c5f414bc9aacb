import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-segmentation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    class Merged:
        """Read-only union of several .npz fixture files."""
        def __init__(self, *files):
            self.files = [np.load(os.path.join(d, f)) for f in files]

        def __getitem__(self, key):
            for f in self.files:
                if key in f.files:
                    return f[key]
            raise KeyError(key)

        def keys(self):
            return {k for f in self.files for k in f.files}
    return {"losses": np.load(os.path.join(d, "losses.npz")), "models": Merged("models.npz", "variants.npz")}
