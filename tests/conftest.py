import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import i2r_amd  # noqa: E402,F401  (registers the package under its importable name)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# -m gpu runs oracle / golden comparisons FIRST, the subprocess-driven bench / collective tests LAST, so that `-x` can never hide the
# parity evidence behind a launcher problem
_ORDER = ["test_model_gpu", "test_kernels_gpu", "test_post_gpu", "test_input_gpu", "test_lanes", "test_dist_gpu"]


def _rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    return _ORDER.index(name) if name in _ORDER else -1      # CPU-side files keep their place in front


def pytest_collection_modifyitems(config, items):
    import torch
    items.sort(key=_rank)                                    # stable: the order inside one file is unchanged
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
