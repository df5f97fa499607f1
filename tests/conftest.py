"""pytest config: registers the `gpu` marker and puts the product package
(`strotss-tensorflow_amd/`, which holds the reference-named `nn` package and
`run_strotss.py`) and the repo root (for `oracle`) on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "strotss-tensorflow_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
