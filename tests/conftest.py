import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-fingerprinting_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _native_library_is_built():
    """The HIP library is built in-tree by __graft_entry__.build(); build it (hipcc cross-compiles without a
    GPU) if a fresh checkout reaches the tests first.  Tests never fall back to anything else."""
    import __graft_entry__ as ge
    newest = max(os.path.getmtime(os.path.join(ge.CSRC, f)) for f in os.listdir(ge.CSRC))
    if not os.path.exists(ge.LIB) or os.path.getmtime(ge.LIB) < newest:
        ge.build()
    yield


def natural_frame():
    """The reference's own 1920x1080 test image (tests/media/imgs/frame63.jpeg, a data file), decoded with Pillow
    exactly as tools/make_golden.py did."""
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, "frame63.jpeg")).convert("RGB"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def golden_cases():
    """DCT codec cases."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and f != "payload_codecs.npz" and not f.endswith("_digest.npz")
                  and not f.startswith("svd_"))


def svd_golden_cases():
    """DwtDctSvd codec cases."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("svd_") and f.endswith(".npz"))
