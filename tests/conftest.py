import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG_NAME = "nuclei-feature-extraction_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a checkout without the built artefacts (they are git-ignored), or with a libnfx.so that was built from OTHER sources than the ones
    # next to it (nfx_build_stamp() against csrc/stamp.py, asked of a child process), is built once, exactly as __graft_entry__.build()
    # does; the product itself never builds or falls back at run time -- it fails loudly when libnfx.so is missing
    pkg = os.path.join(ROOT, PKG_NAME)
    built = [os.path.join(pkg, "libnfx.so"), os.path.join(pkg, "bin", "nuclei-feature-extraction"), os.path.join(pkg, "bin", "filter-geojson")]
    fresh = all(os.path.exists(b) for b in built)
    if fresh:
        import bench
        fresh = bench.library_is_fresh()
    if not fresh and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def nfx():
    """The product package (host mirror of the reference's FeatureSet interface over the C ABI)."""
    # torch carries its own HIP runtime: when both live in one process torch's has to come up first (the other order
    # leaves torch with "No HIP GPUs are available"), so bring it up before libnfx makes its first HIP call
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG_NAME + ".synth")
