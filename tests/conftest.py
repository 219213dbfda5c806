import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


# Parity runs are made in the library's deterministic mode (include/vaeseg.h, vs_set_deterministic): the per-(n,c) statistics are then summed
# with commuting integer atomics and two runs of a test agree bit for bit.  The default (fp64-atomic) mode — the one the benchmark runs — is
# exercised by tests/test_gpu_parity_report.py (same checks, with the run-to-run spread stated) and by every bf16 / throughput test that asks
# for it through the `atomic_mode` fixture.
os.environ.setdefault("VS_DETERMINISTIC", "1")


@pytest.fixture
def atomic_mode():
    """run one test in the default (non-deterministic, fp64-atomic) statistics mode"""
    from vae_segmentation_amd import ops
    was = ops.is_deterministic()
    ops.set_deterministic(False)
    yield
    ops.set_deterministic(was)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. in the build container."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
