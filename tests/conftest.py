import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "automatic-speech-recognition-asr-onnx_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # The CPU oracles are torch f32 graphs over 13..137-row matrices: on a 256-core host torch's default intra-op pool (one thread per core) makes them
    # slower by orders of magnitude (bench.py measures 0.04 audio-s/s at 256 threads against 57 at 16), and the oracle calls are most of the GPU suite's wall time.
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except Exception:          # noqa: BLE001 -- torch is only needed by the oracle-backed tests
        pass


def sub(name: str):
    """Import a sub-module of the (hyphenated) product package."""
    return importlib.import_module(f"{PKG_NAME}.{name}")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG_NAME)
