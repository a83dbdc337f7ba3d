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


def sub(name: str):
    """Import a sub-module of the (hyphenated) product package."""
    return importlib.import_module(f"{PKG_NAME}.{name}")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG_NAME)
