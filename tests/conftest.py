"""Shared pytest setup: the `gpu` marker, the package under its on-disk name, and the
oracle binding (test infrastructure only)."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def rtk():
    return importlib.import_module("raytracer-rust_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding

    oracle_binding.build()
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
