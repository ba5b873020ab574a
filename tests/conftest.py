import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
PKG = "anticipated-vins-mono_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def mod(name):
    return importlib.import_module(PKG + "." + name)


@pytest.fixture(scope="session")
def abi():
    return mod("abi")


@pytest.fixture(scope="session")
def synth():
    return mod("synth")


@pytest.fixture(scope="session")
def buffers():
    return mod("buffers")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py

    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def ctx():
    """avm_ctx on cuda:0 — fails loudly (no CPU fallback) if the HIP library or the GPU is missing."""
    return mod("lib").Context(0)


@pytest.fixture(scope="session")
def estimator(ctx, abi):
    opt = abi.default_options()
    opt.marginalization_flag = abi.MARGIN_NONE
    return mod("estimator").Estimator(ctx=ctx, options=opt)


@pytest.fixture(scope="session")
def selector(ctx):
    return mod("feature_selector").FeatureSelector(ctx=ctx)
