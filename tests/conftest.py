import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.pyoracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libalp_ref.so not built (needs /root/reference; run oracle/Makefile `ref`)")
    return Reference()


@pytest.fixture(scope="session")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no GPU is visible (alp_amd has no CPU fallback)")
    from alp_amd import capi
    c = capi.Context(0)
    yield c
    c.close()
