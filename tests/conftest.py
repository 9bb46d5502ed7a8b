import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on demand with gcc."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def bpg():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import proof_protocol_decoder_amd as pkg
    pkg.lib()  # raises if the HIP library is missing: no silent fallback
    return pkg
