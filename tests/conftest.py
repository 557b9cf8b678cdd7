import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
ALIST = os.path.join(ROOT, "short_ldpc_decoding_osd_amd", "data", "CCSDS_ldpc_n128_k64.alist")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The suites need libldpcosd.so (host entry points on CPU, kernels on the GPU box) and the C oracle.
    Both are normally built by __graft_entry__.build(); build them here if a fresh checkout has neither
    (hipcc cross-compiles gfx950 without a GPU).  A failed build is reported by the tests that need it."""
    try:
        from short_ldpc_decoding_osd_amd import build as hip_build
        hip_build.build()
        from oracle import c_oracle
        c_oracle.build()
    except Exception as exc:   # noqa: BLE001
        print(f"conftest: automatic build failed: {exc}", file=sys.stderr)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def alist_path():
    return ALIST


@pytest.fixture(scope="session")
def np_code():
    from oracle import np_oracle
    return np_oracle.Code(ALIST)
