"""pytest configuration: markers and import paths.

`-m "not gpu"`: oracle vs. golden vectors, host logic, C-ABI symbol checks (no GPU needed).
`-m gpu`      : parity of the HIP path (through the C-ABI) against the oracle; needs an MI355X.
"""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle", ROOT / "open-msspe-design_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def oracle_tables(oracle):
    return oracle.Tables()
