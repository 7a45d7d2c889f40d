import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # under pytest-xdist every worker would start a full-width OpenMP team: N workers x all cores spin against each other and the
    # CPU oracle's convolutions slow down by orders of magnitude, so split the cores between the workers
    workers = int(os.environ.get("PYTEST_XDIST_WORKER_COUNT", "0") or 0)
    if workers > 1:
        import torch
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // workers))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
