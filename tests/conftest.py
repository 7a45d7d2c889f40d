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
    cpus = _cpu_share()
    if workers > 1 or cpus < (os.cpu_count() or 1):
        import torch
        torch.set_num_threads(max(1, cpus // max(workers, 1)))


def _cpu_share() -> int:
    """CPUs this process may really use: the affinity mask, capped by the container's cgroup CPU bandwidth.  A GPU box shows all
    256 logical CPUs of the host but grants 16 of them: torch's default OpenMP team (one thread per visible core) then spins against
    its own quota and the CPU oracle runs several times slower (bench.py's thread sweep: 16 threads 12.5 tiles/s, 32 threads 5.9)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                    # cgroup v2: "<quota> <period>" or "max <period>"
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:                                # cgroup v1
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
