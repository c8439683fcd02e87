import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


# The suite exercises the Python classes' composition of the per-kernel entry points (what the oracle comparisons walk link by link); the library's own loop over
# the same buffers (agent0_amd/deepq/native_loop.py, the production default for the configurations it covers) is switched on by the tests that compare it, bit for
# bit, with that composition (tests/test_gpu_trainer.py::test_native_loop_*).
os.environ.setdefault("A0_NATIVE_LOOP", "0")


def _usable_cores() -> int:
    """Cores this process may actually use: the affinity mask, further limited by a cgroup CPU quota (a GPU box reports the whole host in os.cpu_count() and grants
    the container a share: torch's default of one intra-op thread per reported core oversubscribes that share and slows the CPU oracle several-fold)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    import torch

    torch.set_num_threads(max(1, min(torch.get_num_threads(), _usable_cores())))


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass without a device: skip them unless selected with -m gpu."""
    import torch

    has_gpu = torch.cuda.is_available()
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU in this container (run through gpurun)"))
