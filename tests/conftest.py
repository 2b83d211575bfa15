import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu under gpurun)")
    # the oracle is plain PyTorch on the CPU: one OpenMP thread per machine core under a container CPU quota (16 of 256 CPUs on
    # the GPU boxes) gets the whole process throttled (sam_road_amd/hostcpu.py)
    import torch
    from sam_road_amd.hostcpu import usable_cpus
    torch.set_num_threads(min(torch.get_num_threads(), usable_cpus()))


def load_golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
