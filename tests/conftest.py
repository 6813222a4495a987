"""pytest configuration: ``gpu`` marker + repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Latency-mode split-K (round 6: on by default for SMALL towers, engine.ImageTower.SPLIT_K_AUTO_PIXELS) changes the summation
# order of the long-K layers — deterministic, batch-invariant inside the small class, 1e-6 from the unsplit launches.  Most GPU tests
# here assert BIT-identity between two kernels or two schedules on miniature towers; they pin the unsplit launches.  The default
# ("auto") is what tests/test_gpu_network.py::test_latency_mode_* , the smoke test and bench.py run.
os.environ.setdefault("WEDETECT_SPLIT_K", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
