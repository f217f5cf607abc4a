import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _gpu_run_heartbeat():
    """On a GPU box a few tests build multi-GB synthetic models for minutes without finishing a test (pytest -q prints
    nothing meanwhile) and the box's watchdog takes 7 silent minutes for a hang: while the session runs, a line is appended
    to gpurun_out/heartbeat.log every minute (the current test's id with it).  A real hang is still bounded: every GPU test
    carries a 20-minute pytest-timeout (below)."""
    import threading
    import time
    import torch
    if not torch.cuda.is_available():
        yield
        return
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    # the oracle's CPU forwards are the bulk of this suite's minutes; a one-GPU box exposes every host thread but grants a
    # share of 16 cores, and torch's default (all of them) oversubscribes that share 4x (15 s instead of 4 s per oracle forward)
    torch.set_num_threads(min(torch.get_num_threads(), int(os.environ.get("DGQ_TEST_THREADS", "16"))))
    stop = threading.Event()

    def beat():
        t0 = time.time()
        while not stop.wait(60.0):
            with open(os.path.join(ROOT, "gpurun_out", "heartbeat.log"), "a") as fh:
                fh.write("%6.0f s  %s\n" % (time.time() - t0, os.environ.get("PYTEST_CURRENT_TEST", "")))
    th = threading.Thread(target=beat, daemon=True)
    th.start()
    yield
    stop.set()


def pytest_collection_modifyitems(config, items):
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(1200))
