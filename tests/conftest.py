import faulthandler
import os
import signal
import sys
import threading

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(HERE, "golden")

# Per-test wall-clock limit (seconds).  A test that overruns fails BY NAME instead of blocking `pytest -x` until the
# caller's limit kills the run with nothing to show for it: first a SIGALRM that raises in the test (covers anything that
# returns to the interpreter: waiting for a child, numpy, the oracle releasing the GIL ...), and 45 s later, for a test
# stuck inside a native call that never returns, faulthandler dumps every thread's Python stack and ends the process.
TEST_LIMIT_S = int(os.environ.get("RPA_TEST_LIMIT_S", "300"))

# GPU suite order: the kernel-vs-reference parity tests first, the tests that start other programs last -- so that
# whatever goes wrong with a child process, the parity evidence has already been produced (the driver runs with -x).
GPU_ORDER = ["test_gpu_parity.py", "test_lca_models.py", "test_gpu_soak.py", "test_gpu_cli.py", "test_gpu_fullsize.py", "test_gpu_dist.py"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "limit(seconds): per-test wall-clock limit instead of RPA_TEST_LIMIT_S")


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.basename(str(item.fspath))
        return GPU_ORDER.index(name) if name in GPU_ORDER else -1   # everything else keeps its place in front
    items.sort(key=key)  # stable: the order inside a file is the file's


# A line per test as it starts and ends, in gpurun_out/ (scratch, merged back from the GPU box): if a run is cut short from
# outside, the file still says which test it was in and how long every earlier one took.
_PROGRESS = os.path.join(ROOT, "gpurun_out", "pytest_progress.log")


def _progress(line):
    try:
        os.makedirs(os.path.dirname(_PROGRESS), exist_ok=True)
        with open(_PROGRESS, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def pytest_sessionstart(session):
    import time
    _progress("SESSION pid=%d %s args=%s" % (os.getpid(), time.strftime("%Y-%m-%dT%H:%M:%S"), " ".join(session.config.invocation_params.args)))


def pytest_runtest_logstart(nodeid, location):
    _progress("START %s" % nodeid)


def pytest_runtest_logreport(report):
    if report.when == "call" or report.outcome != "passed":
        _progress("%s %s %s %.2fs" % (report.outcome.upper(), report.when, report.nodeid, report.duration))


class TestTimeout(Exception):
    pass


@pytest.fixture(autouse=True)
def _per_test_limit(request):
    m = request.node.get_closest_marker("limit")
    limit = int(m.args[0]) if m else TEST_LIMIT_S
    if limit <= 0 or threading.current_thread() is not threading.main_thread() or not hasattr(signal, "SIGALRM"):
        yield
        return

    def on_alarm(signum, frame):
        raise TestTimeout("%s exceeded its %d s limit" % (request.node.nodeid, limit))

    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.setitimer(signal.ITIMER_REAL, limit)
    faulthandler.dump_traceback_later(limit + 45, exit=True, file=sys.stderr)
    try:
        yield
    finally:
        faulthandler.cancel_dump_traceback_later()
        signal.setitimer(signal.ITIMER_REAL, 0)
        signal.signal(signal.SIGALRM, old)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from helpers import oracle as O
    O.lib()  # builds oracle/liboracle.so on first use
    return O
