import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# A test stuck inside a native call (a device wait that never returns, a join) cannot be interrupted by a Python
# signal handler; faulthandler's watchdog THREAD can still dump every thread's Python stack and end the process, so a
# hang fails within minutes with the line it sits on instead of being killed silently by the box.  Re-armed at the
# start of every test (module fixtures are torn down inside the last test's window).  BPG_TEST_WATCHDOG=0 disables.
_WATCHDOG_S = int(os.environ.get("BPG_TEST_WATCHDOG", "360"))
_WATCHDOG_FILE = None
# BPG_TEST_WATCHDOG_ABORT=<seconds>: additionally raise SIGABRT in the process that long into a test -- for a run under
# a debugger (`rocgdb -batch -ex run -ex "thread apply all bt" --args python -m pytest ...`), which then prints the
# NATIVE stacks of the hang.
_ABORT_S = int(os.environ.get("BPG_TEST_WATCHDOG_ABORT", "0"))
_ABORT_TIMER = None


def pytest_runtest_logstart(nodeid, location):
    global _WATCHDOG_FILE, _ABORT_TIMER
    if _ABORT_S > 0:
        import signal
        import threading
        if _ABORT_TIMER is not None:
            _ABORT_TIMER.cancel()
        _ABORT_TIMER = threading.Timer(_ABORT_S, lambda: os.kill(os.getpid(), signal.SIGABRT))
        _ABORT_TIMER.daemon = True
        _ABORT_TIMER.start()
    if _WATCHDOG_S > 0:
        import faulthandler
        if _WATCHDOG_FILE is None:  # pytest captures fd 2 while a test runs: the dump goes to a file of its own
            d = os.path.join(ROOT, "gpurun_out")
            _WATCHDOG_FILE = open(os.path.join(d if os.path.isdir(d) else "/tmp", "test_watchdog.txt"), "w")
        _WATCHDOG_FILE.seek(0)
        _WATCHDOG_FILE.truncate()
        _WATCHDOG_FILE.write("watchdog armed for %s (%d s)\n" % (nodeid, _WATCHDOG_S))
        _WATCHDOG_FILE.flush()
        faulthandler.dump_traceback_later(_WATCHDOG_S, exit=True, file=_WATCHDOG_FILE)


def pytest_sessionfinish(session, exitstatus):
    import faulthandler
    faulthandler.cancel_dump_traceback_later()
    if _ABORT_TIMER is not None:
        _ABORT_TIMER.cancel()


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
