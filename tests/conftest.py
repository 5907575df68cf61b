import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_budget():
    """Cores this container may use (cgroup quota if there is one)."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except (OSError, ValueError):
        pass
    return n


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) spends its time in the lane-per-thread emulator (one OS thread per lane, up to
    1 024 per workgroup): ~35 min on one core, ~6 min over eight.  When pytest-xdist is importable and the caller gave
    no -n, the CPU selection is spread over the cores this container may use; everything a worker would otherwise
    `make` is built once beforehand (eight workers racing on one target is not a test).  The GPU selection is never
    parallelised (one device).  GAR_TESTS_WORKERS=1 keeps the run serial, =<n> picks the count."""
    opt = config.option
    # Never inside a worker: pytest-xdist runs this very hook in each worker with numprocesses reset to None, and a
    # worker that spreads its own run over eight more workers is a fork bomb.  Three independent guards (the worker's
    # own markers, and a sentinel every descendant inherits through the environment).
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER") \
            or os.environ.get("GAR_TESTS_XDIST_PARENT"):
        return None
    if not hasattr(opt, "numprocesses") or opt.numprocesses not in (None, 0):
        return None
    if "not gpu" not in (getattr(opt, "markexpr", "") or "") or getattr(opt, "collectonly", False) \
            or getattr(opt, "usepdb", False):
        return None
    want = os.environ.get("GAR_TESTS_WORKERS", "")
    n = int(want) if want.isdigit() else min(8, _cpu_budget())
    if n <= 1:
        return None
    import subprocess
    for d, target in ((os.path.join(ROOT, "oracle"), None), (os.path.join(ROOT, "tests", "emu"), None),
                      (os.path.join(ROOT, "tests", "cpp"), "emu")):
        if os.path.exists(os.path.join(d, "Makefile")):
            subprocess.run(["make", "-s", "-C", d] + ([target] if target else []), check=False)
    os.environ["GAR_TESTS_XDIST_PARENT"] = str(os.getpid())
    opt.numprocesses = n     # pytest-xdist's own pytest_cmdline_main (which runs after this one) does the rest
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/gar_oracle.c via ctypes); test infrastructure."""
    from oracle import oracle as ora
    ora.lib()
    return ora
