"""The CPU suite spreads itself over pytest-xdist workers (tests/conftest.py::pytest_cmdline_main).  A worker runs the
same hook; if it spread itself again the run would be a fork bomb (it was, once): the guards are tested here."""
import importlib.util
import os
import types

HERE = os.path.dirname(os.path.abspath(__file__))


def _conftest():
    spec = importlib.util.spec_from_file_location("_gar_conftest_under_test", os.path.join(HERE, "conftest.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _config(**extra):
    opt = types.SimpleNamespace(numprocesses=None, markexpr="not gpu", collectonly=False, usepdb=False)
    return types.SimpleNamespace(option=opt, **extra)


def test_worker_never_spreads_itself(monkeypatch):
    ct = _conftest()
    monkeypatch.setenv("GAR_TESTS_WORKERS", "4")
    # (1) the worker's own marker on the config object
    monkeypatch.delenv("PYTEST_XDIST_WORKER", raising=False)
    monkeypatch.delenv("GAR_TESTS_XDIST_PARENT", raising=False)
    cfg = _config(workerinput={"workerid": "gw0"})
    assert ct.pytest_cmdline_main(cfg) is None and cfg.option.numprocesses is None
    # (2) pytest-xdist's environment marker
    monkeypatch.setenv("PYTEST_XDIST_WORKER", "gw0")
    cfg = _config()
    assert ct.pytest_cmdline_main(cfg) is None and cfg.option.numprocesses is None
    # (3) the sentinel the spreading process leaves for every descendant
    monkeypatch.delenv("PYTEST_XDIST_WORKER")
    monkeypatch.setenv("GAR_TESTS_XDIST_PARENT", "1")
    cfg = _config()
    assert ct.pytest_cmdline_main(cfg) is None and cfg.option.numprocesses is None


def test_gpu_selection_and_explicit_n_are_left_alone(monkeypatch):
    ct = _conftest()
    for k in ("PYTEST_XDIST_WORKER", "GAR_TESTS_XDIST_PARENT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("GAR_TESTS_WORKERS", "4")
    cfg = _config()
    cfg.option.markexpr = "gpu"                      # one device: never parallel
    assert ct.pytest_cmdline_main(cfg) is None and cfg.option.numprocesses is None
    cfg = _config()
    cfg.option.numprocesses = 2                      # the caller's own -n wins
    assert ct.pytest_cmdline_main(cfg) is None and cfg.option.numprocesses == 2
    monkeypatch.setenv("GAR_TESTS_WORKERS", "1")     # serial on request
    cfg = _config()
    assert ct.pytest_cmdline_main(cfg) is None and cfg.option.numprocesses is None
