"""The C++ host mirror (include/gar_hip.hpp) through its own C++ test program
(tests/cpp/test_gar.cpp, written like tests/gar/riccati.cpp / tests/gar/parallel.cpp):
on the wave emulator here (CPU), on the real library on the GPU box."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")


def _run(binary, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([os.path.join(CPP, "_build", binary)], capture_output=True, text=True,
                          env=e, timeout=900)


def test_cpp_host_mirror_on_emulator():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    subprocess.run(["make", "-s", "-C", CPP, "emu"], check=True)
    r = _run("test_gar_emu", {"GAR_TEST_SMALL": "1"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all passed" in r.stdout


def test_cpp_host_mirror_links_and_fails_loudly_without_gpu():
    lib = os.path.join(os.path.dirname(HERE), "aligator_amd", "libgar_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libgar_hip.so not built")
    subprocess.run(["make", "-s", "-C", CPP], check=True)
    r = _run("test_gar")
    # 0 on a GPU box; 77 + a clear message where there is no HIP device (no CPU path)
    assert r.returncode in (0, 77), r.stdout + r.stderr
    if r.returncode == 77:
        assert "no HIP device" in r.stdout and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_cpp_host_mirror_on_gpu():
    subprocess.run(["make", "-s", "-C", CPP], check=True)
    r = _run("test_gar")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all passed" in r.stdout


@pytest.mark.gpu
def test_newton_iteration_through_the_seam_on_gpu():
    """bench/lqr.cpp's loop restated on the host boundary (tests/cpp/bench_lqr_loop.cpp): upload of every
    knot, backward, forward, collapseFeedback, every stage's gains -- serial and N/8 legs agree, and the
    bulk read-back keeps an iteration in the sub-millisecond range at the north-star shape."""
    import re
    subprocess.run(["make", "-s", "-C", os.path.join(os.path.dirname(HERE), "oracle")], check=True)
    subprocess.run(["make", "-s", "-C", CPP], check=True)
    r = subprocess.run([os.path.join(CPP, "_build", "bench_lqr_loop"), "256"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout)
    lines = [ln for ln in r.stdout.splitlines() if "north star" in ln]
    assert len(lines) == 2 and "wave_leg<36,12>" in lines[1]
    us = [float(re.search(r"([0-9.]+) us / Newton", ln).group(1)) for ln in lines]
    diff = float(re.search(r"= ([0-9.e+-]+)$", lines[1]).group(1))
    assert diff < 1e-6 and us[1] < 2500.0, r.stdout
