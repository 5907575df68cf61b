"""The drop-in seam, executed with the REFERENCE's own types: the `HipRiccatiSolver :
gar::RiccatiSolverBase<double>` this repository ships as include/aligator/gar/hip-riccati.hpp is compiled against
/root/reference/include (over the test-only Eigen-API stand-in oracle/ref_shim -- Eigen is absent from this image) and
driven through `RiccatiSolverBase<double>*` the way `SolverProxDDP` drives `linear_solver_` (backward, forward,
collapseFeedback, getFeedforward / getFeedback of every stage), next to the reference's own ProximalRiccatiSolver /
ParallelRiccatiSolver on the same `LqrProblemTpl`: serial, padded inside the C ABI, constrained, leg mode, folded
constraints, and the legs split over 2 and 3 (virtual) devices behind the one solver object.  The library underneath is the wave-emulator build (CPU).  Needs /root/reference (build container)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"

needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include", "aligator", "gar")),
                                     reason="/root/reference absent (GPU box)")


@pytest.mark.gpu
def test_gpu_reference_typed_binding_on_the_device():
    """The seam on the MI355X with the reference's OWN types: oracle/_ref/seam_driver_gpu is tests/integration/
    seam_driver.cpp -- `HipRiccatiSolver : RiccatiSolverBase<double>` (the shipped include/aligator/gar/hip-riccati.hpp)
    against the reference's LqrProblemTpl, next to the reference's own ProximalRiccatiSolver / ParallelRiccatiSolver
    in the same process -- compiled where /root/reference exists (tests/integration/build_gpu_driver.sh, from
    __graft_entry__.build()) and linked with the REAL aligator_amd/libgar_hip.so.  Serial, padded (56, 22), nc = 32,
    leg mode, folded constraints, legs over two sub-solvers (devices {0, 0}); solution and every stage's gains to 1e-8
    of their scale (one exception, stated in seam_driver.cpp: the SOLUTION of the constrained knots folded into legs,
    multipliers of order 1 / mu, is held to 1e-5 -- its gains to 1e-8 like the rest), the kernel family that ran; the
    iteration timed on both sides."""
    if os.path.isdir(os.path.join(REF, "include", "aligator", "gar")):
        subprocess.run(["bash", os.path.join(HERE, "integration", "build_gpu_driver.sh")], check=True)
    exe = os.path.join(ROOT, "oracle", "_ref", "seam_driver_gpu")
    assert os.path.exists(exe), ("oracle/_ref/seam_driver_gpu is built by __graft_entry__.build() in the container that "
                                 "holds /root/reference and travels with the snapshot: run build() there first")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(r.stdout)
    assert r.returncode == 0 and "seam ok" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok\n") >= 10 and "MISMATCH" not in r.stdout


@pytest.mark.gpu
def test_gpu_reference_proxddp_loop_on_the_device():
    """BASELINE configs[0] / [4] as stated: the reference's OWN SolverProxDDP loop (compiled unchanged from
    /root/reference over the Eigen stand-in: oracle/ref_ddp_build.sh -> oracle/_ref/libaligator_ddp_ref.so) with
    `linear_solver_` replaced by the shipped HipRiccatiSolver on the real library: tests/lqr.cpp's case converges in
    ONE iteration exactly as with the reference's own solvers (same trajectories to 1e-8), and bench/lqr.cpp's loop
    (dim 56, nu 22: the Talos-walk LQ shape) is timed -- ProxDDP iterations per second, reference SERIAL / PARALLEL
    beside the backend serial / in leg mode, same box (tests/integration/proxddp_lqr_driver.cpp)."""
    if os.path.isdir(os.path.join(REF, "include", "aligator", "gar")):
        subprocess.run(["bash", os.path.join(HERE, "integration", "build_gpu_driver.sh")], check=True)
    exe = os.path.join(ROOT, "oracle", "_ref", "proxddp_lqr_gpu")
    assert os.path.exists(exe), "oracle/_ref/proxddp_lqr_gpu is built by __graft_entry__.build() where /root/reference exists"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print("\n".join(ln for ln in r.stdout.splitlines() if "Warning" not in ln))
    assert r.returncode == 0 and "proxddp ok" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
    # the specialised families ran (the terminal knot ProxDDP builds has nx2 = 0: the binding declares it with nx2 = nx)
    assert "<8,4>  ok" in r.stdout and "kernel wave_leg<8,4>" in r.stdout and "kernel pair<56,24>" in r.stdout and "kernel pair_leg<56,24>" in r.stdout


@needs_reference
def test_the_reference_proxddp_loop_runs_on_the_backend(tmp_path):
    """the same driver on CPU: linked with the wave-emulator build, tests/lqr.cpp's case in full (num_iters == 1 with
    the reference's solvers and with HipRiccatiSolver, serial and 4 legs) and a small bench/lqr.cpp-shaped loop"""
    subprocess.run(["bash", os.path.join(ROOT, "oracle", "ref_ddp_build.sh")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    refdir, emu = os.path.join(ROOT, "oracle", "_ref"), os.path.join(HERE, "emu", "_build")
    exe = tmp_path / "proxddp_lqr_emu"
    cmd = ["g++", "-std=c++17", "-O1", "-fopenmp", "-DFMT_HEADER_ONLY", "-include", "aligator/context.hpp",
           "-Wno-deprecated-declarations", "-I", os.path.join(ROOT, "oracle", "ref_shim"), "-I", os.path.join(REF, "include"),
           "-I", os.path.join(refdir, "fmt_only"), "-I", os.path.join(ROOT, "include"), "-o", str(exe),
           os.path.join(HERE, "integration", "proxddp_lqr_driver.cpp"), "-L", refdir, "-laligator_ddp_ref",
           f"-Wl,-rpath,{refdir}", "-L", emu, "-lgar_hip_emu", f"-Wl,-rpath,{emu}", "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([str(exe), "--quick"], capture_output=True, text=True, timeout=3000)
    out = "\n".join(ln for ln in r.stdout.splitlines() if "Warning" not in ln)
    print(out)
    assert r.returncode == 0 and "proxddp ok" in r.stdout, out[-3000:] + r.stderr[-2000:]
    assert out.count("num_iters 1 ") == 4 and "kernel wave<8,4>" in out and "kernel wave_leg<8,4>" in out


@needs_reference
def test_the_shipped_binding_compiles_against_the_reference_and_matches_its_solvers(tmp_path):
    import torch
    header = os.path.join(ROOT, "include", "aligator", "gar", "hip-riccati.hpp")
    assert "class HipRiccatiSolver : public RiccatiSolverBase<double>" in open(header).read()
    # INTEGRATION.md points at the file instead of printing a copy of it
    assert "include/aligator/gar/hip-riccati.hpp" in open(os.path.join(ROOT, "INTEGRATION.md")).read()
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    fmt = tmp_path / "fmt_only"
    fmt.mkdir()
    os.symlink(os.path.join(os.path.dirname(torch.__file__), "include", "fmt"), fmt / "fmt")
    exe = tmp_path / "seam_driver"
    emu = os.path.join(HERE, "emu", "_build")
    # (the reference's include/ first: every aligator/... header but hip-riccati.hpp is found there)
    cmd = ["g++", "-std=c++17", "-O1", "-fopenmp", "-DFMT_HEADER_ONLY", "-Wno-deprecated-declarations",
           "-I", os.path.join(ROOT, "oracle", "ref_shim"), "-I", os.path.join(REF, "include"), "-I", str(fmt),
           "-I", os.path.join(ROOT, "include"), "-o", str(exe),
           os.path.join(HERE, "integration", "seam_driver.cpp"), os.path.join(REF, "src", "utils", "exceptions.cpp"),
           "-L", emu, "-lgar_hip_emu", f"-Wl,-rpath,{emu}", "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=1800)
    print(r.stdout)
    assert r.returncode == 0 and "seam ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
