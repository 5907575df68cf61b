"""The drop-in seam, executed with the REFERENCE's own types: the `HipRiccatiSolver :
gar::RiccatiSolverBase<double>` that INTEGRATION.md prints is extracted VERBATIM from the document, compiled against
/root/reference/include (over the test-only Eigen-API stand-in oracle/ref_shim -- Eigen is absent from this image) and
driven through `RiccatiSolverBase<double>*` the way `SolverProxDDP` drives `linear_solver_` (backward, forward,
collapseFeedback, getFeedforward / getFeedback of every stage), next to the reference's own ProximalRiccatiSolver /
ParallelRiccatiSolver on the same `LqrProblemTpl`: serial, padded inside the C ABI, constrained, leg mode, folded
constraints.  The library underneath is the wave-emulator build (CPU).  Needs /root/reference (build container)."""
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include", "aligator", "gar")),
                                reason="/root/reference absent (GPU box)")


def test_the_printed_binding_compiles_against_the_reference_and_matches_its_solvers(tmp_path):
    import torch
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", doc, flags=re.S)
    binding = next(b for b in blocks if "class HipRiccatiSolver" in b)
    # the only addition: a way for the test to read the kernel family behind the base-class pointer
    binding += ('\ninline const char *gar_hip_kernel_name_of(aligator::gar::RiccatiSolverBase<double> &s) {\n'
                '  return static_cast<aligator::gar::HipRiccatiSolver &>(s).kernelName();\n}\n')
    assert "const char *kernelName() const" in binding, "INTEGRATION.md's class must expose kernelName()"
    (tmp_path / "hip_riccati_binding.hpp").write_text(binding)
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    fmt = tmp_path / "fmt_only"
    fmt.mkdir()
    os.symlink(os.path.join(os.path.dirname(torch.__file__), "include", "fmt"), fmt / "fmt")
    exe = tmp_path / "seam_driver"
    emu = os.path.join(HERE, "emu", "_build")
    cmd = ["g++", "-std=c++17", "-O1", "-fopenmp", "-DFMT_HEADER_ONLY", "-Wno-deprecated-declarations",
           "-I", os.path.join(ROOT, "oracle", "ref_shim"), "-I", os.path.join(REF, "include"), "-I", str(fmt),
           "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path), "-o", str(exe),
           os.path.join(HERE, "integration", "seam_driver.cpp"), os.path.join(REF, "src", "utils", "exceptions.cpp"),
           "-L", emu, "-lgar_hip_emu", f"-Wl,-rpath,{emu}", "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900)
    print(r.stdout)
    assert r.returncode == 0 and "seam ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
