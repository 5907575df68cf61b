"""The Python scripting surface beyond the solvers (bindings/python/src/gar/expose-utils.cpp:26-37,
expose-prox-riccati.cpp:30-31, 48-52): lqrCreateSparseMatrix, StageFactor.kktChol, kkt0.mat / kkt0.chol."""
import os
import subprocess

import numpy as np
import pytest

from aligator_amd import synth
from aligator_amd.gar import (BunchKaufman, ProximalRiccatiSolver, lqrCreateSparseMatrix, lqrInitializeSolution,
                              lqrNumRows)
import parity_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "_build", "libgar_hip_emu.so")


def test_bunch_kaufman_mirror_takes_the_reference_pivots():
    """The host-side BunchKaufman (aligator_amd/lqr.py) against the oracle's restatement of core/bunchkaufman.hpp
    (itself pinned on the reference's compiled code, tests/test_ref_pin.py): identical pivot sequences -- 1x1, with
    interchanges, 2x2 --, L D L^T reproduces the matrix, solve() solves."""
    from oracle import oracle as ora
    rng = np.random.default_rng(0)
    two_by_two = swaps = 0
    for trial in range(120):
        n = int(rng.integers(1, 24))
        A = rng.standard_normal((n, n))
        A = A + A.T
        if trial % 3 == 0:
            A[np.diag_indices(n)] *= 0.01          # small diagonal: 2x2 pivots
        if trial % 4 == 1:
            A = A @ A.T + 1e-3 * np.eye(n)         # definite: natural order
        bk = BunchKaufman(np.tril(A) + 7.0 * np.triu(np.ones((n, n)), 1))   # only the lower triangle is read
        assert bk.info
        opiv = ora.BunchKaufman(A).pivots
        assert np.array_equal(bk.pivots, opiv), (trial, bk.pivots, opiv)
        two_by_two += int((bk.pivots < 0).sum()) // 2
        swaps += int(sum(1 for k, p in enumerate(bk.pivots) if p >= 0 and p != k))
        B = rng.standard_normal((n, 3))
        X = bk.solve(B)
        assert np.abs(A @ X - B).max() <= 1e-9 * max(1.0, np.abs(X).max()) * np.linalg.cond(A)
        assert np.allclose(bk.solve(B[:, 0]), X[:, 0])
    assert two_by_two > 20 and swaps > 20
    assert not BunchKaufman(np.zeros((3, 3))).info


def test_sparse_kkt_matrix_follows_the_reference_layout():
    """lqrCreateSparseMatrix: sizes, symmetry, the blocks where gar/utils.hxx:8-86 puts them -- and, with the
    reference's own sign convention for the x_{t+1} coupling undone, solving it reproduces the Riccati solution."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(3)
    nx, nu, nc, N, mu = 5, 2, 2, 6, 1e-4
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), N, nx, nu, nc=nc, mode="W")
    mat, rhs = lqrCreateSparseMatrix(prob, mu, False)
    n = lqrNumRows(prob)
    assert sp.issparse(mat) and mat.shape == (n, n) and rhs.shape == (n,)
    M = mat.toarray()
    assert np.array_equal(M, M.T)
    k0 = prob.stages[0]
    assert np.array_equal(M[:nx, nx:2 * nx], prob.G0) and np.array_equal(rhs[:nx], prob.g0)
    assert np.array_equal(M[nx:2 * nx, nx:2 * nx], k0.Q) and np.array_equal(M[nx:2 * nx, 2 * nx:2 * nx + nu], k0.S)
    i1 = 2 * nx + nu
    assert np.array_equal(M[i1:i1 + nc, i1:i1 + nc], -mu * np.eye(nc))
    i2 = i1 + nc
    assert np.array_equal(M[i2:i2 + nx, nx:2 * nx], k0.A) and np.array_equal(M[i2:i2 + nx, i2 + nx:i2 + 2 * nx], np.eye(nx))
    # the reference writes +I where its own residual convention (lqrComputeKktError, the tests' dense builder) has -I:
    # flip those blocks and the matrix is the KKT matrix of the problem
    K = M.copy()
    idx = nx
    for t, k in enumerate(prob.stages[:-1]):
        i2 = idx + k.nx + k.nu + k.nc
        i3 = i2 + k.nx2
        K[i2:i2 + k.nx2, i3:i3 + k.nx2] *= -1.0
        K[i3:i3 + k.nx2, i2:i2 + k.nx2] *= -1.0
        idx = i3
    z = -spla.spsolve(sp.csc_matrix(K), rhs)
    _, _, ref = pc.oracle_serial(prob, mu)
    idx, sc = nx, pc.scale_of(ref)
    assert np.abs(z[:nx] - ref[3][0]).max() <= 1e-8 * sc
    for t, k in enumerate(prob.stages):
        assert np.abs(z[idx:idx + k.nx] - ref[0][t]).max() <= 1e-8 * sc
        idx += k.nx + k.nu + k.nc + (k.nx2 if t < N else 0)


def test_kkt_chol_and_kkt0_views_on_the_emulator():
    """datas[t].kktChol factorises datas[t].kktMat with the pivots the oracle's stage factorisation took; kkt0.mat is
    [Vxx0 G0^T; G0 0] and kkt0.chol solves it to kkt0.ff."""
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    from oracle import oracle as ora
    rng = np.random.default_rng(5)
    nx, nu, nc, N, mu = 6, 3, 2, 4, 1e-6
    prob = synth.generate_lq_problem(rng, rng.standard_normal(nx), N, nx, nu, nc=nc, mode="W")
    s = ProximalRiccatiSolver(prob, lib_path=EMU)
    assert s.backward(mu)
    os_ = ora.ProximalRiccatiSolver(pc.to_oracle(prob))
    os_.backward(mu)
    for t in range(N):
        f = s.datas[t]
        chol = f.kktChol
        assert chol.info and np.array_equal(chol.pivots, ora.BunchKaufman(os_.datas(t).kktMat).pivots)
        K = np.tril(f.kktMat) + np.tril(f.kktMat, -1).T
        rhs = -np.concatenate([np.zeros(nu), np.ones(nc)])
        assert np.abs(K @ chol.solve(rhs) - rhs).max() <= 1e-9 * np.linalg.cond(K)
    k0 = s.kkt0
    M = k0.mat
    assert M.shape == (2 * nx, 2 * nx) and np.array_equal(M[nx:, :nx], prob.G0) and not M[nx:, nx:].any()
    V0, v0 = s.datas[0].vm.Vxx, s.datas[0].vm.vx
    x_l = k0.chol.solve(-np.concatenate([v0, prob.g0]))       # proximal-riccati.hxx:44-52
    assert np.abs(x_l - k0.ff).max() <= 1e-9 * max(1.0, np.abs(k0.ff).max())
    assert np.array_equal(np.tril(M[:nx, :nx]), np.tril(V0))


def test_compat_shim_serves_aligator_gar():
    """compat/aligator: `import aligator.gar` answered by aligator_amd (opt-in, for scripts written against the
    reference's bindings).  The submodule carries every top-level name bindings/python/src/gar/expose-*.cpp registers
    (scanned from the reference when it is present), the classes ARE the mirror's, and a script in the reference's
    idiom -- LqrKnot / LqrProblem / ProximalRiccatiSolver(problem).backward(mu) / forward(xs, us, vs, lbdas) /
    lqrComputeKktError -- runs through it (on the emulator library here)."""
    import re
    import sys
    root = os.path.dirname(HERE)
    code = f"""
import numpy as np
import aligator
from aligator import gar
import aligator.gar as gar2
import aligator_amd.gar as mirror
assert gar is gar2 and gar.ProximalRiccatiSolver is mirror.ProximalRiccatiSolver
nx, nu, N = 8, 4, 6
rng = np.random.default_rng(1)
knots = []
for t in range(N + 1):
    k = gar.LqrKnot(nx, nu if t < N else 0, 0)
    k.Q[:] = np.eye(nx) * 2.0
    k.q[:] = rng.standard_normal(nx)
    if t < N:
        k.R[:] = np.eye(nu)
        k.r[:] = rng.standard_normal(nu)
        k.A[:] = np.eye(nx) + 0.1 * rng.standard_normal((nx, nx))
        k.B[:] = rng.standard_normal((nx, nu))
        k.f[:] = 0.1 * rng.standard_normal(nx)
    knots.append(k)
prob = gar.LqrProblem(knots, nx)
prob.G0[:] = -np.eye(nx)
prob.g0[:] = rng.standard_normal(nx)
solver = gar.ProximalRiccatiSolver(prob, lib_path={EMU!r})
xs, us, vs, lbdas = gar.lqrInitializeSolution(prob)
assert solver.backward(1e-12) and solver.forward(xs, us, vs, lbdas)
err = gar.lqrComputeKktError(prob, xs, us, vs, lbdas, 1e-12, None, False)
assert max(err) < 1e-9, err
print("NAMES", " ".join(sorted(gar.__all__)))
"""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, os.path.join(root, "compat")]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    names = set(out.stdout.split("NAMES", 1)[1].split())
    expected = {"LqrKnot", "LqrProblem", "RiccatiSolverBase", "ProximalRiccatiSolver", "ParallelRiccatiSolver",
                "RiccatiSolverDense", "lqrComputeKktError", "lqrCreateSparseMatrix", "lqrInitializeSolution"}
    assert expected <= names
    ref = "/root/reference/bindings/python/src/gar"
    if os.path.isdir(ref):  # what the reference registers at the top level of its gar module
        found = set()
        for fn in os.listdir(ref):
            src = open(os.path.join(ref, fn)).read()
            found |= set(re.findall(r'bp::def\(\s*"(\w+)"', src))
            found |= set(re.findall(r'bp::class_<[^;]*?>\s*\(\s*"(\w+)"', src, re.S))
        top = {n for n in found if n in expected or n.startswith(("lqr", "Riccati", "Proximal", "Parallel", "Lqr"))}
        assert top <= names, top - names
