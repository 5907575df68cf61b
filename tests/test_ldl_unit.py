"""The workgroup-level L D L^T building blocks (aligator_amd/csrc/gar_device.hpp) on matrices chosen here:
wg_ldl_definite_factor (blocked elimination without pivoting, used where the block is definite) against
wg_bk_factor (the reference's Bunch-Kaufman, core/bunchkaufman.hpp:23-169), and the blocked MFMA substitution
wg_bk_solve_mfma (:451-518) on factors with interchanges and 2x2 pivots.  Wave emulator here; the same
translation unit is compiled with hipcc for the GPU run (tests/test_gpu_parity.py)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def load_emu():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True, capture_output=True)
    return ctypes.CDLL(os.path.join(HERE, "emu", "_build", "libgar_ldl_unit_emu.so"))


def run_unit(lib, A, X, definite_first, x_rowmajor=False, blocked=0, threads=256):
    n, nc = A.shape[0], X.shape[1]
    a = np.asfortranarray(A, dtype=np.float64).copy(order="F")
    x = np.array(X, dtype=np.float64, order="C" if x_rowmajor else "F")
    sub = np.zeros(n)
    piv = np.zeros(n, dtype=np.int32)
    info = np.zeros(4, dtype=np.int32)
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)
    rc = lib.gar_ldl_unit(n, nc, int(definite_first), int(x_rowmajor), int(blocked), int(threads), a.ctypes.data_as(dp), x.ctypes.data_as(dp),
                          sub.ctypes.data_as(dp), piv.ctypes.data_as(ip), info.ctypes.data_as(ip))
    assert rc == 0
    return a, x, sub, piv, info


def spd(rng, n, cond=1e3):
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (q * np.geomspace(1.0, cond, n)) @ q.T


def check_solution(A, X0, X, tol):
    Afull = np.tril(A) + np.tril(A, -1).T
    res = np.abs(Afull @ X - X0).max() / (np.abs(Afull).max() * np.abs(X).max() + np.abs(X0).max())
    assert res <= tol, res


CASES = [(8, 16), (12, 37), (24, 56), (36, 36), (44, 45), (56, 56), (57, 19), (64, 64), (30, 300)]


@pytest.mark.parametrize("n,ncols", CASES)
@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_definite_factor_matches_bunch_kaufman(n, ncols, sign, lib=None):
    lib = lib or load_emu()
    rng = np.random.default_rng(100 * n + ncols)
    A = sign * spd(rng, n)
    X0 = rng.standard_normal((n, ncols))
    a1, x1, sub1, piv1, info1 = run_unit(lib, A, X0, True)
    a2, x2, sub2, piv2, info2 = run_unit(lib, A, X0, False, x_rowmajor=True)   # (and the other layout of X)
    assert info1[0] == 0 and info1[1] == -1      # definite path taken, Bunch-Kaufman not run
    assert info2[0] == -1 and info2[1] == 0
    assert (piv1 == np.arange(n)).all() and (sub1 == 0).all()
    check_solution(A, X0, x1, 1e-13)
    check_solution(A, X0, x2, 1e-13)
    assert np.abs(x1 - x2).max() <= 1e-10 * np.abs(x2).max()     # cond = 1e3
    # the stored form: unit-lower L, inverse pivots on the diagonal
    L = np.tril(a1, -1) + np.eye(n)
    D = np.diag(1.0 / np.diag(a1))
    assert np.abs(L @ D @ L.T - A).max() <= 1e-12 * np.abs(A).max()


@pytest.mark.parametrize("n,ncols", [(12, 16), (44, 45), (56, 56), (64, 20)])
def test_wrong_sign_pivot_sends_the_block_to_bunch_kaufman(n, ncols, lib=None):
    lib = lib or load_emu()
    rng = np.random.default_rng(n)
    A = spd(rng, n)
    k = n // 2
    A[k:, k:] -= 2.0 * A[k:, k:]          # [[P, B], [B^T, -Q]]: quasi-definite, the pivots change sign at k
    X0 = rng.standard_normal((n, ncols))
    a1, x1, _, piv1, info1 = run_unit(lib, A, X0, True)
    a2, x2, _, piv2, info2 = run_unit(lib, A, X0, False)
    assert info1[0] == 1 and info1[1] == 0       # refused, then factorised by Bunch-Kaufman from the restored block
    assert (a1 == a2).all() and (x1 == x2).all() and (piv1 == piv2).all()
    check_solution(A, X0, x1, 1e-13)


@pytest.mark.parametrize("n,ncols", [(16, 16), (44, 45), (56, 57), (116, 37)])
def test_blocked_substitution_with_interchanges_and_2x2_pivots(n, ncols, lib=None):
    """a KKT-like indefinite matrix with a zero block: Bunch-Kaufman must interchange and take 2x2 pivots"""
    lib = lib or load_emu()
    rng = np.random.default_rng(7 * n)
    m = n // 3
    A = np.zeros((n, n))
    A[:n - m, :n - m] = spd(rng, n - m, 50.0)
    A[:m, :m] = 0.0                                # zero leading block: the first columns fail the diagonal test
    J = rng.standard_normal((m, n - m))
    A[n - m:, :n - m] = J
    A[:n - m, n - m:] = J.T
    A[n - m:, n - m:] = -1e-3 * np.eye(m)
    X0 = rng.standard_normal((n, ncols))
    a, x, sub, piv, info = run_unit(lib, A, X0, False)
    xr = run_unit(lib, A, X0, False, x_rowmajor=True)[1]
    assert info[1] == 0 and np.abs(xr - x).max() <= 1e-12 * np.abs(x).max()
    assert (piv < 0).any() and (piv[piv >= 0] != np.arange(n)[piv >= 0]).any()
    ref = np.linalg.solve(A, X0)
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()
    check_solution(A, X0, x, 1e-12)


def load_gpu():
    so = os.path.join(HERE, "unit", "_build", "libgar_ldl_unit.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-s", "-C", os.path.join(HERE, "unit")], check=True, capture_output=True)
    return ctypes.CDLL(so)


@pytest.mark.gpu
def test_ldl_building_blocks_on_the_gpu():
    lib = load_gpu()
    for n, ncols in CASES:
        for sign in (1.0, -1.0):
            test_definite_factor_matches_bunch_kaufman(n, ncols, sign, lib=lib)
    for n, ncols in [(12, 16), (44, 45), (56, 56), (64, 20)]:
        test_wrong_sign_pivot_sends_the_block_to_bunch_kaufman(n, ncols, lib=lib)
    for n, ncols in [(16, 16), (44, 45), (56, 57), (116, 37)]:
        test_blocked_substitution_with_interchanges_and_2x2_pivots(n, ncols, lib=lib)
    for n, ncols, kind in [(44, 45, "kkt"), (72, 20, "kkt"), (116, 37, "dense-stage"), (128, 16, "random")]:
        for packed in (False, True):
            test_blocked_bunch_kaufman_matches_the_unblocked_one(n, ncols, kind, packed, lib=lib)
    for n, ncols, kind in [(44, 45, "kkt"), (116, 37, "dense-stage"), (56, 57, "kkt")]:
        test_workgroup_size_does_not_change_the_factorisation(n, ncols, kind, lib=lib)


@pytest.mark.gpu
def test_ldl_building_block_cycles_are_reported():
    """not a pass/fail on speed: prints the cycle counts DESIGN.md quotes (pytest -s)"""
    lib = load_gpu()
    rng = np.random.default_rng(1)
    for n, ncols in [(24, 56), (44, 45), (56, 56), (56, 1), (116, 37)]:
        A = spd(rng, n) if n <= 64 else None
        if A is None:
            A = spd(rng, n)
        X0 = rng.standard_normal((n, ncols))
        rows = []
        for definite in ((True, False) if n <= 64 else (False,)):
            best = None
            for _ in range(3):
                info = run_unit(lib, A, X0, definite)[4]
                best = info[2:4] if best is None else np.minimum(best, info[2:4])
            rows.append(("definite" if definite else "bunch-kaufman", int(best[0]), int(best[1])))
        print(f"n={n} ncols={ncols}: " + "; ".join(f"{k}: factor {f} solve {s_} cycles" for k, f, s_ in rows))
        assert all(f > 0 for _, f, _ in rows)
    for n, ncols, kind in [(44, 45, "kkt"), (72, 1, "kkt"), (116, 37, "dense-stage")]:
        A = indefinite(np.random.default_rng(3), n, kind)
        X0 = rng.standard_normal((n, ncols))
        rows = []
        for name, blocked in (("column at a time", 0), ("blocked", 1), ("blocked, packed", 2)):
            best = None
            for _ in range(3):
                info = run_unit(lib, A, X0, False, blocked=blocked)[4]
                best = info[2:4] if best is None else np.minimum(best, info[2:4])
            rows.append((name, int(best[0]), int(best[1])))
        print(f"indefinite n={n} ({kind}): " + "; ".join(f"{k}: factor {f} solve {s_} cycles" for k, f, s_ in rows))


def indefinite(rng, n, kind):
    if kind == "kkt":                       # [[H, J^T], [J, -eps I]] with a zero leading block in H
        m = n // 3
        A = np.zeros((n, n))
        A[:n - m, :n - m] = spd(rng, n - m, 50.0)
        A[:m, :m] = 0.0
        J = rng.standard_normal((m, n - m))
        A[n - m:, :n - m] = J
        A[:n - m, n - m:] = J.T
        A[n - m:, n - m:] = -1e-3 * np.eye(m)
        return A
    if kind == "random":                    # dense symmetric indefinite
        B = rng.standard_normal((n, n))
        return B + B.T
    if kind == "dense-stage":               # the stage-dense solver's matrix (csrc/gar_dense.hpp), nu, nc, nx
        nu, nc = n // 8, n // 4
        nx = (n - nu - nc) // 2
        n2 = nu + nc + 2 * nx
        A = np.zeros((n, n))
        A[:nu, :nu] = spd(rng, nu, 10.0)
        D = rng.standard_normal((nc, nu)); Bm = rng.standard_normal((nx, nu))
        A[nu:nu + nc, :nu] = D; A[:nu, nu:nu + nc] = D.T
        A[nu:nu + nc, nu:nu + nc] = -1e-8 * np.eye(nc)
        o2, o3 = nu + nc, nu + nc + nx
        A[o2:o3, :nu] = Bm; A[:nu, o2:o3] = Bm.T
        A[o3:o3 + nx, o2:o3] = -np.eye(nx); A[o2:o3, o3:o3 + nx] = -np.eye(nx)
        A[o3:o3 + nx, o3:o3 + nx] = spd(rng, nx, 100.0)
        for i in range(n2, n):
            A[i, i] = 1.0 + i
        return A
    raise ValueError(kind)


@pytest.mark.parametrize("n,ncols,kind", [(9, 3, "random"), (16, 16, "kkt"), (44, 45, "kkt"), (44, 37, "random"),
                                          (72, 1, "random"), (72, 20, "kkt"), (116, 37, "dense-stage"),
                                          (128, 16, "random"), (100, 5, "kkt")])
@pytest.mark.parametrize("packed", [False, True])
def test_blocked_bunch_kaufman_matches_the_unblocked_one(n, ncols, kind, packed, lib=None):
    """wg_bk_factor_blocked (panel by one wave, MFMA trailing update) against wg_bk_factor (column at a time): the
    same pivots, factors equal to rounding, the same solution"""
    lib = lib or load_emu()
    rng = np.random.default_rng(1000 * n + ncols)
    A = indefinite(rng, n, kind)
    X0 = rng.standard_normal((n, ncols))
    a1, x1, sub1, piv1, info1 = run_unit(lib, A, X0, False, blocked=2 if packed else 1)
    a2, x2, sub2, piv2, info2 = run_unit(lib, A, X0, False, blocked=0)
    assert info1[1] == 0 and info2[1] == 0
    assert (piv1 == piv2).all(), (piv1, piv2)
    if kind != "random" or n > 9:
        assert (piv1 < 0).any() or (piv1 != np.arange(n)).any()      # the case does pivot
    L1, L2 = np.tril(a1), np.tril(a2)
    tol = max(1e-9, 50 * np.linalg.cond(A) * np.finfo(float).eps)   # the two orders of summation: cond * eps apart
    assert np.abs(L1 - L2).max() <= tol * max(1.0, np.abs(L2).max())
    assert np.abs(sub1 - sub2).max() <= tol * max(1.0, np.abs(sub2).max())
    ref = np.linalg.solve(A, X0)
    scale = np.abs(ref).max()
    assert np.abs(x1 - ref).max() <= 10 * tol * scale and np.abs(x1 - x2).max() <= 10 * tol * scale
    assert np.abs(x1 - ref).max() <= 3 * max(np.abs(x2 - ref).max(), 1e-12 * scale)   # as accurate as the unblocked one
    check_solution(A, X0, x1, 1e-11)


def test_blocked_bunch_kaufman_reports_a_zero_column():
    lib = load_emu()
    rng = np.random.default_rng(5)
    A = indefinite(rng, 20, "random")
    A[7:, 7:] = 0.0                          # after seven steps... not necessarily zero: make the WHOLE matrix zero from a column on
    A[:, :] = 0.0
    A[:3, :3] = spd(rng, 3)
    X0 = rng.standard_normal((20, 4))
    for blocked in (0, 1):
        info = run_unit(lib, A, X0, False, blocked=blocked)[4]
        assert info[1] == 1                  # NumericalIssue (bunchkaufman.hpp:58-59)


@pytest.mark.parametrize("n,ncols,kind", [(44, 45, "kkt"), (40, 16, "random"), (116, 37, "dense-stage"), (56, 57, "kkt")])
def test_workgroup_size_does_not_change_the_factorisation(n, ncols, kind, lib=None):
    """The kernels that call these routines run on 256- and on 1 024-thread workgroups: the column-at-a-time
    Bunch-Kaufman parks per-wave partial results of its pivot search in `subdiag` (n doubles) -- with 16 waves only
    the waves that can hold a row may do so (a 44 x 44 block has room for 14 entries, not 48)."""
    lib = lib or load_emu()
    rng = np.random.default_rng(n)
    A = indefinite(rng, n, kind)
    X0 = rng.standard_normal((n, ncols))
    for blocked in (0, 1):
        a1, x1, sub1, piv1, info1 = run_unit(lib, A, X0, False, blocked=blocked, threads=256)
        a2, x2, sub2, piv2, info2 = run_unit(lib, A, X0, False, blocked=blocked, threads=1024)
        assert info1[1] == 0 and info2[1] == 0
        assert (piv1 == piv2).all()
        assert (np.tril(a1) == np.tril(a2)).all() and (sub1 == sub2).all()
        scale = np.abs(x1).max()
        assert np.abs(x1 - x2).max() <= 1e-12 * scale     # (the substitution's tiles are dealt out differently)
    if n <= 64:
        A = spd(rng, n)
        r1 = run_unit(lib, A, X0, True, threads=256)
        r2 = run_unit(lib, A, X0, True, threads=1024)
        assert r1[4][0] == 0 and r2[4][0] == 0 and (np.tril(r1[0]) == np.tril(r2[0])).all()
