"""ctypes front-end of oracle/_ref/libgar_ref.so -- the REFERENCE's own gar sources (core/bunchkaufman.hpp,
gar/riccati-kernel.hxx, gar/proximal-riccati.hxx, gar/parallel-solver.hxx, gar/block-tridiagonal.hpp,
gar/lqr-problem.hxx), compiled UNCHANGED from /root/reference over the minimal Eigen-API stand-in oracle/ref_shim
(oracle/ref_build.sh; Eigen is absent from this image).  TEST INFRASTRUCTURE, not the product: it pins
oracle/gar_oracle.c (tests/test_ref_pin.py) and generated tests/golden/ref_*.npz (tests/golden/make_ref_golden.py).
The .so is prebuilt in the build container (where /root/reference exists) and travels to the GPU box.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libgar_ref.so")
_PD = C.POINTER(C.c_double)
_BLOCKS = ("Q", "S", "R", "q", "r", "A", "B", "f", "C", "D", "d", "Gth", "Gx", "Gu", "Gv", "gamma")
_lib = None


def available() -> bool:
    return os.path.exists(PATH) or os.path.isdir("/root/reference/include/aligator/gar")


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["bash", os.path.join(_HERE, "ref_build.sh")], check=True)
        L = C.CDLL(PATH)
        L.ref_last_error.restype = C.c_char_p
        L.ref_problem_new.restype = C.c_void_p
        L.ref_problem_new.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]
        L.ref_problem_free.argtypes = [C.c_void_p]
        L.ref_problem_set_knot.argtypes = [C.c_void_p, C.c_int] + [_PD] * 16
        L.ref_problem_set_init.argtypes = [C.c_void_p, _PD, _PD]
        L.ref_problem_get_knot.argtypes = [C.c_void_p, C.c_int, C.c_int, _PD]
        for nm in ("ref_serial_new",):
            getattr(L, nm).restype = C.c_void_p
            getattr(L, nm).argtypes = [C.c_void_p]
        L.ref_parallel_new.restype = C.c_void_p
        L.ref_parallel_new.argtypes = [C.c_void_p, C.c_int]
        L.ref_serial_free.argtypes = [C.c_void_p]
        L.ref_parallel_free.argtypes = [C.c_void_p]
        L.ref_serial_backward.argtypes = [C.c_void_p, C.c_double]
        L.ref_parallel_backward.argtypes = [C.c_void_p, C.c_double]
        L.ref_serial_forward.argtypes = [C.c_void_p, C.c_void_p, _PD, _PD, _PD, _PD, _PD]
        L.ref_parallel_forward.argtypes = [C.c_void_p, C.c_void_p, _PD, _PD, _PD, _PD]
        L.ref_serial_factor.argtypes = [C.c_void_p, C.c_int, C.c_int, _PD]
        L.ref_parallel_factor.argtypes = [C.c_void_p, C.c_int, C.c_int, _PD]
        L.ref_serial_initial.argtypes = [C.c_void_p, C.c_int, _PD]
        L.ref_parallel_set_refinement.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.ref_parallel_collapse_feedback.argtypes = [C.c_void_p]
        L.ref_parallel_condensed_dim.argtypes = [C.c_void_p]
        L.ref_parallel_condensed_solution.argtypes = [C.c_void_p, _PD]
        L.ref_dense_new.restype = C.c_void_p
        L.ref_dense_new.argtypes = [C.c_void_p]
        L.ref_dense_free.argtypes = [C.c_void_p]
        L.ref_dense_backward.argtypes = [C.c_void_p, C.c_double]
        L.ref_dense_forward.argtypes = [C.c_void_p, C.c_void_p, _PD, _PD, _PD, _PD, _PD]
        L.ref_dense_factor.argtypes = [C.c_void_p, C.c_int, C.c_int, _PD]
        L.ref_dense_initial.argtypes = [C.c_void_p, C.c_int, _PD]
        L.ref_problem_cycle.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        for nm in ("ref_serial_cycle_append", "ref_parallel_cycle_append", "ref_dense_cycle_append"):
            getattr(L, nm).argtypes = [C.c_void_p, C.c_void_p]
        L.ref_kkt_error.argtypes = [C.c_void_p, _PD, _PD, _PD, _PD, C.c_double, _PD, _PD]
        L.ref_bk_compute.argtypes = [C.c_int, _PD, _PD, _PD, C.POINTER(C.c_int)]
        L.ref_bk_solve.argtypes = [C.c_int, _PD, C.c_int, _PD]
        L.ref_block_tridiag_solve.argtypes = [C.c_int, C.POINTER(C.c_int), _PD, _PD, _PD, _PD, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(_PD)


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


class Problem:
    """aligator::gar::LqrProblemTpl<double> filled from an aligator_amd.lqr.LqrProblem-like object."""

    def __init__(self, prob):
        L = lib()
        self.dims = np.ascontiguousarray([k.dims for k in prob.stages], dtype=np.int32)
        self.N = len(prob.stages) - 1
        self.nc0 = int(prob.nc0)
        self._h = L.ref_problem_new(self.N, self.dims.ctypes.data_as(C.POINTER(C.c_int)), self.nc0)
        keep = []
        for t, k in enumerate(prob.stages):
            blocks = [_f(getattr(k, nm)) for nm in _BLOCKS]
            keep.append(blocks)
            L.ref_problem_set_knot(self._h, t, *[_p(b) for b in blocks])
        G0, g0 = _f(prob.G0), _f(prob.g0)
        L.ref_problem_set_init(self._h, _p(G0), _p(g0))

    def __del__(self):
        try:
            lib().ref_problem_free(self._h)
        except Exception:
            pass

    def cycle(self, knot):
        """The caller's side of an MPC cycle (solvers/proxddp/workspace.hxx:122-126): rotate_vec_left(stages, 0, 1),
        a fresh knot of `knot`'s dimensions in the last-but-one slot, filled with its blocks."""
        L = lib()
        d = np.ascontiguousarray(knot.dims, dtype=np.int32)
        L.ref_problem_cycle(self._h, d.ctypes.data_as(C.POINTER(C.c_int)))
        blocks = [_f(getattr(knot, nm)) for nm in _BLOCKS]
        L.ref_problem_set_knot(self._h, self.N - 1, *[_p(b) for b in blocks])
        nd = self.dims.copy()
        nd[:self.N - 1] = self.dims[1:self.N]
        nd[self.N - 1] = d
        self.dims = nd

    def sizes(self):
        d = self.dims
        nx, nu, nc, nl = int(d[:, 0].sum()), int(d[:self.N, 1].sum()) + (int(d[self.N, 1]) if d[self.N, 1] else 0), \
            int(d[:, 2].sum()), self.nc0 + int(d[:self.N, 3].sum())
        return nx, nu, nc, nl

    def split(self, xs, us, vs, lbdas):
        d, N = self.dims, self.N
        X, U, V, Lm = [], [], [], [lbdas[:self.nc0].copy()]
        px = pu = pv = 0
        pl = self.nc0
        for t in range(N + 1):
            X.append(xs[px:px + d[t, 0]].copy()); px += d[t, 0]
            if not (t == N and d[t, 1] == 0):
                U.append(us[pu:pu + d[t, 1]].copy()); pu += d[t, 1]
            V.append(vs[pv:pv + d[t, 2]].copy()); pv += d[t, 2]
            if t < N:
                Lm.append(lbdas[pl:pl + d[t, 3]].copy()); pl += d[t, 3]
        return X, U, V, Lm


class _Factor:
    pass


class _SolverBase:
    _factor = None

    def datas(self, t, nth=None):
        """StageFactor of stage t as plain arrays: ff, fb (row-major like the reference: shape (nu+nc+nx2, nx)),
        fth, Vxx, vx, Vxt, Vtt, vt, kktMat, Rhat, pivots."""
        nx, nu, nc, nx2, nth0 = (int(v) for v in self.problem.dims[t])
        nth = self.nth(t) if nth is None else nth
        nr, nk = nu + nc + nx2, nu + nc
        shapes = [(nr,), (nr, nx), (nr, nth), (nx, nx), (nx,), (nx, nth), (nth, nth), (nth,), (nk, nk), (nu, nu), (nk,)]
        f = _Factor()
        for what, (nm, shp) in enumerate(zip(("ff", "fb", "fth", "Vxx", "vx", "Vxt", "Vtt", "vt", "kktMat", "Rhat",
                                              "pivots"), shapes)):
            a = np.zeros(shp, order="F")
            if a.size:
                self._factor(self._h, t, what, _p(a))
            setattr(f, nm, np.ascontiguousarray(a))
        f.pivots = f.pivots.astype(np.int64)
        return f


class ProximalRiccatiSolver(_SolverBase):
    """aligator::gar::ProximalRiccatiSolver<double> (gar/proximal-riccati.hxx)."""

    def __init__(self, problem: Problem):
        self.problem = problem
        self._h = lib().ref_serial_new(problem._h)
        self._factor = lib().ref_serial_factor

    def __del__(self):
        try:
            lib().ref_serial_free(self._h)
        except Exception:
            pass

    def nth(self, t):
        return int(self.problem.dims[t, 4])

    def cycleAppend(self):
        """ProximalRiccatiSolver::cycleAppend (proximal-riccati.hxx:79-86) on the problem's last-but-one knot, as
        solver-proxddp.hxx:208 calls it (after Problem.cycle)."""
        if lib().ref_serial_cycle_append(self._h, self.problem._h):
            raise RuntimeError(lib().ref_last_error().decode())

    def backward(self, mueq: float) -> bool:
        rc = lib().ref_serial_backward(self._h, float(mueq))
        if rc < 0:
            raise RuntimeError(lib().ref_last_error().decode())
        return rc == 0

    def forward(self, theta=None):
        nx, nu, nc, nl = self.problem.sizes()
        xs, us, vs, ls = np.zeros(nx), np.zeros(nu), np.zeros(nc), np.zeros(nl)
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        lib().ref_serial_forward(self._h, self.problem._h, _p(th) if th is not None else None, _p(xs), _p(us), _p(vs), _p(ls))
        return self.problem.split(xs, us, vs, ls)

    def initial(self):
        nx0, nth = int(self.problem.dims[0, 0]), int(self.problem.dims[0, 4])
        n0 = nx0 + self.problem.nc0
        out = []
        for what, shp in enumerate(((n0,), (n0, nth), (nth,), (nth, nth))):
            a = np.zeros(shp, order="F")
            if a.size:
                lib().ref_serial_initial(self._h, what, _p(a))
            out.append(np.ascontiguousarray(a))
        return out


class ParallelRiccatiSolver(_SolverBase):
    """aligator::gar::ParallelRiccatiSolver<double> (gar/parallel-solver.hxx); mutates `problem` like the reference."""

    def __init__(self, problem: Problem, num_threads: int):
        self.problem = problem
        self.num_threads = int(num_threads)
        self._h = lib().ref_parallel_new(problem._h, self.num_threads)
        if not self._h:
            raise RuntimeError(lib().ref_last_error().decode())
        self._factor = lib().ref_parallel_factor

    def __del__(self):
        try:
            if self._h:
                lib().ref_parallel_free(self._h)
        except Exception:
            pass

    def nth(self, t):
        N, J = self.problem.N, self.num_threads
        for i in range(J):
            b, e = i * (N + 1) // J, (i + 1) * (N + 1) // J
            if b <= t < e:
                return 0 if i == J - 1 else int(self.problem.dims[e - 1, 3])
        raise IndexError(t)

    def cycleAppend(self):
        """ParallelRiccatiSolver::cycleAppend (parallel-solver.hxx:246-258): drops every parameterisation and
        initialises again (the problem's dims change under it: re-read them with Problem.refresh_nth)."""
        if lib().ref_parallel_cycle_append(self._h, self.problem._h):
            raise RuntimeError(lib().ref_last_error().decode())

    def set_refinement(self, thr, steps):
        lib().ref_parallel_set_refinement(self._h, float(thr), int(steps))

    def backward(self, mueq: float) -> bool:
        rc = lib().ref_parallel_backward(self._h, float(mueq))
        if rc < 0:
            raise RuntimeError(lib().ref_last_error().decode())
        return rc == 0

    def forward(self):
        nx, nu, nc, nl = self.problem.sizes()
        xs, us, vs, ls = np.zeros(nx), np.zeros(nu), np.zeros(nc), np.zeros(nl)
        lib().ref_parallel_forward(self._h, self.problem._h, _p(xs), _p(us), _p(vs), _p(ls))
        return self.problem.split(xs, us, vs, ls)

    def collapseFeedback(self):
        lib().ref_parallel_collapse_feedback(self._h)

    def condensed_solution(self):
        out = np.zeros(lib().ref_parallel_condensed_dim(self._h))
        lib().ref_parallel_condensed_solution(self._h, _p(out))
        return out


class RiccatiSolverDense:
    """aligator::gar::RiccatiSolverDense<double> (gar/dense-riccati.hxx over gar/dense-kernel.hpp)."""

    def __init__(self, problem: Problem):
        self.problem = problem
        self._h = lib().ref_dense_new(problem._h)

    def __del__(self):
        try:
            lib().ref_dense_free(self._h)
        except Exception:
            pass

    def cycleAppend(self):
        """RiccatiSolverDense::cycleAppend (dense-riccati.hxx:118-146)."""
        if lib().ref_dense_cycle_append(self._h, self.problem._h):
            raise RuntimeError(lib().ref_last_error().decode())

    def backward(self, mueq: float) -> bool:
        rc = lib().ref_dense_backward(self._h, float(mueq))
        if rc < 0:
            raise RuntimeError(lib().ref_last_error().decode())
        return rc == 0

    def forward(self, theta=None):
        nx, nu, nc, nl = self.problem.sizes()
        xs, us, vs, ls = np.zeros(nx), np.zeros(nu), np.zeros(nc), np.zeros(nl)
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        lib().ref_dense_forward(self._h, self.problem._h, _p(th) if th is not None else None, _p(xs), _p(us), _p(vs), _p(ls))
        return self.problem.split(xs, us, vs, ls)

    def datas(self, t):
        """ff, fb, ft with rows [K; Z; L; Y] and Pxx, px, Pxt, Ptt, pt of stage t."""
        nx, nu, nc, nx2, nth = (int(v) for v in self.problem.dims[t])
        n = nu + nc + 2 * nx2
        f = _Factor()
        for what, (nm, shp) in enumerate((("ff", (n,)), ("fb", (n, nx)), ("ft", (n, nth)), ("Pxx", (nx, nx)), ("px", (nx,)),
                                          ("Pxt", (nx, nth)), ("Ptt", (nth, nth)), ("pt", (nth,)))):
            a = np.zeros(shp, order="F")
            if a.size:
                lib().ref_dense_factor(self._h, t, what, _p(a))
            setattr(f, nm, np.ascontiguousarray(a))
        return f

    def initial(self):
        nx0, nth = int(self.problem.dims[0, 0]), int(self.problem.dims[0, 4])
        n0 = nx0 + self.problem.nc0
        out = []
        for what, shp in enumerate(((n0,), (n0, nth), (nth,), (nth, nth))):
            a = np.zeros(shp, order="F")
            if a.size:
                lib().ref_dense_initial(self._h, what, _p(a))
            out.append(np.ascontiguousarray(a))
        return out


def kkt_error(problem: Problem, xs, us, vs, lbdas, mueq, theta=None):
    """aligator::gar::lqrComputeKktError (gar/utils.hxx:88-182): (dynErr, cstErr, dualErr)."""
    cat = lambda v: np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in v])
                                         if len(v) else np.zeros(0))
    X, U, V, Lm = cat(xs), cat(us), cat(vs), cat(lbdas)
    th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
    out = np.zeros(3)
    if lib().ref_kkt_error(problem._h, _p(X), _p(U), _p(V), _p(Lm), float(mueq), _p(th) if th is not None else None,
                           _p(out)):
        raise RuntimeError(lib().ref_last_error().decode())
    return tuple(out)


def bk_compute(A):
    A = _f(A)
    n = A.shape[0]
    ldlt, sub, piv = np.zeros((n, n), order="F"), np.zeros(n), np.zeros(n, dtype=np.int32)
    info = lib().ref_bk_compute(n, _p(A), _p(ldlt), _p(sub), piv.ctypes.data_as(C.POINTER(C.c_int)))
    return info, ldlt, sub, piv


def bk_solve(A, B):
    A, X = _f(A), np.array(B, dtype=np.float64, order="F", copy=True).reshape(A.shape[0], -1, order="F")
    lib().ref_bk_solve(A.shape[0], _p(A), X.shape[1], _p(X))
    return X


def block_tridiag_solve(sub, diag, sup, rhs, down=False):
    dims = np.array([d.shape[0] for d in diag], dtype=np.int32)
    cat = lambda blks: np.concatenate([_f(b).ravel(order="F") for b in blks]) if blks else np.zeros(0)
    s, d, u, r = cat(sub), cat(diag), cat(sup), np.concatenate([np.ravel(x) for x in rhs]).astype(np.float64)
    rc = lib().ref_block_tridiag_solve(len(dims), dims.ctypes.data_as(C.POINTER(C.c_int)), _p(s), _p(d), _p(u), _p(r), int(down))
    out, p = [], 0
    for n in dims:
        out.append(r[p:p + n].copy())
        p += n
    return rc == 0, out
