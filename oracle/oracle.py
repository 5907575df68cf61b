"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It wraps oracle/gar_oracle.c (a plain-C restatement of the
reference's gar algorithm, see gar_oracle.h for the file:line map).

Parity status: PINNED against the reference's own gar sources compiled unchanged over
oracle/ref_shim (oracle/ref.py, tests/test_ref_pin.py, tests/golden/ref/*.npz); see gar_oracle.h.
Also by the reference's test thresholds and by dense_kkt.py (LAPACK).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BLOCKS = ("Q", "S", "R", "q", "r", "A", "B", "f", "C", "D", "d",
           "Gth", "Gx", "Gu", "Gv", "gamma")
_PD = C.POINTER(C.c_double)
_PPD = C.POINTER(_PD)


def build(native: bool = False) -> str:
    """Compile the oracle (gcc) and return the .so path."""
    lib = "libgar_oracle_native.so" if native else "libgar_oracle.so"
    args = ["make", "-s", "-C", _HERE, f"LIB={lib}"]
    if native:
        args.append("ARCH=native")
    subprocess.run(args, check=True)
    return os.path.join(_HERE, "_build", lib)


class _Knot(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nx", "nu", "nc", "nx2", "nth")] + \
               [(n, _PD) for n in _BLOCKS]


class _Problem(C.Structure):
    _fields_ = [("N", C.c_int), ("nc0", C.c_int), ("G0", _PD), ("g0", _PD),
                ("stages", C.POINTER(_Knot))]


class _Bk(C.Structure):
    _fields_ = [("n", C.c_int), ("L", _PD), ("subdiag", _PD),
                ("piv", C.POINTER(C.c_int)), ("W", _PD), ("blocksize", C.c_int),
                ("info", C.c_int), ("pivot_count", C.c_int)]


class _Value(C.Structure):
    _fields_ = [(n, _PD) for n in ("Vxx", "vx", "Vxt", "Vtt", "vt")]


class _Factor(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nx", "nu", "nc", "nx2", "nth")] + \
               [(n, _PD) for n in ("Qhat", "Rhat", "Shat", "qhat", "rhat", "AtV",
                                   "BtV", "Gxhat", "Guhat", "ff", "fb", "fth",
                                   "kktMat")] + \
               [("kktChol", C.POINTER(_Bk)), ("vm", _Value)]


class _Prox(C.Structure):
    _fields_ = [("problem", C.POINTER(_Problem)), ("N", C.c_int),
                ("datas", C.POINTER(_Factor)), ("n0", C.c_int),
                ("kkt0_mat", _PD), ("kkt0_ff", _PD), ("kkt0_fth", _PD),
                ("kkt0_chol", C.POINTER(_Bk)), ("thGrad", _PD), ("thHess", _PD)]


class _Par(C.Structure):
    _fields_ = [("problem", C.POINTER(_Problem)), ("N", C.c_int),
                ("num_threads", C.c_int), ("datas", C.POINTER(_Factor)),
                ("nblk", C.c_int), ("dims", C.POINTER(C.c_int)),
                ("sub", _PPD), ("diag", _PPD), ("super", _PPD),
                ("diagFacs", _PPD), ("upFacs", _PPD),
                ("ldlt", C.POINTER(C.POINTER(_Bk))),
                ("rhs", _PD), ("sol", _PD), ("err", _PD),
                ("rhs_blk", _PPD), ("sol_blk", _PPD), ("err_blk", _PPD),
                ("condensedThreshold", C.c_double),
                ("maxRefinementSteps", C.c_int),
                ("last_refinement_steps", C.c_int),
                ("last_residual", C.c_double)]


_lib = None


def lib(native: bool = False):
    global _lib
    if native:
        return _bind(C.CDLL(build(native=True)))
    if _lib is None:
        _lib = _bind(C.CDLL(build()))
    return _lib


def _bind(L):
    L.ora_problem_new.restype = C.POINTER(_Problem)
    L.ora_problem_new.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]
    L.ora_problem_copy.restype = C.POINTER(_Problem)
    L.ora_problem_copy.argtypes = [C.POINTER(_Problem)]
    L.ora_problem_free.argtypes = [C.POINTER(_Problem)]
    L.ora_problem_add_parameterization.argtypes = [C.POINTER(_Problem), C.c_int]
    L.ora_bk_new.restype = C.POINTER(_Bk)
    L.ora_bk_new.argtypes = [C.c_int]
    L.ora_bk_free.argtypes = [C.POINTER(_Bk)]
    L.ora_bk_compute.argtypes = [C.POINTER(_Bk), _PD, C.c_int]
    L.ora_bk_solve_in_place.argtypes = [C.POINTER(_Bk), _PD, C.c_int, C.c_int, C.c_int]
    L.ora_prox_new.restype = C.POINTER(_Prox)
    L.ora_prox_new.argtypes = [C.POINTER(_Problem)]
    L.ora_prox_free.argtypes = [C.POINTER(_Prox)]
    L.ora_prox_backward.argtypes = [C.POINTER(_Prox), C.c_double]
    L.ora_prox_forward.argtypes = [C.POINTER(_Prox), _PPD, _PPD, _PPD, _PPD, _PD]
    L.ora_prox_cycle_append.argtypes = [C.POINTER(_Prox), C.POINTER(_Knot)]
    L.ora_par_new.restype = C.POINTER(_Par)
    L.ora_par_new.argtypes = [C.POINTER(_Problem), C.c_int]
    L.ora_par_free.argtypes = [C.POINTER(_Par)]
    L.ora_par_backward.argtypes = [C.POINTER(_Par), C.c_double]
    L.ora_par_forward.argtypes = [C.POINTER(_Par), _PPD, _PPD, _PPD, _PPD]
    L.ora_par_collapse_feedback.argtypes = [C.POINTER(_Par)]
    L.ora_lqr_kkt_error.argtypes = [C.POINTER(_Problem), _PPD, _PPD, _PPD, _PPD,
                                    C.c_double, _PD, _PD]
    L.ora_get_work.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                               C.POINTER(C.c_int)]
    L.ora_blocktridiag_solve.argtypes = [C.c_int, C.POINTER(C.c_int), _PPD, _PPD,
                                         _PPD, _PPD, C.POINTER(C.POINTER(_Bk))]
    L.ora_blocktridiag_solve_down.argtypes = L.ora_blocktridiag_solve.argtypes
    L.ora_batch_sweep.argtypes = [C.POINTER(C.POINTER(_Prox)), C.c_int, C.c_double,
                                  C.POINTER(_PPD), C.POINTER(_PPD),
                                  C.POINTER(_PPD), C.POINTER(_PPD), C.c_int]
    L.ora_batch_sweep_local.argtypes = [C.POINTER(C.POINTER(_Problem)), C.c_int, C.c_double, C.c_int,
                                        C.c_int, _PD]
    L.ora_omp_max_threads.restype = C.c_int
    return L


def _view(ptr, shape, order="F"):
    n = int(np.prod(shape)) if len(shape) else 1
    if n == 0:
        return np.zeros(shape, order=order)
    a = np.ctypeslib.as_array(ptr, shape=(n,))
    return a.reshape(shape, order=order)


def _block_shapes(nx, nu, nc, nx2, nth):
    return dict(Q=(nx, nx), S=(nx, nu), R=(nu, nu), q=(nx,), r=(nu,),
                A=(nx2, nx), B=(nx2, nu), f=(nx2,), C=(nc, nx), D=(nc, nu),
                d=(nc,), Gth=(nth, nth), Gx=(nx, nth), Gu=(nu, nth),
                Gv=(nc, nth), gamma=(nth,))


class KnotView:
    """numpy views (column-major) into one oracle knot."""

    def __init__(self, k: _Knot):
        self._k = k
        self.nx, self.nu, self.nc, self.nx2, self.nth = k.nx, k.nu, k.nc, k.nx2, k.nth
        for name, shp in _block_shapes(k.nx, k.nu, k.nc, k.nx2, k.nth).items():
            setattr(self, name, _view(getattr(k, name), shp))


class Problem:
    """Oracle-side LqrProblemTpl. Owns its memory (C side)."""

    def __init__(self, dims: Sequence[Sequence[int]], nc0: int, _ptr=None, native=False):
        self._L = lib(native)
        if _ptr is None:
            d = np.ascontiguousarray(np.asarray(dims, dtype=np.int32).reshape(-1, 5))
            _ptr = self._L.ora_problem_new(d.shape[0] - 1,
                                           d.ctypes.data_as(C.POINTER(C.c_int)), nc0)
        self._p = _ptr

    @classmethod
    def from_knots(cls, knots, G0, g0, native=False) -> "Problem":
        """Copy a product-side problem (objects with nx..nth + block attrs)."""
        dims = [(k.nx, k.nu, k.nc, k.nx2, k.nth) for k in knots]
        g0 = np.asarray(g0, dtype=np.float64)
        p = cls(dims, g0.shape[0], native=native)
        for t, k in enumerate(knots):
            kv = p.knot(t)
            for name in _BLOCKS:
                src = np.asarray(getattr(k, name), dtype=np.float64)
                dst = getattr(kv, name)
                if dst.size:
                    dst[...] = src.reshape(dst.shape)
        if p.nc0:
            p.G0[...] = np.asarray(G0, dtype=np.float64).reshape(p.G0.shape)
            p.g0[...] = g0
        return p

    def copy(self) -> "Problem":
        return Problem(None, 0, _ptr=self._L.ora_problem_copy(self._p))

    def __del__(self):
        try:
            self._L.ora_problem_free(self._p)
        except Exception:
            pass

    @property
    def N(self):
        return self._p.contents.N

    @property
    def nc0(self):
        return self._p.contents.nc0

    @property
    def G0(self):
        return _view(self._p.contents.G0, (self.nc0, self._p.contents.stages[0].nx))

    @property
    def g0(self):
        return _view(self._p.contents.g0, (self.nc0,))

    def knot(self, t) -> KnotView:
        return KnotView(self._p.contents.stages[t])

    def add_parameterization(self, nth: int):
        self._L.ora_problem_add_parameterization(self._p, nth)

    def dims(self):
        return [(k.nx, k.nu, k.nc, k.nx2, k.nth)
                for k in (self._p.contents.stages[t] for t in range(self.N + 1))]

    def initialize_solution(self):
        """lqrInitializeSolution (gar/utils.hpp:114-142)."""
        dm = self.dims()
        N = self.N
        xs = [np.zeros(d[0]) for d in dm]
        us = [np.zeros(d[1]) for d in dm]
        vs = [np.zeros(d[2]) for d in dm]
        lbdas = [np.zeros(self.nc0)] + [np.zeros(dm[t][3]) for t in range(N)]
        if dm[-1][1] == 0:
            us.pop()
        return xs, us, vs, lbdas


def _pp(vecs: List[np.ndarray], n: int):
    arr = (_PD * n)()
    for i in range(n):
        if i < len(vecs) and vecs[i] is not None:
            arr[i] = vecs[i].ctypes.data_as(_PD)
    return arr


class FactorView:
    def __init__(self, f: _Factor):
        nx, nu, nc, nx2, nth = f.nx, f.nu, f.nc, f.nx2, f.nth
        self.nx, self.nu, self.nc, self.nx2, self.nth = nx, nu, nc, nx2, nth
        nr = nu + nc + nx2
        self.ff = _view(f.ff, (nr,))
        self.fb = _view(f.fb, (nr, nx), order="C")
        self.fth = _view(f.fth, (nr, nth), order="C")
        self.Qhat = _view(f.Qhat, (nx, nx))
        self.Rhat = _view(f.Rhat, (nu, nu))
        self.Shat = _view(f.Shat, (nx, nu))
        self.qhat = _view(f.qhat, (nx,))
        self.rhat = _view(f.rhat, (nu,))
        self.kktMat = _view(f.kktMat, (nu + nc, nu + nc))   # [Rhat D^T; D -mu I] as handed to Bunch-Kaufman
        self.Vxx = _view(f.vm.Vxx, (nx, nx))
        self.vx = _view(f.vm.vx, (nx,))
        self.Vxt = _view(f.vm.Vxt, (nx, nth))
        self.Vtt = _view(f.vm.Vtt, (nth, nth))
        self.vt = _view(f.vm.vt, (nth,))


class ProximalRiccatiSolver:
    """Oracle-side gar::ProximalRiccatiSolver."""

    def __init__(self, problem: Problem):
        self.problem = problem
        self._L = problem._L
        self._s = self._L.ora_prox_new(problem._p)

    def __del__(self):
        try:
            self._L.ora_prox_free(self._s)
        except Exception:
            pass

    def backward(self, mueq: float) -> bool:
        return bool(self._L.ora_prox_backward(self._s, float(mueq)))

    def forward(self, xs, us, vs, lbdas, theta: Optional[np.ndarray] = None) -> bool:
        n = self.problem.N + 1
        th = None
        if theta is not None:
            theta = np.ascontiguousarray(theta, dtype=np.float64)
            th = theta.ctypes.data_as(_PD)
        return bool(self._L.ora_prox_forward(self._s, _pp(xs, n), _pp(us, n),
                                             _pp(vs, n), _pp(lbdas, n), th))

    def datas(self, t) -> FactorView:
        return FactorView(self._s.contents.datas[t])

    @property
    def kkt0_ff(self):
        return _view(self._s.contents.kkt0_ff, (self._s.contents.n0,))

    @property
    def kkt0_fth(self):
        nth = self._s.contents.datas[0].nth
        return _view(self._s.contents.kkt0_fth, (self._s.contents.n0, nth), order="C")

    @property
    def thGrad(self):
        return _view(self._s.contents.thGrad, (self._s.contents.datas[0].nth,))

    @property
    def thHess(self):
        nth = self._s.contents.datas[0].nth
        return _view(self._s.contents.thHess, (nth, nth))

    def getFeedforward(self, t):
        return self.datas(t).ff

    def getFeedback(self, t):
        return self.datas(t).fb


class ParallelRiccatiSolver:
    """Oracle-side gar::ParallelRiccatiSolver (MUTATES the problem)."""

    def __init__(self, problem: Problem, num_threads: int):
        self.problem = problem
        self._L = problem._L
        self._s = self._L.ora_par_new(problem._p, int(num_threads))
        if not self._s:
            raise RuntimeError("numThreads should be greater than or equal to 2")

    def __del__(self):
        try:
            self._L.ora_par_free(self._s)
        except Exception:
            pass

    @property
    def maxRefinementSteps(self):
        return self._s.contents.maxRefinementSteps

    @maxRefinementSteps.setter
    def maxRefinementSteps(self, v):
        self._s.contents.maxRefinementSteps = int(v)

    @property
    def last_residual(self):
        return self._s.contents.last_residual

    @property
    def last_refinement_steps(self):
        return self._s.contents.last_refinement_steps

    def backward(self, mueq: float) -> bool:
        return bool(self._L.ora_par_backward(self._s, float(mueq)))

    def forward(self, xs, us, vs, lbdas) -> bool:
        n = self.problem.N + 1
        return bool(self._L.ora_par_forward(self._s, _pp(xs, n), _pp(us, n),
                                            _pp(vs, n), _pp(lbdas, n)))

    def collapseFeedback(self):
        self._L.ora_par_collapse_feedback(self._s)

    def datas(self, t) -> FactorView:
        return FactorView(self._s.contents.datas[t])

    def condensed_solution(self):
        s = self._s.contents
        tot = sum(s.dims[i] for i in range(s.nblk))
        return _view(s.sol, (tot,)).copy()


def lqr_kkt_error(problem: Problem, xs, us, vs, lbdas, mueq=0.0, theta=None):
    """lqrComputeKktError (gar/utils.hxx:88-182) -> (dyn, cstr, dual)."""
    n = problem.N + 1
    out = np.zeros(3)
    th = None
    if theta is not None:
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        th = theta.ctypes.data_as(_PD)
    problem._L.ora_lqr_kkt_error(problem._p, _pp(xs, n), _pp(us, n), _pp(vs, n),
                                 _pp(lbdas, n), float(mueq), th,
                                 out.ctypes.data_as(_PD))
    return tuple(out)


def get_work(horz, tid, nthreads):
    b, e = C.c_int(), C.c_int()
    lib().ora_get_work(horz, tid, nthreads, C.byref(b), C.byref(e))
    return b.value, e.value


class BunchKaufman:
    """Oracle-side aligator::BunchKaufman (core/bunchkaufman.hpp)."""

    def __init__(self, a: np.ndarray):
        self._L = lib()
        a = np.asfortranarray(a, dtype=np.float64)
        self.n = a.shape[0]
        self._bk = self._L.ora_bk_new(self.n)
        self.info = self._L.ora_bk_compute(self._bk, a.ctypes.data_as(_PD), self.n)

    def __del__(self):
        try:
            self._L.ora_bk_free(self._bk)
        except Exception:
            pass

    @property
    def pivots(self):
        return np.array([self._bk.contents.piv[i] for i in range(self.n)])

    @property
    def matrixLDLT(self):
        return _view(self._bk.contents.L, (self.n, self.n)).copy()

    @property
    def subdiag(self):
        return _view(self._bk.contents.subdiag, (self.n,)).copy()

    def solve(self, b: np.ndarray) -> np.ndarray:
        x = np.array(b, dtype=np.float64, order="F", copy=True)
        ncols = 1 if x.ndim == 1 else x.shape[1]
        self._L.ora_bk_solve_in_place(self._bk, x.ctypes.data_as(_PD), 1, self.n, ncols)
        return x


def block_tridiag_solve(sub, diag, sup, rhs, down=False):
    """symmetricBlockTridiagSolve[DownLooking] on lists of numpy blocks.
    Returns (ok, solution blocks); inputs are copied."""
    L = lib()
    nb = len(diag)
    dims = np.array([d.shape[0] for d in diag], dtype=np.int32)
    subc = [np.array(m, dtype=np.float64, order="F", copy=True) for m in sub]
    diagc = [np.array(m, dtype=np.float64, order="F", copy=True) for m in diag]
    supc = [np.array(m, dtype=np.float64, order="F", copy=True) for m in sup]
    rhsc = [np.array(v, dtype=np.float64, copy=True) for v in rhs]
    facs = (C.POINTER(_Bk) * nb)(*[L.ora_bk_new(int(n)) for n in dims])
    fn = L.ora_blocktridiag_solve_down if down else L.ora_blocktridiag_solve
    ok = fn(nb, dims.ctypes.data_as(C.POINTER(C.c_int)), _pp(subc, nb),
            _pp(diagc, nb), _pp(supc, nb), _pp(rhsc, nb), facs)
    for i in range(nb):
        L.ora_bk_free(facs[i])
    return bool(ok), rhsc


class BatchSweep:
    """CPU baseline: OpenMP parallel-for over independent
    problems, one serial backward+forward sweep each."""

    def __init__(self, problems: List[Problem], native=True):
        self.problems = problems
        self._L = problems[0]._L
        self.solvers = [self._L.ora_prox_new(p._p) for p in problems]
        nb = len(problems)
        self._sols = [p.initialize_solution() for p in problems]
        self._arr = (C.POINTER(_Prox) * nb)(*self.solvers)
        self._pp = []
        for k in range(4):
            arrs = [_pp(self._sols[b][k], problems[b].N + 1) for b in range(nb)]
            self._pp.append((arrs, (_PPD * nb)(*[C.cast(a, _PPD) for a in arrs])))

    def max_threads(self):
        return self._L.ora_omp_max_threads()

    def sweep(self, mueq: float, nthreads: int) -> int:
        return self._L.ora_batch_sweep(self._arr, len(self.problems), float(mueq),
                                       self._pp[0][1], self._pp[1][1],
                                       self._pp[2][1], self._pp[3][1], int(nthreads))

    def sweep_local(self, mueq: float, nthreads: int, reps: int):
        """`reps` sweeps of every problem with thread-local (first-touch) copies of the data
        (ora_batch_sweep_local) -> (failed sweeps, seconds of the timed region)."""
        arr = (C.POINTER(_Problem) * len(self.problems))(*[p._p for p in self.problems])
        sec = C.c_double(0.0)
        fails = self._L.ora_batch_sweep_local(arr, len(self.problems), float(mueq), int(nthreads),
                                              int(reps), C.byref(sec))
        return int(fails), float(sec.value)

    def solution(self, b):
        return self._sols[b]

    def __del__(self):
        try:
            for s in self.solvers:
                self._L.ora_prox_free(s)
        except Exception:
            pass
