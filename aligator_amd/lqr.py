"""Host-side data model of the gar LQ problem (Python mirror).

Mirrors the reference's `aligator.gar` scripting surface
(bindings/python/src/gar/expose-gar.cpp:51-121) and the C++ types behind it:

* ``LqrKnot``     <- gar::LqrKnotTpl     (include/aligator/gar/lqr-problem.hpp:34-103)
* ``LqrProblem``  <- gar::LqrProblemTpl  (lqr-problem.hpp:105-195)
* ``lqrInitializeSolution`` <- gar/utils.hpp:114-142
* ``lqrComputeKktError``    <- gar/utils.hxx:88-182
* ``lqrNumRows``            <- gar/utils.hpp:65-77
* ``lqrCreateSparseMatrix`` <- gar/utils.hxx:8-86 (bindings/python/src/gar/expose-utils.cpp:26-37)
* ``BunchKaufman``          <- core/bunchkaufman.hpp (what StageFactor.kktChol / kkt0.chol expose)

Every block is a column-major (Fortran-order) float64 numpy array with the
same name and shape as in the reference, so packing into the device record
(include/gar_hip.h) is a plain concatenation.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

BLOCK_NAMES = ("Q", "S", "R", "q", "r", "A", "B", "f", "C", "D", "d",
               "Gth", "Gx", "Gu", "Gv", "gamma")


def block_shapes(nx: int, nu: int, nc: int, nx2: int, nth: int) -> dict:
    """Shapes of the 16 per-knot blocks (lqr-problem.hxx:28-72)."""
    return dict(Q=(nx, nx), S=(nx, nu), R=(nu, nu), q=(nx,), r=(nu,),
                A=(nx2, nx), B=(nx2, nu), f=(nx2,),
                C=(nc, nx), D=(nc, nu), d=(nc,),
                Gth=(nth, nth), Gx=(nx, nth), Gu=(nu, nth), Gv=(nc, nth),
                gamma=(nth,))


class LqrKnot:
    """One stage of a constrained LQ problem (gar::LqrKnotTpl).

    cost  1/2 [x;u]^T [Q S; S^T R] [x;u] + q^T x + r^T u,
    dynamics x' = A x + B u + f, constraint C x + D u + d = mu * v,
    optional parameterisation (Gth, Gx, Gu, Gv, gamma) of dimension nth.
    """

    __slots__ = ("nx", "nu", "nc", "nx2", "nth") + BLOCK_NAMES

    def __init__(self, nx: int, nu: int, nc: int = 0, nx2: Optional[int] = None,
                 nth: int = 0):
        self.nx, self.nu, self.nc = int(nx), int(nu), int(nc)
        self.nx2 = int(nx if nx2 is None else nx2)
        self.nth = int(nth)
        for name, shp in block_shapes(self.nx, self.nu, self.nc, self.nx2,
                                      self.nth).items():
            setattr(self, name, np.zeros(shp, order="F"))

    @property
    def dims(self) -> Tuple[int, int, int, int, int]:
        return (self.nx, self.nu, self.nc, self.nx2, self.nth)

    def addParameterization(self, nth: int) -> "LqrKnot":
        """lqr-problem.hxx:232-241: resize and ZERO the parametric blocks."""
        self.nth = int(nth)
        shp = block_shapes(*self.dims)
        for name in ("Gth", "Gx", "Gu", "Gv", "gamma"):
            setattr(self, name, np.zeros(shp[name], order="F"))
        return self

    def copy(self) -> "LqrKnot":
        out = LqrKnot(self.nx, self.nu, self.nc, self.nx2, self.nth)
        for name in BLOCK_NAMES:
            setattr(out, name, np.array(getattr(self, name), order="F", copy=True))
        return out

    def assign(self, other: "LqrKnot") -> None:
        """lqr-problem.hxx:74-103."""
        self.nx, self.nu, self.nc, self.nx2, self.nth = other.dims
        for name in BLOCK_NAMES:
            setattr(self, name, np.array(getattr(other, name), order="F", copy=True))

    def isApprox(self, other: "LqrKnot", prec: float = np.finfo(float).eps) -> bool:
        """lqr-problem.hxx:243-264 (Eigen isApprox: ||a-b|| <= prec*min(||a||,||b||))."""
        if self.dims != other.dims:
            return False
        for name in BLOCK_NAMES:
            a, b = getattr(self, name), getattr(other, name)
            if np.linalg.norm(a - b) > prec * min(np.linalg.norm(a), np.linalg.norm(b)):
                return False
        return True

    def __eq__(self, other):
        return isinstance(other, LqrKnot) and self.isApprox(other)

    def __repr__(self):
        s = f"LqrKnot {{\n  nx:  {self.nx}\n  nu:  {self.nu}\n  nc:  {self.nc}"
        if self.nth > 0:
            s += f"\n  nth: {self.nth}"
        return s + "\n}"


class LqrProblem:
    """gar::LqrProblemTpl: ``G0 x0 + g0 = 0`` and a list of N+1 knots."""

    def __init__(self, knots: Sequence[LqrKnot], nc0: int):
        self.stages: List[LqrKnot] = list(knots)
        nx0 = self.stages[0].nx if self.stages else 0
        self.G0 = np.zeros((int(nc0), nx0), order="F")
        self.g0 = np.zeros(int(nc0))

    @property
    def horizon(self) -> int:
        return len(self.stages) - 1

    @property
    def nc0(self) -> int:
        return int(self.g0.shape[0])

    @property
    def isInitialized(self) -> bool:
        return len(self.stages) > 0

    @property
    def isParameterized(self) -> bool:
        return self.isInitialized and self.stages[0].nth > 0

    @property
    def ntheta(self) -> int:
        return self.stages[0].nth

    def addParameterization(self, nth: int) -> None:
        for k in self.stages:
            k.addParameterization(nth)

    def copy(self) -> "LqrProblem":
        out = LqrProblem([k.copy() for k in self.stages], self.nc0)
        out.G0[...] = self.G0
        out.g0[...] = self.g0
        return out

    def evaluate(self, xs, us, theta=None) -> float:
        """lqr-problem.hxx:285-319."""
        N = self.horizon
        if len(xs) != N + 1 or len(us) < N or not self.stages:
            return 0.0
        ret = 0.0
        for i, k in enumerate(self.stages):
            ret += 0.5 * xs[i] @ (k.Q @ xs[i]) + xs[i] @ k.q
            if i == N:
                break
            ret += 0.5 * us[i] @ (k.R @ us[i]) + us[i] @ k.r
            ret += xs[i] @ (k.S @ us[i])
        if self.isParameterized and theta is not None:
            for i, k in enumerate(self.stages):
                ret += 0.5 * theta @ (k.Gth @ theta) + theta @ (k.Gx.T @ xs[i])
                ret += theta @ k.gamma
                if i == N:
                    break
                ret += theta @ (k.Gu.T @ us[i])
        return float(ret)


def lqrInitializeSolution(problem: LqrProblem):
    """gar/utils.hpp:114-142 -> (xs, us, vs, lbdas), zero-filled."""
    N = problem.horizon
    xs = [np.zeros(k.nx) for k in problem.stages]
    us = [np.zeros(k.nu) for k in problem.stages]
    vs = [np.zeros(k.nc) for k in problem.stages]
    lbdas = [np.zeros(problem.nc0)] + [np.zeros(problem.stages[i].nx2)
                                       for i in range(N)]
    if problem.stages[-1].nu == 0:
        us.pop()
    return xs, us, vs, lbdas


def lqrNumRows(problem: LqrProblem) -> int:
    """gar/utils.hpp:65-77."""
    N = problem.horizon
    n = problem.nc0
    for t, k in enumerate(problem.stages):
        n += k.nx + k.nu + k.nc
        if t != N:
            n += k.nx
    return n


def lqrComputeKktError(problem: LqrProblem, xs, us, vs, lbdas, mueq: float = 0.0,
                       theta=None, verbose: bool = False):
    """gar/utils.hxx:88-182 -> (dynErr, cstErr, dualErr), infinity norms."""
    def inf(v):
        return float(np.max(np.abs(v))) if v.size else 0.0

    N = problem.horizon
    dyn_err = inf(problem.g0 + problem.G0 @ xs[0])
    cst_err = 0.0
    dual_err = 0.0
    for t, k in enumerate(problem.stages):
        cst = k.C @ xs[t] + k.d - mueq * vs[t]
        gx = k.q + k.Q @ xs[t] + k.C.T @ vs[t]
        gu = k.r + k.S.T @ xs[t] + k.D.T @ vs[t]
        if k.nu > 0:
            cst = cst + k.D @ us[t]
            gx = gx + k.S @ us[t]
            gu = gu + k.R @ us[t]
        if t == 0:
            gx = gx + problem.G0.T @ lbdas[0]
        else:
            gx = gx - lbdas[t]
        if t < N:
            dyn = k.A @ xs[t] + k.B @ us[t] + k.f - xs[t + 1]
            gx = gx + k.A.T @ lbdas[t + 1]
            gu = gu + k.B.T @ lbdas[t + 1]
            dyn_err = max(dyn_err, inf(dyn))
        if theta is not None:
            gx = gx + k.Gx @ theta
            gu = gu + k.Gu @ theta
        if verbose:
            print(f"[{t:>2d}] |gx| = {inf(gx):.3e} | |gu| = {inf(gu):.3e} | "
                  f"|cst| = {inf(cst):.3e}")
        dual_err = max(dual_err, inf(gx), inf(gu))
        cst_err = max(cst_err, inf(cst))
    return dyn_err, cst_err, dual_err


def lqrCreateSparseMatrix(problem: LqrProblem, mueq: float, update: bool = False):
    """-> (mat, rhs): the global KKT matrix of the LQ problem as a scipy.sparse CSC matrix and its right-hand side,
    laid out exactly as the reference's gar::lqrCreateSparseMatrix (gar/utils.hxx:8-86; Python:
    expose-utils.cpp:26-37): unknowns [lbda0; x0 u0 v0; lbda1; x1 u1 v1; ...], row block of a knot [q; r; d], then f;
    the coupling of x_{t+1} with lbda_{t+1} is written as +I like the reference does (:75-80 -- the residual
    convention of lqrComputeKktError and of the reference's dense test builder is -I).
    `update` (re-use of an existing sparsity pattern) changes nothing here: the matrix is rebuilt."""
    import scipy.sparse as sp
    n = lqrNumRows(problem)
    N = problem.horizon
    rows, cols, vals = [], [], []

    def put(i0, j0, blk):
        blk = np.asarray(blk)
        if blk.size:
            ii, jj = np.nonzero(np.ones(blk.shape, dtype=bool))
            rows.append(ii + i0)
            cols.append(jj + j0)
            vals.append(blk[ii, jj])

    rhs = np.zeros(n)
    nc0 = problem.nc0
    rhs[:nc0] = problem.g0
    put(0, nc0, problem.G0)
    put(nc0, 0, problem.G0.T)
    idx = nc0
    for t, k in enumerate(problem.stages):
        nk = k.nx + k.nu + k.nc
        rhs[idx:idx + k.nx] = k.q
        rhs[idx + k.nx:idx + k.nx + k.nu] = k.r
        rhs[idx + k.nx + k.nu:idx + nk] = k.d
        i0, i1 = idx + k.nx, idx + k.nx + k.nu
        i2 = i1 + k.nc
        put(idx, idx, k.Q)
        put(i0, idx, k.S.T)
        put(idx, i0, k.S)
        put(i0, i0, k.R)
        put(i1, idx, k.C)
        put(idx, i1, k.C.T)
        put(i1, i0, k.D)
        put(i0, i1, k.D.T)
        put(i1, i1, -mueq * np.eye(k.nc))
        if t != N:
            rhs[idx + nk:idx + nk + k.nx2] = k.f
            put(i2, idx, k.A)
            put(idx, i2, k.A.T)
            put(i2, i0, k.B)
            put(i0, i2, k.B.T)
            i3 = i2 + k.nx2
            put(i2, i3, np.eye(k.nx2))
            put(i3, i2, np.eye(k.nx2))
            idx += nk + k.nx2
    mat = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)) \
        if vals else sp.csc_matrix((n, n))
    return mat, rhs


class BunchKaufman:
    """L D L^T of a symmetric (possibly indefinite) matrix with Bunch-Kaufman partial pivoting, LOWER triangle --
    the factorisation behind StageFactor.kktChol and kkt0.chol (core/bunchkaufman.hpp; the textbook algorithm of
    LAPACK's dsytf2 with uplo = 'L', alpha = (1 + sqrt 17) / 8).  A host-side mirror of what the reference's Python
    module exposes for inspection; the device sweeps factorise in registers / LDS and do not keep this object.

    `pivots` follows the reference: pivots[k] = p (0-based) for a 1x1 pivot with rows k and p interchanged,
    pivots[k] = pivots[k+1] = -1 - p for a 2x2 pivot with rows k+1 and p interchanged (bunchkaufman.hpp:155-161).
    `matrixLDLT` holds the unit-lower factor below the block diagonal and D's blocks on it (D itself; the reference
    stores the blocks' inverses).  Only the lower triangle of the input is read."""

    def __init__(self, A=None):
        self.pivots = np.zeros(0, dtype=np.int64)
        self.matrixLDLT = np.zeros((0, 0))
        self.info = True
        if A is not None:
            self.compute(A)

    def compute(self, A):
        a = np.tril(np.array(A, dtype=np.float64, order="F", copy=True))
        n = a.shape[0]
        a = a + np.tril(a, -1).T          # work on the full symmetric matrix: the updates stay one-liners
        piv = np.zeros(n, dtype=np.int64)
        alpha = (1.0 + np.sqrt(17.0)) / 8.0
        self.info = True
        k = 0
        while k < n:
            step = 1
            akk = abs(a[k, k])
            if k + 1 < n:
                imax = k + 1 + int(np.argmax(np.abs(a[k + 1:, k])))
                colmax = abs(a[imax, k])
            else:
                imax, colmax = k, 0.0
            if max(akk, colmax) == 0.0:
                self.info = False
                break
            if akk >= alpha * colmax:
                kp = k
            else:
                rowmax = max(np.abs(a[imax, k:imax]).max(initial=0.0), np.abs(a[imax + 1:, imax]).max(initial=0.0))
                if akk >= (alpha * colmax) * (colmax / rowmax):
                    kp = k
                elif abs(a[imax, imax]) >= alpha * rowmax:
                    kp = imax
                else:
                    kp, step = imax, 2
            kk = k + step - 1
            if kp != kk:                   # symmetric interchange of rows / columns kk and kp of the ACTIVE block
                sub = a[k:, k:]            # (the columns of L already computed are not touched: solve() interleaves)
                sub[[kk - k, kp - k], :] = sub[[kp - k, kk - k], :]
                sub[:, [kk - k, kp - k]] = sub[:, [kp - k, kk - k]]
            if step == 1:
                d = a[k, k]
                l = a[k + 1:, k] / d
                a[k + 1:, k + 1:] -= np.outer(l, a[k + 1:, k])
                a[k + 1:, k] = l
                a[k, k + 1:] = l
                piv[k] = kp
            else:
                D = a[k:k + 2, k:k + 2].copy()
                W = np.linalg.solve(D, a[k:k + 2, k + 2:]).T     # L's two columns
                a[k + 2:, k + 2:] -= W @ a[k:k + 2, k + 2:]
                a[k + 2:, k:k + 2] = W
                a[k:k + 2, k + 2:] = W.T
                piv[k] = piv[k + 1] = -1 - kp
            k += step
        self.pivots = piv
        self.matrixLDLT = np.tril(a)
        return self

    def _blocks(self):
        n, k = len(self.pivots), 0
        while k < n:
            step = 2 if self.pivots[k] < 0 else 1
            yield k, step
            k += step

    def solve(self, B):
        """x with A x = B (the caller's A, before any interchange)."""
        x = np.array(B, dtype=np.float64, copy=True)
        one = x.ndim == 1
        if one:
            x = x[:, None]
        a, n = self.matrixLDLT, len(self.pivots)
        for k, step in self._blocks():            # P, then L^{-1}, block column by block column
            kk = k + step - 1
            kp = int(self.pivots[k]) if step == 1 else -1 - int(self.pivots[k])
            if kp != kk:
                x[[kk, kp]] = x[[kp, kk]]
            x[k + step:] -= a[k + step:, k:k + step] @ x[k:k + step]
        for k, step in self._blocks():            # D^{-1}
            if step == 1:
                x[k] /= a[k, k]
            else:
                D = np.array([[a[k, k], a[k + 1, k]], [a[k + 1, k], a[k + 1, k + 1]]])
                x[k:k + 2] = np.linalg.solve(D, x[k:k + 2])
        for k, step in reversed(list(self._blocks())):   # L^{-T}, then P^T
            x[k:k + step] -= a[k + step:, k:k + step].T @ x[k + step:]
            kk = k + step - 1
            kp = int(self.pivots[k]) if step == 1 else -1 - int(self.pivots[k])
            if kp != kk:
                x[[kk, kp]] = x[[kp, kk]]
        return x[:, 0] if one else x
