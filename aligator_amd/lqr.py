"""Host-side data model of the gar LQ problem (Python mirror).

Mirrors the reference's `aligator.gar` scripting surface
(bindings/python/src/gar/expose-gar.cpp:51-121) and the C++ types behind it:

* ``LqrKnot``     <- gar::LqrKnotTpl     (include/aligator/gar/lqr-problem.hpp:34-103)
* ``LqrProblem``  <- gar::LqrProblemTpl  (lqr-problem.hpp:105-195)
* ``lqrInitializeSolution`` <- gar/utils.hpp:114-142
* ``lqrComputeKktError``    <- gar/utils.hxx:88-182
* ``lqrNumRows``            <- gar/utils.hpp:65-77

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
