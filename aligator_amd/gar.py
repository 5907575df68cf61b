"""Host-side mirror of the reference's ``aligator.gar`` solver surface on top of
the C ABI (include/gar_hip.h).

Same names, argument meaning and error behaviour as the reference:

* ``RiccatiSolverBase``       <- gar::RiccatiSolverBase   (gar/riccati-base.hpp:13-37)
* ``ProximalRiccatiSolver``   <- gar::ProximalRiccatiSolver (gar/proximal-riccati.hpp:17-47;
  Python binding bindings/python/src/gar/expose-prox-riccati.cpp:14-54)
* ``ParallelRiccatiSolver``   <- gar::ParallelRiccatiSolver (gar/parallel-solver.hpp:26-110;
  bindings/python/src/gar/expose-parallel.cpp:16-24)
* ``BatchedRiccatiSolver``    -- new: `batch` independent problems of identical
  dimensions swept by one launch (the throughput axis).

All arithmetic happens in the HIP kernels; this file only packs / unpacks.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .lqr import (BLOCK_NAMES, BunchKaufman, LqrKnot, LqrProblem, block_shapes, lqrComputeKktError,
                  lqrCreateSparseMatrix, lqrInitializeSolution, lqrNumRows)

__all__ = ["LqrKnot", "LqrProblem", "RiccatiSolverBase", "ProximalRiccatiSolver",
           "ParallelRiccatiSolver", "RiccatiSolverDense", "BatchedRiccatiSolver", "lqrInitializeSolution",
           "lqrComputeKktError", "lqrCreateSparseMatrix", "lqrNumRows", "BunchKaufman", "get_work"]

_PD = C.POINTER(C.c_double)
GAR_HIP_ERR_FACTOR = -4


def _ptr(a: Optional[np.ndarray]):
    if a is None or a.size == 0:
        return None
    return a.ctypes.data_as(_PD)


def _f64(a, order="F"):
    return np.require(a, dtype=np.float64, requirements=[order])


def set_option(name: str, value: Optional[str], lib_path=None):
    """gar_hip_set_option: a behaviour switch of the library (`BACKWARD`, `PAD`, `SPD_ACCEPT`, ... -- include/gar_hip.h
    lists them) set through the API instead of the `GAR_HIP_<NAME>` environment variable; process-wide, takes
    precedence over the environment, `None` hands the name back to the environment."""
    L = _lib.load(lib_path)
    rc = L.gar_hip_set_option(name.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise ValueError(L.gar_hip_last_error().decode())


def get_option(name: str, lib_path=None) -> Optional[str]:
    v = _lib.load(lib_path).gar_hip_get_option(name.encode())
    return None if v is None else v.decode()


def suggest_num_legs(horizon: int, nx: int, nu: int, lib_path=None) -> int:
    """The library's measured table of leg counts for ONE problem in leg mode on one device
    (gar_hip_suggest_num_legs, include/gar_hip.h; the reference's caller passes num_threads itself)."""
    return int(_lib.load(lib_path).gar_hip_suggest_num_legs(int(horizon), int(nx), int(nu)))


def get_work(horz: int, tid: int, num_threads: int):
    """gar/parallel-solver.hxx:23-28."""
    return (tid * (horz + 1) // num_threads, (tid + 1) * (horz + 1) // num_threads)


class _ValueView:
    """StageFactor::CostToGo (gar/riccati-kernel.hpp:33-39)."""
    __slots__ = ("Vxx", "vx", "Vxt", "Vtt", "vt")


class _FactorView:
    """The surviving outputs of gar::StageFactor (riccati-kernel.hpp:88-101).  `kktMat` is formed on the device on
    first access (gar_hip_get_kkt: a kernel launch and a synchronisation nobody pays who does not read it)."""
    __slots__ = ("nx", "nu", "nc", "nx2", "nth", "ff", "fb", "fth", "vm", "_kkt", "_kkt_fn")

    @property
    def kktMat(self):
        if self._kkt is None:
            self._kkt = self._kkt_fn()
        return self._kkt

    @property
    def kktChol(self):
        """StageFactor::kktChol (expose-prox-riccati.cpp:31): the Bunch-Kaufman factorisation of kktMat, formed on the
        host on request (the device sweeps factorise in registers / LDS and keep only the gains)."""
        from .lqr import BunchKaufman
        return BunchKaufman(self.kktMat)


class _Kkt0View:
    """ProximalRiccatiSolver::kkt0 (proximal-riccati.hpp:40-43; expose-prox-riccati.cpp:48-52): ff, fth, and -- on
    request, from stage 0's value function and the problem's G0 -- mat = [Vxx0 G0^T; G0 0] and its factorisation."""
    __slots__ = ("ff", "fth", "_mat_fn")

    @property
    def mat(self):
        return self._mat_fn()

    @property
    def chol(self):
        from .lqr import BunchKaufman
        return BunchKaufman(self.mat)


class BatchedRiccatiSolver:
    """`batch` LQ problems with the same per-stage dimensions on one GPU.

    dims: (horizon+1) x (nx, nu, nc, nx2, nth) -- the caller's dimensions.  num_legs = 1 is the serial
    ProximalRiccatiSolver algorithm, >= 2 the ParallelRiccatiSolver one.  Kernel selection and padding onto a
    specialised shape happen inside the C ABI (include/gar_hip.h, gar_hip_solver_create): everything here speaks
    the caller's dimensions; `device_dims` / `device_layout()` report the device records (device-resident producers).
    rank_of: (rank, world) -- horizon sharding, one process per GPU: this solver owns legs [rank J / W, (rank+1) J / W).
    devices: a list of device ids -- horizon sharding inside THIS process (gar_hip_multi_create): device r owns legs
    [r J / W, (r+1) J / W), the boundary exchange happens inside backward(); the same device may be named more than once.
    """

    def __init__(self, dims, nc0: int, batch: int = 1, num_legs: int = 1, device: int = 0,
                 rank_of=None, lib_path: Optional[str] = None, dense: bool = False, devices=None):
        self._L = _lib.load(lib_path)
        self.dense = bool(dense)   # RiccatiSolverDense's algorithm (csrc/gar_dense.hpp): serial in time
        if self.dense:
            assert num_legs == 1 and rank_of is None, "the stage-dense solver is serial in time"
        self.dims = np.ascontiguousarray(np.asarray(dims, dtype=np.int32).reshape(-1, 5))
        self.nc0 = int(nc0)
        self.horizon = self.dims.shape[0] - 1
        self.batch, self.num_legs = int(batch), int(num_legs)
        rank, world = rank_of if rank_of is not None else (0, 1)
        self.devices = [int(d) for d in devices] if devices is not None else None
        if self.dense:
            self._h = self._L.gar_hip_solver_create_dense(
                int(device), self.horizon, self.dims.ctypes.data_as(C.POINTER(C.c_int32)),
                self.nc0, self.batch)
        elif self.devices is not None:
            assert rank_of is None, "devices= shards inside this process; rank_of= is one process per GPU"
            ids = (C.c_int * len(self.devices))(*self.devices)
            self._h = self._L.gar_hip_multi_create(
                len(self.devices), ids, self.horizon, self.dims.ctypes.data_as(C.POINTER(C.c_int32)),
                self.nc0, self.batch, self.num_legs)
        else:
            self._h = self._L.gar_hip_solver_create_ranked(
                int(device), self.horizon, self.dims.ctypes.data_as(C.POINTER(C.c_int32)),
                self.nc0, self.batch, self.num_legs, int(rank), int(world))
        if not self._h:
            raise RuntimeError(self._err())
        self._refresh_layout()

    # ---- plumbing ------------------------------------------------------------
    def _err(self) -> str:
        return self._L.gar_hip_last_error().decode()

    def _check(self, rc: int):
        if rc == GAR_HIP_ERR_FACTOR:
            # the reference throws ALIGATOR_RUNTIME_ERROR (riccati-kernel.hxx:239-241)
            raise RuntimeError(self._err())
        if rc != 0:
            raise RuntimeError(f"gar_hip error {rc}: {self._err()}")

    def _refresh_layout(self):
        L, h = self._L, self._h
        self.problem_doubles = L.gar_hip_problem_doubles(h)
        self.factors_doubles = L.gar_hip_factors_doubles(h)
        self.solution_doubles = L.gar_hip_solution_doubles(h)
        offs = np.zeros((self.horizon + 1, 6), dtype=np.int64)
        for t in range(self.horizon + 1):
            self._check(L.gar_hip_stage_offsets(h, t, offs[t].ctypes.data_as(C.POINTER(C.c_int64))))
        self.stage_offsets = offs
        io = np.zeros(2, dtype=np.int64)
        self._check(L.gar_hip_init_offsets(h, io.ctypes.data_as(C.POINTER(C.c_int64))))
        self.G0_off, self.g0_off = int(io[0]), int(io[1])
        self.kernel_name = L.gar_hip_kernel_name(h).decode()
        # the device side (identical unless the library padded the shape onto a specialised family)
        dl = np.zeros((self.horizon + 1, 11), dtype=np.int64)
        for t in range(self.horizon + 1):
            self._check(L.gar_hip_device_stage_layout(h, t, dl[t].ctypes.data_as(C.POINTER(C.c_int64))))
        ds = np.zeros(8, dtype=np.int64)
        self._check(L.gar_hip_device_sizes(h, ds.ctypes.data_as(C.POINTER(C.c_int64))))
        self.device_dims = dl[:, :5].astype(np.int32)
        self.device_stage_offsets = dl[:, 5:].copy()   # knot, factor, x, u, v, lbda
        self.device_problem_doubles, self.device_factors_doubles, self.device_solution_doubles = (int(v) for v in ds[:3])
        self.device_nc0, self.device_G0_off, self.device_g0_off, self.padded = int(ds[3]), int(ds[4]), int(ds[5]), bool(int(ds[6]) & 1)
        # the device knots t < N keep Q and R as packed lower triangles (csrc/gar_layout.h: the headline sweep's records)
        self.record_format = int(L.gar_hip_device_record_format(h))   # GAR_HIP_FMT_* flags (include/gar_hip.h)
        # dimensions of the packed records' stages: the caller's, but for a terminal knot given with nx2 = 0 (kept as
        # nx2 = nx by the library: gar_hip_packed_stage_dims)
        pd = np.zeros((self.horizon + 1, 5), dtype=np.int32)
        for t in range(self.horizon + 1):
            self._check(L.gar_hip_packed_stage_dims(h, t, pd[t].ctypes.data_as(C.POINTER(C.c_int32))))
        self.packed_dims = pd
        self.qr_packed = bool(self.record_format & 1)
        self._factors_cache = {}
        self._mueq = None   # of the last backward (datas[t].kktMat is formed on request)

    def close(self):
        if getattr(self, "_h", None):
            self._L.gar_hip_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def set_stream(self, hip_stream: int):
        self._check(self._L.gar_hip_set_stream(self._h, C.c_void_p(hip_stream)))

    def sync(self):
        self._check(self._L.gar_hip_sync(self._h))

    def set_pipeline(self, halves: int = 2):
        """gar_hip_set_pipeline: the forward sweep of one half of the batch beside the backward sweep of the other
        half (same results, bit for bit; serial one-wave-per-problem family only -- raises otherwise).  halves = -1:
        the library's own choice from batch size and CU count (what a new solver starts with; never raises)."""
        self._check(self._L.gar_hip_set_pipeline(self._h, int(halves)))

    @property
    def pipeline(self) -> int:
        return int(self._L.gar_hip_pipeline(self._h))

    # ---- packing ---------------------------------------------------------------
    def effective_nth(self, t: int) -> int:
        """nth the kernels use for stage t (leg mode re-parameterises, like
        ParallelRiccatiSolver::initialize, parallel-solver.hxx:52-60)."""
        if self.num_legs == 1:
            return int(self.dims[t, 4])
        for i in range(self.num_legs):
            b, e = get_work(self.horizon, i, self.num_legs)
            if b <= t < e:
                return 0 if i == self.num_legs - 1 else int(self.dims[e - 1, 3])
        raise IndexError(t)

    def pack(self, problem: LqrProblem) -> np.ndarray:
        """One problem as the contiguous device record (csrc/gar_layout.h)."""
        buf = np.zeros(self.problem_doubles)
        nx0 = int(self.dims[0, 0])
        G0, g0 = problem.G0, problem.g0
        buf[self.G0_off:self.G0_off + self.nc0 * nx0] = _f64(G0).ravel(order="F")
        buf[self.g0_off:self.g0_off + self.nc0] = g0
        for t, k in enumerate(problem.stages):
            nx, nu, nc, nx2, nth = (int(v) for v in self.dims[t])
            if (k.nx, k.nu, k.nc, k.nx2) != tuple(int(v) for v in self.dims[t, :4]):
                raise ValueError(f"knot {t}: dimensions differ from the solver's")
            stored = nth if self.num_legs == 1 else 0
            p = int(self.stage_offsets[t, 0])
            nx2p = int(self.packed_dims[t, 3])    # (the record's nx2: the knot's own, or nx for a terminal nx2 = 0)
            for name, shp in block_shapes(nx, nu, nc, nx2p, stored).items():
                n = int(np.prod(shp))
                if name in ("Gth", "Gx", "Gu", "Gv", "gamma") and stored == 0:
                    continue
                blk = _f64(getattr(k, name))
                if blk.shape != tuple(shp):       # the terminal A (0 x nx) / f (0): zeros in the record
                    full = np.zeros(shp, order="F")
                    full[tuple(slice(0, d) for d in blk.shape)] = blk
                    blk = full
                buf[p:p + n] = blk.ravel(order="F")
                p += n
        return buf

    def pack_device(self, problem: LqrProblem) -> np.ndarray:
        """One problem in the DEVICE's record format, for producers that write knots in place
        (gar_hip_device_problems / gar_hip_upload_packed_device): pack(), then -- when the solver keeps the lower
        triangles of Q and R packed (`qr_packed`, csrc/gar_layout.h) -- those blocks of every knot t < N repacked."""
        if self.padded:
            raise NotImplementedError("device records of a padded solver: ask device_dims / device_stage_offsets")
        buf = self.pack(problem)
        if not self.qr_packed:
            return buf
        for t in range(self.horizon):
            nx, nu = int(self.dims[t, 0]), int(self.dims[t, 1])
            p = int(self.stage_offsets[t, 0])
            for off, n in ((p, nx), (p + nx * nx + nx * nu, nu)):   # Q, then (behind S) R
                full = buf[off:off + n * n].reshape(n, n, order="F").copy()
                buf[off:off + n * n] = 0.0
                k = off
                for j in range(n):
                    buf[k:k + n - j] = full[j:, j]
                    k += n - j
        return buf

    def unpack(self, buf: np.ndarray) -> LqrProblem:
        """Inverse of pack(): a host LqrProblem from one packed device record."""
        knots = []
        for t in range(self.horizon + 1):
            nx, nu, nc, nx2, nth = (int(v) for v in self.dims[t])
            stored = nth if self.num_legs == 1 else 0
            k = LqrKnot(nx, nu, nc, nx2, stored)
            p = int(self.stage_offsets[t, 0])
            for name, shp in block_shapes(nx, nu, nc, int(self.packed_dims[t, 3]), stored).items():
                n = int(np.prod(shp))
                own = getattr(k, name)
                own[...] = buf[p:p + n].reshape(shp, order="F")[tuple(slice(0, d) for d in own.shape)]
                p += n
            knots.append(k)
        prob = LqrProblem(knots, self.nc0)
        nx0 = int(self.dims[0, 0])
        prob.G0[...] = buf[self.G0_off:self.G0_off + self.nc0 * nx0].reshape((self.nc0, nx0), order="F")
        prob.g0[...] = buf[self.g0_off:self.g0_off + self.nc0]
        return prob

    def upload(self, problems: Sequence[LqrProblem], b0: int = 0):
        packed = np.concatenate([self.pack(p) for p in problems])
        self.upload_packed(packed, b0, len(problems))

    def upload_packed(self, packed: np.ndarray, b0: int = 0, nb: Optional[int] = None):
        packed = np.ascontiguousarray(packed, dtype=np.float64)
        nb = packed.size // self.problem_doubles if nb is None else nb
        self._check(self._L.gar_hip_upload_packed(self._h, b0, nb, _ptr(packed)))

    def upload_packed_device(self, dev_ptr: int, b0: int = 0, nb: Optional[int] = None,
                             record_format: Optional[int] = None):
        """`dev_ptr`: device address of nb packed DEVICE records (`device_problem_doubles` each, laid out by
        `device_dims` / `device_stage_offsets`; e.g. a torch tensor's data_ptr()).  `record_format`: the GAR_HIP_FMT_*
        flags the producer wrote the records in (Q / R full or packed lower triangles): a mismatch with the solver's
        `record_format` is refused instead of swept."""
        nb = self.batch - b0 if nb is None else nb
        if record_format is None:
            self._check(self._L.gar_hip_upload_packed_device(self._h, b0, nb, C.c_void_p(dev_ptr)))
        else:
            self._check(self._L.gar_hip_upload_packed_device_fmt(self._h, b0, nb, C.c_void_p(dev_ptr), int(record_format)))

    def device_pointers(self):
        """(problems, factors, solutions) device addresses for device-resident producers."""
        L, h = self._L, self._h
        return (L.gar_hip_device_problems(h), L.gar_hip_device_factors(h),
                L.gar_hip_device_solutions(h))

    def download_packed(self, b0: int = 0, nb: Optional[int] = None) -> np.ndarray:
        """nb packed problems back from HBM (diagnostics / tests)."""
        nb = self.batch - b0 if nb is None else nb
        out = np.zeros(nb * self.problem_doubles)
        self._check(self._L.gar_hip_download_packed(self._h, b0, nb, _ptr(out)))
        return out

    # ---- device-resident LQ assembly (updateLQSubproblem, solver-proxddp.hxx:734-805) ----
    DERIV_BLOCKS = ("Lxx", "Lxu", "Luu", "Lx", "Lu", "Jx", "Ju", "slack", "Cx", "Cu", "Lv",
                    "Hxx", "Hxu", "Huu", "lx_corr", "lu_corr")

    @staticmethod
    def deriv_shapes(nx, nu, nc, nx2):
        return dict(Lxx=(nx, nx), Lxu=(nx, nu), Luu=(nu, nu), Lx=(nx,), Lu=(nu,), Jx=(nx2, nx),
                    Ju=(nx2, nu), slack=(nx2,), Cx=(nc, nx), Cu=(nc, nu), Lv=(nc,), Hxx=(nx, nx),
                    Hxu=(nx, nu), Huu=(nu, nu), lx_corr=(nx,), lu_corr=(nu,))

    @property
    def deriv_doubles(self) -> int:
        return int(self._L.gar_hip_deriv_doubles(self._h))

    def pack_derivs(self, derivs, init) -> np.ndarray:
        """One problem's derivative buffer (csrc/gar_layout.h, gar_deriv_layout): header
        G0 | g0 | init Hxx, then one record per stage in DERIV_BLOCKS order -- in the CALLER's dimensions, also when
        the library padded the shape (the device kernel scatters into the padded knots)."""
        buf = np.zeros(self.deriv_doubles)
        off = np.zeros(4, dtype=np.int64)
        for t, d in enumerate(derivs):
            self._check(self._L.gar_hip_deriv_offsets(self._h, t, off.ctypes.data_as(C.POINTER(C.c_int64))))
            p = int(off[0])
            nx, nu, nc, nx2, _ = (int(v) for v in self.dims[t])
            for name, shp in self.deriv_shapes(nx, nu, nc, nx2).items():
                n = int(np.prod(shp))
                buf[p:p + n] = _f64(np.asarray(d[name]).reshape(shp)).ravel(order="F")
                p += n
        nx0 = int(self.dims[0, 0])
        buf[int(off[1]):int(off[1]) + self.nc0 * nx0] = _f64(init["Jx"]).ravel(order="F")
        buf[int(off[2]):int(off[2]) + self.nc0] = init["value"]
        buf[int(off[3]):int(off[3]) + nx0 * nx0] = _f64(init["Hxx"]).ravel(order="F")
        return buf

    def update_lq_subproblem_device(self, deriv_device_ptr: int, preg: float, hess_exact: bool):
        """The knots of every problem are rebuilt ON THE DEVICE from a device-resident
        derivative buffer (batch x deriv_doubles); asynchronous on the solver's stream."""
        self._factors_cache = {}
        self._check(self._L.gar_hip_update_lq_subproblem_device(
            self._h, C.c_void_p(deriv_device_ptr), float(preg), int(bool(hess_exact))))

    def upload_knot(self, b: int, t: int, k: LqrKnot):
        """gar_hip_upload_stage: the 16 separately allocated blocks of LqrKnotTpl."""
        a = {n: _f64(getattr(k, n)) for n in BLOCK_NAMES}
        self._check(self._L.gar_hip_upload_stage(self._h, b, t, *[_ptr(a[n]) for n in BLOCK_NAMES]))

    def set_init(self, b: int, G0, g0):
        G0, g0 = _f64(np.asarray(G0)), _f64(np.asarray(g0))
        self._check(self._L.gar_hip_set_init(self._h, b, _ptr(G0), _ptr(g0)))

    # ---- the sweep ---------------------------------------------------------------
    def backward(self, mueq: float) -> bool:
        self._factors_cache = {}
        self._mueq = float(mueq)
        self._check(self._L.gar_hip_backward(self._h, float(mueq)))
        return True

    def forward(self, theta: Optional[np.ndarray] = None) -> bool:
        th = None
        if theta is not None:
            theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(-1)
            nth0 = self.effective_nth(0) if self.num_legs == 1 else 0
            if nth0 > 0 and theta.size != nth0 * self.batch:  # the C ABI reads nth0 * batch doubles
                raise ValueError(f"theta has {theta.size} entries, expected nth * batch = {nth0 * self.batch}")
            th = _ptr(theta)
        self._check(self._L.gar_hip_forward(self._h, th))
        return True

    def backward_blocks(self, problem: LqrProblem, mueq: float) -> bool:
        """gar_hip_backward_blocks: the caller's whole problem (batch = 1) and backward(mueq) in one call."""
        self._factors_cache = {}
        self._mueq = float(mueq)
        keep, ptrs = [], (C.c_void_p * (16 * (self.horizon + 1)))()
        for t, k in enumerate(problem.stages):
            stored = self.num_legs == 1 and k.nth > 0
            for i, name in enumerate(BLOCK_NAMES):
                a = getattr(k, name)
                if a.size == 0 or (i >= 11 and not stored):
                    ptrs[16 * t + i] = None
                    continue
                a = _f64(a)
                keep.append(a)
                ptrs[16 * t + i] = a.ctypes.data
        G0, g0 = _f64(problem.G0), _f64(problem.g0)
        self._check(self._L.gar_hip_backward_blocks(self._h, ptrs, _ptr(G0), _ptr(g0), float(mueq)))
        return True

    def backward_async(self, mueq: float):
        self._factors_cache = {}
        self._mueq = float(mueq)
        self._check(self._L.gar_hip_backward_async(self._h, float(mueq)))

    def forward_async(self, theta_device_ptr: int = 0):
        self._check(self._L.gar_hip_forward_async(self._h, C.c_void_p(theta_device_ptr)))

    def num_failed(self) -> int:
        return self._L.gar_hip_num_failed(self._h)

    def slow_path_stages(self):
        """(stages of the last backward that left the register LDL^T because Rhat failed the first
        Bunch-Kaufman test, those of them where Bunch-Kaufman really pivoted), summed over the batch."""
        out = np.zeros(2, dtype=np.int64)
        self._check(self._L.gar_hip_slow_path_stages(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return int(out[0]), int(out[1])

    def constrained_bk_stages(self):
        """Constrained wave kernels, last backward, summed over the batch: (stages run as the coupled stage --
        register LDL^T of the (nu+nc) reduced KKT matrix --, stages run with the LDS Bunch-Kaufman); the rest
        ran as the decoupled D = 0 stage."""
        out = np.zeros(2, dtype=np.int64)
        self._check(self._L.gar_hip_constrained_bk_stages(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return int(out[0]), int(out[1])

    def set_refinement(self, threshold: float, max_steps: int, backward_ok: Optional[float] = None):
        self._check(self._L.gar_hip_set_refinement(self._h, float(threshold), int(max_steps)))
        if backward_ok is not None:
            self._check(self._L.gar_hip_set_condensed_backward_ok(self._h, float(backward_ok)))

    def condensed_info(self, b: int = 0):
        """(infinity norm of the last condensed residual evaluated, refinement steps taken)."""
        out = (C.c_double * 2)()
        self._check(self._L.gar_hip_condensed_info(self._h, int(b), out))
        return float(out[0]), int(out[1])

    def condensed_backward_error(self, b: int = 0) -> float:
        """Componentwise backward error of the block-cyclic-reduction solve of the condensed system (gar_hip.h)."""
        out = (C.c_double * 1)()
        self._check(self._L.gar_hip_condensed_backward_error(self._h, int(b), out))
        return float(out[0])

    def condensed_resolved(self, b: int = 0) -> bool:
        """True when the fast condensed solve of problem b (cyclic reduction / leg-parallel state elimination)
        missed its residual check and the system was solved again in the reference's order."""
        out = C.c_int(0)
        self._check(self._L.gar_hip_condensed_resolved(self._h, int(b), C.byref(out)))
        return bool(out.value)

    @property
    def condensed_solver_name(self) -> str:
        """Which fast solver the condensed system goes through before the gated chain (gar_hip.h)."""
        return self._L.gar_hip_condensed_solver_name(self._h).decode()

    def collapse_feedback(self):
        self._factors_cache = {}
        self._check(self._L.gar_hip_collapse_feedback(self._h))

    # ---- results -------------------------------------------------------------------
    def solution(self, b: int = 0):
        """-> (xs, us, vs, lbdas) as lists of per-stage vectors
        (the shape lqrInitializeSolution returns, gar/utils.hpp:114-142)."""
        d = self.dims
        N = self.horizon
        X = np.zeros(int(d[:, 0].sum()))
        U = np.zeros(int(d[:, 1].sum()))
        V = np.zeros(int(d[:, 2].sum()))
        Lb = np.zeros(self.nc0 + int(d[:N, 3].sum()))
        self._check(self._L.gar_hip_get_solution(self._h, b, _ptr(X), _ptr(U), _ptr(V), _ptr(Lb)))
        xs = list(np.split(X, np.cumsum(d[:, 0])[:-1]))
        us = [u for u in np.split(U, np.cumsum(d[:, 1])[:-1])]
        vs = list(np.split(V, np.cumsum(d[:, 2])[:-1]))
        ldim = np.concatenate([[self.nc0], d[:N, 3]])
        lbdas = list(np.split(Lb, np.cumsum(ldim)[:-1]))
        if d[N, 1] == 0:
            us.pop()
        return xs, us, vs, lbdas

    def fetch_results(self, b: int = 0, solution: bool = True, gains: bool = True):
        """gar_hip_fetch_results: ONE device-side gather + ONE device-to-host copy + ONE synchronisation
        for the whole solution and/or the gains of every stage of problem b (instead of 4 + 3 (N+1)
        small copies).  Returns (solution_record, ff_all, fb_all) as numpy views of the library's
        pinned host buffer (valid until the next fetch); None for what was not requested."""
        what = (1 if solution else 0) | (2 if gains else 0)
        self._check(self._L.gar_hip_fetch_results(self._h, int(b), what))
        offs = np.zeros(3, dtype=np.int64)
        ptr = self._L.gar_hip_host_results(self._h, offs.ctypes.data_as(C.POINTER(C.c_int64)))
        gd = np.zeros(2, dtype=np.int64)
        self._check(self._L.gar_hip_gains_doubles(self._h, gd.ctypes.data_as(C.POINTER(C.c_int64))))
        total = int(offs[2] + gd[1])
        buf = np.ctypeslib.as_array((C.c_double * total).from_address(ptr))
        sol = buf[:int(offs[1])] if solution else None
        ff = buf[int(offs[1]):int(offs[2])] if gains else None
        fb = buf[int(offs[2]):total] if gains else None
        return sol, ff, fb

    def gains_all(self, b: int = 0):
        """-> ([ff_t], [fb_t]) for every stage t, fb_t of shape (nu+nc+nx2, nx) (StageFactor's row-major
        fb), through the bulk path; the same values gar_hip_get_gains returns stage by stage."""
        _, ff, fb = self.fetch_results(b, solution=False, gains=True)
        ffs, fbs = [], []
        off = np.zeros(2, dtype=np.int64)
        for t in range(self.horizon + 1):
            nx, nu, nc, nx2, _ = (int(v) for v in self.dims[t])
            nr = nu + nc + (2 * nx2 if self.dense else nx2)
            self._check(self._L.gar_hip_gains_offsets(self._h, t, off.ctypes.data_as(C.POINTER(C.c_int64))))
            f = ff[int(off[0]):int(off[0]) + nr].copy()
            g = fb[int(off[1]):int(off[1]) + nr * nx].reshape(nr, nx).copy()
            ffs.append(f)
            fbs.append(g)
        return ffs, fbs

    def factor(self, t: int, b: int = 0) -> _FactorView:
        key = (b, t)
        if key in self._factors_cache:
            return self._factors_cache[key]
        nx, nu, nc, nx2, _ = (int(v) for v in self.dims[t])
        nth = self.effective_nth(t)
        nr = nu + nc + (2 * nx2 if self.dense else nx2)   # dense: block rows [K; Z; L; Y]
        f = _FactorView()
        f.nx, f.nu, f.nc, f.nx2, f.nth = nx, nu, nc, nx2, nth
        f.ff = np.zeros(nr)
        f.fb = np.zeros((nr, nx))
        f.fth = np.zeros((nr, nth))
        self._check(self._L.gar_hip_get_gains(self._h, b, t, _ptr(f.ff), _ptr(f.fb), _ptr(f.fth)))
        vm = _ValueView()
        vm.Vxx = np.zeros((nx, nx), order="F")
        vm.vx = np.zeros(nx)
        vm.Vxt = np.zeros((nx, nth), order="F")
        vm.Vtt = np.zeros((nth, nth), order="F")
        vm.vt = np.zeros(nth)
        self._check(self._L.gar_hip_get_value(self._h, b, t, _ptr(vm.Vxx), _ptr(vm.vx),
                                              _ptr(vm.Vxt), _ptr(vm.Vtt), _ptr(vm.vt)))
        f.vm = vm
        # StageFactor::kktMat = [Rhat D^T; D -mu I] (expose-prox-riccati.cpp:30-31): not kept by the sweeps, formed
        # on the device on request from the knot and stage t+1's Vxx (gar_hip_get_kkt)
        mueq = self._mueq

        def kkt():
            out = np.zeros((nu + nc, nu + nc), order="F")
            if not self.dense and nu + nc > 0 and mueq is not None:
                self._check(self._L.gar_hip_get_kkt(self._h, b, t, float(mueq), _ptr(out)))
            return out
        f._kkt, f._kkt_fn = None, kkt
        self._factors_cache[key] = f
        return f

    def initial(self, b: int = 0):
        """-> (kkt0.ff, kkt0.fth, thGrad, thHess) (proximal-riccati.hpp:40-43)."""
        nx0 = int(self.dims[0, 0])
        nth = self.effective_nth(0)
        n0 = nx0 + self.nc0
        ff, fth = np.zeros(n0), np.zeros((n0, nth))
        g, H = np.zeros(nth), np.zeros((nth, nth), order="F")
        self._check(self._L.gar_hip_get_initial(self._h, b, _ptr(ff), _ptr(fth), _ptr(g), _ptr(H)))
        return ff, fth, g, H

    def cycle_append(self, dims5):
        d = np.ascontiguousarray(np.asarray(dims5, dtype=np.int32))
        self._check(self._L.gar_hip_cycle_append(self._h, d.ctypes.data_as(C.POINTER(C.c_int32))))
        N = self.horizon
        if N >= 1:
            nd = self.dims.copy()
            nd[:N - 1] = self.dims[1:N]
            nd[N - 1] = d
            self.dims = nd
        self._refresh_layout()


class RiccatiSolverBase:
    """gar::RiccatiSolverBase (gar/riccati-base.hpp:13-37)."""

    def backward(self, mueq: float) -> bool:
        raise NotImplementedError

    def forward(self, xs, us, vs, lbdas, theta=None) -> bool:
        raise NotImplementedError

    def cycleAppend(self, knot: LqrKnot) -> None:
        raise NotImplementedError

    def collapseFeedback(self) -> None:
        pass

    def getFeedforward(self, i: int) -> np.ndarray:
        raise NotImplementedError

    def getFeedback(self, i: int) -> np.ndarray:
        raise NotImplementedError


class _HipSolver(RiccatiSolverBase):
    def __init__(self, problem: LqrProblem, num_legs: int, device: int, lib_path, devices=None):
        self.problem_ = problem  # non-owning, re-read on every backward()
        dims = [k.dims for k in problem.stages]
        self._num_legs = num_legs
        self._device, self._lib_path, self._devices = device, lib_path, devices
        self._make(dims)

    _dense = False

    def _make(self, dims):
        if self._num_legs > 1:
            dims = [(nx, nu, nc, nx2, 0) for (nx, nu, nc, nx2, _) in dims]
        self._impl = BatchedRiccatiSolver(dims, self.problem_.nc0, 1, self._num_legs,
                                          self._device, lib_path=self._lib_path, dense=self._dense,
                                          devices=self._devices)

    class _Datas:
        def __init__(self, impl):
            self._impl = impl

        def __getitem__(self, t) -> _FactorView:
            if t < 0:
                t += self._impl.horizon + 1
            return self._impl.factor(t)

        def __len__(self):
            return self._impl.horizon + 1

    @property
    def datas(self):
        return _HipSolver._Datas(self._impl)

    def _upload(self):
        p = self.problem_
        if p.horizon != self._impl.horizon:
            raise ValueError("problem horizon changed; create a new solver")
        for t, k in enumerate(p.stages):
            self._impl.upload_knot(0, t, k)
        self._impl.set_init(0, p.G0, p.g0)

    def forward(self, xs, us, vs, lbdas, theta=None) -> bool:
        self._impl.forward(theta)
        X, U, V, Lb = self._impl.solution(0)
        for dst, src in ((xs, X), (us, U), (vs, V), (lbdas, Lb)):
            for i in range(min(len(dst), len(src))):
                dst[i][...] = src[i]
        return True

    def getFeedforward(self, i: int) -> np.ndarray:
        return self._impl.factor(i).ff

    def getFeedback(self, i: int) -> np.ndarray:
        return self._impl.factor(i).fb

    @property
    def kernel_name(self):
        return self._impl.kernel_name


class ProximalRiccatiSolver(_HipSolver):
    """gar::ProximalRiccatiSolver on the MI355X backend (serial in time)."""

    def __init__(self, problem: LqrProblem, device: int = 0, lib_path=None):
        super().__init__(problem, 1, device, lib_path)

    def backward(self, mueq: float) -> bool:
        self._upload()
        return self._impl.backward(mueq)

    @property
    def kkt0(self):
        k = _Kkt0View()
        k.ff, k.fth, _, _ = self._impl.initial(0)

        def mat():   # [Vxx0 G0^T; G0 0] (proximal-riccati.hxx:44-47), the lower triangle being what is factorised
            V, G0 = self._impl.factor(0).vm.Vxx, np.asarray(self.problem_.G0)
            n, m = V.shape[0], G0.shape[0]
            M = np.zeros((n + m, n + m), order="F")
            M[:n, :n] = V
            M[n:, :n] = G0
            M[:n, n:] = G0.T
            return M
        k._mat_fn = mat
        return k

    @property
    def thGrad(self):
        return self._impl.initial(0)[2]

    @property
    def thHess(self):
        return self._impl.initial(0)[3]

    def cycleAppend(self, knot: LqrKnot) -> None:
        """proximal-riccati.hxx:79-86."""
        self._impl.cycle_append(knot.dims)


class RiccatiSolverDense(_HipSolver):
    """gar::RiccatiSolverDense on the MI355X backend (gar/dense-riccati.hpp:19-56): per stage one
    Bunch-Kaufman factorisation of the whole (nu+nc+2 nx2)^2 matrix instead of the condensation.
    `datas[t]` holds ff / fb / fth with block rows [K; Z; L; Y] (the reference's stage_factors[t].ff,
    .fb, .ft) and, as `vm.Vxx, vx, Vxt, Vtt, vt`, the reference's Pxx[t], px[t], Pxt[t], Ptt[t], pt[t]."""
    _dense = True

    def __init__(self, problem: LqrProblem, device: int = 0, lib_path=None):
        super().__init__(problem, 1, device, lib_path)

    def backward(self, mueq: float) -> bool:
        self._upload()
        return self._impl.backward(mueq)

    @property
    def kkt0(self):
        k = _Kkt0View()
        k.ff, k.fth, _, _ = self._impl.initial(0)

        def mat():   # [Vxx0 G0^T; G0 0] (proximal-riccati.hxx:44-47), the lower triangle being what is factorised
            V, G0 = self._impl.factor(0).vm.Vxx, np.asarray(self.problem_.G0)
            n, m = V.shape[0], G0.shape[0]
            M = np.zeros((n + m, n + m), order="F")
            M[:n, :n] = V
            M[n:, :n] = G0
            M[:n, n:] = G0.T
            return M
        k._mat_fn = mat
        return k

    @property
    def thGrad(self):
        return self._impl.initial(0)[2]

    @property
    def thHess(self):
        return self._impl.initial(0)[3]

    def cycleAppend(self, knot: LqrKnot) -> None:
        """dense-riccati.hxx:118-146."""
        self._impl.cycle_append(knot.dims)


class ParallelRiccatiSolver(_HipSolver):
    """gar::ParallelRiccatiSolver: `num_threads` legs, one workgroup per leg.

    Like the reference, construction MUTATES the caller's problem: every knot of
    a non-final leg is re-parameterised with nth = nx (parallel-solver.hxx:52-60)
    and backward() rewrites Gx, Gu, Gth, gamma of each leg-end knot (:136-147).
    The device records keep this parameterisation implicit.
    """

    def __init__(self, problem: LqrProblem, num_threads: int, device: int = 0, lib_path=None, devices=None):
        """devices: a list of device ids -- the legs are split over these devices inside this one object
        (include/gar_hip.h, gar_hip_multi_create): `linear_solver_` stays ONE RiccatiSolverBase."""
        if num_threads < 2:
            raise RuntimeError(f"(ParallelRiccatiSolver) numThreads ({num_threads}) should be "
                               "greater than or equal to 2.")  # parallel-solver.hxx:42-46
        self.numThreads_ = int(num_threads)
        self.condensedThreshold = 1e-10   # parallel-solver.hpp:92
        self.maxRefinementSteps = 5       # parallel-solver.hpp:94
        self._parameterize(problem)
        super().__init__(problem, self.numThreads_, device, lib_path, devices=devices)

    def getNumThreads(self) -> int:
        return self.numThreads_

    def _parameterize(self, problem: LqrProblem):
        N = problem.horizon
        for i in range(self.numThreads_ - 1):
            i0, i1 = get_work(N, i, self.numThreads_)
            nth = problem.stages[i1 - 1].nx2
            for t in range(i0, i1):
                problem.stages[t].addParameterization(nth)

    def backward(self, mueq: float) -> bool:
        p = self.problem_
        N = p.horizon
        for i in range(self.numThreads_ - 1):  # configure_knot (:136-147)
            _, end = get_work(N, i, self.numThreads_)
            k = p.stages[end - 1]
            k.Gx[...] = k.A.T
            k.Gu[...] = k.B.T
            k.Gth[...] = 0.0
            k.gamma[...] = k.f
        self._impl.set_refinement(self.condensedThreshold, self.maxRefinementSteps)
        if p.horizon != self._impl.horizon:
            raise ValueError("problem horizon changed; create a new solver")
        return self._impl.backward_blocks(p, mueq)   # upload + sweep in one call

    def forward(self, xs, us, vs, lbdas, theta=None) -> bool:
        return super().forward(xs, us, vs, lbdas, None)  # theta ignored (:209-212)

    def collapseFeedback(self) -> None:
        self._impl.collapse_feedback()

    def cycleAppend(self, knot: LqrKnot) -> None:
        """parallel-solver.hxx:246-258: drop the parameterisation, re-initialise."""
        self.problem_.addParameterization(0)
        self._parameterize(self.problem_)
        self._make([k.dims for k in self.problem_.stages])
