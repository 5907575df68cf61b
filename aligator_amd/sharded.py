"""Horizon sharding of ONE LQ problem (or one batch of them) over the ranks of a
``torch.distributed`` process group: one process per GPU, RCCL over xGMI.

This is the multi-GPU form of gar::ParallelRiccatiSolver
(gar/parallel-solver.hxx:132-243).  In the reference the legs are OpenMP threads
of one process and the "boundary exchange" is the implicit barrier that closes
the parallel region (:150-164 -> assembleCondensedSystem at :169).  Here rank r
owns legs [r J / W, (r+1) J / W) of J = `num_legs` over W ranks (any W <= J, like get_work over threads):

  1. per-rank leg-parallel backward          gar_hip_backward_legs_async
  2. ONE all-gather of the per-leg boundary tuples (Vxx | Vxt | Vtt | vx | vt of
     the leg's first stage: 3 nx^2 + 2 nx doubles, 31.7 KB at nx = 36)
  3. the small condensed block-tridiagonal system is solved redundantly on every
     rank (no broadcast)                      gar_hip_condensed_solve_async
  4. per-rank leg-parallel forward            gar_hip_forward_legs_async

xGMI is point-to-point and the payload is tens of KB, so the exchange is
latency-bound: a single fused collective per sweep, no O(log J) rounds.

torch.distributed is plumbing here (process group + the collective on the
library's own device buffers); all arithmetic stays in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from .gar import BatchedRiccatiSolver, get_work

__all__ = ["ShardedRiccatiSolver"]


class _DevArray:
    """A raw device pointer as a __cuda_array_interface__ object (float64, 1-D)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False),
                                         "version": 2, "strides": None}


def _view(ptr: int, n: int, on_device: bool) -> torch.Tensor:
    """A torch tensor ALIASING n doubles at `ptr` (no copy)."""
    if on_device:
        return torch.as_tensor(_DevArray(ptr, n), device=torch.device("cuda", torch.cuda.current_device()))
    buf = (C.c_double * n).from_address(ptr)  # host memory (the CPU test build of the library)
    return torch.from_numpy(np.ctypeslib.as_array(buf))


class ShardedRiccatiSolver:
    """`batch` problems of identical dimensions, horizon split into `num_legs` legs,
    the legs split over the ranks of `group` (any number of ranks <= num_legs)."""

    def __init__(self, dims, nc0: int, num_legs: int, batch: int = 1, device: int = 0,
                 group: Optional[dist.ProcessGroup] = None, lib_path: Optional[str] = None,
                 on_device: Optional[bool] = None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if num_legs < 2 or num_legs < self.world:
            # (the reference takes any thread count >= 2, parallel-solver.hxx:42-46; every rank needs a leg)
            raise ValueError("num_legs must be >= 2 and >= the number of ranks")
        self.num_legs = num_legs
        # rank r owns legs [r J / W, (r+1) J / W) -- any split, like get_work over threads (:23-28); the
        # all-gather moves equal chunks of ceil(J / W) tuples, short chunks leave their tail unused
        self.leg_range = (self.rank * num_legs // self.world, (self.rank + 1) * num_legs // self.world)
        self.legs_per_rank = -(-num_legs // self.world)
        lo = self.leg_range[0]
        self.impl = BatchedRiccatiSolver(dims, nc0, batch=batch, num_legs=num_legs, device=device,
                                         rank_of=(self.rank, self.world), lib_path=lib_path)
        self.on_device = torch.cuda.is_available() if on_device is None else on_device
        L, h = self.impl._L, self.impl.handle
        tup = int(L.gar_hip_boundary_doubles(h))
        n_local = batch * self.legs_per_rank * tup
        self._local = _view(L.gar_hip_device_boundary_local(h), n_local, self.on_device)
        self._all = _view(L.gar_hip_device_boundary_all(h), n_local * self.world, self.on_device)
        # gloo moves host memory only: with device buffers (two ranks sharing one GPU in the tests, bench.py
        # --backend gloo) the exchange is staged through the host -- a test path; RCCL gathers in place
        self._staged = self.on_device and dist.get_backend(group) == "gloo"
        self.stream = None
        if self.on_device:
            # one (non-null) stream for the kernels and the collective: no host round trip inside a
            # sweep (gar_hip_set_stream(NULL) would select the library's private stream)
            self.stream = torch.cuda.Stream()
            self.impl.set_stream(self.stream.cuda_stream)
        self.stage_range = (get_work(self.impl.horizon, lo, num_legs)[0],
                            get_work(self.impl.horizon, self.leg_range[1] - 1, num_legs)[1])

    # ---- the sweep -----------------------------------------------------------------
    def backward(self, mueq: float, check: bool = True) -> bool:
        """Leg sweep -> ONE all-gather of the boundary tuples -> redundant condensed solve.
        On the GPU nothing here synchronises the host: the library's kernels are enqueued on torch's
        current stream (set in __init__), the RCCL collective orders itself after them and the
        condensed solve after the collective through torch's own stream events.  `check` (the
        reference throws on a failed stage factorisation, riccati-kernel.hxx:239-241) costs one
        device-to-host copy and an all-reduce of the flag, so that EVERY rank raises together."""
        L, h = self.impl._L, self.impl.handle
        self.impl._factors_cache = {}
        self.impl._mueq = float(mueq)   # datas[t].kktMat is formed on request from this sweep's mueq
        import contextlib
        with (torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()):
            self.impl._check(L.gar_hip_backward_legs_async(h, float(mueq)))   # parallel-solver.hxx:150-164
            # the boundary exchange: one all-gather, rank-major == the layout the condensed solve reads
            self.exchange()
            self.impl._check(L.gar_hip_condensed_solve_async(h))             # :169-202, redundant per rank
        if check:
            nf = torch.tensor([self.impl.num_failed()], dtype=torch.int32,
                              device=self._local.device if self.on_device else "cpu")
            dist.all_reduce(nf, op=dist.ReduceOp.MAX, group=self.group)
            if int(nf.item()) != 0:
                raise RuntimeError("Failed stage LDL factorization")
        return True

    def exchange(self):
        """ONE all-gather of the boundary tuples into the buffer the condensed solve reads (equal chunks of
        ceil(J / W) tuples per rank)."""
        if not self._staged:
            dist.all_gather_into_tensor(self._all, self._local, group=self.group)
            return
        self.stream.synchronize()
        host = torch.empty(self._all.numel(), dtype=torch.float64)
        dist.all_gather_into_tensor(host, self._local.cpu(), group=self.group)
        self._all.copy_(host)

    def forward(self, sync: bool = True) -> bool:
        L, h = self.impl._L, self.impl.handle
        self.impl._check(L.gar_hip_forward_legs_async(h))                # :209-243
        if sync:
            self.impl.sync()
        return True

    # ---- results ---------------------------------------------------------------------
    def local_solution(self, b: int = 0):
        """(xs, us, vs, lbdas) of problem b; only this rank's stages
        [stage_range[0], stage_range[1]) carry results."""
        return self.impl.solution(b)

    def gather_solution(self, b: int = 0):
        """Every rank's stages merged (an all_gather of the per-stage vectors; test/diagnostic
        helper, not part of the timed path)."""
        sol = self.impl.solution(b)
        flat = torch.from_numpy(np.concatenate([np.concatenate([np.ravel(v) for v in part])
                                                if part else np.zeros(0) for part in sol]))
        if self.on_device:
            flat = flat.cuda()
        parts = [torch.empty_like(flat) for _ in range(self.world)]
        dist.all_gather(parts, flat, group=self.group)
        parts = [p.cpu().numpy() for p in parts]
        N = self.impl.horizon
        owner = np.zeros(N + 1, dtype=int)
        for r in range(self.world):
            l0, l1 = r * self.num_legs // self.world, (r + 1) * self.num_legs // self.world
            s0 = get_work(N, l0, self.num_legs)[0]
            s1 = get_work(N, l1 - 1, self.num_legs)[1]
            owner[s0:s1] = r
        out = []
        pos = 0
        for pi, part in enumerate(sol):
            merged = []
            for t, v in enumerate(part):
                # xs[t], us[t], vs[t] belong to stage t; lbdas[t] (t >= 1) is produced with
                # x_t by the leg holding stage t-1 ... except at a leg start, where it comes
                # from the condensed solution every rank holds (parallel-solver.hxx:215-220)
                r = owner[min(t, N)]
                merged.append(parts[r][pos:pos + v.size].reshape(v.shape).copy())
                pos += v.size
            out.append(merged)
        return tuple(out)
