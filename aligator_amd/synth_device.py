"""Device-side synthetic LQ problems (torch is plumbing here: RNG + batched
matmul on the GPU, written straight into the packed device records).

Same distributions as aligator_amd.synth (which restates
tests/gar/test_util.cpp:14-76): generator "F" (faithful) or "W"
(well-conditioned), every knot of every problem drawn independently, so the
timed sweep streams distinct data from HBM (no cache-resident replicas).
Uniform-dimension problems (nth = 0): unconstrained, or -- the reference's own benchmark shape,
bench/gar-riccati.cpp:19-22 -- with nc equality constraints on every knot, `C = [I 0]`, `d ~ U[-1,1]`
(test_util.cpp:41-44) and `D = 0` or (`coupled=True`) `D ~ U[-1,1]`: the coupled reduced-KKT stage.
"""
from __future__ import annotations

import numpy as np
import torch

from .lqr import LqrProblem


def _colmajor(m: torch.Tensor) -> torch.Tensor:
    """[..., r, c] -> [..., r*c] in column-major order."""
    return m.transpose(-1, -2).reshape(*m.shape[:-2], m.shape[-2] * m.shape[-1])


def fill_problems(solver, seed: int = 1234, mode: str = "W", singular: bool = True,
                  chunk: int = 64, keep=(0, -1), coupled: bool = False):
    """Generate `solver.batch` problems on the GPU and load them into the solver.
    Keeps host copies of the problems listed in `keep` (for oracle spot checks)."""
    d = solver.dims
    N = solver.horizon
    nx, nu, nc = int(d[0, 0]), int(d[0, 1]), int(d[0, 2])
    assert N >= 1 and (d[:N] == d[0]).all() and d[0, 4] == 0 and d[N, 1] == 0 and d[N, 2] == nc
    assert solver.nc0 == nx
    assert nc == 0 or not solver.padded, "constrained shapes are generated in their own dimensions"
    # the DEVICE records (= the caller's unless the library padded the shape onto a specialised family: then the
    # dummy states / controls are filled in as the library itself does -- Q = I, R = I on them, everything else 0,
    # pinned by the extra rows [0 -I] of G0)
    dd = solver.device_dims
    NX, NU = int(dd[0, 0]), int(dd[0, 1])
    # (without a GPU -- the test-only emulator build of the library -- the same records are built in host memory)
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    f64 = torch.float64
    so = solver.device_stage_offsets
    rec = int(so[1, 0] - so[0, 0]) if N > 1 else int(so[N, 0] - so[0, 0])
    off0 = int(so[0, 0])
    offN = int(so[N, 0])
    P = solver.device_problem_doubles
    nc0d, G0_off = solver.device_nc0, solver.device_G0_off
    nw = nx + nu
    keep_idx = sorted({k % solver.batch for k in keep})
    solver._host_samples = {}

    def randn(*s):
        return torch.randn(*s, generator=gen, device=dev, dtype=f64)

    def randu(*s):
        return torch.rand(*s, generator=gen, device=dev, dtype=f64) * 2.0 - 1.0

    for b0 in range(0, solver.batch, chunk):
        nb = min(chunk, solver.batch - b0)
        buf = torch.zeros(nb, P, device=dev, dtype=f64)
        # G0 = -I, g0 = x0 = 0   (tests/gar/test_util.cpp:72-74)
        G0 = -torch.eye(nc0d, NX, device=dev, dtype=f64)
        buf[:, G0_off:G0_off + nc0d * NX] = _colmajor(G0)
        root = randn(nb, N, nw, nw + 1)
        qsr = root @ root.transpose(-1, -2) / max(nx, nu)
        Q = qsr[..., :nx, :nx]
        if singular:
            r2 = randn(nb, N, nx, int(0.8 * nw))
            Q = r2 @ r2.transpose(-1, -2)
        S = qsr[..., :nx, nx:]
        R = qsr[..., nx:, nx:].clone()
        R.diagonal(dim1=-2, dim2=-1).mul_(1.0 + 1e-6)
        if mode == "F":
            A, B = randu(nb, N, nx, nx), randu(nb, N, nx, nu)
        else:
            A = torch.eye(nx, device=dev, dtype=f64) + 0.1 * randu(nb, N, nx, nx)
            B = 0.5 * randu(nb, N, nx, nu)
        def pad(m, R, C, diag=0.0):
            """[..., r, c] -> [..., R, C] with `diag` on the new part of the diagonal."""
            if m.shape[-2:] == (R, C):
                return m
            out = torch.zeros(*m.shape[:-2], R, C, device=dev, dtype=f64)
            out[..., :m.shape[-2], :m.shape[-1]] = m
            if diag:
                i = torch.arange(min(m.shape[-2], m.shape[-1]), min(R, C), device=dev)
                out[..., i, i] = diag
            return out

        def padv(v, R):
            return torch.nn.functional.pad(v, (0, R - v.shape[-1]))
        def sym_block(m, n):
            """The device's Q / R block: full column-major, or (solver.qr_packed, csrc/gar_layout.h) the lower
            triangle packed column after column in the first n (n + 1) / 2 doubles, the rest of the block untouched."""
            if not getattr(solver, "qr_packed", False):
                return _colmajor(m)
            iu = torch.triu_indices(n, n, device=dev)           # (r, c), r <= c, ordered by r then c ...
            low = m[..., iu[1], iu[0]]                           # ... = column r of the lower triangle, rows c >= r
            return torch.nn.functional.pad(low, (0, n * n - low.shape[-1]))
        stage = torch.cat([sym_block(pad(Q, NX, NX, 1.0), NX), _colmajor(pad(S, NX, NU)), sym_block(pad(R, NU, NU, 1.0), NU),
                           padv(randu(nb, N, nx), NX), padv(randu(nb, N, nu), NU), _colmajor(pad(A, NX, NX)),
                           _colmajor(pad(B, NX, NU)), padv(randn(nb, N, nx), NX)], dim=-1)
        if nc > 0:   # C = [I 0], D = 0 or U[-1, 1], d ~ U[-1, 1]   (test_util.cpp:41-44)
            Cm = torch.eye(nc, nx, device=dev, dtype=f64).expand(nb, N, nc, nx)
            Dm = randu(nb, N, nc, nu) if coupled else torch.zeros(nb, N, nc, nu, device=dev, dtype=f64)
            stage = torch.cat([stage, _colmajor(Cm), _colmajor(Dm), randu(nb, N, nc)], dim=-1)
        assert stage.shape[-1] <= rec
        view = buf[:, off0:off0 + N * rec].view(nb, N, rec)
        view[..., :stage.shape[-1]] = stage
        # terminal knot: nu = 0, non-singular Q (test_util.cpp:69)
        rt = randn(nb, nx, nx + 1)
        Qt = rt @ rt.transpose(-1, -2) / nx
        At = randu(nb, nx, nx) if mode == "F" else (torch.eye(nx, device=dev, dtype=f64) + 0.1 * randu(nb, nx, nx))
        term = torch.cat([_colmajor(pad(Qt, NX, NX, 1.0)), padv(randu(nb, nx), NX), _colmajor(pad(At, NX, NX)),
                          padv(randn(nb, nx), NX)], dim=-1)
        if nc > 0:   # the terminal knot carries its constraint too (nu = 0: no D block)
            term = torch.cat([term, _colmajor(torch.eye(nc, nx, device=dev, dtype=f64).expand(nb, nc, nx)), randu(nb, nc)], dim=-1)
        buf[:, offN:offN + term.shape[-1]] = term
        if dev.type == "cuda":
            torch.cuda.synchronize()  # generation done before the solver's stream copies it
        # (the format these records were just written in: refused, not swept, if the solver's differs)
        solver.upload_packed_device(buf.data_ptr(), b0, nb, record_format=1 if getattr(solver, "qr_packed", False) else 0)
        solver.sync()
        for k in keep_idx:
            if b0 <= k < b0 + nb:
                solver._host_samples[k] = buf[k - b0].cpu().numpy()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        del buf, root, qsr, stage
    solver.sync()


def download_problem(solver, b: int) -> LqrProblem:
    """Host LqrProblem (the caller's dimensions) of a sampled problem."""
    k = b % solver.batch
    # (gar_hip_download_packed converts from whatever the device records are -- padded, packed triangles -- to the
    # caller's packed layout; the host copies kept by fill_problems serve the problems listed in `keep` only)
    if solver.padded or getattr(solver, "qr_packed", False) or k not in getattr(solver, "_host_samples", {}):
        return solver.unpack(solver.download_packed(k, 1))
    return solver.unpack(solver._host_samples[k])
