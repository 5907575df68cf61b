#!/usr/bin/env python3
"""bench.py -- batched Riccati sweeps/s on MI355X (BASELINE.json metric).

One *step* = one pass of the hot path over one batch of synthetic LQ problems:
``backward(mueq)`` + ``forward(xs,us,vs,lbdas)`` on every problem of the batch
(the timed body of the reference's bench/gar-riccati.cpp:46-49), inputs already
resident in HBM.  Workload = BASELINE.json configs[1]: N=256, nx=36, nu=12,
nc=0, fp64, serial in time, `--batch` independent problems per GPU.

N>1 GPUs: the batch axis shards with no data-path collective (weak scaling:
`--batch` problems PER GPU); RCCL is used only for the timing barrier/max.
`--gpus N` without a torchrun environment (WORLD_SIZE unset) spawns its own N ranks through
torch.distributed.run on 127.0.0.1 -- like ParallelRiccatiSolver(problem, num_threads) spawns its own team
(parallel-solver.hxx:150) -- and fails loudly when fewer than N devices are visible.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# The pipelined sweep (gar_hip_set_pipeline) runs the two halves of the batch on two HIP streams; the runtime maps
# streams onto at most GPU_MAX_HW_QUEUES (default 4) hardware queues and two streams that share one run their kernels
# one after the other.  Read by the HIP runtime when it initialises: before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# thread placement of the CPU baseline (read by libgomp when the oracle library is first loaded)
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_PROC_BIND", "close")

from aligator_amd import synth_device  # noqa: E402
from aligator_amd.gar import BatchedRiccatiSolver  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec; 6.29e12 measured copy)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # scripts/gpu_r5_evidence.sh, step "pmc" (rocprofv3 --pmc, reduced by scripts/pmc_reduce.py)


def pmc_traffic(kernel, batch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, corrected as MI355X_MICROARCH.md prescribes, with the
    factors calibrated on a known-size copy in the same run); None if not collected for this batch."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        e = d["kernels"][kernel]
        if int(e["batch"]) != int(batch):
            return None
        return float(e["hbm_bytes_per_launch"])
    except Exception:
        return None


def pmc_traffic_in_run(args, N, nx, nu, timeout=240):
    """HBM bytes per launch of the backward sweep kernel measured IN THIS RUN when rocprofv3 is on PATH: two
    counter-only passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; `--pmc X --kernel-trace`, nothing else, as
    MI355X_MICROARCH.md prescribes) over a short child run of this script (`--pmc-child`: the same solver, batch and
    data, one warm-up and two steps, then the library's own streaming kernel on the same number of records).  The
    counters' units are calibrated in the same pass on that streaming kernel, whose byte counts are known exactly
    and whose access pattern is the sweep's (16 B per lane, one record in flight per wave): bytes = counts x
    (known bytes / counts of gar_stream_sweep).  Returns (bytes, detail) or (None, reason)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    knot_b = 8 * knot_doubles_read(nx, nu)
    fac_b = 8 * ((nu + nx) * (nx + 1) + nx * (nx + 1) // 2 + nx)   # what pmc_child asks the streaming kernel to write
    known = {"FETCH_SIZE": float(args.batch) * N * (-(-knot_b // 16) * 16), "WRITE_SIZE": float(args.batch) * N * (-(-fac_b // 16) * 16)}
    got, detail = {}, {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="gar_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--batch", str(args.batch), "--horizon", str(N),
               "--nx", str(nx), "--nu", str(nu), "--generator", args.generator]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited {r.returncode}: {r.stdout.decode(errors='replace')[-200:]}"
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter:
                        per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            bwd = [v for k, v in per.items() if "gar_backward" in k]
            cal = [v for k, v in per.items() if "gar_stream_sweep" in k]
            if not bwd or not cal or not sum(cal[0]):
                return None, f"no {counter} rows for the sweep / calibration kernels"
            per_count = known[counter] / (sum(cal[0]) / len(cal[0]))
            got[counter] = sum(bwd[0]) / len(bwd[0]) * per_count
            detail[counter] = {"counts_per_launch": sum(bwd[0]) / len(bwd[0]), "bytes_per_count": per_count,
                               "launches": len(bwd[0])}
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {counter} timed out after {timeout} s"
        except Exception as e:  # the measurement never fails the bench line
            return None, f"{type(e).__name__}: {str(e)[:120]}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    detail["read_bytes"], detail["write_bytes"] = got["FETCH_SIZE"], got["WRITE_SIZE"]
    return got["FETCH_SIZE"] + got["WRITE_SIZE"], detail


def pmc_child(args):
    """The workload of pmc_traffic_in_run's counter passes (run under rocprofv3): the bench's solver and data, one
    warm-up and two steps, then the streaming kernel the counters are calibrated on."""
    N, nx, nu = args.horizon, args.nx, args.nu
    dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
    torch.cuda.set_device(0)
    solver = BatchedRiccatiSolver(dims, nx, batch=args.batch, num_legs=1, device=0)
    synth_device.fill_problems(solver, seed=1234, mode=args.generator, keep=())
    solver.set_pipeline(0)   # whole-batch launches: the counters are reduced per launch of `batch` problems
    for _ in range(3):
        solver.backward_async(1e-14)
        solver.forward_async()
    solver.sync()
    knot_b = 8 * knot_doubles_read(nx, nu, solver.qr_packed)
    fac_b = 8 * ((nu + nx) * (nx + 1) + nx * (nx + 1) // 2 + nx)   # (packed Vxx: moved_bytes)
    solver.close()
    ms = solver._L.gar_hip_stream_ceiling_ms(0, int(args.batch), int(N), knot_b, fac_b, 1)
    print(json.dumps({"pmc_child": True, "stream_ms": ms}))


def spawn_ranks(args):
    """`--gpus N` without a torchrun environment: re-exec through torch.distributed.run with N local ranks."""
    import socket, subprocess
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n < args.gpus and not args.same_device:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n} HIP device(s) visible")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc = 1
    for attempt in range(3):
        # the rendezvous store listens on EVERY local address: probe the port there, not on 127.0.0.1 alone (a port
        # free on loopback can be held on another address: EADDRINUSE); a launch that dies at once is tried again
        with socket.socket() as so:
            so.bind(("", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        t0 = time.perf_counter()
        rc = subprocess.run(cmd, env=env).returncode
        if rc == 0 or time.perf_counter() - t0 > 45.0:
            break
        print(f"bench.py: launch on port {port} failed within {time.perf_counter() - t0:.0f} s (rc {rc}); "
              f"{'trying another port' if attempt < 2 else 'giving up'}", file=sys.stderr)
    raise SystemExit(rc)


def algorithmic_bytes(N, nx, nu):
    """per stage, nc = nth = 0, doubles."""
    knot = 2 * nx * nx + 2 * nx * nu + nu * nu + 2 * nx + nu          # 3684
    fac = (nu + nx) * nx + (nu + nx) + nx * nx + nx                    # 3108
    fwd_out = 2 * nx + nu                                              # 84
    bwd = 8 * (knot + fac) * N
    fwd = 8 * (fac + fwd_out) * N
    return bwd, fwd


def knot_doubles_read(nx, nu, qr_packed=True):
    """Doubles of a knot record the backward sweep reads: everything, or (csrc/gar_layout.h, `qr_packed`: the headline
    one-wave sweep's records keep Q and R as their packed lower triangles) 2 988 instead of 3 684 at (36, 12)."""
    sym = (nx * (nx + 1) // 2 + nu * (nu + 1) // 2) if qr_packed else (nx * nx + nu * nu)
    return sym + nx * nx + 2 * nx * nu + 2 * nx + nu


def moved_bytes(N, nx, nu, qr_packed=True):
    """What the serial one-wave kernels actually move per sweep: the factor record keeps the symmetric Vxx as its
    lower triangle, packed (csrc/gar_layout.h: gar_sym_index) -- nx (nx + 1) / 2 doubles instead of nx^2, written by
    the backward sweep and read by the forward sweep -- and (round 4, `qr_packed`) the knot records keep Q and R as
    their packed lower triangles: the upper triangles never influence a result (riccati-kernel.hxx:216, :232)."""
    knot = knot_doubles_read(nx, nu, qr_packed)
    fac = (nu + nx) * nx + (nu + nx) + nx * (nx + 1) // 2 + nx        # 2478 at (36, 12)
    fwd_out = 2 * nx + nu
    return 8 * (knot + fac) * N, 8 * (fac + fwd_out) * N


def cpu_quota_cores():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota); None if unlimited."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(-(-int(q) // int(p))))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, -(-q // p))
    except Exception:
        return None


def cpu_baseline(args, nx, nu, N, mueq):
    """The oracle (restated reference, NOT the Eigen build) timed on this box's host cores
    (BASELINE.md section 2): C2 = OpenMP over independent problems, every thread sweeping thread-local
    (first-touch, NUMA-local) copies of its problems with OMP_PLACES=cores / OMP_PROC_BIND=close; C1 = one
    thread, one problem; C3 = the leg-parallel solver (J threads) on one problem, J in {2, 3, 4, 6}
    (bench/gar-riccati.cpp:87-90); C4 = the reference's own benchmark shape nc = 32 (:19-22).
    A bounded sample (~15 s of CPU work); a reported baseline, never the target."""
    from aligator_amd import synth
    from oracle import oracle as ora

    budget = max(args.cpu_seconds, 0.5)
    L = ora.lib(native=True)  # rebuilt with -march=native on THIS host
    base = []
    t0 = time.time()
    for i in range(8):
        p = synth.generate_lq_problem(np.random.default_rng(1234 + i), np.zeros(nx), N, nx, nu,
                                      mode=args.generator)
        base.append(ora.Problem.from_knots(p.stages, p.G0, p.g0, native=True))
        if i == 0:
            first = p
        if time.time() - t0 > 20:
            break
    threads = ora.BatchSweep(base[:1]).max_threads()
    # the GPU boxes of this pool expose 256 logical CPUs but grant the container a 16-CPU quota: more threads than
    # that only get throttled (128 threads measured 8 % parallel efficiency), so the team is sized to the quota
    quota = cpu_quota_cores()
    if quota is not None:
        threads = max(1, min(threads, quota))
    # C1: one thread
    bs1 = ora.BatchSweep(base[:2])
    bs1.sweep_local(mueq, 1, 1)
    f1, s1 = bs1.sweep_local(mueq, 1, 3)
    lat = s1 / 6.0
    # C2: all threads, two problems per thread, repetitions sized to the budget
    probs = [base[i % len(base)] for i in range(2 * threads)]
    bs = ora.BatchSweep(probs[:1])
    bs.problems = probs
    _, sw = bs.sweep_local(mueq, threads, 1)
    reps = int(max(1, min(50, 0.6 * budget / max(sw, 1e-3))))
    fails, sec = bs.sweep_local(mueq, threads, reps)
    assert fails == 0 and f1 == 0
    rate = len(probs) * reps / sec
    out = {"value": rate, "unit": "sweeps/s", "cores": threads, "kind": "port",
           "sample": f"{len(probs)} problems ({len(base)} distinct) x {reps} reps, one OpenMP thread per "
                     f"problem pair, thread-local copies, OMP_PLACES={os.environ.get('OMP_PLACES')} "
                     f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}; oracle/gar_oracle.c -O3 "
                     f"-march=native (restated reference, not the Eigen build)",
           "logical_cpus": os.cpu_count(), "cpu_quota_cores": quota,
           "one_thread_ms_per_sweep": lat * 1e3,
           "parallel_efficiency": rate * lat / threads}
    # C3: leg-parallel, one problem (its OpenMP team = J threads)
    legs = {}
    t_c3 = time.time()
    for J in (2, 3, 4, 6):
        if J > threads or time.time() - t_c3 > 0.2 * budget + 2:
            break
        op = ora.Problem.from_knots(first.stages, first.G0, first.g0, native=True)  # (the solver mutates it)
        par = ora.ParallelRiccatiSolver(op, J)
        sol = op.initialize_solution()
        best = 1e30
        for _ in range(3):
            t1 = time.perf_counter()
            par.backward(mueq)
            par.forward(*sol)
            best = min(best, time.perf_counter() - t1)
        legs[str(J)] = best * 1e3
    out["leg_parallel_ms_per_sweep"] = legs
    # C4: nc = 32 on every knot (C = [I 0], the reference's generator), mu = 1e-11
    try:
        pc_ = synth.generate_lq_problem(np.random.default_rng(99), np.zeros(nx), N, nx, nu, nc=32,
                                        mode=args.generator)
        oc = ora.Problem.from_knots(pc_.stages, pc_.G0, pc_.g0, native=True)
        bc = ora.BatchSweep([oc])
        bc.problems = [oc] * threads
        bc.sweep_local(1e-11, threads, 1)
        _, sc = bc.sweep_local(1e-11, threads, 2)
        out["nc32_sweeps_per_s"] = threads * 2 / sc
    except Exception as e:  # the baseline never fails the bench line
        out["nc32_sweeps_per_s"] = None
        out["nc32_error"] = str(e)[:100]
    # C5 (round 6): the reference's OWN compiled code beside the port -- oracle/_ref/libgar_ref.so, its gar sources
    # compiled unchanged over the naive Eigen stand-in (oracle/ref_build.sh; prebuilt where /root/reference exists, it
    # travels to the GPU box) -- on the first problem above, one thread: ProximalRiccatiSolver backward + forward, and
    # ParallelRiccatiSolver with 4 threads.  An UPPER bound on the Eigen build's time (the stand-in's products are
    # plain loops), flagged as such; `kind` stays "port".
    try:
        from oracle import ref as oref
        if not os.path.exists(oref.PATH):
            raise RuntimeError("oracle/_ref/libgar_ref.so is not in the snapshot")
        rp = oref.Problem(first)
        rs = oref.ProximalRiccatiSolver(rp)
        best = 1e30
        for _ in range(3):
            t1 = time.perf_counter()
            rs.backward(mueq)
            rsol = rs.forward()
            best = min(best, time.perf_counter() - t1)
        # ... the same problem through the port, one thread, and the two solutions against each other
        op1 = ora.Problem.from_knots(first.stages, first.G0, first.g0, native=True)
        os1 = ora.ProximalRiccatiSolver(op1)
        os1.backward(mueq)
        psol = op1.initialize_solution()
        os1.forward(*psol)
        scale = max(1.0, max(float(np.abs(v).max()) for part in psol for v in part if v.size))
        dref = max(float(np.abs(a - b).max()) for A, B in zip(rsol, psol) for a, b in zip(A, B) if a.size) / scale
        rpar = oref.Problem(first)
        rpp = oref.ParallelRiccatiSolver(rpar, 4)
        bestp = 1e30
        for _ in range(3):
            t1 = time.perf_counter()
            rpp.backward(mueq)
            rpp.forward()
            bestp = min(bestp, time.perf_counter() - t1)
        out["reference_standin"] = {
            "what": "the reference's own ProximalRiccatiSolver / ParallelRiccatiSolver, sources compiled unchanged "
                    "(oracle/ref_build.sh) over the naive Eigen stand-in: an UPPER BOUND on the Eigen build's time "
                    "(naive products), same problem, this box's host",
            "serial_one_thread_ms_per_sweep": best * 1e3, "serial_sweeps_per_s_one_thread": 1.0 / best,
            "parallel_4_threads_ms_per_sweep": bestp * 1e3,
            "port_one_thread_ms_per_sweep": lat * 1e3,
            "max_rel_diff_reference_vs_port": dref}
    except Exception as e:  # the baseline never fails the bench line
        out["reference_standin"] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
    del L
    return out


def parallel_in_time(args, device, stream, nx, nu, mueq, N=2048, legs=256, reps=10):
    """Secondary figure (not `value`): ONE problem of the configs[3] shape swept serially and
    parallel in time (one wave per (problem, leg), leg-boundary system by block cyclic reduction)
    on this GPU; the two solutions are compared."""
    dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
    out, ref = {}, None
    for J in (1, legs):
        s = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=J, device=device)
        s.set_stream(stream.cuda_stream)
        synth_device.fill_problems(s, seed=4242, mode=args.generator, keep=())
        for _ in range(2):
            s.backward_async(mueq)
            s.forward_async()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            s.backward_async(mueq)
            s.forward_async()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        sol = s.solution(0)
        if J == 1:
            ref, out["serial_ms_per_sweep"], out["serial_kernel"] = sol, ms, s.kernel_name
        else:
            scale = max(1.0, max(float(np.abs(v).max()) for v in ref[3]))
            diff = max(float(np.abs(a - c).max()) for A, B in zip(sol, ref) for a, c in zip(A, B) if a.size)
            resid, steps = s.condensed_info(0)
            out.update({"legs": J, "ms_per_sweep": ms, "kernel": s.kernel_name,
                        "speedup_vs_serial": out["serial_ms_per_sweep"] / ms,
                        "max_rel_diff_vs_serial": diff / scale,
                        "condensed_residual": resid, "refinement_steps": steps})
    out["workload"] = f"one problem, N={N} nx={nx} nu={nu} fp64 (BASELINE.json configs[3] shape), one GPU"
    return out


def horizon_sharded(args, world, rank, device, dist, N=2048, legs=256, reps=10):
    """BASELINE.json configs[3]: ONE problem (N=2048, nx=36, nu=12) whose horizon is split into `legs`
    legs, the legs split evenly over the ranks (aligator_amd/sharded.py): leg sweep -> ONE RCCL
    all-gather of the 31.7 KB/leg boundary tuples -> redundant condensed solve -> leg roll-out, no
    host synchronisation inside a sweep.  ms per sweep = max over ranks; the all-gather is also timed
    on its own.  Rank 0 checks its own stages against a serial sweep of the same problem."""
    from aligator_amd.sharded import ShardedRiccatiSolver
    nx, nu, mueq = 36, 12, 1e-14
    dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
    s = ShardedRiccatiSolver(dims, nx, legs, batch=1, device=device)
    synth_device.fill_problems(s.impl, seed=4242, mode=args.generator, keep=())  # every rank: the same problem
    torch.cuda.synchronize()
    for _ in range(2):
        s.backward(mueq, check=False)
        s.forward(sync=False)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.backward(mueq, check=False)
        s.forward(sync=False)
    torch.cuda.synchronize()
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], device="cuda" if dist.get_backend() == "nccl" else "cpu",
                      dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ms = float(el.item()) / reps * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s.stream):
        e0.record()
        for _ in range(reps):
            s.exchange()
        e1.record()
    torch.cuda.synchronize()
    ag_ms = e0.elapsed_time(e1) / reps
    s.backward(mueq, check=True)   # the checked form: raises on every rank if any stage failed
    s.forward()
    resid, steps = s.impl.condensed_info(0)
    out = {"workload": f"one problem, N={N} nx={nx} nu={nu} fp64 (BASELINE.json configs[3]), horizon sharded",
           "ranks": world, "legs": legs, "legs_per_rank": s.legs_per_rank, "ms_per_sweep": ms,
           "all_gather_ms": ag_ms, "all_gather_bytes_per_rank": 8 * (3 * nx * nx + 2 * nx) * s.legs_per_rank,
           "condensed_residual": resid, "refinement_steps": steps, "kernel": s.impl.kernel_name,
           "host_syncs_per_sweep": 0}
    if rank == 0:
        ref = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=1, device=device)
        synth_device.fill_problems(ref, seed=4242, mode=args.generator, keep=())
        ref.backward(mueq)
        ref.forward()
        a, b = s.local_solution(0), ref.solution(0)
        lo, hi = s.stage_range
        scale = max(1.0, max(float(np.abs(v).max()) for v in b[3]))
        out["max_rel_diff_vs_serial_on_rank0_stages"] = max(
            float(np.abs(a[0][t] - b[0][t]).max()) for t in range(lo, hi)) / scale
    return out


def horizon_single_process(args, N=2048, legs=256, reps=10):
    """BASELINE.json configs[3] from the seam the reference calls: ONE process, ONE solver object
    (include/gar_hip.h, gar_hip_multi_create -- what `HipRiccatiSolver(problem, num_legs, devices)` holds behind
    `SolverProxDDP::linear_solver_`), its legs split over `--gpus` devices; the boundary exchange happens inside
    backward() (csrc/gar_multi.hpp: peer gather kernel or hipMemcpyPeerAsync, ordered by HIP events).  No torchrun,
    no torch.distributed.  --same-device: every sub-solver on cuda:0 (how a one-GPU box exercises the path)."""
    from aligator_amd import synth
    nx, nu, mueq = 36, 12, 1e-14
    W = args.gpus
    if not args.same_device and torch.cuda.device_count() < W:
        raise SystemExit(f"bench.py --single-process --gpus {W}: only {torch.cuda.device_count()} device(s) visible")
    devices = [0] * W if args.same_device else list(range(W))
    prob = synth.generate_lq_problem(4242, np.zeros(nx), N, nx, nu, mode=args.generator)
    dims = [k.dims for k in prob.stages]
    s = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs, devices=devices)
    s.upload([prob])
    for _ in range(2):
        s.backward_async(mueq)
        s.forward_async()
    s.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.backward_async(mueq)
        s.forward_async()
    s.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    assert s.backward(mueq) and s.forward()   # the checked form: raises if any stage on any device failed
    resid, steps = s.condensed_info(0)
    out = {"workload": f"one problem, N={N} nx={nx} nu={nu} fp64 (BASELINE.json configs[3]), horizon sharded, ONE process",
           "devices": devices, "legs": legs, "ms_per_sweep": ms,
           "exchange": s._L.gar_hip_multi_exchange_name(s.handle).decode() or "none (one device)",
           "boundary_bytes_read_per_device": 8 * (3 * nx * nx + 2 * nx) * legs,
           "condensed_residual": resid, "refinement_steps": steps, "kernel": s.kernel_name, "host_syncs_per_sweep": 0}
    ref = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=1, device=devices[0])
    ref.upload([prob])
    ref.backward(mueq)
    ref.forward()
    a, b = s.solution(0), ref.solution(0)
    scale = max(1.0, max(float(np.abs(v).max()) for v in b[3]))
    out["max_rel_diff_vs_serial"] = max(float(np.abs(x - y).max()) for A, B in zip(a, b) for x, y in zip(A, B) if x.size) / scale
    t0 = time.perf_counter()
    for _ in range(reps):
        ref.backward_async(mueq)
        ref.forward_async()
    ref.sync()
    out["serial_one_device_ms_per_sweep"] = (time.perf_counter() - t0) / reps * 1e3
    return out


def batch_scan(args, device, stream, N, nx, nu, mueq, main_batch, main_value, reps=5):
    """the headline workload at Bsz in {1, 256, 1024} (and `--batch`, the timed region itself): sweeps/s
    of the serial-in-time solver (num_legs = 1, what `value` is quoted on) and, for ONE problem, of the
    parallel-in-time solver with N/8 legs -- the configuration a caller with a single problem would pick."""
    dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
    out = {}
    for B in (1, 256, 1024):
        if B >= main_batch:
            continue
        row = {}
        for legs in ((1, max(2, N // 8)) if B == 1 else (1,)):
            s = BatchedRiccatiSolver(dims, nx, batch=B, num_legs=legs, device=device)
            s.set_stream(stream.cuda_stream)
            synth_device.fill_problems(s, seed=777, mode=args.generator, keep=())
            for _ in range(2):
                s.backward_async(mueq)
                s.forward_async()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                s.backward_async(mueq)
                s.forward_async()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            key = "serial" if legs == 1 else f"legs{legs}"
            row[key] = {"sweeps_per_s": B / dt, "ms_per_step": dt * 1e3, "kernel": s.kernel_name}
            s.close()
        out[str(B)] = row
    out[str(main_batch)] = {"serial": {"sweeps_per_s": main_value, "note": "the timed region of this line"}}
    return out


def config3(args, device, stream, reps=10):
    """BASELINE.json configs[2] / SURVEY 8(d) "Config 3": ONE problem, N = 1024, nx = 12, nu = 6 (padded inside the C
    ABI onto the (12, 8) family), ParallelRiccatiSolver semantics with legs in {2, ..., 64}, mu = 1e-9, up to 10
    refinement steps (tests/gar/parallel.cpp:185-245), beside the serial-in-time solver; every configuration's
    solution against the serial ORACLE's (relative to the largest multiplier) and the reference's residual
    lqrComputeKktError, the reference's bar being 1e-7 (parallel.cpp:221, 234-235)."""
    from aligator_amd import synth
    from aligator_amd.gar import lqrComputeKktError
    from oracle import oracle as ora
    N, nx, nu, mueq = 1024, 12, 6, 1e-9
    prob = synth.generate_lq_problem(33, np.zeros(nx), N, nx, nu, mode="W")
    dims = [k.dims for k in prob.stages]
    osol = ora.ProximalRiccatiSolver(ora.Problem.from_knots(prob.stages, prob.G0, prob.g0))
    osol.backward(mueq)
    from aligator_amd.gar import lqrInitializeSolution
    ref = lqrInitializeSolution(prob)
    osol.forward(*ref)
    scale = max(1.0, max(float(np.abs(v).max()) for part in ref for v in part if v.size))
    out = {"workload": f"one problem, N={N} nx={nx} nu={nu} fp64, mu={mueq} (BASELINE.json configs[2])", "legs": {}}
    for legs in (1, 2, 4, 8, 16, 32, 64):
        s = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs, device=device)
        s.set_stream(stream.cuda_stream)
        if legs > 1:
            s.set_refinement(1e-10, 10)
        s.upload([prob])
        for _ in range(2):
            s.backward_async(mueq)
            s.forward_async()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            s.backward_async(mueq)
            s.forward_async()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        sol = s.solution(0)
        err = max(float(np.abs(a - b).max()) for A, B in zip(sol, ref) for a, b in zip(A, B) if a.size) / scale
        kkt = max(lqrComputeKktError(prob, *sol, mueq=mueq)) / scale
        row = {"ms_per_sweep": ms, "kernel": s.kernel_name, "max_rel_err_vs_serial_oracle": err, "max_kkt_rel": kkt,
               "failed_factorisations": s.num_failed()}
        if legs > 1:
            resid, steps = s.condensed_info(0)
            row.update({"condensed_solver": s.condensed_solver_name, "condensed_residual": resid, "refinement_steps": steps})
            out["legs"][str(legs)] = row
        else:
            out["serial"] = row
        s.close()
    best = min(out["legs"], key=lambda k: out["legs"][k]["ms_per_sweep"])
    out["best_legs"] = int(best)
    out["speedup_vs_serial"] = out["serial"]["ms_per_sweep"] / out["legs"][best]["ms_per_sweep"]
    return out


def seam():
    """One Newton iteration of bench/lqr.cpp's ProxDDP loop through the RiccatiSolverBase seam, phase by phase --
    tests/cpp/bench_lqr_loop.cpp built without its oracle leg (tests/cpp/_build/seam_bench, made by
    __graft_entry__.build()): the call sequence of include/aligator/gar/hip-riccati.hpp over the C ABI, (36, 12) and
    bench/lqr.cpp's / the Talos walk's (56, 22), N = 256, serial and in leg mode with the leg count the library's measured
    table suggests (gar_hip_suggest_num_legs: 64 / 32), best of three runs of 20 iterations.  None when the binary is absent."""
    import subprocess
    # (bench_lqr_loop: the same program WITH its oracle leg -- the restated reference's iteration on this box's host,
    # one thread, beside every GPU figure; seam_bench is the build without it)
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "bench_lqr_loop")
    if not os.path.exists(exe):
        exe = os.path.join(ROOT, "tests", "cpp", "_build", "seam_bench")
    if not os.path.exists(exe):
        return {"skipped": "tests/cpp/_build/seam_bench not built"}
    try:
        r = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=240)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001  (a secondary figure must not take the line down)
        return {"skipped": f"{type(e).__name__}: {e}"}


def proxddp_loop():
    """BASELINE configs[4] with the loop the reference itself runs: its OWN SolverProxDDP (compiled unchanged from
    /root/reference over the Eigen stand-in, oracle/ref_ddp_build.sh) on tests/lqr.cpp's case and on bench/lqr.cpp's
    (dim 56, nu 22: the Talos-walk LQ shape; Talos itself needs Pinocchio + example-robot-data, absent), with
    `linear_solver_` = the reference's SERIAL / PARALLEL solvers and = the shipped HipRiccatiSolver, same box, same
    process: convergence, iterations, wall clock per run, ProxDDP iterations per second
    (tests/integration/proxddp_lqr_driver.cpp; the executable is built where /root/reference exists and travels)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "proxddp_lqr_gpu")
    if not os.path.exists(exe):
        return {"skipped": "oracle/_ref/proxddp_lqr_gpu not built (__graft_entry__.build() where /root/reference exists)"}
    try:
        r = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=300)
        # (the reference's logger colours its warnings: the JSON object may sit behind an ANSI reset on its line)
        line = [ln for ln in r.stdout.splitlines() if '{"what"' in ln][-1]
        return json.loads(line[line.index('{"what"'):])
    except Exception as e:   # noqa: BLE001
        return {"skipped": f"{type(e).__name__}: {e}"}


def secondary_shapes(device, batch=1024):
    """Two more shapes of the same hot path, outside the timed region (rank 0, one GPU), so that the round's
    bench record carries them: the reference's OWN gar benchmark shape (bench/gar-riccati.cpp:19-22: nx=36,
    nu=12, nc=32 equality constraints on every knot, mu=1e-11, generator with D=0) and the Talos-walk LQ shape
    of BASELINE.json configs[4] (bench/talos-walk.cpp:20-28: nx=56, nu=22, N=275).  sweeps/s and the backward
    kernels' fraction of the HBM roofline on each shape's own algorithmic bytes."""
    import ctypes as C
    from aligator_amd import synth
    from aligator_amd.gar import BatchedRiccatiSolver
    out = {}
    for key, (nx, nu, nc, N, mu, what) in {
            "reference_bench_shape_nc32": (36, 12, 32, 256, 1e-11, "bench/gar-riccati.cpp: nx=36 nu=12 nc=32 N=256, D=0 (its generator)"),
            # the reference's GENERAL constrained stage (riccati-kernel.hxx:224-277): the same shape with a random D
            # on every knot -- the 44 x 44 reduced KKT matrix [Rhat D^T; D -mu I] is then really coupled
            "reference_bench_shape_nc32_coupled": (36, 12, 32, 256, 1e-11, "bench/gar-riccati.cpp's shape with D != 0 (U[-1,1]) on every knot: "
                                                   "the coupled reduced KKT stage, nx=36 nu=12 nc=32 N=256"),
            "talos_walk_lq_shape": (56, 22, 0, 275, 1e-10, "bench/talos-walk.cpp LQ sub-problem shape: nx=56 nu=22 N=275")}.items():
        # every problem of the batch is its own draw, generated on the device like the headline's (round 6: the two
        # host problems replicated 512 times that stood here were served from the last-level cache, and their
        # per-problem uploads made rocprofv3 counter passes over this code impractical)
        dims = [(nx, nu, nc, nx, 0)] * N + [(nx, 0, nc, nx, 0)]
        s = BatchedRiccatiSolver(dims, nx, batch=batch, device=device)
        synth_device.fill_problems(s, seed=100 + 7 * nc + nx, mode="W", coupled=key.endswith("_coupled"))
        torch.cuda.synchronize()
        s.backward(mu); s.forward(); s.sync()
        failed = s.num_failed()
        s._check(s._L.gar_hip_set_timing(s.handle, 1))
        reps, kb = 3, 0.0
        t0 = time.perf_counter()
        for _ in range(reps):
            s.backward_async(mu); s.forward_async()
            o = (C.c_double * 3)()
            s._check(s._L.gar_hip_last_kernel_ms(s.handle, o))
            kb += o[0]
        s.sync()
        dt = (time.perf_counter() - t0) / reps
        knot = 2 * nx * nx + 2 * nx * nu + nu * nu + 2 * nx + nu + nc * (nx + nu + 1)
        fac = (nu + nc + nx) * (nx + 1) + nx * nx + nx
        # parity of what was just timed, in the same run: two problems of the batch (the first and the last) against
        # the oracle -- solution, relative to the largest multiplier (O(1/mu) on the constrained shape), and the
        # reference's own residual lqrComputeKktError (gar/utils.hxx:88-182) relative to the same scale
        from aligator_amd.gar import lqrComputeKktError
        from oracle import oracle as ora
        err = kkt = 0.0
        for b in (0, batch - 1):
            prob = synth_device.download_problem(s, b)
            sol = s.solution(b)
            op = ora.Problem.from_knots(prob.stages, prob.G0, prob.g0)
            osol = ora.ProximalRiccatiSolver(op)
            osol.backward(mu)
            ref = op.initialize_solution()
            osol.forward(*ref)
            scale = max(1.0, max(float(np.abs(v).max()) for part in ref for v in part if v.size))
            err = max(err, max(float(np.abs(a - c).max()) for A, B in zip(sol, ref) for a, c in zip(A, B) if a.size) / scale)
            kkt = max(kkt, max(lqrComputeKktError(prob, *sol, mueq=mu)) / scale)
        chain = s.constrained_bk_stages() if nc > 0 else None
        out[key] = {"workload": what, "batch": batch, "kernel": s.kernel_name, "sweeps_per_s": batch / dt,
                    "backward_ms": kb / reps, "failed_factorisations": failed,
                    **({"stages_on_the_coupled_kernel_and_on_lds_bunch_kaufman": [int(v) for v in chain]} if chain is not None else {}),
                    "backward_frac_of_hbm_roofline": 8 * (knot + fac) * N * batch / (kb / reps * 1e-3) / HBM_PEAK,
                    "max_rel_err_vs_oracle": err, "max_kkt_rel": kkt}
        s.close()
        if nc > 0:
            # ... and the way the reference itself benchmarks this shape: ONE problem, ParallelRiccatiSolver (BM_parallel,
            # bench/gar-riccati.cpp:64-90): D = 0 through the fold onto the wave-leg kernels (csrc/gar_fold.hpp); D != 0 on the
            # constrained segment legs (csrc/gar_cstr_seg.hpp, round 6), beside the any-dimension leg kernels that served such
            # problems before (GAR_HIP_CSTR_SEG_LEGS=0) and the serial chain; x, u against the serial oracle in the same run
            lat, names = {}, {}
            for legs, seg in ((1, "1"), (32, "1")) + (((32, "0"),) if key.endswith("_coupled") else ()):
                os.environ["GAR_HIP_CSTR_SEG_LEGS"] = seg
                s1 = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, device=device)
                os.environ.pop("GAR_HIP_CSTR_SEG_LEGS")
                s1.upload([prob])
                for _ in range(2):
                    s1.backward_async(mu); s1.forward_async()
                s1.sync()
                t0 = time.perf_counter()
                for _ in range(10):
                    s1.backward_async(mu); s1.forward_async()
                s1.sync()
                lat[(legs, seg)] = (time.perf_counter() - t0) / 10 * 1e3
                names[(legs, seg)] = s1.kernel_name
                if legs > 1 and seg == "1":
                    sol = s1.solution(0)
                    xu = max(float(np.abs(a - c).max()) for A, B in zip(sol[:2], ref[:2]) for a, c in zip(A, B) if a.size)
                    xu /= max(1.0, max(float(np.abs(v).max()) for v in ref[0]))
                    out[key]["parallel_mode_one_problem"] = {
                        "legs": legs, "kernel": s1.kernel_name, "condensed_solver": s1.condensed_solver_name,
                        "ms_per_sweep": lat[(legs, seg)], "serial_ms_per_sweep": lat[(1, "1")],
                        "max_rel_err_x_u_vs_serial_oracle": xu,
                        "note": "multipliers are of order 1/mu at mu = 1e-11: x, u are what a leg-parallel solve of this problem "
                                "pins (DESIGN.md 2, conditioning bound; tests/test_gpu_parity.py checks v, lambda against it)"}
                s1.close()
            if (32, "0") in lat:
                out[key]["parallel_mode_one_problem"]["ms_per_sweep_on_the_any_dimension_leg_kernels"] = lat[(32, "0")]
        if nc == 0:
            # ... and the way the reference itself benchmarks this shape: ONE problem, LQSolverChoice::PARALLEL
            # (bench/talos-walk.cpp:102-127, bench/lqr.cpp:112-134) -- latency of one backward + forward sweep in leg
            # mode against the serial kernel on the same problem, the leg-mode solution against the serial oracle's
            # (tests/gar/parallel.cpp:211-235) in the same run
            # (`prob`, `ref` and `scale` are those of the batch's last problem)
            lat = {}
            for legs in (1, 34):
                s1 = BatchedRiccatiSolver([k.dims for k in prob.stages], nx, batch=1, num_legs=legs, device=device)
                s1.upload([prob])
                for _ in range(2):
                    s1.backward_async(mu); s1.forward_async()
                s1.sync()
                t0 = time.perf_counter()
                for _ in range(10):
                    s1.backward_async(mu); s1.forward_async()
                s1.sync()
                lat[legs] = (time.perf_counter() - t0) / 10 * 1e3
                if legs > 1:
                    sol = s1.solution(0)
                    perr = max(float(np.abs(a - c).max()) for A, B in zip(sol, ref) for a, c in zip(A, B) if a.size) / scale
                    out[key]["parallel_mode_one_problem"] = {
                        "legs": legs, "kernel": s1.kernel_name, "condensed_solver": s1.condensed_solver_name,
                        "condensed_redone_by_the_chain": bool(s1.condensed_resolved(0)),
                        "ms_per_sweep": lat[legs], "serial_ms_per_sweep": lat[1],
                        "max_rel_err_vs_serial_oracle": perr,
                        "max_kkt_rel": max(lqrComputeKktError(prob, *sol, mueq=mu)) / scale}
                s1.close()
    return out


def stream_ceiling(solver, device, batch, N, nx, nu, bwd_bytes, bwd_ms):
    """bwd_bytes: the bytes the backward sweep MOVES per problem (moved_bytes), which is what these kernels move."""
    knot_b = 8 * knot_doubles_read(nx, nu, solver.qr_packed)
    fac_b = 8 * ((nu + nx) * (nx + 1) + nx * (nx + 1) // 2 + nx)
    if knot_b > 32 * 1024 or fac_b > 28 * 1024:
        return None
    ms = solver._L.gar_hip_stream_ceiling_ms(int(device), int(batch), int(N), knot_b, fac_b, 3)
    if ms <= 0:
        return None
    out = {"ms": ms, "GBps": bwd_bytes * batch / (ms * 1e-3) / 1e9, "frac_of_peak": bwd_bytes * batch / (ms * 1e-3) / HBM_PEAK,
           "kernel_over_stream": bwd_ms / ms,
           "note": "a kernel that only moves the bytes the backward sweep moves (same waves, same walk; packed Vxx, packed lower Q / R), "
                   "measured in this run"}
    # ... and a PLAIN grid-stride 16 B/lane copy of the same number of bytes (four nontemporal loads in flight per
    # lane, 64 workgroups per CU: the best of scripts/ubench/copy_variants.cpp), in the same process: what this
    # box's HBM gives the kernel the guide's 6.3 TB/s describes.  The gap between the two is what the sweep's walk
    # (4 096 sequential streams, one record in flight per wave, 60:40 read:write) costs.
    ms2 = solver._L.gar_hip_stream_ceiling_ms(int(device), int(batch), int(N), knot_b, fac_b, -3)
    if ms2 > 0:   # the same walk with two knots requested ahead: what a deeper prefetch could buy the sweep
        out["two_ahead"] = {"ms": ms2, "GBps": bwd_bytes * batch / (ms2 * 1e-3) / 1e9, "kernel_over_stream": bwd_ms / ms2}
    cms = solver._L.gar_hip_copy_ceiling_ms(int(device), int(bwd_bytes * batch), 3)
    if cms > 0:
        out["plain_copy"] = {"ms": cms, "GBps": bwd_bytes * batch / (cms * 1e-3) / 1e9,
                             "frac_of_peak": bwd_bytes * batch / (cms * 1e-3) / HBM_PEAK,
                             "kernel_over_copy": bwd_ms / cms, "stream_over_copy": ms / cms}
    return out


def parity_check(solver, args, mueq, nsample=8):
    """Spot-check the timed data: pull problems spread over the batch back, solve with the oracle."""
    from aligator_amd.gar import lqrComputeKktError
    from oracle import oracle as ora

    worst, worst_kkt = 0.0, 0.0
    for b in np.linspace(0, solver.batch - 1, nsample).astype(int):
        prob = synth_device.download_problem(solver, int(b))
        sol = solver.solution(int(b))
        op = ora.Problem.from_knots(prob.stages, prob.G0, prob.g0)
        os_ = ora.ProximalRiccatiSolver(op)
        os_.backward(mueq)
        ref = op.initialize_solution()
        os_.forward(*ref)
        scale = max(1.0, max(float(np.abs(v).max()) for v in ref[3]))
        for A, B in zip(sol, ref):
            for a, c in zip(A, B):
                if a.size:
                    worst = max(worst, float(np.abs(a - c).max()) / scale)
        worst_kkt = max(worst_kkt, max(lqrComputeKktError(prob, *sol, mueq=mueq)))
    return worst, worst_kkt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="problems per GPU (59 GB of the 288 GB at the default)")
    ap.add_argument("--horizon", type=int, default=256)
    ap.add_argument("--nx", type=int, default=36)
    ap.add_argument("--nu", type=int, default=12)
    ap.add_argument("--generator", default="W", choices=["W", "F"])
    ap.add_argument("--mode", default="batch", choices=["batch", "horizon"],
                    help="batch: the headline (independent problems per GPU, weak scaling); horizon: ONE "
                         "problem's horizon sharded over the ranks (configs[3]) -- also reported as "
                         "`horizon_sharded` inside the batch-mode line whenever there is more than one rank")
    ap.add_argument("--single-generator", action="store_true",
                    help="skip the second measurement on the other generator (value_F / value_W)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary parallel-in-time figure")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary shapes (constrained, Talos)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the timing barrier (gloo + --same-device lets two "
                         "ranks share one GPU to exercise the N>1 path on a 1-GPU box)")
    ap.add_argument("--same-device", action="store_true", help="testing: every rank uses cuda:0")
    ap.add_argument("--single-process", action="store_true",
                    help="--mode horizon without torchrun: ONE process, ONE solver object whose legs are split over "
                         "--gpus devices (gar_hip_multi_create), the boundary exchange inside the library")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"],
                    help="auto: roofline.traffic from rocprofv3 counter passes run inside this bench (rank 0, one GPU) "
                         "when rocprofv3 is on PATH, the committed profiles/pmc_traffic.json otherwise; off: the file")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "0", "2"],
                    help="the schedule of the timed steps: 0 = backward then forward sweep of the whole batch on one "
                         "stream; 2 = gar_hip_set_pipeline(2), the forward sweep of one half of the batch beside the "
                         "backward sweep of the other half (same results, bit for bit); auto: `value` is timed in the "
                         "schedule the LIBRARY chooses by itself (gar_hip_set_pipeline(-1), what a new solver starts "
                         "with: a caller's default) -- the other schedule is timed beside it (K steps, same data) and "
                         "both figures are in the line")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.pmc_child:
        return pmc_child(args)
    if args.single_process:
        if args.mode != "horizon":
            raise SystemExit("--single-process belongs to --mode horizon (the batch axis needs no exchange)")
        if int(os.environ.get("RANK", "0")) != 0:
            return  # launched under torchrun all the same: rank 0 drives every device, the others have nothing to do
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the backend has no CPU path")
        hs = horizon_single_process(args)
        big = horizon_single_process(args, N=16384, legs=512, reps=5)
        print(json.dumps({
            "metric": "Riccati sweeps/sec (bwd+fwd), ONE problem N=2048 nx=36 nu=12, horizon sharded",
            "value": 1e3 / hs["ms_per_sweep"], "unit": "sweeps/s", "n_gpus": args.gpus, "steps": 10, "warmup": 2,
            "ms_per_step": hs["ms_per_sweep"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": f"synthetic (generator {args.generator})",
            "config": {"workload": hs["workload"],
                       "parallelism": f"horizon-sharded x{args.gpus} inside one process, one in-library boundary gather per sweep"},
            "horizon_sharded": hs, "horizon_sharded_N16384": big}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and "WORLD_SIZE" in os.environ and int(os.environ.get("RANK", "0")) == 0 and args.gpus != 1:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the backend has no CPU path")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.mode == "horizon":
        import torch.distributed as dist
        if world == 1 and "MASTER_ADDR" not in os.environ:  # --mode horizon on one GPU: a 1-rank group
            # (an in-process store: a one-rank group needs no rendezvous socket, so no port can be in use)
            dist.init_process_group("nccl", store=dist.HashStore(), world_size=1, rank=0,
                                    device_id=torch.device("cuda", local_rank))
        elif args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    if args.mode == "horizon":
        hs = horizon_sharded(args, world, rank, local_rank, dist)
        big = horizon_sharded(args, world, rank, local_rank, dist, N=16384, legs=512, reps=5)
        if rank == 0:
            print(json.dumps({
                "metric": "Riccati sweeps/sec (bwd+fwd), ONE problem N=2048 nx=36 nu=12, horizon sharded",
                "value": 1e3 / hs["ms_per_sweep"], "unit": "sweeps/s", "n_gpus": world, "steps": 10, "warmup": 2,
                "ms_per_step": hs["ms_per_sweep"], "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f64", "data": f"synthetic (generator {args.generator})",
                "config": {"workload": hs["workload"], "parallelism": f"horizon-sharded x{world}, one RCCL all-gather per sweep"},
                "horizon_sharded": hs, "horizon_sharded_N16384": big}))
        dist.destroy_process_group()
        return

    # The Newton-iteration seam (host data, one problem) is measured FIRST, before this process allocates anything
    # large: for seconds after a process has freed tens of GB of HBM the driver's scrubbing of that memory keeps the
    # copy engines busy and every host<->device copy of the seam runs several times slower (measured: 0.92 -> 2.0 ms
    # and 2.3 -> 4.9 ms right after a 57 GB process, back to 0.92 / 2.3 twenty seconds later).
    seam_line = seam() if (world == 1 and rank == 0 and not args.no_extras) else None
    ddp_line = proxddp_loop() if (world == 1 and rank == 0 and not args.no_extras) else None
    N, nx, nu, mueq = args.horizon, args.nx, args.nu, 1e-14
    dims = [(nx, nu, 0, nx, 0)] * N + [(nx, 0, 0, nx, 0)]
    solver = BatchedRiccatiSolver(dims, nx, batch=args.batch, num_legs=1, device=local_rank)
    # a dedicated (non-null) stream: the kernels, the HIP events and the timing all
    # live on it (gar_hip_set_stream(NULL) would select the solver's private stream)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    solver.set_stream(stream.cuda_stream)

    def step():
        solver.backward_async(mueq)
        solver.forward_async()

    def kernel_ms():
        """per-kernel durations: HIP events the library records on the launch stream(s) around the backward sweep
        kernel, the initial-stage kernel and the forward sweep kernel, averaged over a few extra untimed steps"""
        solver._check(solver._L.gar_hip_set_timing(solver.handle, 1))
        kms = np.zeros(3)
        reps = max(3, min(args.steps, 10))
        for _ in range(reps):
            step()
            out3 = (C.c_double * 3)()
            solver._check(solver._L.gar_hip_last_kernel_ms(solver.handle, out3))
            kms += np.array(list(out3))
        solver._check(solver._L.gar_hip_set_timing(solver.handle, 0))
        return kms / reps

    def timed(pipe):
        """W untimed warm-up steps, then EXACTLY `steps` steps between barrier + synchronize on both sides (max over
        ranks), in the given schedule"""
        solver.set_pipeline(pipe)
        for _ in range(args.warmup):
            step()
        solver.sync()
        torch.cuda.synchronize()
        if solver.num_failed() != 0:
            raise SystemExit("a stage factorisation failed during warm-up")
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            solver.backward_async(mueq)
            solver.forward_async()
        solver.sync()   # (pipelined: orders the timing stream behind the half streams; a host wait either way)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def measure(generator, seed):
        """the timed steps in the plain schedule and -- where the solver has it -- in the pipelined one, on the same
        data; then the per-kernel durations of each and the slow-path counters of one backward"""
        synth_device.fill_problems(solver, seed=seed, mode=generator)
        torch.cuda.synchronize()
        res = {}
        for pipe in ((0, 2) if args.pipeline == "auto" else (int(args.pipeline),)):
            if pipe == 2:
                try:
                    solver.set_pipeline(2)
                except RuntimeError as e:   # another kernel family (small batch, other shape): plain only
                    res["pipelined_unavailable"] = str(e)[:160]
                    continue
            elapsed = timed(pipe)
            res[pipe] = {"elapsed": elapsed, "kms": kernel_ms()}
        solver.set_pipeline(0)
        step()
        slow, pivoted = solver.slow_path_stages()
        res["slow"], res["pivoted"], res["failed"] = slow / (args.batch * N), pivoted / (args.batch * N), solver.num_failed()
        return res

    # the reference's own generator F first (secondary figure), then the headline generator: the
    # parity spot check and everything below run on the data of the headline measurement
    other = "F" if args.generator == "W" else "W"
    second = None if args.single_generator else measure(other, 4321 + 7919 * rank)
    main_res = measure(args.generator, 1234 + 7919 * rank)

    # `value` is the schedule a caller gets WITHOUT asking: the library's own choice for this batch on this device
    solver.set_pipeline(-1)
    library_default = solver.pipeline
    solver.set_pipeline(0)

    def best_of(res):
        """(historical name) the schedule `value` is quoted in: the library's default when it was timed"""
        sched = library_default if library_default in res else min((p for p in (0, 2) if p in res), key=lambda p: res[p]["elapsed"])
        return sched, res[sched]["elapsed"]
    sched, elapsed = best_of(main_res)
    # the roofline's kernel figures are those of the PLAIN schedule whenever it ran (full-batch launches, each kernel
    # alone on the chip); the pipelined schedule's per-half-batch launches are reported beside them
    kref = main_res[0] if 0 in main_res else main_res[2]
    launch_batch = args.batch if 0 in main_res else (args.batch + 1) // 2
    bwd_ms, init_ms, fwd_ms = (float(v) for v in kref["kms"])
    slow_frac, piv_frac, failed = main_res["slow"], main_res["pivoted"], main_res["failed"]

    err, kkt = parity_check(solver, args, mueq)
    pit = None
    if world == 1 and not args.no_legs and (nx, nu) == (36, 12):
        pit = parallel_in_time(args, local_rank, stream, nx, nu, mueq)
    hs = None
    if world > 1 and not args.no_legs and (nx, nu) == (36, 12):
        # secondary figure (configs[3]); it must never cost the headline line: every rank runs the same code, so an
        # exception is raised (and caught) on every rank alike
        try:
            hs = horizon_sharded(args, world, rank, local_rank, dist)
        except Exception as e:  # noqa: BLE001
            hs = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    traffic, traffic_src = pmc_traffic("backward", args.batch), None
    if rank == 0:
        traffic_src = ("committed rocprofv3 --pmc passes of this kernel at this batch (profiles/pmc_traffic.json, "
                       "scripts/gpu_r5_evidence.sh), NOT collected in this run; null when the batch differs")
        if world == 1 and args.pmc == "auto":
            solver.sync()
            live, detail = pmc_traffic_in_run(args, N, nx, nu)
            if live is not None:
                traffic, traffic_src = live, {"collected": "in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                                              "--kernel-trace only) over a child run of the same solver, batch and data; units "
                                              "calibrated in the same pass on gar_stream_sweep's known byte counts", **detail}
            else:
                traffic_src += f" [in-run collection unavailable: {detail}]"
    if rank == 0:
        sweeps = args.batch * world * args.steps
        bwd_b, fwd_b = algorithmic_bytes(N, nx, nu)
        bwd_mv, fwd_mv = moved_bytes(N, nx, nu, solver.qr_packed)
        achieved = bwd_b * launch_batch / (bwd_ms * 1e-3)

        def sched_obj(res):
            """both schedules of the timed steps, side by side (same data, K steps each)"""
            o = {}
            for p_, name in ((0, "plain"), (2, "pipelined")):
                if p_ in res:
                    k = res[p_]["kms"]
                    o[name] = {"value": sweeps / res[p_]["elapsed"], "ms_per_step": res[p_]["elapsed"] / args.steps * 1e3,
                               "sweep_frac_of_hbm_roofline": (sweeps / res[p_]["elapsed"]) * (bwd_b + fwd_b) / HBM_PEAK,
                               "kernel_ms_per_launch": {"backward_sweep": float(k[0]), "forward_sweep": float(k[2])},
                               "problems_per_launch": args.batch if p_ == 0 else (args.batch + 1) // 2}
            if "pipelined_unavailable" in res:
                o["pipelined_unavailable"] = res["pipelined_unavailable"]
            if "pipelined" in o:
                o["pipelined"]["note"] = ("gar_hip_set_pipeline(2): the forward sweep of one half of the batch (gar_forward_lean: "
                                          "LDS-DMA, 66 registers) resident on every SIMD beside the backward wave of the other half; "
                                          "per-launch durations are those of HALF-batch launches running together")
            return o
        out = {
            "metric": "Riccati sweeps/sec (bwd+fwd), N=256 nx=36 nu=12",
            "value": sweeps / elapsed,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": f"synthetic (generator {args.generator}, generated on device)",
            "config": {"workload": f"batched serial-in-time Riccati N={N} nx={nx} nu={nu} nc=0 fp64 "
                                   f"(BASELINE.json configs[1])",
                       "batch_per_gpu": args.batch, "kernel": solver.kernel_name,
                       "schedule": ("pipelined (two half-batches, forward beside backward)" if sched == 2 else "plain") +
                                   (" = the library's default for this batch (gar_hip_set_pipeline(-1))" if sched == library_default
                                    else " (forced by --pipeline; the library's default is the other one)"),
                       "parallelism": f"batch-sharded x{world} (no data-path collective)"},
            "schedules": sched_obj(main_res),
            "kernel_ms": {"backward_sweep": bwd_ms, "initial_stage": init_ms, "forward_sweep": fwd_ms,
                          "problems_per_launch": launch_batch},
            "roofline": {"bound": "hbm", "kernel": f"gar_backward_{solver.kernel_name}",
                         "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK,
                         "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bwd_b * launch_batch,
                         # `achieved` is the contract's figure: SURVEY 8(d)'s algorithmic bytes (the reference's records,
                         # full symmetric Vxx) over the kernel's time.  The kernel itself stores Vxx as its packed
                         # lower triangle and so MOVES fewer bytes: this is the bandwidth it actually draws
                         "moved_bytes_per_launch": bwd_mv * launch_batch,
                         "moved_GBps": bwd_mv * launch_batch / (bwd_ms * 1e-3) / 1e9,
                         "moved_frac_of_peak": bwd_mv * launch_batch / (bwd_ms * 1e-3) / HBM_PEAK,
                         # what this box's HBM sustains for the backward sweep's bytes alone: the same number of
                         # one-wave-per-problem streams, per stage the knot read (one ahead in flight) and the
                         # factor record written, no arithmetic (gar_hip_stream_ceiling_ms, csrc/gar_generic.hpp)
                         "stream_ceiling": stream_ceiling(solver, local_rank, launch_batch, N, nx, nu, bwd_mv, bwd_ms),
                         "forward_GBps": fwd_b * launch_batch / (fwd_ms * 1e-3) / 1e9,
                         "forward_moved_GBps": fwd_mv * launch_batch / (fwd_ms * 1e-3) / 1e9,
                         "sweep_frac_of_hbm_roofline": (sweeps / elapsed) * (bwd_b + fwd_b) / HBM_PEAK},
            "parity": {"max_rel_err_vs_oracle": err, "max_kkt": kkt, "failed_factorisations": failed},
            # stages (fraction of batch x N) whose Rhat failed the first Bunch-Kaufman test and left the
            # register LDL^T for the out-of-line path; "pivoted": those where Bunch-Kaufman interchanged
            "slow_path_stage_frac": {args.generator: slow_frac},
            "pivoted_stage_frac": {args.generator: piv_frac},
        }
        if second is not None:
            sched2, el2 = best_of(second)
            k2 = second[0] if 0 in second else second[2]
            kms2 = k2["kms"]
            out[f"value_{other}"] = sweeps / el2
            out[f"ms_per_step_{other}"] = el2 / args.steps * 1e3
            out[f"schedules_{other}"] = sched_obj(second)
            out[f"kernel_ms_{other}"] = {"backward_sweep": float(kms2[0]), "initial_stage": float(kms2[1]),
                                         "forward_sweep": float(kms2[2])}
            out[f"roofline_frac_{other}"] = bwd_b * launch_batch / (float(kms2[0]) * 1e-3) / HBM_PEAK
            out["slow_path_stage_frac"][other] = second["slow"]
            out["pivoted_stage_frac"][other] = second["pivoted"]
        if pit is not None:
            out["parallel_in_time"] = pit
        if hs is not None:
            out["horizon_sharded"] = hs
        if world == 1 and not args.no_extras:
            out["secondary_shapes"] = secondary_shapes(local_rank)
            out["batch_scan"] = batch_scan(args, local_rank, stream, N, nx, nu, mueq, args.batch, sweeps / elapsed)
            out["config3"] = config3(args, local_rank, stream)
            out["seam"] = seam_line
            out["proxddp_loop"] = ddp_line
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args, nx, nu, N, mueq)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
