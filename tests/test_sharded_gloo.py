"""The N > 1 path on CPU: world_size = 2 and 3, backend gloo, 127.0.0.1.

Two (three) processes run the REAL sharded flow of aligator_amd.sharded
(gar_hip_solver_create_sharded -> backward_legs -> all_gather of the boundary
tuples -> condensed solve -> forward_legs) with the kernel sources executing on
the wave emulator (tests/emu: test-only build, host memory), and bench.py's batch
sharding (independent problems per rank, no data-path collective).  Results are
checked against the serial CPU oracle and the golden fixture.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU = os.path.join(HERE, "emu", "_build", "libgar_hip_emu.so")

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
emu, mode = sys.argv[2], sys.argv[3]
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
from aligator_amd.lqr import lqrComputeKktError
import parity_cases as pc
if mode == "horizon":
    from aligator_amd.sharded import ShardedRiccatiSolver
    from test_golden import load_fixture, assert_matches
    prob, mueq, _, gold = load_fixture(os.path.join(sys.argv[1], "tests", "golden", "parallel_shape_nx8_N17.npz"))
    for legs in (2, 3, 4, 5, 6):          # odd leg counts: an UNEVEN split over the ranks (1 + 2, 2 + 3; 1 + 2 + 2, ...)
        if legs < world:                   # 1 <= W <= J (gar_hip_solver_create_ranked)
            continue
        s = ShardedRiccatiSolver([k.dims for k in prob.stages], prob.nc0, legs, batch=1,
                                 lib_path=emu, on_device=False)
        s.impl.upload([prob])
        s.backward(mueq)
        s.forward()
        sol = s.gather_solution(0)
        assert_matches(sol, gold, 1e-8)
        assert max(lqrComputeKktError(prob, *sol, mueq=mueq)) <= 1e-8
        # this rank really only computed its own stages: legs [r J / W, (r+1) J / W) of get_work(17, ., J)
        lo, hi = s.stage_range
        l0, l1 = rank * legs // world, (rank + 1) * legs // world
        assert s.leg_range == (l0, l1), (s.leg_range, l0, l1)
        assert (lo, hi) == (l0 * 18 // legs, l1 * 18 // legs), (lo, hi, l0, l1)
        # datas[t].kktMat of a sharded solver is formed from THIS sweep's mueq (ADVICE r2)
        t_own = lo
        f = s.impl.factor(t_own, 0)
        if f.nu > 0:
            assert np.abs(np.diag(f.kktMat)).max() > 0.0
    # coupled constraints (D != 0) over the ranks: every rank runs the constrained segment legs (gar_cstr_seg.hpp) on ITS
    # legs -- leg_begin > 0 on all but rank 0 -- against the serial oracle
    crng = np.random.default_rng(12)
    cprob = synth.generate_lq_problem(crng, crng.standard_normal(8), 17, 8, 4, nc=4, mode="W")
    for k in cprob.stages[:-1]:
        k.D[...] = crng.uniform(-1, 1, k.D.shape)
    _, _, cref = pc.oracle_serial(cprob, 1e-6)
    csc = pc.scale_of(cref)
    for legs in (4, 5):
        s = ShardedRiccatiSolver([k.dims for k in cprob.stages], cprob.nc0, legs, batch=1, lib_path=emu, on_device=False)
        assert "wave_seg<8,4,4>" in s.impl.kernel_name, s.impl.kernel_name
        s.impl.upload([cprob])
        s.backward(1e-6)
        s.forward()
        csol = s.gather_solution(0)
        for A, B in zip(csol, cref):
            assert pc.maxdiff(A, B) <= 1e-7 * csc
        assert max(lqrComputeKktError(cprob, *csol, mueq=1e-6)) <= 1e-7 * csc
    # the any-dimension leg kernels over two ranks: the gathered tuples go through the leg-parallel state elimination
    # and the block cyclic reduction of the reduced condensed system (gar_condensed_cr.hpp), redundantly on every rank
    os.environ["GAR_HIP_FORCE_GENERIC"] = "1"
    os.environ["GAR_HIP_PAD"] = "0"
    for legs in (5, 6) if world == 2 else (5,):
        s = ShardedRiccatiSolver([k.dims for k in prob.stages], prob.nc0, legs, batch=1,
                                 lib_path=emu, on_device=False)
        assert s.impl.kernel_name == "generic" and s.impl.condensed_solver_name == "reduced+cyclic"
        s.impl.upload([prob])
        s.backward(mueq)
        s.forward()
        sol = s.gather_solution(0)
        assert_matches(sol, gold, 1e-8)
        assert not s.impl.condensed_resolved(0)
    print(f"rank {rank}: horizon sharding ok")
else:
    # batch sharding (bench.py --gpus N): each rank sweeps its own problems; the only
    # collective is the reduction of a scalar summary
    nx, nu, N, B = 8, 4, 6, 3
    probs = [synth.generate_lq_problem(1000 + 17 * rank + i, np.zeros(nx), N, nx, nu, mode="W") for i in range(B)]
    s = pc.check_batched(probs, 1e-12, 1e-9, lib_path=emu)
    assert s.kernel_name == "wave<8,4>"
    t = torch.tensor([float(B)], dtype=torch.float64)
    dist.all_reduce(t)
    assert t.item() == B * world
    print(f"rank {rank}: batch sharding ok")
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("", 0))   # (the rendezvous store listens on every local address)
        return s.getsockname()[1]


@pytest.fixture(scope="module", autouse=True)
def build_emu():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)


def _run_ranks(tmp_path, mode, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, EMU, mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert "ok" in out


@pytest.mark.parametrize("mode", ["horizon", "batch"])
def test_two_ranks_gloo(tmp_path, mode):
    _run_ranks(tmp_path, mode, 2)


def test_three_ranks_gloo_horizon(tmp_path):
    """An odd world: legs 3 (one per rank), 4, 5, 6 over three ranks -- per-rank counts 1+1+2, 1+2+2, 2+2+2 -- through
    the chunked all-gather of gar_hip_solver_create_ranked's layout; the any-dimension leg kernels at 5 legs."""
    _run_ranks(tmp_path, "horizon", 3)
