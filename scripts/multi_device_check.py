"""The one-process multi-device solver (gar_hip_multi_create, csrc/gar_multi.hpp) at BASELINE configs[3]'s shape --
N = 2048, nx = 36, nu = 12 -- over W devices, both exchange forms (peer-mapped "pull" gather / hipMemcpyPeerAsync
"copy"), against the ONE-device solver with the same legs BIT FOR BIT (solution and the bulk read-back of every gain) and
against the serial oracle; three sweeps in a row (the readers of sweep k gate the writers of sweep k + 1).
usage: multi_device_check.py [--devices 0,1,...] [--same-device W]   (default: every visible device)
Prints one line per (legs, exchange) and `multi-device ok` / exits 1."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from aligator_amd import _lib, synth
from aligator_amd.gar import BatchedRiccatiSolver

ap = argparse.ArgumentParser()
ap.add_argument("--devices", default=None)
ap.add_argument("--same-device", type=int, default=0, help="W sub-solvers sharing device 0 (a 1-GPU box)")
ap.add_argument("--horizon", type=int, default=2048)
args = ap.parse_args()
L = _lib.load()
ndev = L.gar_hip_device_count()
devs = [0] * args.same_device if args.same_device else ([int(d) for d in args.devices.split(",")] if args.devices else list(range(ndev)))
W = len(devs)
print(f"visible devices {ndev}; sub-solvers on {devs}", flush=True)
if W < 2:
    print("fewer than two sub-solvers: nothing to exchange (use --same-device W on a 1-GPU box)")
    sys.exit(0)
nx, nu, N, mu = 36, 12, args.horizon, 1e-10
prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, mode="W")
dims = [k.dims for k in prob.stages]
import parity_cases as pc
_, _, ref = pc.oracle_serial(prob, mu)
sc = pc.scale_of(ref)
flat = lambda sol: np.concatenate([np.concatenate([np.ravel(v) for v in part]) if len(part) else np.zeros(0) for part in sol])
bad = 0
for legs in sorted({W, 32 * W}):   # one leg per device (configs[3] as stated) and 32 legs per device
    one = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs, device=devs[0])
    one.upload([prob])
    assert one.backward(mu) and one.forward()
    for exchange in ("pull", "copy"):
        if exchange == "copy":
            L.gar_hip_set_option(b"MULTI_EXCHANGE", b"copy")
        try:
            many = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs, devices=devs)
            got = many._L.gar_hip_multi_exchange_name(many.handle).decode()
            many.upload([prob])
            for _ in range(3):
                assert many.backward(mu) and many.forward()
            same = np.array_equal(flat(one.solution(0)), flat(many.solution(0)))
            bulk = all(np.array_equal(x, y) for x, y in zip(many.fetch_results(0), one.fetch_results(0)))
            err = max(pc.maxdiff(a, b) for a, b in zip(many.solution(0), ref)) / sc
            ok = same and bulk and err <= 1e-8
            # (a node whose devices lack peer access serves "pull" as "copy": reported, not an error)
            print(f"W={W} legs={legs} asked {exchange} ran {got}: bitwise == one-device {same}, bulk read-back {bulk}, "
                  f"|x - serial oracle| {err:.1e}, kernel {many.kernel_name}  {'ok' if ok else 'MISMATCH'}", flush=True)
            bad += not ok
            many.close()
        finally:
            L.gar_hip_set_option(b"MULTI_EXCHANGE", None)
    one.close()
print("multi-device ok" if not bad else f"{bad} case(s) FAILED")
sys.exit(1 if bad else 0)
