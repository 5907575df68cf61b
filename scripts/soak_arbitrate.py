"""Replay one soak draw (SOAK_SEED INNER_SEED [focus]) on the GPU and print the 4-way arbitration of
tests/parity_cases.check_parallel / check_serial: HIP, oracle leg-parallel, oracle serial, LAPACK on the dense KKT."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from soak_draws import draws
import parity_cases as pc
soak_seed, inner = sys.argv[1], int(sys.argv[2])
focus = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].endswith(".so") else None
lib_path = next((a for a in sys.argv[3:] if a.endswith(".so")), None)
if lib_path:   # an older build of the library (regression hunting): bind only the symbols it has
    import ctypes
    from aligator_amd import _lib
    have = ctypes.CDLL(os.path.abspath(lib_path))
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(have, k)}
    from aligator_amd.gar import BatchedRiccatiSolver

    def _old_backward(self, p, mu):   # what the mirror did before gar_hip_backward_blocks existed
        for t, k in enumerate(p.stages):
            self.upload_knot(0, t, k)
        self.set_init(0, p.G0, p.g0)
        return self.backward(mu)
    BatchedRiccatiSolver.backward_blocks = _old_backward
for i, d in enumerate(draws(soak_seed, focus, version=2)):
    if d["seed"] == inner:
        break
    assert i < 40000
os.environ["GAR_HIP_BACKWARD"] = d["backward"]; os.environ["GAR_HIP_WIDE"] = d["wide"]
print("draw", i, {k: d[k] for k in ("nx", "nu", "nc", "horz", "legs", "mode", "mu", "backward")})
rep = {}
try:
    par = pc.check_parallel(d["prob"], d["mu"], d["legs"], 1e-8, lib_path, conditioned=True, report=rep)
    print("passes on", par._impl.kernel_name)
except AssertionError:
    print("FAILS the conditioned tolerance")
for k, v in rep.items():
    print(f"  {k:28s}", ["%.1e" % x for x in v] if isinstance(v, list) else v)
try:
    from aligator_amd.gar import ParallelRiccatiSolver, lqrInitializeSolution
    for ok in (None, 1e-13, 1e-14, 0.0):
        p2 = ParallelRiccatiSolver(d["prob"].copy(), d["legs"], lib_path=lib_path)
        p2.maxRefinementSteps = 10
        if ok is not None:
            p2._impl._check(p2._impl._L.gar_hip_set_condensed_backward_ok(p2._impl.handle, ok))
        p2.backward(d["mu"])
        sol = lqrInitializeSolution(d["prob"]); p2.forward(*sol)
        _, _, ref = pc.oracle_serial(d["prob"], d["mu"])
        sc = pc.scale_of(ref)
        print(f"backward_ok={ok}: omega={p2._impl.condensed_backward_error(0):.2e} info={p2._impl.condensed_info(0)} resolved={p2._impl.condensed_resolved(0)} "
              f"err vs serial oracle (x,u,v,lbd)={[('%.1e' % (pc.maxdiff(a, b) / sc)) for a, b in zip(sol, ref)]}")
except Exception as e:
    print("omega probe:", type(e).__name__, e)
