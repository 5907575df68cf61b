import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from aligator_amd import synth
from aligator_amd.gar import BatchedRiccatiSolver
import parity_cases as pc
nx, nu, N = 36, 12, 2048
prob = synth.generate_lq_problem(7, np.zeros(nx), N, nx, nu, mode="W")
dims = [k.dims for k in prob.stages]
_, _, ref = pc.oracle_serial(prob, 1e-10)
sc = pc.scale_of(ref)
for W, legs in ((8, 256), (8, 8), (5, 37), (16, 64)):
    for ex in ("pull", "copy"):
        import os
        if ex == "copy": os.environ["GAR_HIP_MULTI_EXCHANGE"] = "copy"
        else: os.environ.pop("GAR_HIP_MULTI_EXCHANGE", None)
        s = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs, devices=[0] * W)
        s.upload([prob])
        for _ in range(3):
            assert s.backward(1e-10) and s.forward()
        err = max(pc.maxdiff(a, b) for a, b in zip(s.solution(0), ref)) / sc
        one = BatchedRiccatiSolver(dims, nx, batch=1, num_legs=legs)
        one.upload([prob]); one.backward(1e-10); one.forward()
        same = all(np.array_equal(x, y) for A, B in zip(s.solution(0), one.solution(0)) for x, y in zip(A, B))
        ra = s.fetch_results(0); rb = one.fetch_results(0)
        same_bulk = all(np.array_equal(x, y) for x, y in zip(ra, rb))
        print(f"W={W} legs={legs} {ex}: err vs serial oracle {err:.1e}, bitwise == one-device: {same}, bulk read-back: {same_bulk}, exchange {s._L.gar_hip_multi_exchange_name(s.handle).decode()}")
        s.close(); one.close()
