"""Find a logged soak failure by the inner seed its FAIL line prints and save the exact problem as a fixture
(tests/golden/soak_*.npz) -- the soak's draw sequence is a function of (SOAK_SEED, SOAK_FOCUS) only.
usage: python scripts/soak_replay.py SOAK_SEED INNER_SEED [constrained]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from soak_draws import draws
from aligator_amd.lqr import BLOCK_NAMES


def save_problem(path, prob, **meta):
    out = {f"k{t}_{nm}": getattr(k, nm) for t, k in enumerate(prob.stages) for nm in BLOCK_NAMES if getattr(k, nm).size}
    out["dims"] = np.array([k.dims for k in prob.stages], dtype=np.int32)
    out["G0"], out["g0"] = prob.G0, prob.g0
    for k, v in meta.items():
        out["meta_" + k] = np.asarray(v)
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    soak_seed, inner = int(sys.argv[1]), int(sys.argv[2])
    focus = sys.argv[3] if len(sys.argv) > 3 else None
    for i, d in enumerate(draws(soak_seed, focus)):
        if d["seed"] == inner:
            path = os.path.join(ROOT, "tests", "golden", f"soak_{soak_seed}_{inner}.npz")
            save_problem(path, d["prob"], mu=max(d["mu"], 1e-8), legs=d["legs"], soak_seed=soak_seed, inner_seed=inner, draw=i)
            print(f"draw {i}: nx={d['nx']} nu={d['nu']} nc={d['nc']} N={d['horz']} legs={d['legs']} mu={d['mu']:.3e} -> {path}")
            break
        if i > 40000:
            sys.exit("not found in 40000 draws")
