"""The soak's draw sequence (scripts/soak.py) as a replayable generator: it depends on (soak_seed, focus)
only -- never on timing or on what the checks found -- so a logged failure is reproduced exactly from the
SOAK_SEED of its run and the inner seed its FAIL line prints (tests: test_soak_failures_replayed_*)."""
import numpy as np

from aligator_amd import synth

SHAPES = [(36, 12, 0), (32, 12, 0), (16, 8, 0), (12, 8, 0), (12, 4, 0), (8, 4, 0), (12, 6, 0), (8, 3, 0),
          (36, 12, 32), (16, 8, 8), (8, 4, 4), (6, 3, 2), (5, 2, 0),
          (30, 10, 0), (13, 5, 0), (10, 3, 0), (7, 2, 0), (33, 11, 0), (40, 9, 0), (20, 14, 0),  # padded
          (56, 22, 0), (56, 24, 0), (50, 20, 0)]  # the wide family (two waves per problem / one wave)


def draws(soak_seed, focus=None, build=True, version=1):
    """Generator over the soak's draws: dicts with nx, nu, nc, horz, mode, mu, legs, seed, backward, wide,
    dense_draw (the uniform the dense/Riccati choice compares with SOAK_DENSE) and `prob`."""
    rng = np.random.default_rng(int(soak_seed))
    shapes = [sh for sh in SHAPES if sh[2] > 0] if focus == "constrained" else SHAPES
    while True:
        nx, nu, nc = shapes[rng.integers(len(shapes))]
        horz = int(rng.integers(3, 70))
        mode = "F" if (rng.random() < 0.3 and nc == 0) else "W"      # the reference's generator, every shape
        # the THROUGHPUT kernel (one wave per problem; what the bench line runs: the library picks it for
        # batch > #CUs) on half of the serial draws, the latency kernel (one workgroup per problem) else
        backward = ("wave", "wg4", "pair")[int(rng.integers(3))] if nx <= 36 else "wave"
        wide = "pair" if rng.random() < 0.7 else "single"
        # (constrained problems below mu ~ 1e-10 are conditioned like 1/mu: the oracle and the kernels then
        # differ by cond * eps > 1e-6 from each other on EVERY kernel family, generic included)
        mu = 10.0 ** rng.uniform(-12 if nc == 0 else -10, -5)
        legs = 1 if (nx > 36 or rng.random() < (0.8 if nc > 0 else 0.4)) else int(rng.integers(2, max(3, min(9, horz // 2))))
        if version >= 2 and nx > 36 and rng.random() < 0.3:   # (round 3: the wide shape in leg mode, gar_leg_seg.hpp;
            legs = int(rng.integers(2, max(3, min(6, horz // 2))))   # version 1 keeps the draw sequence of the logged runs)
        seed = int(rng.integers(1 << 30))
        prob = synth.generate_lq_problem(np.random.default_rng(seed), rng.standard_normal(nx), horz, nx, nu, nc=nc, mode=mode)
        if nc > 0:
            # D = 0 everywhere (the reference's generator) / on every knot / on a random subset: the sweep then
            # moves along the chain decoupled stage -> coupled stage (-> LDS Bunch-Kaufman where a knot's R, S are
            # scaled down so that the reduced KKT matrix pivots); a dense C half of the time
            what = rng.random()
            for k in prob.stages[:-1]:
                if what > 0.35 and (what > 0.7 or rng.random() < 0.3):
                    k.D[...] = rng.uniform(-1, 1, k.D.shape)
                    if rng.random() < 0.15:
                        k.R[...] *= 1e-3
                        k.S[...] *= 1e-3
            if rng.random() < 0.5:
                for k in prob.stages:
                    k.C[...] = rng.uniform(-1, 1, k.C.shape)
        dense_draw = rng.random()
        yield dict(nx=nx, nu=nu, nc=nc, horz=horz, mode=mode, mu=mu, legs=legs, seed=seed, backward=backward,
                   wide=wide, dense_draw=dense_draw, prob=prob)


def find_draw(soak_seed, inner_seed, focus=None, limit=40000):
    for i, d in enumerate(draws(soak_seed, focus)):
        if d["seed"] == inner_seed:
            d["draw"] = i
            return d
        if i >= limit:
            raise LookupError(f"inner seed {inner_seed} not among the first {limit} draws of soak seed {soak_seed}")
