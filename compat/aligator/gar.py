"""`aligator.gar` as the reference's bindings expose it (bindings/python/src/gar/expose-gar.cpp:23-178,
expose-prox-riccati.cpp, expose-parallel.cpp, expose-dense.cpp, expose-utils.cpp), served by aligator_amd: the same
class and function names, constructor arguments, attributes and in-place behaviour; the solvers run on the HIP
library behind include/gar_hip.h and fail loudly without it (there is no CPU fallback)."""
from aligator_amd.lqr import (LqrKnot, LqrProblem, lqrComputeKktError, lqrCreateSparseMatrix,  # noqa: F401
                              lqrInitializeSolution)
from aligator_amd.gar import (ParallelRiccatiSolver, ProximalRiccatiSolver, RiccatiSolverBase,  # noqa: F401
                              RiccatiSolverDense)

__all__ = ["LqrKnot", "LqrProblem", "RiccatiSolverBase", "ProximalRiccatiSolver", "ParallelRiccatiSolver",
           "RiccatiSolverDense", "lqrComputeKktError", "lqrCreateSparseMatrix", "lqrInitializeSolution"]
