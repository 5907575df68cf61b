#!/bin/bash
# TEST INFRASTRUCTURE: the reference's OWN SolverProxDDP -- include/aligator/solvers/proxddp, core/, the quadratic cost
# and linear dynamics models, utils/logger, gar/ -- compiled UNCHANGED from the sources where they lie under
# /root/reference, over the Eigen-API stand-in oracle/ref_shim (Eigen, Boost, mimalloc and the generated config header
# are absent from this image; the reference's own build system is not run), into oracle/_ref/libaligator_ddp_ref.so
# (git-ignored, travels to the GPU box).  tests/integration/proxddp_lqr_driver.cpp links it: the reference's ProxDDP
# loop with `linear_solver_` as the reference sets it up AND replaced by the shipped HipRiccatiSolver
# (tests/lqr.cpp:29-75, bench/lqr.cpp:23-57).  A no-op where /root/reference is absent.
set -eu
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${GAR_REFERENCE_ROOT:-/root/reference}"
[ -d "$REF/include/aligator/solvers/proxddp" ] || { echo "ref_ddp_build: $REF absent (GPU box): keeping the prebuilt oracle/_ref"; exit 0; }
FMT="$(python3 -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "include"))')"
[ -f "$FMT/fmt/format.h" ] || { echo "ref_ddp_build: no header-only fmt under $FMT"; exit 1; }
mkdir -p "$HERE/_ref"
FMTDIR="$HERE/_ref/fmt_only"; mkdir -p "$FMTDIR"; ln -sfn "$FMT/fmt" "$FMTDIR/fmt"
OUT="$HERE/_ref/libaligator_ddp_ref.so"
SRCS="$REF/src/utils/exceptions.cpp $REF/src/utils/logger.cpp $(ls $REF/src/core/*.cpp) $REF/src/solvers/results-base.cpp
  $REF/src/solvers/value-function.cpp $REF/src/solvers/workspace-base.cpp $(ls $REF/src/solvers/proxddp/*.cpp)
  $REF/src/modelling/costs/quad-costs.cpp $REF/src/modelling/state-error.cpp $REF/src/gar/lqr-problem.cpp
  $REF/src/gar/proximal-riccati.cpp $REF/src/gar/riccati-base.cpp $REF/src/gar/riccati-kernel.cpp
  $REF/src/gar/parallel-solver.cpp $REF/src/gar/dense-riccati.cpp $REF/src/gar/dense-kernel.cpp"
FLAGS="-std=c++17 -O2 -fPIC -fopenmp -DFMT_HEADER_ONLY -DALIGATOR_MULTITHREADING -include aligator/context.hpp
  -Wno-deprecated-declarations -I $HERE/ref_shim -I $REF/include -I $FMTDIR"
if [ -f "$OUT" ] && [ -z "$(find $SRCS "$HERE/ref_shim" "$REF/include/aligator" -newer "$OUT" 2>/dev/null | head -1)" ]; then
  exit 0
fi
OBJ="$HERE/_ref/ddp_obj"; mkdir -p "$OBJ"
pids=""
for s in $SRCS; do
  o="$OBJ/$(echo "${s#$REF/src/}" | tr '/' '_' | sed 's/\.cpp$/.o/')"
  g++ $FLAGS '-DALIGATOR_TRACY_SET_THREAD_NAME(x)=delete[](x)' -c -o "$o" "$s" &
  pids="$pids $!"
  while [ "$(jobs -rp | wc -l)" -ge "${GAR_BUILD_JOBS:-8}" ]; do wait -n; done
done
for p in $pids; do wait $p; done
g++ -shared -fopenmp -o "$OUT" "$OBJ"/*.o
echo "built $OUT"
