#!/bin/bash
# TEST INFRASTRUCTURE: tests/integration/seam_driver.cpp -- the shipped binding include/aligator/gar/hip-riccati.hpp
# compiled against the REFERENCE's own headers, next to the reference's own solvers -- linked with the REAL
# aligator_amd/libgar_hip.so.  Needs /root/reference (this container); the executable goes to oracle/_ref/ (git-ignored,
# travels to the GPU box with the snapshot like oracle/_ref/libgar_ref.so), where the -m gpu test of
# tests/test_integration_binding.py runs it.  A no-op where /root/reference is absent.
set -eu
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REF="${GAR_REFERENCE_ROOT:-/root/reference}"
[ -d "$REF/include/aligator/gar" ] || { echo "build_gpu_driver: $REF absent (GPU box): keeping the prebuilt oracle/_ref/seam_driver_gpu"; exit 0; }
FMT="$(python3 -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "include"))')"
[ -f "$FMT/fmt/format.h" ] || { echo "build_gpu_driver: no header-only fmt under $FMT"; exit 1; }
mkdir -p "$ROOT/oracle/_ref"
FMTDIR="$(mktemp -d)"
trap 'rm -rf "$FMTDIR"' EXIT
ln -sfn "$FMT/fmt" "$FMTDIR/fmt"
OUT="$ROOT/oracle/_ref/seam_driver_gpu"
SRC="$HERE/seam_driver.cpp"
if [ -f "$OUT" ] && [ -f "$ROOT/oracle/_ref/proxddp_lqr_gpu" ] && [ -z "$(find "$SRC" "$HERE/proxddp_lqr_driver.cpp" "$ROOT/include" "$ROOT/oracle/ref_shim" "$ROOT/aligator_amd/libgar_hip.so" -newer "$ROOT/oracle/_ref/proxddp_lqr_gpu" 2>/dev/null | head -1)" ]; then
  exit 0
fi
g++ -std=c++17 -O2 -fopenmp -DSEAM_GPU -DFMT_HEADER_ONLY -Wno-deprecated-declarations \
  -I "$ROOT/oracle/ref_shim" -I "$REF/include" -I "$FMTDIR" -I "$ROOT/include" -o "$OUT" \
  "$SRC" "$REF/src/utils/exceptions.cpp" -L "$ROOT/aligator_amd" -lgar_hip \
  -Wl,-rpath,'$ORIGIN/../../aligator_amd' -Wl,-rpath,/opt/rocm/lib -Wl,--allow-shlib-undefined
echo "built $OUT"
# ... and the reference's own SolverProxDDP loop on the backend (tests/integration/proxddp_lqr_driver.cpp over
# oracle/_ref/libaligator_ddp_ref.so, oracle/ref_ddp_build.sh)
bash "$ROOT/oracle/ref_ddp_build.sh"
OUT2="$ROOT/oracle/_ref/proxddp_lqr_gpu"
g++ -std=c++17 -O2 -fopenmp -DDDP_GPU -DFMT_HEADER_ONLY -include aligator/context.hpp -Wno-deprecated-declarations \
  -I "$ROOT/oracle/ref_shim" -I "$REF/include" -I "$FMTDIR" -I "$ROOT/include" -o "$OUT2" \
  "$HERE/proxddp_lqr_driver.cpp" -L "$ROOT/oracle/_ref" -laligator_ddp_ref -L "$ROOT/aligator_amd" -lgar_hip \
  -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../aligator_amd' -Wl,-rpath,/opt/rocm/lib -Wl,--allow-shlib-undefined
echo "built $OUT2"
