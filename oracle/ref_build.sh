#!/bin/bash
# TEST INFRASTRUCTURE: compile the REFERENCE's own gar sources, unchanged, where they lie under /root/reference,
# over the minimal Eigen-API stand-in oracle/ref_shim (Eigen, Boost and the jrl-cmakemodules generated headers are
# absent from this image; the reference's own build system is not run).  Output only into oracle/_ref/ (git-ignored,
# travels to the GPU box with the snapshot).  fmt comes header-only from the PyTorch wheel's include directory.
set -eu
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${GAR_REFERENCE_ROOT:-/root/reference}"
[ -d "$REF/include/aligator/gar" ] || { echo "ref_build: $REF absent (GPU box): keeping the prebuilt oracle/_ref"; exit 0; }
FMT="$(python3 -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "include"))')"
[ -f "$FMT/fmt/format.h" ] || { echo "ref_build: no header-only fmt under $FMT"; exit 1; }
mkdir -p "$HERE/_ref"
FMTDIR="$(mktemp -d)"                               # expose fmt alone, not the rest of torch's include tree
trap 'rm -rf "$FMTDIR"' EXIT
ln -sfn "$FMT/fmt" "$FMTDIR/fmt"
OUT="$HERE/_ref/libgar_ref.so"
SRCS="$HERE/ref_driver.cpp $REF/src/utils/exceptions.cpp"
if [ -f "$OUT" ] && [ -z "$(find $SRCS "$HERE/ref_shim" "$REF/include/aligator/gar" "$REF/include/aligator/core" -newer "$OUT" 2>/dev/null | head -1)" ]; then
  exit 0
fi
g++ -std=c++17 -O2 -fPIC -shared -fopenmp -DFMT_HEADER_ONLY -Wno-deprecated-declarations \
  -I "$HERE/ref_shim" -I "$REF/include" -I "$FMTDIR" -o "$OUT" $SRCS
echo "built $OUT"
