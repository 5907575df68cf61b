#!/bin/bash
# The first run on more than one physical GPU, self-describing: every step prints what it is and its verdict, no step
# depends on another having passed.  On a box with ONE device (or with --same-device) the same steps run with every rank /
# sub-solver on device 0 (gloo for the timing barrier; the exchange is then a device-to-device gather): that is what
# `-m gpu` exercises (tests/test_multi_device.py::test_gpu_first_multi_gpu_script).
#   1. the peer-access matrix (hipDeviceCanAccessPeer): which exchange form gar_hip_multi_create will pick
#   2. ONE process, one RiccatiSolverBase object, legs over the devices (gar_hip_multi_create): pull vs copy exchange,
#      bitwise the one-device solver at BASELINE configs[3]'s shape (parallel-solver.hxx:150-169 is the reference's
#      "exchange": the barrier closing its OpenMP region)
#   3. one process per GPU over RCCL, --mode horizon, 2 / 4 / 8 ranks (ONE all_gather_into_tensor per sweep)
#   4. the batch axis: bench.py --gpus N (weak scaling; carries horizon_sharded inside its line)
# usage: scripts/first_multi_gpu.sh [--same-device] [--quick]     output: stdout + gpurun_out/first_multi_gpu/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
O=$R/gpurun_out/first_multi_gpu; mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
SAME=0; QUICK=0
for a in "$@"; do [ "$a" == "--same-device" ] && SAME=1; [ "$a" == "--quick" ] && QUICK=1; done
NDEV=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
echo "devices visible: $NDEV"
if [ "$NDEV" -lt 1 ]; then echo "no HIP device: nothing to run"; exit 1; fi
if [ "$NDEV" -lt 2 ]; then SAME=1; echo "one device: every step runs with all ranks / sub-solvers on device 0 (--same-device)"; fi
RC=0
step() { echo; echo "== $1 =="; }
step "1. peer-access matrix"
python - <<'PY' | tee $O/peer_matrix.log
import ctypes, torch
n = torch.cuda.device_count()
hip = ctypes.CDLL("libamdhip64.so")
print("     " + " ".join(f"{j:2d}" for j in range(n)))
allp = True
for i in range(n):
    row = []
    for j in range(n):
        c = ctypes.c_int(0)
        if i != j:
            hip.hipDeviceCanAccessPeer(ctypes.byref(c), i, j)
            allp &= bool(c.value)
        row.append(" ." if i == j else f"{c.value:2d}")
    print(f"  {i:2d} " + " ".join(row))
print("every pair has peer access: the library gathers boundary tuples with ONE kernel over peer-mapped buffers (\"pull\")" if allp and n > 1
      else "no complete peer access (or one device): hipMemcpyPeerAsync exchange (\"copy\")" if n > 1 else "one device")
PY
step "2. one process, legs over the devices (gar_hip_multi_create): pull vs copy, bitwise vs one device, configs[3]'s shape"
if [ $SAME == 1 ]; then W=$([ $QUICK == 1 ] && echo 2 || echo 8); ARGS="--same-device $W"; else ARGS=""; fi
timeout 900 python scripts/multi_device_check.py $ARGS 2>&1 | grep -v amdgpu.ids | tee $O/multi_device_check.log
[ ${PIPESTATUS[0]} == 0 ] || { echo "STEP 2 FAILED"; RC=1; }
timeout 600 python bench.py --mode horizon --single-process --gpus $([ $SAME == 1 ] && echo "${W} --same-device" || echo $NDEV) 2> $O/horizon_single_process.err | tail -1 | tee $O/horizon_single_process.json | cut -c1-400
step "3. one process per GPU, RCCL all-gather, --mode horizon"
if [ $SAME == 1 ]; then RANKS=$([ $QUICK == 1 ] && echo "2" || echo "2 4 8"); EXTRA="--same-device --backend gloo"; else RANKS=$(for n in 2 4 8; do [ $n -le $NDEV ] && echo -n "$n "; done); EXTRA=""; fi
for n in $RANKS; do
  echo "-- $n ranks --"
  timeout 900 python bench.py --gpus $n --mode horizon $EXTRA 2> $O/horizon_${n}ranks.err | tail -1 | tee $O/horizon_${n}ranks.json | cut -c1-400
  grep -q '"horizon_sharded"' $O/horizon_${n}ranks.json || { echo "STEP 3 ($n ranks) FAILED: $(tail -3 $O/horizon_${n}ranks.err)"; RC=1; }
done
step "4. the batch axis: bench.py --gpus N (weak scaling), horizon_sharded inside the line"
for n in $RANKS; do
  echo "-- $n ranks --"
  B=$([ $SAME == 1 ] && echo "--batch 64 --steps 2 --warmup 1" || echo "--steps 10 --warmup 2")
  timeout 1200 python bench.py --gpus $n $EXTRA $B --single-generator --no-cpu --no-extras --pmc off 2> $O/batch_${n}ranks.err | tail -1 > $O/batch_${n}ranks.json
  python - <<PY || { echo "STEP 4 ($n ranks) FAILED: $(tail -3 $O/batch_${n}ranks.err)"; RC=1; }
import json; d = json.loads(open("$O/batch_${n}ranks.json").read())
hs = d.get("horizon_sharded")
print("n_gpus", d["n_gpus"], "value", round(d["value"]), d["unit"], "scaling", d["scaling"], "parity", d["parity"]["max_rel_err_vs_oracle"],
      "| horizon_sharded:", {k: hs[k] for k in ("ms_per_sweep", "all_gather_ms", "max_rel_diff_vs_serial_on_rank0_stages") if k in hs} if isinstance(hs, dict) else hs)
assert d["n_gpus"] == $n and d["value"] > 0 and d["parity"]["failed_factorisations"] == 0
assert hs is None or "error" not in hs, hs
PY
done
echo; [ $RC == 0 ] && echo "first_multi_gpu: every step ok" || echo "first_multi_gpu: FAILURES above"
exit $RC
