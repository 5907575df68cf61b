#!/bin/bash
# Round-6 box visits (every step under its own timeout; STEPS selects): tests | abpair | tracepair | bench | absh (generic
# A/B: SHAPE, LIBS) ...  Writes under gpurun_out/r6/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
STEPS=${STEPS:-"tests abpair tracepair bench"}
cd $R
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has tests; then
  echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -q -rA -p no:xdist > $O/pytest_gpu.log 2>&1; echo "rc=$?"; grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" $O/pytest_gpu.log | grep -vE "^PASSED" | tail -15
  echo "== smoke =="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error" | tee $O/smoke.log
fi
if has abpair; then
  echo "== A/B pair<56,24>: 8-byte F loads | wide F loads | wide + consumption order =="
  SHAPE=talos timeout 900 python scripts/ab_shape.py ${LIBS:-base=libgar_hip_pairbase.so wide=libgar_hip_pairwide.so wide+order=libgar_hip.so} 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab_pair.log
fi
if has absh; then
  echo "== A/B $SHAPE: $LIBS =="
  timeout 900 python scripts/ab_shape.py $LIBS 2>&1 | grep -vE "amdgpu.ids" | tee -a $O/ab_${SHAPE}.log
fi
if has tracepair; then
  echo "== trace pair (tracing build of the current sources) =="
  timeout 300 python scripts/trace_pair.py 1024 2>&1 | grep -vE "amdgpu.ids" | tee $O/trace_pair.log
fi
if has bench; then
  echo "== bench default =="; timeout 900 python bench.py --steps 20 --warmup 2 2> $O/bench.err | tail -1 > $O/bench_default_batch4096.json
  python - <<PY
import json; d=json.loads(open("$O/bench_default_batch4096.json").read())
print(d["value"], d["config"]["schedule"], d["roofline"]["frac"], d["kernel_ms"], {k: (v["value"] if isinstance(v, dict) else v) for k, v in d["schedules"].items()})
print("traffic", d["roofline"]["traffic"], "parity", d["parity"])
for k, v in d["secondary_shapes"].items(): print(k, v["kernel"], v["sweeps_per_s"], v["backward_ms"], v["backward_frac_of_hbm_roofline"], v["max_rel_err_vs_oracle"])
print("seam", {k: v["legs"]["us_per_newton_iteration"] for k, v in d["seam"].items() if isinstance(v, dict) and "legs" in v})
PY
fi
if has ab2; then
  echo "== A/B talos: full records + register LDL | packed records | packed + blocked LDL =="
  SHAPE=talos timeout 900 python scripts/ab_shape.py full=libgar_hip_pairfull.so packed=libgar_hip_noblk.so packed+blocked=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab2_talos.log
  echo "== A/B coupled nc32 (D != 0): register 44 x 44 LDL | blocked =="
  SHAPE=nc32c timeout 900 python scripts/ab_shape.py register=libgar_hip_noblk.so blocked=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab2_nc32c.log
  echo "== A/B legs (the seam's device side): register 36 x 36 LDL in the cyclic reduction | blocked =="
  timeout 300 python scripts/ab_legs.py register=libgar_hip_noblk.so blocked=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab2_legs.log
  LEGS=64 timeout 300 python scripts/ab_legs.py register=libgar_hip_noblk.so blocked=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee -a $O/ab2_legs.log
  SHAPE=talos timeout 300 python scripts/ab_legs.py register=libgar_hip_noblk.so blocked=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee -a $O/ab2_legs.log
fi
if has tracecstr; then
  echo "== trace coupled stage (tracing build) =="
  DNONZERO=1 timeout 300 python scripts/trace_cstr.py 1024 2>&1 | grep -vE "amdgpu.ids" | tee $O/trace_cstr_coupled.log
  timeout 300 python scripts/trace_cstr.py 1024 2>&1 | grep -vE "amdgpu.ids" | tee $O/trace_cstr_decoupled.log
fi
if has ab3; then
  echo "== A/B coupled nc32 (D != 0): register 44 x 44 LDL | blocked, LDS addresses kept out of the loop-invariant spills =="
  SHAPE=nc32c timeout 900 python scripts/ab_shape.py register=libgar_hip.so blocked=libgar_hip_blk2.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab3_nc32c.log
  echo "== A/B talos: register 24 x 24 | blocked =="
  SHAPE=talos timeout 900 python scripts/ab_shape.py register=libgar_hip.so blocked=libgar_hip_blk2.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab3_talos.log
fi
if has ab4; then
  echo "== A/B coupled nc32 (D != 0): as it was | register LDL + lane offsets re-derived per stage | blocked LDL + re-derived =="
  SHAPE=nc32c timeout 900 python scripts/ab_shape.py old=libgar_hip_cpl_old.so register+refresh=libgar_hip_cpl_reg_refresh.so blocked+refresh=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab4_nc32c.log
fi
if has ab5; then
  echo "== A/B lane offsets re-derived per stage: pair<56,24> =="
  SHAPE=talos timeout 900 python scripts/ab_shape.py parked=libgar_hip_pairnorefresh.so rederived=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab5_talos.log
  echo "== ... the decoupled constrained stage wave<36,12,32> =="
  SHAPE=nc32 timeout 900 python scripts/ab_shape.py parked=libgar_hip.so rederived=libgar_hip_cstrrefresh.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab5_nc32.log
  echo "== ... the headline sweep wave<36,12>, batch 4096 =="
  SHAPE=north BATCH=4096 timeout 900 python scripts/ab_shape.py parked=libgar_hip.so rederived=libgar_hip_sweeprefresh.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab5_north.log
fi
if has ab6; then
  echo "== A/B roll-out of the wide shape: plain loop | software-pipelined =="
  SHAPE=talos timeout 900 python scripts/ab_shape.py plain=libgar_hip_fwdplain.so pipelined=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab6_talos_forward.log
fi
if has ab7; then
  echo "== A/B coupled stage: C operands reloaded behind the KKT solve | kept in registers =="
  SHAPE=nc32c timeout 900 python scripts/ab_shape.py reload=libgar_hip.so keep=libgar_hip_norelc.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab7_nc32c.log
fi
if has ab8; then
  echo "== A/B coupled stage (C operands kept in registers): 44-row register LDL | hybrid: Rhat columns as a DPP panel + MFMA Schur complement, rest in registers =="
  SHAPE=nc32c timeout 900 python scripts/ab_shape.py register=libgar_hip_nohybrid.so hybrid=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab8_nc32c.log
fi
if has ab9; then
  echo "== A/B decoupled constrained stage wave<36,12,32>: C requested behind the factorisation | at the start of the stage (+ lane offsets re-derived) | at the start only =="
  SHAPE=nc32 timeout 900 python scripts/ab_shape.py late=libgar_hip.so early+refresh=libgar_hip_earlyc.so early=libgar_hip_earlyc_norefresh.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab9_nc32.log
fi
if has ab10; then
  echo "== A/B decoupled constrained stage: C requested in one burst at the start | slotted behind the first half's MFMAs =="
  SHAPE=nc32 timeout 900 python scripts/ab_shape.py burst=libgar_hip_noslotc.so slotted=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab10_nc32.log
fi
if has ab11; then
  echo "== A/B pair<56,24>: even first half | uneven (Rhat columns + factorisation on wave 1) with lane offsets re-derived | uneven with the spills =="
  SHAPE=talos timeout 900 python scripts/ab_shape.py even=libgar_hip_pair_even.so uneven=libgar_hip.so uneven_spilling=libgar_hip_pair_uneven_spill.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab11_talos.log
fi
if has ab12; then
  echo "== A/B pair<56,24> (uneven first half): next-knot operands all at the end of the stage | what the first half released, right behind it =="
  SHAPE=talos timeout 900 python scripts/ab_shape.py late=libgar_hip_pair_late.so early=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee $O/ab12_talos.log
fi
if has ab13; then
  echo "== A/B the 4-wave latency kernel mfma<36,12>: factorisation on wave 3 between the barriers | on the last worker wave behind its export =="
  for B in 256 64 1; do SHAPE=north BATCH=$B REPS=10 timeout 600 python scripts/ab_shape.py late=libgar_hip_mfma_late.so early=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids"; done | tee $O/ab13_mfma.log
fi
if has ab14; then
  echo "== A/B mfma<36,12>: Hessian tiles of the next knot requested at the start of its stage | before the last barrier of the stage before =="
  for B in 256 1; do SHAPE=north BATCH=$B REPS=10 timeout 600 python scripts/ab_shape.py late=libgar_hip_mfma_lateh.so early=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids"; done | tee $O/ab14_mfma.log
fi
if has ab15; then
  echo "== A/B mfma<36,12>: Vxx flush in 8-byte pieces behind an index division | 16-byte pieces (VxxOut), chunks dealt over the workers =="
  for B in 256 1; do SHAPE=north BATCH=$B REPS=10 timeout 600 python scripts/ab_shape.py flush8=libgar_hip_mfma_flush8.so flush16=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids"; done | tee $O/ab15_mfma.log
fi
if has ab16; then
  echo "== A/B mfma<36,12>: __launch_bounds__(256, 2) | (256, 1) =="
  for B in 256 1; do SHAPE=north BATCH=$B REPS=10 timeout 600 python scripts/ab_shape.py lb2=libgar_hip_mfma_lb2.so lb1=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids"; done | tee $O/ab16_mfma.log
fi
if has cseg; then
  echo "== leg mode with coupled constraints: the constrained segment legs (gar_cstr_seg.hpp) against the any-dimension leg kernels =="
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "segment_legs or fold" 2>&1 | tail -5 | tee $O/cseg_tests.log
  timeout 900 python scripts/time_coupled_legs.py 2>&1 | grep -vE "amdgpu.ids" | tee $O/cseg_time.log
fi
if has csegprof; then
  echo "== kernel split of leg mode with coupled constraints (LEGS=${LEGS:-32}) =="
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/csegprof -o cseg -- python $R/scripts/prof_coupled_legs.py 2>&1 | grep -E "done|rror")
  find $O/csegprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cseg_kernel_stats_${LEGS:-32}legs.csv; head -24 $O/cseg_kernel_stats_${LEGS:-32}legs.csv < /dev/null | cut -c1-150; rm -rf $O/csegprof
fi
if has csegbatch; then
  echo "== leg mode with coupled constraints, a batch of problems =="
  BATCH=${BATCH:-64} timeout 900 python scripts/time_coupled_legs.py 2>&1 | grep -vE "amdgpu.ids" | tee $O/cseg_time_batch${BATCH:-64}.log
fi
if has csegab; then
  echo "== constrained segment legs: threads of the per-stage parameter kernel (256 | 512 | 1024), 32 and 6 legs =="
  for L in 32 6; do for lib in libgar_hip_cseg_st256.so libgar_hip.so libgar_hip_cseg_st1024.so; do LEGS=$L LIB=$lib timeout 200 python scripts/prof_coupled_legs.py 2>&1 | grep done; done; done | tee $O/cseg_ab_stage_threads.log
fi
if has csegab2; then
  echo "== constrained segment legs, per-stage parameter kernel: Bunch-Kaufman of the 44 x 44 kktMat column by column | panel-blocked =="
  for L in 32 6; do for lib in libgar_hip_cseg_bkplain.so libgar_hip.so; do LEGS=$L LIB=$lib timeout 200 python scripts/prof_coupled_legs.py 2>&1 | grep done; done; done | tee $O/cseg_ab_bk_blocked.log
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "segment_legs or fold" 2>&1 | tail -2 | tee -a $O/cseg_ab_bk_blocked.log
fi
if has cycab; then
  echo "== A/B condensed solve (cyclic reduction): leg 0's wave also forming S_0, r_0, C_0 and reading G0 row by row | two extra waves for the initial condition's row, G0 staged in LDS =="
  for L in 32 64 8; do LEGS=$L timeout 300 python scripts/ab_legs.py prev=libgar_hip_cycprev.so new=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids"; done | tee $O/ab_cyclic_setup_recover.log
  SHAPE=talos timeout 300 python scripts/ab_legs.py prev=libgar_hip_cycprev.so new=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee -a $O/ab_cyclic_setup_recover.log
  echo "== kernel split, 32 legs, N = 256 =="
  (cd /tmp && HORIZON=256 LEGS=32 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cycprof -o cyc -- python $R/scripts/prof_legs.py 2>&1 | grep -E "done|rror")
  find $O/cycprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cyc_kernel_stats_32legs.csv; head -16 $O/cyc_kernel_stats_32legs.csv | cut -c1-150; rm -rf $O/cycprof
  timeout 900 python -m pytest tests -m gpu -q -x -p no:xdist -k "parallel or leg or condensed or golden or cycle or multi or seam or binding" 2>&1 | tail -3 | tee $O/cyc_tests.log
fi
if has cycab2; then
  echo "== A/B condensed solve: as it was | initial row on its own waves | + a third wave per survivor for the new coupling =="
  for L in 32 64 8; do LEGS=$L timeout 300 python scripts/ab_legs.py prev=libgar_hip_cycprev.so setup=libgar_hip_cyc1.so third=${LIB3:-libgar_hip_cyc2.so} 2>&1 | grep -vE "amdgpu.ids"; done | tee $O/ab_cyclic_third_wave.log
fi
if has cycab3; then
  echo "== A/B condensed solve: three waves per survivor | seven (products dealt by tile column; 256 registers per wave) | seven, S_i committed before the inverse =="
  for L in 32 64; do LEGS=$L timeout 300 python scripts/ab_legs.py three=libgar_hip_cyc2.so seven=libgar_hip_cyc3.so seven_early=libgar_hip_cyc4.so 2>&1 | grep -vE "amdgpu.ids"; done | tee $O/ab_cyclic_seven_waves.log
  for S in "32 12" "16 8" "12 4"; do set -- $S; NX=$1 NU=$2 LEGS=32 timeout 300 python scripts/ab_legs.py three=libgar_hip_cyc2.so seven=libgar_hip_cyc3.so seven_early=libgar_hip_cyc4.so 2>&1 | grep -vE "amdgpu.ids"; done | tee -a $O/ab_cyclic_seven_waves.log
fi
if has seamab; then
  echo "== A/B the seam (tests/cpp/_build/bench_lqr_loop --json, alternating, the library file swapped): before | set-up kernel's operands requested before the inverse + status words cleared while the host packs =="
  L=$R/aligator_amd
  cp $L/libgar_hip.so /tmp/new.so; cp $L/${PREV:-libgar_hip_cyc2b.so} /tmp/prev.so
  for rep in 1 2 3; do for w in prev new; do cp /tmp/$w.so $L/libgar_hip.so
    timeout 200 $R/tests/cpp/_build/bench_lqr_loop --json 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
for k in ('nx36_nu12','nx56_nu22'):
    v=d[k]['legs']; print('$w', k, v['us_per_newton_iteration'], v['host_us']['backward_blocks_pack_h2d_sweep_status_sync'], v['device_ms'])
"; done; done | tee $O/ab_seam_setup_prefetch_status_clear.log
  cp /tmp/new.so $L/libgar_hip.so
  for L2 in 32 64; do LEGS=$L2 timeout 300 python scripts/ab_legs.py prev=${PREV:-libgar_hip_cyc2b.so} new=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids"; done | tee -a $O/ab_seam_setup_prefetch_status_clear.log
  NX=32 NU=12 LEGS=32 timeout 300 python scripts/ab_legs.py prev=${PREV:-libgar_hip_cyc2b.so} new=libgar_hip.so 2>&1 | grep -vE "amdgpu.ids" | tee -a $O/ab_seam_setup_prefetch_status_clear.log
fi
