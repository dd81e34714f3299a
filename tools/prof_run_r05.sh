#!/bin/bash
# Round-5 evidence run on the GPU box (through gpurun): rocprofv3 kernel traces + PMC passes of the headline bench
# command, of BASELINE config 4 (width 32) and of the accuracy-family leg; summaries land in gpurun_out/prof_r05/.
# Usage: bash tools/prof_run_r05.sh [headline] [wide] [accuracy] [sweeps] [grad]
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_${PROF_TAG:-r05}
mkdir -p "$OUT"
WHAT="${*:-headline wide accuracy}"
BENCH="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-shared-series --no-accuracy-family --no-gradient --no-object-api --sharded 0 --no-config3 --steady-seconds 0 --settle-seconds 0"

trace() {  # name, command...
  local name=$1; shift
  rm -rf "$OUT/$name"
  timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o trace -- "$@" > "$OUT/$name.log" 2>&1
  local db; db=$(find "$OUT/$name" -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" "$OUT/${name}_kernel_trace_stats.txt" > /dev/null
}
pmc() {  # name, counters, command...
  local name=$1 counters=$2; shift 2
  rm -rf "$OUT/$name"
  timeout 240 rocprofv3 --kernel-trace --pmc $counters -d "$OUT/$name" -o pmc -- "$@" > "$OUT/$name.log" 2>&1
  local db; db=$(find "$OUT/$name" -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > "$OUT/${name}_summary.txt"
}

for w in $WHAT; do
  case $w in
    headline)
      trace headline $BENCH
      pmc headline_fetch "FETCH_SIZE" $BENCH
      pmc headline_write "WRITE_SIZE" $BENCH
      pmc headline_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" $BENCH
      ;;
    wide)
      trace wide python tools/gpu_wide_profile.py
      pmc wide_fetch "FETCH_SIZE" python tools/gpu_wide_profile.py
      pmc wide_write "WRITE_SIZE" python tools/gpu_wide_profile.py
      pmc wide_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" python tools/gpu_wide_profile.py
      ;;
    wide64)
      trace wide64 python tools/gpu_wide64_profile.py
      pmc wide64_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" python tools/gpu_wide64_profile.py
      ;;
    accuracy)
      trace accuracy python tools/gpu_accuracy_profile.py
      pmc accuracy_fetch "FETCH_SIZE" python tools/gpu_accuracy_profile.py
      pmc accuracy_write "WRITE_SIZE" python tools/gpu_accuracy_profile.py
      pmc accuracy_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" python tools/gpu_accuracy_profile.py
      ;;
    grad)
      trace grad python tools/gpu_grad_profile.py
      pmc grad_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" python tools/gpu_grad_profile.py
      pmc grad_fetch "FETCH_SIZE" python tools/gpu_grad_profile.py
      pmc grad_write "WRITE_SIZE" python tools/gpu_grad_profile.py
      ;;
    rows)
      trace rows python tools/gpu_rows_profile.py
      ;;
    small)
      trace small python tools/gpu_small_profile.py
      pmc small_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" python tools/gpu_small_profile.py
      ;;
    widegrad)
      trace widegrad python tools/gpu_wide_grad_profile.py
      ;;
    singlewide)
      trace singlewide python tools/gpu_single_wide_profile.py
      ;;
    singlenarrow)
      trace singlenarrow python tools/gpu_single_narrow_profile.py
      ;;
    singlegrad)
      trace singlegrad python tools/gpu_single_grad_profile.py
      ;;
    sweeps8)
      trace sweeps_w8 python tools/gpu_sweep_profile2.py 2 3
      trace sweeps_w16 python tools/gpu_sweep_profile2.py 2 7
      ;;
    general)
      trace general python tools/gpu_general_profile.py
      ;;
    sweeps)
      trace sweeps_w64 python tools/gpu_sweep_profile.py 0 32
      trace sweeps_w40 python tools/gpu_sweep_profile.py 0 20
      ;;
  esac
done
ls -la "$OUT" | head -60
