# rocprofv3 passes of the headline bench: kernel trace + stats; FETCH_SIZE; WRITE_SIZE; SQ set -- counters in
# their own passes, every pass under its own timeout (a combined FETCH_SIZE + WRITE_SIZE pass once hung).
mkdir -p gpurun_out
R=$PWD
BENCH="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-shared-series --steady-seconds 0 --settle-seconds 0"
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace -o trace -- $BENCH > $R/gpurun_out/trace.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o pmc -- $BENCH > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o pmc -- $BENCH > $R/gpurun_out/pmc_write.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/pmc_sq -o pmc -- $BENCH > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
f=$(find gpurun_out/trace -name "*.db" | head -1); python tools/rocpd_summary.py $f > gpurun_out/trace_summary.txt
for d in pmc_fetch pmc_write pmc_sq; do f=$(find gpurun_out/$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f > gpurun_out/${d}_summary.txt; done
find gpurun_out -name "*.db" -delete
head -8 gpurun_out/trace_summary.txt | cut -c1-150
