# rocprofv3 passes of the headline bench (kernel trace + stats; FETCH_SIZE; WRITE_SIZE; SQ set) and of
# BASELINE config 4 (width 32: the wide kernels).  Counters in their own passes, never with --sys-trace.
set -x
mkdir -p gpurun_out
R=$PWD
BENCH="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --steady-seconds 0"
WIDE="python $R/tools/gpu_wide_profile.py"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace -o trace -- $BENCH > $R/gpurun_out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o pmc -- $BENCH > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o pmc -- $BENCH > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/pmc_sq -o pmc -- $BENCH > $R/gpurun_out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wtrace -o trace -- $WIDE > $R/gpurun_out/wtrace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d $R/gpurun_out/wpmc_mem -o pmc -- $WIDE > $R/gpurun_out/wpmc_mem.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/wpmc_sq -o pmc -- $WIDE > $R/gpurun_out/wpmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/wpmc_mfma -o pmc -- $WIDE > $R/gpurun_out/wpmc_mfma.log 2>&1
cd $R
for d in trace wtrace; do f=$(find gpurun_out/$d -name "*.db" | head -1); python tools/rocpd_summary.py $f > gpurun_out/${d}_summary.txt; done
for d in pmc_fetch pmc_write pmc_sq wpmc_mem wpmc_sq wpmc_mfma; do f=$(find gpurun_out/$d -name "*.db" | head -1); python tools/rocpd_pmc.py $f > gpurun_out/${d}_summary.txt; done
find gpurun_out -name "*.db" -delete
head -12 gpurun_out/trace_summary.txt; head -12 gpurun_out/wtrace_summary.txt; tail -2 gpurun_out/wpmc_mfma.log
