# rocprofv3 passes of BASELINE config 4 (width 32): kernel trace, then MFMA / LDS counters in their own pass.
# Every pass under its own timeout (a combined FETCH_SIZE + WRITE_SIZE pass once hung for 25 minutes).
mkdir -p gpurun_out
R=$PWD
WIDE="python $R/tools/gpu_wide_profile.py"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wtrace -o trace -- $WIDE > $R/gpurun_out/wtrace.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $R/gpurun_out/wpmc_mfma -o pmc -- $WIDE > $R/gpurun_out/wpmc_mfma.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/wpmc_fetch -o pmc -- $WIDE > $R/gpurun_out/wpmc_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/wpmc_write -o pmc -- $WIDE > $R/gpurun_out/wpmc_write.log 2>&1
cd $R
f=$(find gpurun_out/wtrace -name "*.db" | head -1); python tools/rocpd_summary.py $f > gpurun_out/wtrace_summary.txt
for d in wpmc_mfma wpmc_fetch wpmc_write; do f=$(find gpurun_out/$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f > gpurun_out/${d}_summary.txt; done
find gpurun_out -name "*.db" -delete
head -9 gpurun_out/wtrace_summary.txt | cut -c1-150; grep "prefix32\|wide_scan_kernel<32, true, 1>\|wide_correct" gpurun_out/wpmc_mfma_summary.txt | cut -c1-160; tail -2 gpurun_out/wpmc_mfma.log | cut -c1-200
