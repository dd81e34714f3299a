set -x
mkdir -p gpurun_out
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace -o trace -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
for d in trace; do f=$(find gpurun_out/$d -name "*.db" | head -1); python tools/rocpd_summary.py $f > gpurun_out/trace_summary.txt; done
for d in pmc_fetch pmc_write pmc_sq; do f=$(find gpurun_out/$d -name "*.db" | head -1); python tools/rocpd_pmc.py $f > gpurun_out/${d}_summary.txt; done
find gpurun_out -name "*.db" -delete
cat gpurun_out/bench.json; head -20 gpurun_out/trace_summary.txt; cat gpurun_out/pmc_fetch_summary.txt gpurun_out/pmc_write_summary.txt | grep -v "finalize\|relayout"
