R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/pmc_sq -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_sq2 -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_sq2.log 2>&1
cd $R
for d in pmc_sq pmc_sq2; do f=$(find gpurun_out/$d -name "*.db" | head -1); python tools/rocpd_pmc.py $f > gpurun_out/${d}_summary.txt; done
find gpurun_out -name "*.db" -delete
grep "summarize" gpurun_out/pmc_sq_summary.txt gpurun_out/pmc_sq2_summary.txt; tail -3 gpurun_out/pmc_sq2.log
