# rocprofv3 kernel trace of the object API's stored-factor sweeps (width 32 and width 8, N = 1e5)
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/strace32 -o trace -- python $R/tools/gpu_sweep_profile.py 0 16 > $R/gpurun_out/strace32.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/strace8 -o trace -- python $R/tools/gpu_sweep_profile.py 2 3 > $R/gpurun_out/strace8.log 2>&1
cd $R
for w in 32 8; do f=$(find gpurun_out/strace$w -name "*.db" | head -1); python tools/rocpd_summary.py $f > gpurun_out/strace${w}_summary.txt; done
find gpurun_out -name "*.db" -delete
cut -c1-170 gpurun_out/strace32_summary.txt | head -14; cut -c1-170 gpurun_out/strace8_summary.txt | head -14
