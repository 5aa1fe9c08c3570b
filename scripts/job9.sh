cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "saturate" 2>&1 | tail -3
bash scripts/ab.sh "" _nosat 2>&1 | tee gpurun_out/r02i/ab_saturate.log
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $ROOT/gpurun_out/r02i/prof_preprocess -o pre -- python $ROOT/scripts/bench_preprocess.py > $ROOT/gpurun_out/r02i/preprocess.log 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $ROOT/gpurun_out/r02i/prof_fargan -o fg -- python $ROOT/bench.py --model fargan --steps 2 --warmup 1 --sustain 0 --no-cpu-baseline > $ROOT/gpurun_out/r02i/fargan_prof.log 2>&1
cd $ROOT
find gpurun_out/r02i -name "*kernel_trace.csv" -size +1M -delete
head -12 gpurun_out/r02i/prof_preprocess/*/pre_kernel_stats.csv 2>/dev/null || find gpurun_out/r02i -name "*kernel_stats.csv"
tail -3 gpurun_out/r02i/preprocess.log
