cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r02b/pytest.log
AB_FILTER= bash scripts/ab.sh "" _p128 _p256 _lb2 2>&1 | tee gpurun_out/r02b/ab.log
timeout 300 python bench.py --model fargan --dtype fp32 --steps 5 --warmup 1 > gpurun_out/r02b/bench_fargan.json 2> gpurun_out/r02b/bench_fargan.err; echo "fargan rc $?"; tail -c 1500 gpurun_out/r02b/bench_fargan.json
