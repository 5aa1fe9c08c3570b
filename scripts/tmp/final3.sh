mkdir -p gpurun_out/r04r; O=gpurun_out/r04r
(timeout 700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log); tail -3 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 200 python scripts/bench_preprocess.py 2>/dev/null | tail -1 > $O/bench_preprocess.json; head -c 300 $O/bench_preprocess.json; echo
