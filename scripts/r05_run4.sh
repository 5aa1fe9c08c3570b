#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05d; mkdir -p $OUT; cd $ROOT
timeout 300 python scripts/latency_table.py > $OUT/latency_table.txt 2>&1; tail -50 $OUT/latency_table.txt
PM_RECORD_ERRORS=1 timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
