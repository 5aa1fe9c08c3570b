cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02r
timeout 120 python scripts/rccl_sanity.py > gpurun_out/r02r/rccl_sanity.txt 2>&1; grep "rccl ok" gpurun_out/r02r/rccl_sanity.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02r/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02r/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time bash scripts/gpu_round.sh r02r bench ) 2>&1 | tail -12
