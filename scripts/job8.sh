cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02h/pytest.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r02h/pytest.log
AB_FILTER=pair_c256 bash scripts/ab.sh "" _g4 2>&1 | tee gpurun_out/r02h/ab_g.log
bash scripts/gpu_round.sh r02h bench
bash scripts/profile_gpu.sh r02 > gpurun_out/r02h/profile.log 2>&1; tail -5 gpurun_out/r02h/profile.log
