# round 6 evidence, last pass (after the FULL GEMM launches): whole GPU suite, headline refresh, the other configurations' bench lines + kernel tables, the quantile PMC passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r06/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r06/pytest_gpu_full.log
R=r06 bash tools/refresh_profiles.sh > gpurun_out/r06/refresh.log 2>&1; echo "refresh rc=$?"
R=r06 bash tools/prof_configs.sh 2>&1 | tail -6
R=r06 bash tools/pmc_quantile.sh > gpurun_out/r06/pmc_quantile.log 2>&1; echo "pmc_quantile rc=$?"
