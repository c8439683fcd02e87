# round 6 evidence, part D: the whole GPU suite on the final tree, the headline refresh (bench lines, kernel statistics, PMC traffic), the learning curves, the C host, the soak runs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r06/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r06/pytest_gpu_full.log
R=r06 bash tools/refresh_profiles.sh > gpurun_out/r06/refresh.log 2>&1; echo "refresh rc=$?"
python3 tests/learning_runs.py gpurun_out/r06/learning.json > gpurun_out/r06/learning.log 2>&1; echo "learning rc=$?"; grep -c curve gpurun_out/r06/learning.log
R=r06 bash tools/c_host_bench.sh > gpurun_out/r06/c_host_bench.log 2>&1; echo "c_host rc=$?"; tail -3 gpurun_out/r06/c_host_bench.log | cut -c1-400
bash tools/soak.sh > gpurun_out/r06/soak.txt 2>&1; echo "soak rc=$?"; tail -12 gpurun_out/r06/soak.txt | cut -c1-200
