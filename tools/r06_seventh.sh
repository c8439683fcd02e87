# round 6: the staggered eight-wave GEMM tile — bit equality (GEMM tests + quantile tests) and same-box timing against the unstaggered build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout -k 10 900 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_engine.py -m gpu -q -x --timeout 600 > gpurun_out/r06/pytest_stag.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r06/pytest_stag.log
bash tools/build_variant.sh nostag -DA0_X9_STAGGER=0 > /dev/null 2>&1
for i in 0 1; do
  python3 tools/ubench_wplanes.py 8192 2>&1 | grep rows
  python3 tools/with_lib.py tools/variants/libagent0_hip_nostag.so tools/ubench_wplanes.py 8192 2>&1 | grep rows | sed 's/^/NOSTAG /'
done
