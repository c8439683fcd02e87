# round 6: FULL launches of the split-operand GEMM (no validity masks in the staging) — bit equality and same-box A/B (A0_X9_NO_FULL=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout -k 10 900 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_engine.py tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 > gpurun_out/r06/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r06/pytest_full.log
for i in 0 1; do
  python3 tools/ubench_wplanes.py 8192 32768 2>&1 | grep rows
  A0_X9_NO_FULL=1 python3 tools/ubench_wplanes.py 8192 32768 2>&1 | grep rows | sed 's/^/MASKED /'
done
ab() {
  if [ $3 = 1 ]; then export A0_X9_NO_FULL=1; else unset A0_X9_NO_FULL; fi
  python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps $4 --warmup 2 --algo $1 --env $2 2> gpurun_out/r06/abf_$1_$3.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 masked=$3', d['ms_per_step'], d['value'], d['last_loss'], r['family'], r['avg_us'], r['frac'])"
}
for p in 1 0 1 0; do ab iqn Asterix $p 4; done
for p in 1 0; do ab fqf Asterix $p 4; done
for p in 1 0 1 0; do ab dqn Breakout $p 20; done
