# round 6, second GPU call: new tests + A/B of the four-wave 128 x 128 tile (A0_X9_BIG4) on configs[3] / [4]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout -k 10 900 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_reference_vectors.py -m gpu -q -x --timeout 600 -k "six_and_nine or g8 or g10" > gpurun_out/r06/pytest_new1.log 2>&1; echo "pytest1 rc=$?"; tail -3 gpurun_out/r06/pytest_new1.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_trainer.py -m gpu -q -x --timeout 600 -k "native_loop_equals_the_python_classes or g12 or learner_handle_exchanges or one_rank_rccl" > gpurun_out/r06/pytest_new2.log 2>&1; echo "pytest2 rc=$?"; tail -3 gpurun_out/r06/pytest_new2.log
ab() {  # algo env knobval steps
  A0_X9_BIG4=$3 python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps $4 --warmup 2 --algo $1 --env $2 2> gpurun_out/r06/ab4_$1_$3.err | tee gpurun_out/r06/ab4_$1_$3.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 big4=$3', d['ms_per_step'], d['value'], d['last_loss'], r['family'], r['avg_us'], r['frac'], [(c['family'], c['avg_us'], c['ms_per_iteration']) for c in r['candidates']])"
}
for p in 0 1 0 1; do ab iqn Asterix $p 4; done
for p in 0 1; do ab fqf Asterix $p 4; done
