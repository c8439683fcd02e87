# round 6, third GPU call: the merged quantile actor step (a0_tau_cos_features / a0_fqf_taus_cos, a0_actor_quantile_tail_env_step_enc)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout -k 10 1000 python3 -m pytest tests/test_gpu_trainer.py tests/test_gpu_trace.py tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -k "iqn or fqf or quant or rollout or handle" > gpurun_out/r06/pytest_q.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06/pytest_q.log
ab() {  # algo env knob steps
  A0_TAU_COS=$3 A0_STEP_ENC=$3 python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps $4 --warmup 2 --algo $1 --env $2 2> gpurun_out/r06/abq_$1_$3.err | tee gpurun_out/r06/abq_$1_$3.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 merged=$3', d['ms_per_step'], d['value'], d['last_loss'], d['config']['host_loop'][:20], r['family'], r['avg_us'], r['frac'])"
}
for p in 0 1 0 1; do ab iqn Asterix $p 4; done
for p in 0 1 0 1; do ab fqf Asterix $p 4; done
