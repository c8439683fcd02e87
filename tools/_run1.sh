cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab && python -m pytest tests/test_gpu_trainer.py tests/test_gpu_trace.py -x -q -m gpu -k "actor_rollout_matches_oracle or native_loop_equals or native_handles or library_handle_loop" > gpurun_out/ab/t.log 2>&1; tail -3 gpurun_out/ab/t.log
C51="--algo c51 learner.double_q=true learner.dueling_head=true learner.noisy_net=true learner.n_step_q=3 replay.policy=prioritize"
for i in 1 2; do
A0_STEP_ENC=1 python bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 10 --warmup 3 $C51 > gpurun_out/ab/c_se$i.json 2> gpurun_out/ab/c_se$i.err || exit 1
A0_STEP_ENC=0 python bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 10 --warmup 3 $C51 > gpurun_out/ab/c_no$i.json 2> gpurun_out/ab/c_no$i.err || exit 1
A0_STEP_ENC=1 python bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 10 --warmup 3 --algo qr > gpurun_out/ab/q_se$i.json 2> gpurun_out/ab/q_se$i.err || exit 1
A0_STEP_ENC=0 python bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 10 --warmup 3 --algo qr > gpurun_out/ab/q_no$i.json 2> gpurun_out/ab/q_no$i.err || exit 1
done
python - <<'PY'
import json
for f in ("c_se1","c_no1","c_se2","c_no2","q_se1","q_no1","q_se2","q_no2"):
    d=json.loads(open(f"gpurun_out/ab/{f}.json").read().strip().splitlines()[-1])
    r=d["roofline"]; print(f, d["ms_per_step"], r["frac"], r["avg_us"], r["launches"], r.get("actor_step_kernel",{}).get("avg_us"), d["last_loss"])
PY
