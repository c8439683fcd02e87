cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab && python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_trainer.py -x -q -m gpu -k "one_launch or side_by_side or native_learner or native_c51 or native_qr or update_full_size or native_loop_equals" > gpurun_out/ab/t.log 2>&1; tail -3 gpurun_out/ab/t.log
C51="--algo c51 learner.double_q=true learner.dueling_head=true learner.noisy_net=true learner.n_step_q=3 replay.policy=prioritize"
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 20 --warmup 3 $C51 > gpurun_out/ab/c_tr$i.json 2> gpurun_out/ab/c_tr$i.err || exit 1
A0_NO_HEAD_PAIR=1 python bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 20 --warmup 3 $C51 > gpurun_out/ab/c_no$i.json 2> gpurun_out/ab/c_no$i.err || exit 1
python bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 20 --warmup 3 --algo qr > gpurun_out/ab/q_tr$i.json 2> gpurun_out/ab/q_tr$i.err || exit 1
A0_NO_HEAD_PAIR=1 python bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 20 --warmup 3 --algo qr > gpurun_out/ab/q_no$i.json 2> gpurun_out/ab/q_no$i.err || exit 1
done
python - <<'PY'
import json
for f in ("c_tr1","c_no1","c_tr2","c_no2","c_tr3","c_no3","q_tr1","q_no1","q_tr2","q_no2","q_tr3","q_no3"):
    d=json.loads(open(f"gpurun_out/ab/{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["last_loss"])
PY
