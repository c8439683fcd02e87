# Tuning aid: same-box A/B of the whole bench line — a baseline build (agent0_amd/lib/variants/libagent0_hip_base.so, selected with A0_LIB)
# against the in-tree library, alternating twice (box-to-box spread is 2-3 %, larger than most single changes).
mkdir -p gpurun_out/r02
for i in 1 2; do
  for v in base cur; do
    if [ $v = base ]; then export A0_LIB=$PWD/agent0_amd/lib/variants/libagent0_hip_base.so; else unset A0_LIB; fi
    python bench.py --no-cpu-baseline --no-ratio320 --no-other-entry 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  done
done
