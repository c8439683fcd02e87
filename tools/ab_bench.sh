# Tuning aid: same-box A/B of the whole bench line — a baseline build (tools/variants/libagent0_hip_base.so from tools/build_variant.sh, loaded through tools/with_lib.py)
# against the in-tree library, alternating twice (box-to-box spread is 2-3 %, larger than most single changes).
mkdir -p gpurun_out/${R:-r03}
for i in 1 2; do
  for v in base cur; do
    if [ $v = base ]; then run="python tools/with_lib.py tools/variants/libagent0_hip_base.so bench.py"; else run="python bench.py"; fi
    $run --no-cpu-baseline --no-ratio320 --no-other-entry 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  done
done
