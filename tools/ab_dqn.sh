# same-box A/B of one environment knob on the headline configuration (BASELINE configs[1]): bash tools/ab_dqn.sh A0_PIPELINE_TARGET=0   (knob=value is the B side)
cd $GRAFT_REPO_ROOT; R=${R:-r04}
mkdir -p gpurun_out/$R
kv=$1; knob=${kv%%=*}; val=${kv#*=}
for off in ${SEQ:-0 1 0 1}; do
  if [ $off = 1 ]; then export $knob=$val; else unset $knob; fi
  python3 bench.py --no-cpu-baseline --no-ratio320 --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS:-} 2> gpurun_out/$R/ab_dqn.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dqn $knob=' + ('$val' if $off else 'unset'), 'main', d['ms_per_step'], 'launch', (d.get('other_entry') or {}).get('ms_per_step'), d['last_loss'])"
done
unset $knob
