# same-box A/B of one environment knob on the headline configuration (configs[1], both schedules): bash tools/ab_knob_dqn.sh A0_NO_FC1_FRAMES
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${R:-r03}
knob=$1
for off in 0 1 0 1; do
  if [ $off = 1 ]; then export $knob=1; else unset $knob; fi
  python3 bench.py --no-cpu-baseline --no-ratio320 2> gpurun_out/${R:-r03}/ab_knob_dqn.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d.get('other_entry') or {}; print('$knob=$off', d['value'], d['ms_per_step'], 'launch', o.get('value'), o.get('ms_per_step'), 'roofline', d['roofline']['frac'])"
done
