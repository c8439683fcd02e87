# same-box A/B of one environment knob on the quantile configurations (configs[3], configs[4]): bash tools/ab_knob.sh A0_NO_DGRAD_HADAMARD
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${R:-r03}
knob=$1
for algo in ${ALGOS:-iqn fqf}; do
  for off in 0 1 0 1; do
    if [ $off = 1 ]; then export $knob=1; else unset $knob; fi
    python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 --algo $algo --env Asterix 2> gpurun_out/${R:-r03}/ab_knob.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$algo $knob=$off', d['value'], d['ms_per_step'], d['updates_per_sec'], d['last_loss'])"
  done
done
