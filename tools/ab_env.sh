# same-box A/B of one environment setting on the quantile configurations: bash tools/ab_env.sh NAME=VALUE  (alternates unset / set, twice)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${R:-r03}
kv=$1; name=${kv%%=*}
for algo in ${ALGOS:-iqn fqf}; do
  for on in 0 1 0 1; do
    if [ $on = 1 ]; then export "$kv"; else unset $name; fi
    python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 --algo $algo --env Asterix 2> gpurun_out/${R:-r03}/ab_env.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$algo', '$kv' if $on else '(default)', d['value'], d['ms_per_step'], d['updates_per_sec'], d['last_loss'])"
  done
done
