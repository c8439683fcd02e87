# same-box A/B of one environment setting on the 8-game suite configuration (fqf + double-Q + dueling + prioritized sum-tree replay, README.md:62-112), one game:
# bash tools/ab_suite.sh NAME=VALUE [game]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${R:-r04}
kv=$1; name=${kv%%=*}; game=${2:-Asterix}
for on in ${SEQ:-0 1 0 1}; do
  if [ $on = 1 ]; then export "$kv"; else unset $name; fi
  python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 3 --warmup 2 --algo fqf --env $game learner.double_q=true learner.dueling_head=true replay.policy=prioritize 2> gpurun_out/${R:-r04}/ab_suite.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$game suite', '$kv' if $on else '(default)', d['value'], d['ms_per_step'], d['updates_per_sec'], d['last_loss'])"
done
unset $name
