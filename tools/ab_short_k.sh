# same-box A/B of the short-reduction forward kernel on the quantile configurations (configs[3], configs[4])
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${R:-r03}
for algo in iqn fqf; do
  for off in 0 1 0 1; do
    if [ $off = 1 ]; then export A0_NO_SHORT_K=1; else unset A0_NO_SHORT_K; fi
    python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 --algo $algo --env Asterix 2> gpurun_out/${R:-r03}/ab_sk.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$algo no_short_k=$off', d['value'], d['ms_per_step'], d['updates_per_sec'], d['last_loss'])"
  done
done
