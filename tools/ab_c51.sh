# same-box A/B of one environment knob on BASELINE configs[2] (c51 rainbow-lite): bash tools/ab_c51.sh A0_C51_SEPARATE [rounds]; then the kernel table of the default build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; R=${R:-r04}
mkdir -p gpurun_out/$R
kv=$1; knob=${kv%%=*}; val=${kv#*=}; [ "$val" = "$kv" ] && val=1       # KNOB or KNOB=VALUE (the B side)
ARGS="--no-cpu-baseline --no-ratio320 --no-other-entry --steps 6 --warmup 2 --algo c51 learner.double_q=true learner.dueling_head=true learner.noisy_net=true learner.n_step_q=3 replay.policy=prioritize"
for off in ${SEQ:-0 1 0 1}; do
  if [ $off = 1 ]; then export $knob=$val; else unset $knob; fi
  python3 bench.py $ARGS 2> gpurun_out/$R/ab_c51.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c51 $knob=' + ('$val' if $off else 'unset'), d['value'], d['ms_per_step'], d['updates_per_sec'], d['last_loss'])"
done
unset $knob
if [ "${PROF:-1}" = 1 ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_c51 -- python3 bench.py $ARGS > gpurun_out/$R/c51_prof.log 2>&1 || exit 1
  f=$(ls gpurun_out/$R/prof_c51/*/*kernel_stats.csv | head -1); cp $f gpurun_out/$R/c51_kernel_stats.csv; rm -rf gpurun_out/$R/prof_c51
fi
