# rocprofv3 kernel statistics of BASELINE configs[2..4] at full size (c51 rainbow-lite, Asterix iqn, Asterix fqf): one bench line each (with the roofline of
# that configuration's dominant kernel) and the per-kernel table of the same command.  Output: gpurun_out/<round>/{c51,iqn,fqf}_{bench.json,kernel_stats.csv}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; R=${R:-r05}
mkdir -p gpurun_out/$R
run() {   # name, bench arguments...
  name=$1; shift
  python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 "$@" > gpurun_out/$R/${name}_bench.json 2> gpurun_out/$R/${name}_bench.err || return 1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_$name -- python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 "$@" > gpurun_out/$R/${name}_prof.log 2>&1 || return 1
  f=$(ls gpurun_out/$R/prof_$name/*/*kernel_stats.csv | head -1); cp $f gpurun_out/$R/${name}_kernel_stats.csv; rm -rf gpurun_out/$R/prof_$name
  python3 -c "
import json; d=json.loads(open('gpurun_out/$R/${name}_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$name', d['value'], d['ms_per_step'], d['updates_per_sec'], r['kernel'][:40], r['achieved'], r['frac'], r['avg_us'])"
}
run c51 --algo c51 learner.double_q=true learner.dueling_head=true learner.noisy_net=true learner.n_step_q=3 replay.policy=prioritize &&
run iqn --algo iqn --env Asterix &&
run fqf --algo fqf --env Asterix &&
run qr --algo qr &&
run mdqn --algo mdqn
