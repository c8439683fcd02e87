# headline configuration (BASELINE configs[1]): bench line + rocprofv3 kernel table of the same command -> gpurun_out/$R/dqn_{bench.json,kernel_stats.csv}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; R=${R:-r04}
mkdir -p gpurun_out/$R
ARGS="--no-cpu-baseline --no-ratio320 --steps ${STEPS:-10} --warmup 3"
python3 bench.py $ARGS > gpurun_out/$R/dqn_bench.json 2> gpurun_out/$R/dqn_bench.err || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/$R/dqn_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('dqn main', d['value'], d['ms_per_step'], 'launch', (d.get('other_entry') or {}).get('ms_per_step'), 'roofline', r['achieved'], r['frac'], r['avg_us'])"
if [ "${PROF:-1}" = 1 ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_dqn -- python3 bench.py $ARGS --no-other-entry > gpurun_out/$R/dqn_prof.log 2>&1 || exit 1
  f=$(ls gpurun_out/$R/prof_dqn/*/*kernel_stats.csv | head -1); cp $f gpurun_out/$R/dqn_kernel_stats.csv; rm -rf gpurun_out/$R/prof_dqn
fi
