# rocprofv3 kernel statistics of one bench configuration: tools/prof_one.sh <name> <bench arguments...>
# Output: gpurun_out/$R/<name>_{bench.json,kernel_stats.csv}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; R=${R:-r05}
mkdir -p gpurun_out/$R
name=$1; shift
python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 "$@" > gpurun_out/$R/${name}_bench.json 2> gpurun_out/$R/${name}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_$name -- python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 "$@" > gpurun_out/$R/${name}_prof.log 2>&1 || exit 1
f=$(ls gpurun_out/$R/prof_$name/*/*kernel_stats.csv | head -1); cp $f gpurun_out/$R/${name}_kernel_stats.csv; rm -rf gpurun_out/$R/prof_$name
python3 -c "
import json; d=json.loads(open('gpurun_out/$R/${name}_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$name', d['value'], d['ms_per_step'], d['updates_per_sec'], r['kernel'][:40], r['achieved'], r['frac'], r['avg_us'])"
