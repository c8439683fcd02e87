# round 6 evidence, part A: the whole GPU suite, then the headline configuration's bench lines, kernel statistics and PMC traffic (tools/refresh_profiles.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r06/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06/pytest_gpu_full.log
R=r06 bash tools/refresh_profiles.sh > gpurun_out/r06/refresh.log 2>&1; echo "refresh rc=$?"
python3 -c "
import json
for f in ('bench','bench_launch'):
    d=json.loads(open('gpurun_out/r06/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f, d['ms_per_step'], d['value'], (d.get('other_entry') or {}).get('ms_per_step'), (d.get('at_reference_update_ratio') or {}).get('value'), r['family'], r['avg_us'], r['frac'], r['iteration'], (d.get('cpu_baseline') or {}).get('value'))
"
