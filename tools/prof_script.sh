# diagnostics: rocprofv3 kernel statistics of one Python script.  usage: tools/prof_script.sh <name> <script.py> [args...]  -> gpurun_out/$R/<name>_kernel_stats.csv + top of the table on stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; R=${R:-r03}; name=$1; shift
mkdir -p gpurun_out/$R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_$name -- python3 "$@" > gpurun_out/$R/${name}_prof.log 2>&1 || { tail -5 gpurun_out/$R/${name}_prof.log; exit 1; }
f=$(ls gpurun_out/$R/prof_$name/*/*kernel_stats.csv | head -1); cp $f gpurun_out/$R/${name}_kernel_stats.csv; rm -rf gpurun_out/$R/prof_$name
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/$R/${name}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:${TOP:-14}]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
