# per-(kernel, grid) launch durations of one bench configuration from a rocprofv3 kernel trace: bash tools/trace_by_grid.sh <out-name> <bench args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; R=${R:-r03}
name=$1; shift
mkdir -p gpurun_out/$R
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$R/trace_$name -- python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 3 --warmup 2 "$@" > gpurun_out/$R/trace_$name.log 2>&1
python3 - gpurun_out/$R/trace_$name gpurun_out/$R/${name}_by_grid.json <<'PY'
import csv, glob, collections, json, sys
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"][:110], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(((sum(v), k, v) for k, v in per.items()), reverse=True)
tot = sum(t for t, _, _ in rows)
out = [{"kernel": k[0], "workgroups": k[1], "launches": len(v), "avg_us": round(sum(v) / len(v), 2), "min_us": round(min(v), 2), "share": round(t / tot, 4)} for t, k, v in rows[:40]]
json.dump(out, open(sys.argv[2], "w"), indent=1)
for o in out[:24]: print(o["share"], o["launches"], o["avg_us"], o["workgroups"], o["kernel"][:90])
PY
rm -rf gpurun_out/$R/trace_$name
