cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A0_PROBE=none rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ratio320 --no-other-entry --replay-size 100000 > gpurun_out/prof_k.log 2>&1
f=$(ls gpurun_out/prof_k/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot/1e6)
for r in rows[:40]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):7d} avg_us {float(r["AverageNs"])/1e3:9.2f} tot_ms {float(r["TotalDurationNs"])/1e6:9.2f} {float(r["Percentage"]):6.2f}%')
PY
