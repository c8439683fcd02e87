"""Idle time between consecutive kernels of a rocprofv3 kernel trace: python tools/gap_analysis.py <kernel_trace.csv> [skip_fraction]
Prints, over the last part of the run (the timed iterations), the GPU-busy time, the idle time and the largest classes of gaps keyed by (previous kernel -> next kernel)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
rows = rows[int(len(rows) * skip):]
short = lambda k: k.split("(")[0].replace("void ", "")[:60]
busy = idle = 0
gaps = collections.defaultdict(lambda: [0, 0.0])
end_prev, k_prev = None, None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end_prev is not None and s > end_prev:
        g = (s - end_prev) / 1e3
        idle += g
        key = (short(k_prev), short(r["Kernel_Name"]))
        gaps[key][0] += 1; gaps[key][1] += g
    busy += (e - s) / 1e3
    if end_prev is None or e > end_prev:
        end_prev, k_prev = e, r["Kernel_Name"]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"kernels {len(rows)}  span {span/1e3:.2f} ms  busy {busy/1e3:.2f} ms  idle {idle/1e3:.2f} ms ({100*idle/span:.1f} %)")
for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t/1e3:8.3f} ms  n={n:5d}  avg {t/n:6.2f} us   {a}  ->  {b}")
