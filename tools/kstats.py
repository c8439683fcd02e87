"""Prints a rocprofv3 kernel_stats.csv as a short table: python tools/kstats.py gpurun_out/r04/c51_kernel_stats.csv [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['AverageNs'])/1000:8.2f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):6.2f} %")
print(f"total {tot/1e6:.1f} ms")
