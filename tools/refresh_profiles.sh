# Regenerates the evidence under gpurun_out/<round>/ (R=r02 by default) that profiles/ is built from: default bench line (+ the launch entry point), kernel-trace
# stats of the same command, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) for the HBM traffic of the fused kernels.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; R=${R:-r03}
mkdir -p gpurun_out/${R:-r03}
python3 bench.py > gpurun_out/${R:-r03}/bench.json 2> gpurun_out/${R:-r03}/bench.err
python3 bench.py --entry launch --no-cpu-baseline --no-other-entry > gpurun_out/${R:-r03}/bench_launch.json 2> gpurun_out/${R:-r03}/bench_launch.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R:-r03}/prof -- python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry > gpurun_out/${R:-r03}/prof.log 2>&1
A0_PROBE=none rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${R:-r03}/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ratio320 --no-other-entry --replay-size 100000 > gpurun_out/${R:-r03}/pmc_fetch.log 2>&1
A0_PROBE=none rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${R:-r03}/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ratio320 --no-other-entry --replay-size 100000 > gpurun_out/${R:-r03}/pmc_write.log 2>&1
R=$R python3 - <<'PY'
import csv, glob, collections, json, os
R = os.environ.get("R", "r02")
def looping(name):      # a0_encoder_fused_kernel<7, 3, 2, 84, true, true>: the last template argument
    return name.split(">")[0].replace(" ", "").endswith("true,true")
out = {}
for name in ("fetch", "write"):
    for f in glob.glob(f"gpurun_out/{R}/pmc_{name}/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            key = ("enc" if ("a0_encoder_fused_kernel" in k or "a0_encoder_fused_multi_kernel" in k) else "dgrad" if "a0_encoder_dgrad_fused" in k else "envcommit" if "a0_env_step_commit" in k else
                   "gather" if "a0_sample_gather_kernel" in k else "qenv" if "a0_actor_qhead_env_kernel" in k else "stepenc" if "a0_actor_step_enc" in k else None)
            gs = r["Grid_Size"]
            if key == "enc" and "a0_encoder_fused_multi_kernel" in k: gs = "multi"     # round 4: the learner's forward passes of one update in one launch (2 x 512 observations for dqn)
            elif key == "enc" and looping(k): gs = "loop"          # the looping instantiation (launches of more observations than CUs: the learner's 512)
            if key: acc[(key, gs)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for (key, gs), d in sorted(acc.items()):
            for c, v in d.items():
                out[f"{key}:{gs}:{c}"] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(out, open(f"gpurun_out/{R}/pmc_summary.json", "w"), indent=1)
# per-grid-size durations of the fused kernels from the kernel trace (the stats file averages over every launch of the run)
per = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/{R}/prof/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "a0_encoder_fused_kernel" in k or "a0_encoder_dgrad_fused" in k or "a0_encoder_fused_multi_kernel" in k or "a0_actor_step_enc" in k:
            per[(k.split("(")[0][:48], "multi" if "a0_encoder_fused_multi_kernel" in k else "loop" if ("a0_encoder_fused_kernel" in k and looping(k)) else r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tr = {f"{k}|grid={g}": {"launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)} for (k, g), v in per.items()}
json.dump(tr, open(f"gpurun_out/{R}/fused_by_grid.json", "w"), indent=1)
print(json.dumps(tr, indent=1))
PY
rm -rf gpurun_out/${R:-r03}/pmc_fetch gpurun_out/${R:-r03}/pmc_write
f=$(ls gpurun_out/${R:-r03}/prof/*/*kernel_stats.csv | head -1); cp $f gpurun_out/${R:-r03}/kernel_stats.csv; rm -rf gpurun_out/${R:-r03}/prof
