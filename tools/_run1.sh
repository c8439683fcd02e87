cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
for i in 1 2; do
A0_STEP_ENC=1 python bench.py --no-cpu-baseline --no-ratio320 --steps 20 --warmup 3 > gpurun_out/ab/se$i.json 2> gpurun_out/ab/se$i.err || exit 1
A0_STEP_ENC=0 python bench.py --no-cpu-baseline --no-ratio320 --steps 20 --warmup 3 > gpurun_out/ab/no$i.json 2> gpurun_out/ab/no$i.err || exit 1
done
python - <<'PY'
import json
for f in ("se1","no1","se2","no2"):
    d=json.loads(open(f"gpurun_out/ab/{f}.json").read().strip().splitlines()[-1])
    r=d["roofline"]; print(f, d["ms_per_step"], d["other_entry"]["ms_per_step"], r["frac"], r["avg_us"], r["launches"], r.get("actor_step_kernel",{}).get("avg_us"), r["traffic"])
PY
