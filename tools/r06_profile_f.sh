# round 6: headline refresh after cpu_baseline gained the measured whole-iteration item
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
R=r06 bash tools/refresh_profiles.sh > gpurun_out/r06/refresh.log 2>&1; echo "refresh rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench.json").read().strip().splitlines()[-1]); c = d["cpu_baseline"]
print(d["ms_per_step"], d["value"], c["value"], c["cores"], c["items"].get("v_bench_workload_iteration"), c["items"]["composed_from_i_and_ii"]["env_frames_per_sec"])
print(c["sample"][:300])
PY
