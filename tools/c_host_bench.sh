# BASELINE configs[1] and configs[2] from the plain C host (tests/c_host_loop.c) next to the Python bench of the same box -> gpurun_out/$R/c_host_loop.json
cd $GRAFT_REPO_ROOT; R=${R:-r04}
mkdir -p gpurun_out/$R
gcc -O2 -D__HIP_PLATFORM_AMD__ tests/c_host_loop.c -I/opt/rocm/include -Iinclude -Lagent0_amd/lib -lagent0_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$GRAFT_REPO_ROOT/agent0_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -o /tmp/c_host_loop || exit 1
/tmp/c_host_loop 40 1000000 0 1 > gpurun_out/$R/c_host_1.json || exit 1
/tmp/c_host_loop 40 1000000 0 2 > gpurun_out/$R/c_host_2.json || exit 1
python3 bench.py --no-cpu-baseline --no-ratio320 --steps 20 --warmup 3 2> /dev/null > gpurun_out/$R/c_host_py1.json || exit 1
python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 10 --warmup 3 --algo c51 learner.double_q=true learner.dueling_head=true learner.noisy_net=true learner.n_step_q=3 replay.policy=prioritize 2> /dev/null > gpurun_out/$R/c_host_py2.json || exit 1
python3 - <<PY
import json
R = "$R"
rd = lambda f: json.loads(open(f"gpurun_out/{R}/{f}").read().strip().splitlines()[-1])
c1, c2, p1, p2 = rd("c_host_1.json"), rd("c_host_2.json"), rd("c_host_py1.json"), rd("c_host_py2.json")
out = {"note": "tests/c_host_loop.c (plain C over a0_actor / a0_rbuf / a0_learner, eager launches, 1 M-slot ring, 40 timed iterations) and bench.py (Python host, hipGraphs, main schedule) on the same box",
       "configs[1]": {"c_host": c1, "python_main_ms": p1["ms_per_step"], "python_launch_ms": (p1.get("other_entry") or {}).get("ms_per_step")},
       "configs[2]": {"c_host": c2, "python_main_ms": p2["ms_per_step"]}}
json.dump(out, open(f"gpurun_out/{R}/c_host_loop.json", "w"), indent=1)
print(json.dumps({k: (v["c_host"]["ms_per_iteration"], v["python_main_ms"]) for k, v in out.items() if k.startswith("configs")}))
PY
