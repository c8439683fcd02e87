# BASELINE configs[1] and configs[2] from the plain C host (tests/c_host_loop.c) next to the Python bench of the same box, ALTERNATING (consecutive runs on one box drift by
# a per cent or two either way) -> gpurun_out/$R/c_host_loop.json
cd $GRAFT_REPO_ROOT; R=${R:-r05}
mkdir -p gpurun_out/$R
gcc -O2 -D__HIP_PLATFORM_AMD__ tests/c_host_loop.c -I/opt/rocm/include -Iinclude -Lagent0_amd/lib -lagent0_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$GRAFT_REPO_ROOT/agent0_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -o /tmp/c_host_loop || exit 1
C51="--algo c51 learner.double_q=true learner.dueling_head=true learner.noisy_net=true learner.n_step_q=3 replay.policy=prioritize"
for rep in 0 1 2; do
  /tmp/c_host_loop 40 1000000 0 1 > gpurun_out/$R/c_host_1_$rep.json || exit 1
  python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 20 --warmup 3 2> /dev/null > gpurun_out/$R/c_host_py1_$rep.json || exit 1
  A0_NATIVE_LOOP=0 python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 20 --warmup 3 2> /dev/null > gpurun_out/$R/c_host_pg1_$rep.json || exit 1
done
for rep in 0 1 2; do
  /tmp/c_host_loop 40 1000000 0 2 > gpurun_out/$R/c_host_2_$rep.json || exit 1
  python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 10 --warmup 3 $C51 2> /dev/null > gpurun_out/$R/c_host_py2_$rep.json || exit 1
  A0_NATIVE_LOOP=0 python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 10 --warmup 3 $C51 2> /dev/null > gpurun_out/$R/c_host_pg2_$rep.json || exit 1
done
/tmp/c_host_loop 6 1000000 0 3 > gpurun_out/$R/c_host_3.json || exit 1
/tmp/c_host_loop 6 1000000 0 4 > gpurun_out/$R/c_host_4.json || exit 1
python3 - <<PY
import json
R = "$R"
def rd(f):      # the bench line / the C host's summary line (it also prints a fingerprint line behind it)
    rows = [json.loads(x) for x in open(f"gpurun_out/{R}/{f}").read().strip().splitlines() if x.startswith("{")]
    return next((r for r in rows if "host" in r), rows[-1])
out = {"note": "one box, alternating runs, ms per iteration: tests/c_host_loop.c (plain C over a0_actor / a0_rbuf / a0_learner, eager launches, no rollout prefetch, 1 M-slot ring, 40 timed "
               "iterations), bench.py with the library's handles over the Trainer's buffers (deepq/native_loop.py, the default), bench.py with A0_NATIVE_LOOP=0 (Python classes + hipGraphs)"}
for c, key in ((1, "configs[1]"), (2, "configs[2]")):
    cs = [rd(f"c_host_{c}_{r}.json") for r in range(3)]
    out[key] = {"c_host_ms": [x["ms_per_iteration"] for x in cs], "python_native_loop_ms": [rd(f"c_host_py{c}_{r}.json")["ms_per_step"] for r in range(3)],
                "python_classes_hipgraph_ms": [rd(f"c_host_pg{c}_{r}.json")["ms_per_step"] for r in range(3)], "c_host_last": cs[-1]}
out["configs[3]"] = {"c_host_last": rd("c_host_3.json")}
out["configs[4] (one GPU's share)"] = {"c_host_last": rd("c_host_4.json")}
json.dump(out, open(f"gpurun_out/{R}/c_host_loop.json", "w"), indent=1)
print(json.dumps({k: ({kk: vv for kk, vv in v.items() if kk != "c_host_last"} or v["c_host_last"]["ms_per_iteration"]) for k, v in out.items() if k != "note"}))
PY
rm -f gpurun_out/$R/c_host_1_*.json gpurun_out/$R/c_host_2_*.json gpurun_out/$R/c_host_py*.json gpurun_out/$R/c_host_pg*.json gpurun_out/$R/c_host_3.json gpurun_out/$R/c_host_4.json
