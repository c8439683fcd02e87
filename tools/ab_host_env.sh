#!/bin/bash
# Same-box A/B of the host-environment front-end's step path (tools/bench_host_env.py, 12 workers, one group): the library calls + host order (default), each
# knob off alone, both off (round 3's path), default again.  usage: tools/ab_host_env.sh [out_dir] [workers]
out=${1:-gpurun_out/r04/host_env_ab}; w=${2:-12}
mkdir -p "$out"
run() { name=$1; shift; env "$@" python tools/bench_host_env.py "$w" 5 1 1 > "$out/$name.json" 2> "$out/$name.err" || exit 1; cat "$out/$name.json"; }
run default_a A0_X=0 && run no_library_calls A0_ENV_POOL_CALLS=0 && run step_by_step A0_HOST_ROLLOUT=0 && run round3_path A0_ENV_POOL_CALLS=0 A0_HOST_ROLLOUT=0 && run default_b A0_X=0
