#!/bin/bash
# One-rank rehearsal of the data-parallel bench, same box, alternating: plain native loop / Python classes with the exchange captured in the update's hipGraph /
# native loop with the exchange issued by the learner handle (A0_NATIVE_LOOP_DP=1).  usage: tools/ab_native_dp.sh [out_file] [rounds]
out=${1:-gpurun_out/r04/native_dp_ab.txt}; rounds=${2:-2}
mkdir -p "$(dirname "$out")"
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29731 A0_PROBE=none
run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ratio320 --no-other-entry 2>/dev/null |
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], '|', d['gradient_exchange'])" "$name" | tee -a "$out" || exit 1; }
for r in $(seq "$rounds"); do
  run plain_native A0_DP_FORCE=0 && run dp_python_graph A0_DP_FORCE=1 && run dp_native_handle A0_DP_FORCE=1 A0_NATIVE_LOOP_DP=1 && run dp_native_handle_one_stream A0_DP_FORCE=1 A0_NATIVE_LOOP_DP=1 A0_DP_ONE_STREAM=1 || exit 1
done
