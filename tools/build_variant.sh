#!/usr/bin/env bash
# Tuning aid: builds a copy of libagent0_hip.so with extra -D flags into tools/variants/ (run a diagnostic against it with python tools/with_lib.py <path> <script> [args]).
# The build reports "<name>: <flags>" through a0_build_info(), so the product loader refuses it..
# usage: tools/build_variant.sh <name> [-DFOO=1 ...]
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME="$1"; shift
SRC="${ROOT}/agent0_amd/csrc"; OBJ="${SRC}/_obj_${NAME}"; OUT="${ROOT}/tools/variants"
mkdir -p "${OBJ}" "${OUT}"
FLAGS=(-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I"${ROOT}/include" "$@" "-DA0_BUILD_VARIANT=\"${NAME}: $*\"")
pids=()
for s in core net encoder_fused conv1_wgrad conv23_wgrad loss quantile optim replay rng synth_env actor dp learner runtime; do
  /opt/rocm/bin/hipcc "${FLAGS[@]}" -c "${SRC}/${s}.hip" -o "${OBJ}/${s}.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "${OUT}/libagent0_hip_${NAME}.so" "${OBJ}"/*.o -ldl
rm -rf "${OBJ}"
echo "built ${OUT}/libagent0_hip_${NAME}.so"
