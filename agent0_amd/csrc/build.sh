#!/usr/bin/env bash
# Builds libagent0_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../lib"
mkdir -p "${OUT}" "${HERE}/_obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I"${HERE}/../../include")
SRCS=(core net encoder_fused conv1_wgrad loss quantile optim replay rng synth_env actor dp)
pids=()
for s in "${SRCS[@]}"; do
  src="${HERE}/${s}.hip"; obj="${HERE}/_obj/${s}.o"
  if [[ ! -f "${obj}" || "${src}" -nt "${obj}" || -n "$(find "${HERE}" -maxdepth 1 -name '*.h' -newer "${obj}" -print -quit)" || "${HERE}/../../include/agent0_hip.h" -nt "${obj}" ]]; then
    "${HIPCC}" "${FLAGS[@]}" -c "${src}" -o "${obj}" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "${p}" ]] && wait "${p}"; done
objs=(); for s in "${SRCS[@]}"; do objs+=("${HERE}/_obj/${s}.o"); done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT}/libagent0_hip.so" "${objs[@]}" -ldl
echo "built ${OUT}/libagent0_hip.so"
