#!/usr/bin/env bash
# Builds libagent0_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../lib"
mkdir -p "${OUT}" "${HERE}/_obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I"${HERE}/../../include")
SRCS=(core net encoder_fused conv1_wgrad conv23_wgrad loss quantile optim replay rng synth_env actor dp learner runtime)
pids=()
for s in "${SRCS[@]}"; do
  src="${HERE}/${s}.hip"; obj="${HERE}/_obj/${s}.o"
  if [[ ! -f "${obj}" || "${src}" -nt "${obj}" || -n "$(find "${HERE}" -maxdepth 1 -name '*.h' -newer "${obj}" -print -quit)" || "${HERE}/../../include/agent0_hip.h" -nt "${obj}" ]]; then
    # the compiler's per-kernel resource remarks go to _obj/<source>.res (registers, scratch, spills): see kernel_resources.txt below
    ( rc=0; "${HIPCC}" "${FLAGS[@]}" -Rpass-analysis=kernel-resource-usage -c "${src}" -o "${obj}" 2> "${HERE}/_obj/${s}.err" || rc=$?
      { grep -E "remark: +(Function Name|VGPRs|ScratchSize|VGPRs Spill|SGPRs Spill|LDS Size)" "${HERE}/_obj/${s}.err" || true; } | sed -E 's/^.*remark: +//; s/ \[-Rpass.*$//' > "${HERE}/_obj/${s}.res"
      # warnings and errors of this unit (everything that is not a resource remark), also when the compile succeeded
      { grep -A2 -E ": (warning|error|fatal error):" "${HERE}/_obj/${s}.err" || true; } >&2
      if [[ $rc -ne 0 ]]; then rm -f "${obj}"; fi
      exit $rc ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "${p}" ]] && wait "${p}"; done
# one line per kernel: name | VGPRs | scratch bytes per lane | VGPR spills | SGPR spills | LDS bytes (tests/test_abi_and_host.py: no kernel may use scratch)
for s in "${SRCS[@]}"; do
  [[ -f "${HERE}/_obj/${s}.res" ]] && awk -v S="${s}" '/^Function Name:/ {if (n) print line; n=$3; line=S" "n} /^VGPRs:/ {line=line" vgprs="$2} /^ScratchSize/ {line=line" scratch="$3} /^VGPRs Spill:/ {line=line" vgpr_spill="$3} /^SGPRs Spill:/ {line=line" sgpr_spill="$3} /^LDS Size/ {line=line" lds="$4} END {if (n) print line}' "${HERE}/_obj/${s}.res"
done > "${OUT}/kernel_resources.txt"
objs=(); for s in "${SRCS[@]}"; do objs+=("${HERE}/_obj/${s}.o"); done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT}/libagent0_hip.so" "${objs[@]}" -ldl
echo "built ${OUT}/libagent0_hip.so"
