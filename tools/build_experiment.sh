#!/usr/bin/env bash
# Diagnostics: a tuning build of the library from a PATCHED copy of the sources (the product tree is not touched).
# usage: tools/build_experiment.sh <name> <patch under tools/patches/> [-DFOO=1 ...]     -> tools/variants/libagent0_hip_<name>.so
# The timing-only switches that used to live in encoder_fused.hip (half the A-fragment reads, half the weight loads, aliased LDS regions:
# results are garbage, the instruction stream is representative) are tools/patches/encoder_timing_experiments.patch.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME="$1"; PATCH="$2"; shift 2
TMP="$(mktemp -d /tmp/a0_exp_XXXX)"
mkdir -p "${TMP}/agent0_amd" "${TMP}/tools"
cp -r "${ROOT}/agent0_amd/csrc" "${TMP}/agent0_amd/csrc"; rm -rf "${TMP}/agent0_amd/csrc/_obj"*
cp -r "${ROOT}/include" "${TMP}/include"
cp "${ROOT}/tools/build_variant.sh" "${TMP}/tools/"
( cd "${TMP}" && patch -p0 < "${ROOT}/tools/patches/${PATCH}" )
bash "${TMP}/tools/build_variant.sh" "${NAME}" "$@"
mkdir -p "${ROOT}/tools/variants" && cp "${TMP}/tools/variants/libagent0_hip_${NAME}.so" "${ROOT}/tools/variants/"
rm -rf "${TMP}"
echo "built ${ROOT}/tools/variants/libagent0_hip_${NAME}.so"
