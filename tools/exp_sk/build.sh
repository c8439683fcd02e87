#!/usr/bin/env bash
# Timing-only variants of the short-reduction forward kernel (agent0_amd/csrc/short_k_fwd.h), built from the product header with small source
# patches into tools/exp_sk/bin (git-ignored; travels to the GPU box): which part of the kernel bounds it?  Results are garbage by design;
# nothing here is linked into the library.  Run on the box: for v in base noload fewmfma noload_fewmfma fewstore; do tools/exp_sk/bin/sk_$v; done
set -e
cd "$(dirname "$0")"
mkdir -p bin
python3 - <<'PY'
import re
s = open('../../agent0_amd/csrc/short_k_fwd.h').read()
load = '''#pragma unroll
        for (int s = 0; s < 4; ++s) { raw[s][0] = *(const a0_f4*)(pn + 16 * s); raw[s][1] = *(const a0_f4*)(pn + 16 * s + 4); }
        request_m(mv, refill);'''
assert load in s
noload = lambda t: t.replace(load, 'asm volatile("" :: "v"(pn));')
fewmfma = lambda t: t.replace("for (int q = 0; q < 9; ++q)", "for (int q = 0; q < 1; ++q)")
fewstore = lambda t: t.replace("if (MODE == 0) *(float*)(y + o) = v;", "if (MODE == 0 && (r & 3) == 0) *(float*)(y + o) = v;")
for name, f in (("base", lambda t: t), ("noload", noload), ("fewmfma", fewmfma), ("noload_fewmfma", lambda t: fewmfma(noload(t))), ("fewstore", fewstore)):
    open(f'bin/short_k_fwd_{name}.h', 'w').write(f(s))
PY
for v in base noload fewmfma noload_fewmfma fewstore; do
  cp bin/short_k_fwd_$v.h bin/short_k_fwd_exp.h
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -I../../agent0_amd/csrc -I../../include -Ibin sk_bench.hip -o bin/sk_$v 2>&1 | grep -E "error" || true
done
ls bin
