#!/usr/bin/env bash
# CPU sanitizer job: AddressSanitizer + UndefinedBehaviorSanitizer builds of every native source that runs without a GPU —
# the oracle's C (oracle/sumtree.c, philox.c, synth_env.c), the host emulation of the layer orchestration (tests/host_emul.cpp, which compiles the product's own
# net_impl.h / operands.h / net_tables.h for the CPU) — run under the tests that exercise them; the two plain-C hosts of the handle API (tests/c_host_demo.c,
# c_host_loop.c: every statement of theirs talks to the GPU) are compiled with the same flags and -Wall -Wextra -Werror.  GPU sanitizers are not available on
# the pool (gpurun refuses ASan / XNACK runs), so this is the sanitizer coverage there is.   usage: bash tools/asan.sh [pytest args]
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "${ROOT}"
make -C oracle -s asan
SAN=(-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined)
mkdir -p tests/_build/asan
for src in c_host_demo c_host_loop; do
  gcc "${SAN[@]}" -Wall -Wextra -Wno-missing-field-initializers -Werror -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -c "tests/${src}.c" -o "tests/_build/asan/${src}.o"
done
LIBASAN="$(gcc -print-file-name=libasan.so)"
# detect_leaks=0: the interpreter is not instrumented and "leaks" its own arenas at exit; halt_on_error so that the first report fails the run
export A0_SANITIZE=1 LD_PRELOAD="${LIBASAN}" ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0" UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
python -m pytest tests/test_oracle_core.py tests/test_engine_emul.py -x -q -p no:cacheprovider "$@"
