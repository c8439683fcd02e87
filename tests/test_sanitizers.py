"""CPU: the sanitizer job (tools/asan.sh; SURVEY.md §5 build stance) — AddressSanitizer + UndefinedBehaviorSanitizer builds of the oracle's C sources and of the host
emulation (the product's own layer orchestration headers compiled for the CPU) under the tests that exercise them, the plain-C hosts compiled with the same flags.
GPU sanitizers do not exist on the pool; this is the coverage there is.  A control shows that the set-up does catch an error."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.pop("A0_SANITIZE", None)
    return env


def test_cpu_sanitizer_job_is_clean():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan.sh")], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    # the instrumented libraries are the ones that were loaded
    probe = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
             "from oracle import core; import cpu_ops\n"
             "core.lib(); cpu_ops.CpuOps()\n"
             "maps = open('/proc/self/maps').read()\n"
             "assert '_build/asan/liba0oracle.so' in maps and '_build/asan/libhost_emul.so' in maps and 'libasan' in maps\n"
             "print('instrumented')\n" % (ROOT, os.path.join(ROOT, "tests")))
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(_env(), A0_SANITIZE="1", LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "instrumented" in r.stdout, r.stdout + r.stderr


def test_the_sanitizer_set_up_catches_an_out_of_bounds_read(tmp_path):
    """Control: a deliberately wrong C function built and loaded exactly as tools/asan.sh builds and loads the real ones must stop the process."""
    src = tmp_path / "bad.c"
    src.write_text("#include <stdlib.h>\nint bad(int n) { int* a = (int*)malloc(4 * sizeof(int)); int v = a[n]; free(a); return v; }\n")
    so = tmp_path / "libbad.so"
    subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fPIC", "-shared", "-o", str(so), str(src)])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(_env(), LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0")
    r = subprocess.run([sys.executable, "-c", "import ctypes; ctypes.CDLL(%r).bad(7)" % str(so)], env=env, capture_output=True, text=True, timeout=120)
    # (UBSan's object-size check fires first on this one; either report stops the process)
    assert r.returncode != 0 and ("heap-buffer-overflow" in r.stderr or "runtime error: load of address" in r.stderr), r.stderr[-2000:]
