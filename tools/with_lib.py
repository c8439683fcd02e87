"""Diagnostics only: run a Python script against a tuning build of the library (tools/build_variant.sh).

usage: python tools/with_lib.py <path/to/libagent0_hip_NAME.so> <script.py> [args...]

The product loader (agent0_amd/_abi.py) refuses any library whose a0_build_info() is not "default"; this wrapper loads the given
file with allow_variant=True BEFORE the script imports anything, so every later _abi.load() returns it.  Results of such a run are
timings for experiments, never parity evidence."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: the library must bind to the HIP runtime PyTorch has loaded, as it does in the product's import order)
from agent0_amd import _abi  # noqa: E402

lib_path, script = os.path.abspath(sys.argv[1]), sys.argv[2]
lib = _abi.load(lib_path, allow_variant=True)
print(f"with_lib: {lib_path} [{(lib.a0_build_info() or b'').decode()}]", file=sys.stderr)
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, run_name="__main__")
