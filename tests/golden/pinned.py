"""A CPU-independent way to run torch's CPU kernels (fixture group G6P) — TEST INFRASTRUCTURE.

The reference's FQF numbers are not reproducible across CPUs from the second train step on: torch's CPU kernels round differently on different hosts (vectorised
exp / softmax per ISA, MKL / oneDNN code paths and blockings per CPU and thread count), the proposed fractions differ in the last ulp, cos(pi i tau) amplifies that
~200x, one Adam step at eps = 1e-2/16 turns it into 1e-4 of some parameters and the fraction loss — a sum of differences of neighbouring quantile values — into 6e-4
(profiles/r03_experiments.md).  A process started with ``ENV`` and configured by ``configure()`` takes the same code path on every x86-64 host:

  ATEN_CPU_CAPABILITY=default   ATen's scalar (non-vectorised) kernels: libm's expf / logf, no per-ISA vector math
  MKL_CBWR=COMPATIBLE           Intel MKL's conditional-numerical-reproducibility mode: one SSE2 code path on every CPU (also non-Intel)
  one thread                    no thread-count-dependent blocking or reduction order
  oneDNN disabled               convolutions fall back to ATen's im2col + MKL sgemm (torch.backends.mkldnn.flags(enabled=False))

``tests/golden/gen_golden.py g6p`` (run with ENV set: ``python tests/golden/pinned.py gen``) records the REFERENCE's FQF train steps in this mode as ``g6p_*.npz``;
``tests/pinned_oracle.py`` runs the oracle in the same mode in a child process wherever the tests run.  The environment variables must be set before torch is imported,
hence the child processes.
"""
import os
import subprocess
import sys

ENV = {"ATEN_CPU_CAPABILITY": "default", "MKL_CBWR": "COMPATIBLE", "OMP_NUM_THREADS": "1", "MKL_NUM_THREADS": "1", "A0_PINNED": "1",
       "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""}          # the child never touches a GPU


def check_env():
    missing = [k for k, v in ENV.items() if os.environ.get(k) != v]
    if missing:
        raise RuntimeError(f"pinned mode needs {missing} in the environment BEFORE torch is imported: start the process through tests/golden/pinned.py")


def configure():
    """Call right after importing torch in a process started with ``ENV``."""
    import torch
    check_env()
    torch.set_num_threads(1)
    torch.backends.mkldnn.enabled = False
    if "DEFAULT" not in torch.backends.cpu.get_cpu_capability().upper():
        raise RuntimeError(f"ATEN_CPU_CAPABILITY=default was not honoured: {torch.backends.cpu.get_cpu_capability()}")


def run(argv, **kw):
    """Runs ``python argv...`` in pinned mode as a child process."""
    env = dict(os.environ)
    env.update(ENV)
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    return subprocess.run([sys.executable] + list(argv), env=env, **kw)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    if sys.argv[1:] == ["gen"]:
        sys.exit(run([os.path.join(here, "gen_golden.py"), "g6p"]).returncode)
    sys.exit(run(sys.argv[1:]).returncode)
