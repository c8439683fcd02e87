"""CPU oracle for the agent0 deepq hot path — TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of every function on the hot
path of /root/reference ``agent0/deepq``.  It is
what the HIP kernels in ``agent0_amd/csrc`` are checked against.

Rules (enforced by tests/test_abi_and_host.py::test_product_never_imports_the_oracle_or_the_emulation):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * nothing under ``agent0_amd/`` or ``agent0/`` imports it — the product path
    raises when the HIP library is missing, it never falls back to this code.

Pinning: every function here is checked against golden vectors produced by
importing the reference itself (tests/golden/gen_golden.py -> tests/golden/*.npz,
see tests/test_oracle_golden.py).  Exception: the sum-tree (oracle/sumtree.c, driven
through oracle/core.py) has NO reference counterpart — the reference samples
uniformly (trainer.py:63-72) and its only proportional sampler is dead code
(replay.py:39-43) — so the sum-tree is "parity unpinned" against the reference;
its contract is defined here and the GPU must match it bit-for-bit.

Floating-point functions are written with torch CPU fp32 ops (the reference's
own arithmetic library, same version as the goldens); integer/byte/index
functions are numpy or plain C.
"""
