"""GPU micro-benchmark (diagnostics): fc1 forward (512 x 3136) on the split-operand kernel at the quantile networks' row counts only (for tuning builds run through tools/with_lib.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agent0_amd.ops import HipOps
hip = HipOps()
K, N = 3136, 512


def timeit(run, n=30):
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for R in (8192, 32768):
    X = torch.randn(R * K, device="cuda"); W = torch.randn(N * K, device="cuda") * 0.02; b = torch.zeros(N, device="cuda"); Y = torch.empty(R * N, device="cuda")
    dY = torch.randn(R * N, device="cuda"); dX = torch.empty(R * K, device="cuda")
    t_f = timeit(lambda: hip.dense_fwd(X, K, W, b, Y, R, N, K, True, None))
    t_d = timeit(lambda: hip.dense_dgrad(dY, W, None, dX, R, N, K))
    print(f"R={R:6d}: fwd {t_f:8.1f} us ({2 * R * N * K / t_f * 1e-6:6.1f} TF/s)  dgrad {t_d:8.1f} us", flush=True)
