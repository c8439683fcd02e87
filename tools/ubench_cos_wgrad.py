"""GPU micro-benchmark (diagnostics): the cosine embedding's weight gradient ([3136][64] over B*N rows) and forward at the quantile networks' row counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agent0_amd.ops import HipOps
hip = HipOps()
N, K = 3136, 64


def timeit(run, n=30):
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for R in (8192, 16384, 32768):
    X = torch.randn(R * K, device="cuda"); dY = torch.randn(R * N, device="cuda"); G = torch.empty(N * K + N, device="cuda")
    sl = torch.empty(max(hip.dense_wgrad_scratch(R, N, K), 4), device="cuda")
    t = timeit(lambda: hip.dense_wgrad(dY, X, K, G, R, N, K, sl))
    print(f"R={R}: cos-embedding wgrad {t:.1f} us ({2 * R * N * K / t * 1e-6:.1f} TFLOP/s), slabs {sl.numel() * 4 / 1e6:.1f} MB")
