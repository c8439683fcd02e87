"""GPU micro-benchmark (diagnostics): the cosine embedding forward ([rows][64] x [3136][64]^T, relu, times the state features) at the quantile
networks' row counts: short-reduction kernel (default) or the general GEMM (A0_NO_SHORT_K=1), in the plain / mul / keep modes."""
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


for R, group in ((8192, 32), (16384, 32), (32768, 64)):
    X = torch.rand(R * K, device="cuda") * 2 - 1; W = torch.randn(N * K, device="cuda") * 0.1; b = torch.randn(N, device="cuda") * 0.1
    M = torch.rand((R // group) * N, device="cuda")
    Y = torch.empty(R * N, device="cuda"); E = torch.empty(R * N, device="cuda")
    t0 = timeit(lambda: hip.dense_fwd(X, K, W, b, Y, R, N, K, True, None))
    t1 = timeit(lambda: hip.dense_fwd_mul(X, K, W, b, M, group, Y, R, N, K, True))
    line = f"R={R}: plain {t0:.1f} us ({R * N * 4 / t0 * 1e-6:.2f} TB/s of output), mul {t1:.1f} us"
    if hip.dense_fwd_mul_keep_ok(R, N, K, K):
        t2 = timeit(lambda: hip.dense_fwd_mul_keep(X, K, W, b, M, group, E, Y, R, N, K, True))
        line += f", keep (two outputs) {t2:.1f} us ({2 * R * N * 4 / t2 * 1e-6:.2f} TB/s)"
    else:
        t2 = timeit(lambda: (hip.dense_fwd(X, K, W, b, E, R, N, K, True, None), hip.hadamard_fwd(E, M, Y, R // group, group, N)))
        line += f", dense + hadamard {t2:.1f} us"
    print(line)
