"""GPU micro-benchmark (diagnostics): the dense GEMMs (fc1 forward / data gradient / weight gradient) at the actor, DQN-learner and
IQN-learner row counts, on both matrix pipes (a0_gemm_mode 1 = split-operand bf16, 0 = fp32 fmaf chain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agent0_amd.ops import HipOps
hip = HipOps()
K, N = 3136, 512


def timeit(run, n=50):
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for R in (256, 512, 8192, 32768):
    X = torch.randn(R * K, device="cuda"); W = torch.randn(N * K, device="cuda") * 0.02; b = torch.zeros(N, device="cuda"); Y = torch.empty(R * N, device="cuda")
    dY = torch.randn(R * N, device="cuda"); dX = torch.empty(R * K, device="cuda"); G = torch.empty(N * K + N, device="cuda")
    sc = torch.empty(max(hip.dense_fwd_scratch(R, N, K), 4), device="cuda")
    sl = torch.empty(max(hip.dense_wgrad_scratch(R, N, K), 4), device="cuda")
    fl = 2 * R * N * K
    for mode in (1, 0):
        hip.gemm_mode(mode)
        t_f = timeit(lambda: hip.dense_fwd(X, K, W, b, Y, R, N, K, True, sc))
        t_d = timeit(lambda: hip.dense_dgrad(dY, W, X, dX, R, N, K))
        t_w = timeit(lambda: hip.dense_wgrad(dY, X, K, G, R, N, K, sl))
        print(f"R={R:6d} mode={'x9 ' if mode else 'f32'}: fwd {t_f:8.1f} us ({fl / t_f * 1e-6:6.1f} TF/s)  dgrad {t_d:8.1f} us ({fl / t_d * 1e-6:6.1f})  "
              f"wgrad {t_w:8.1f} us ({fl / t_w * 1e-6:6.1f})", flush=True)
