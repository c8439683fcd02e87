"""GPU micro-benchmark (diagnostics): fc1 forward (split-K implicit GEMM + slab reduce) at the actor / learner row counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agent0_amd.ops import HipOps
hip = HipOps()
K, N = 3136, 512
for R in (256, 512):
    X = torch.randn(R * K, device="cuda"); W = torch.randn(N * K, device="cuda") * 0.02; b = torch.zeros(N, device="cuda"); Y = torch.empty(R * N, device="cuda")
    sc = torch.empty(max(hip.dense_fwd_scratch(R, N, K), 4), device="cuda")
    def run(): hip.dense_fwd(X, K, W, b, Y, R, N, K, True, sc)
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100): run()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) * 10
    print(f"splits={os.environ.get('A0_FWD_SPLITS', 'auto')} R={R}: {t:.1f} us  ({2 * R * N * K / t * 1e-6:.1f} TF/s)  scratch={sc.numel() * 4 / 1e6:.1f} MB")
