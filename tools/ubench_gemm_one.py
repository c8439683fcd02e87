"""GPU diagnostics: one dense GEMM shape on one matrix pipe, a few launches (driver for tools/pmc_gemm.sh).
usage: ubench_gemm_one.py OP(fwd|dgrad|wgrad) R MODE(1|0) [N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agent0_amd.ops import HipOps
op, R, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N, K = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (512, 3136)
hip = HipOps()
hip.gemm_mode(mode)
X = torch.randn(R * K, device="cuda"); W = torch.randn(N * K, device="cuda") * 0.02; b = torch.zeros(N, device="cuda"); Y = torch.empty(R * N, device="cuda")
dY = torch.randn(R * N, device="cuda"); dX = torch.empty(R * K, device="cuda"); G = torch.empty(N * K + N, device="cuda")
sc = torch.empty(max(hip.dense_fwd_scratch(R, N, K), 4), device="cuda")
sl = torch.empty(max(hip.dense_wgrad_scratch(R, N, K), 4), device="cuda")
run = {"fwd": lambda: hip.dense_fwd(X, K, W, b, Y, R, N, K, True, sc), "dgrad": lambda: hip.dense_dgrad(dY, W, X, dX, R, N, K),
       "wgrad": lambda: hip.dense_wgrad(dY, X, K, G, R, N, K, sl)}[op]
for _ in range(6): run()
torch.cuda.synchronize()
