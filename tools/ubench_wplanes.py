"""GPU diagnostics (round 6): a0_dense_fwd against a0_dense_fwd_wplanes (the weight operand as pre-split bf16 term planes) — bit equality and kernel time.
usage: ubench_wplanes.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agent0_amd.ops import HipOps
hip = HipOps()
N, K = 512, 3136
for R in [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768]:
    X = torch.randn(R * K, device="cuda").clamp_min(0); W = torch.randn(N * K, device="cuda") * 0.02; b = torch.randn(N, device="cuda") * 0.1
    Y0, Y1 = torch.empty(R * N, device="cuda"), torch.empty(R * N, device="cuda")
    planes = torch.empty(hip.weight_planes_words(N, K), dtype=torch.int32, device="cuda")
    assert hip.dense_fwd_wplanes_ok(R, N, K) and hip.dense_fwd_scratch(R, N, K) == 0
    hip.split_planes(W, planes, N, K)
    sc = torch.empty(4, device="cuda")
    res = {}
    for name, fn in (("fp32 W", lambda: hip.dense_fwd(X, K, W, b, Y0, R, N, K, True, sc)), ("planes", lambda: hip.dense_fwd_wplanes(X, K, planes, b, Y1, R, N, K, True)),
                     ("fp32 W", lambda: hip.dense_fwd(X, K, W, b, Y0, R, N, K, True, sc)), ("planes", lambda: hip.dense_fwd_wplanes(X, K, planes, b, Y1, R, N, K, True))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): hip.split_planes(W, planes, N, K)
    e1.record(); torch.cuda.synchronize()
    print(f"rows {R}: us per launch {res}; split_planes {e0.elapsed_time(e1) / 20 * 1e3:.1f} us; bit-identical: {torch.equal(Y0, Y1)}; products {hip.x9_products()}", flush=True)
