"""GPU micro-benchmark (diagnostics): the actor's fc1 GEMM + tail (a0_actor_qhead) at E = 256 / 512 rows, and a whole actor step graph.
Knobs: A0_FC1_VARIANT, A0_FC1_WGS (csrc/net.hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agent0_amd.ops import HipOps
hip = HipOps()
K = 3136


def timeit(run, n=200):
    for _ in range(10): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): run()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n // 20): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for E in (256, 512):
    feat = torch.randn(E * K, device="cuda").abs(); W1 = torch.randn(512 * K, device="cuda") * 0.02; b1 = torch.zeros(512, device="cuda")
    W2 = torch.randn(32 * 512, device="cuda") * 0.05; b2 = torch.zeros(32, device="cuda")
    sc = torch.empty(hip.actor_qhead_scratch(E, K), device="cuda")
    act = torch.zeros(E, dtype=torch.int32, device="cuda"); qm = torch.zeros(E, device="cuda")
    t = timeit(lambda: hip.actor_qhead(feat, E, K, W1, b1, W2, b2, 4, False, sc, 1, 2, 1, 0, 0, 0.1, act, qm))
    print(f"E={E} variant={os.environ.get('A0_FC1_VARIANT', '0')} wgs={os.environ.get('A0_FC1_WGS', '256')}: fc1 GEMM + tail {t:.2f} us (slabs {sc.numel() // (E * 512)})", flush=True)
