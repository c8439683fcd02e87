"""GPU micro-benchmark (diagnostics): sum-tree priority update (512 random leaves) and rollout insert (20 480-leaf ring range), 1 M leaves."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agent0_amd.ops import HipOps
hip = HipOps()
cap2 = 1 << 20
tree = torch.rand(2 * cap2, device="cuda")
hip.sumtree_rebuild(tree, cap2)
idx = torch.randint(0, 1000000, (512,), device="cuda", dtype=torch.int64)
val = torch.rand(512, device="cuda")
v1 = torch.rand(1, device="cuda")
def t(f, n=200):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
print("set 512 random: %.1f us" % t(lambda: hip.sumtree_set(tree, cap2, idx, val, 512)))
print("set range 20480: %.1f us" % t(lambda: hip.sumtree_set_range(tree, cap2, 12345, 20480, 1000000, v1)))
