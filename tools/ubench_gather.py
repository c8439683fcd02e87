"""GPU micro-benchmark (diagnostics): BASELINE metric 2, replay sample GB/s — a0_replay_sample_gather (index + metadata + row copy, one launch)
on a 200 k-row ring (11 GB), batch 512 rows of 56 448 B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from agent0_amd.ops import HipOps
hip = HipOps()
cap, rb, B = 200_000, 2 * 4 * 84 * 84, 512
frames = torch.empty(cap * rb, dtype=torch.uint8, device="cuda"); frames.random_(0, 256)
r_act, r_rew, r_done = hip.zeros(cap, dtype=torch.int32), hip.zeros(cap), hip.zeros(cap)
out = torch.empty(B * rb, dtype=torch.uint8, device="cuda")
io, slot, act, rew, done = hip.zeros(B, dtype=torch.int64), hip.zeros(B, dtype=torch.int32), hip.zeros(B, dtype=torch.int32), hip.zeros(B), hip.zeros(B)
pos = [0]
def run():
    hip.replay_sample_gather(0, pos[0] % (cap - B), cap, 12345, None, 1, None, cap, 0, cap, frames, rb, r_act, r_rew, r_done, None, B, out, io, slot, act, rew, done, None)
    pos[0] += B
for _ in range(5): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
s.record()
for _ in range(n): run()
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) * 1e-3 / n
print(f"sample+gather: {t * 1e6:.1f} us per batch, {B * rb / t / 1e9:.0f} GB/s sampled ({2 * B * rb / t / 1e12:.2f} TB/s of HBM traffic)")
