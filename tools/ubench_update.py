"""GPU micro-benchmark (diagnostics): one whole DQN update at B = 512 (BASELINE configs[1]) replayed from its hipGraph, and its backward half."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import recipe
from agent0_amd.ops import HipOps
from agent0_amd.deepq.engine import DeviceLearner
from agent0_amd.deepq.layout import NetLayout

hip = HipOps()
algo = sys.argv[1] if len(sys.argv) > 1 else "dqn"
spec = recipe.NetSpec(algo, 4, **({"num_atoms": 51} if algo == "c51" else {}))
L = NetLayout.from_spec(spec)
B = 512
dev = DeviceLearner(hip, L, B)
dev.online.load_state_dict(recipe.make_state_dict(spec, 11)); dev.target.load_state_dict(recipe.make_state_dict(spec, 12))
frames = torch.randint(0, 256, (B * 2 * 28224,), dtype=torch.uint8, device="cuda")
a = torch.randint(0, 4, (B,), dtype=torch.int32, device="cuda"); r = torch.randn(B, device="cuda"); d = torch.zeros(B, device="cuda"); w = torch.ones(B, device="cuda")


def graphed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    return g.replay


def timeit(run, n=100):
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


upd = graphed(lambda: dev.update(frames, None, 2 * 28224, a, r, d, w))
fwd = graphed(lambda: dev.forward_dense(frames, None, 2 * 28224, a, r, d, w))
bwd = graphed(lambda: dev.backward_encoder())
app = graphed(lambda: dev.apply())
print(f"{algo} B={B} NO_BRANCH={os.environ.get('A0_NO_BRANCH', '-')}: update {timeit(upd):.1f} us = forward+dense backward {timeit(fwd):.1f} + encoder backward {timeit(bwd):.1f} + apply {timeit(app):.1f}")
