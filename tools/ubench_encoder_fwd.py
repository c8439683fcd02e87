"""GPU micro-benchmark (diagnostics): the fused encoder forward alone, at several launch sizes (against a tuning build: python tools/with_lib.py <lib> tools/ubench_encoder_fwd.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
import recipe
from agent0_amd.ops import HipOps
from agent0_amd.deepq.engine import DeviceNet, Workspace
from agent0_amd.deepq.layout import NetLayout

hip = HipOps()
spec = recipe.NetSpec("dqn", 4)
L = NetLayout.from_spec(spec)
net = DeviceNet(hip, L, hip.net(4, 84, 84))
net.load_state_dict(recipe.make_state_dict(spec, 11))
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
out = []
for B in (128, 256, 512, 1024, 4096):
    frames = torch.randint(0, 256, (B * 28224,), dtype=torch.uint8, device="cuda")
    ws = Workspace(hip, L, B)
    t = timeit(lambda: hip.encoder_fwd_fused(net.net, net.wt, net.encoder_weights(), frames, None, 28224, 0, B, None, None, ws.act3))
    tk = timeit(lambda: hip.encoder_fwd_fused(net.net, net.wt, net.encoder_weights(), frames, None, 28224, 0, B, ws.act1, ws.act2, ws.act3))
    out.append(f"B={B}: {t:.1f} us (+store {tk:.1f})")
print(hip.build_info(), " | ".join(out))
