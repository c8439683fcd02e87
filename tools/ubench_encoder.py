"""GPU micro-benchmark (diagnostics): fused encoder vs the three implicit GEMMs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import recipe
from agent0_amd.ops import HipOps
from agent0_amd.deepq.engine import DeviceNet, Workspace
from agent0_amd.deepq.layout import NetLayout

hip = HipOps()
spec = recipe.NetSpec("dqn", 4)
L = NetLayout.from_spec(spec)
net = DeviceNet(hip, L, hip.net(4, 84, 84))
net.load_state_dict(recipe.make_state_dict(spec, 11))
for B in (256, 512, 4096):
    frames = torch.randint(0, 256, (B * 28224,), dtype=torch.uint8, device="cuda")
    ws = Workspace(hip, L, B)
    def timeit(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    t_f = timeit(lambda: hip.encoder_fwd_fused(net.net, net.wt, net.encoder_weights(), frames, None, 28224, 0, B, None, None, ws.act3))
    t_fk = timeit(lambda: hip.encoder_fwd_fused(net.net, net.wt, net.encoder_weights(), frames, None, 28224, 0, B, ws.act1, ws.act2, ws.act3))
    t_u = timeit(lambda: hip.encoder_fwd(net.net, net.encoder_weights(), frames, None, 28224, 0, B, ws.act1, ws.act2, ws.act3))
    gf = B * 15.47e6 / 1e9
    print(f"B={B} stages={os.environ.get('A0_FUSED_STAGES','7')}: fused {t_f:.1f} us ({gf/t_f*1e-3*1e3:.1f} TF/s), fused+store {t_fk:.1f} us, unfused {t_u:.1f} us")

# fused data gradients (conv3 + conv2), B = 512
B = 512
ws = Workspace(hip, L, B, grads=True)
frames = torch.randint(0, 256, (B * 28224,), dtype=torch.uint8, device="cuda")
net.encode(ws, frames, None, 28224, 0, B, keep=True)
ws.d3.normal_()
t_d = timeit(lambda: hip.encoder_dgrad_fused(net.net, net.wt, ws.d3, ws.act1, ws.act2, B, ws.d2, ws.d1))
print(f"B={B}: fused dgrad {t_d:.1f} us ({B * 12.5e6 / t_d * 1e-6:.1f} TF/s)")
