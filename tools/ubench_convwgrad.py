"""GPU micro-benchmark (diagnostics): the three convolution weight gradients + their reduction (a0_net_encoder_wgrad) at B = 512."""
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
B = 512
frames = torch.randint(0, 256, (B * 2 * 28224,), dtype=torch.uint8, device="cuda")
ws = Workspace(hip, L, B, grads=True)
for t in (ws.act1, ws.act2, ws.d3, ws.d2, ws.d1):
    t.normal_()
g = torch.empty(L.n_params_padded, device="cuda")
g1, g2, g3 = g[L.blocks["conv1"].all], g[L.blocks["conv2"].all], g[L.blocks["conv3"].all]
slabs = torch.empty(max(hip.encoder_bwd_scratch(net.net, B), 4), device="cuda")
run = lambda: hip.encoder_wgrad(net.net, net.encoder_weights(), frames, None, 2 * 28224, 0, B, ws.act1, ws.act2, ws.d3, ws.d2, ws.d1, g1, g2, g3, slabs)
for _ in range(5): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100): run()
e.record(); torch.cuda.synchronize()
print(f"T64={os.environ.get('A0_CONV_WGRAD_T64', '0')} GEMM={os.environ.get('A0_GEMM', 'x9')}: conv1+conv2+conv3 weight gradients + reduction {s.elapsed_time(e) * 10:.1f} us per call, "
      f"slabs {slabs.numel() * 4 / 1e6:.1f} MB")
