import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch, recipe
from agent0_amd.ops import HipOps
from agent0_amd.deepq.engine import DeviceNet, Workspace
from agent0_amd.deepq.layout import NetLayout
hip = HipOps(); spec = recipe.NetSpec("dqn", 4); L = NetLayout.from_spec(spec)
net = DeviceNet(hip, L, hip.net(4, 84, 84)); net.load_state_dict(recipe.make_state_dict(spec, 11))
B = 5
frames = torch.randint(0, 256, (B * 28224,), dtype=torch.uint8, device="cuda")
wf, wu = Workspace(hip, L, B), Workspace(hip, L, B)
hip.encoder_fwd_fused(net.net, net.wt, net.encoder_weights(), frames, None, 28224, 0, B, wf.act1, wf.act2, wf.act3)
hip.encoder_fwd(net.net, net.encoder_weights(), frames, None, 28224, 0, B, wu.act1, wu.act2, wu.act3)
torch.cuda.synchronize()
for name in ("act1", "act2", "act3"):
    a, b = getattr(wf, name), getattr(wu, name)
    d = (a - b).abs()
    print(name, "max err", float(d.max()), "scale", float(b.abs().max()), "frac bad", float((d > 1e-5 * b.abs().max()).float().mean()))
a2f = wf.act2.view(B, 81, 64); a2u = wu.act2.view(B, 81, 64)
bad = ((a2f - a2u).abs() > 1e-4).nonzero()
print("act2 bad count", bad.shape[0], bad[:10].tolist())
a3f = wf.act3.view(B, 49, 64); a3u = wu.act3.view(B, 49, 64)
bad = ((a3f - a3u).abs() > 1e-4)
print("bad by position m:", bad.float().mean(dim=(0, 2)).cpu().numpy().round(2).tolist())
print("bad by channel n:", bad.float().mean(dim=(0, 1)).cpu().numpy().round(2).tolist())
print("bad by obs:", bad.float().mean(dim=(1, 2)).cpu().numpy().round(2).tolist())
print("sample got/want:", a3f[0, 0, :6].tolist(), a3u[0, 0, :6].tolist())
