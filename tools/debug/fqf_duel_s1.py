"""GPU diagnostics: the second step of the g6 fqf_duel fixture — where do device and oracle fraction losses part?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import recipe
from agent0_amd.ops import HipOps
from agent0_amd.deepq.engine import DeviceLearner
from agent0_amd.deepq.layout import NetLayout
from oracle import learner as olearner, nets
from oracle.losses import Hyper

hip = HipOps()
name = sys.argv[1] if len(sys.argv) > 1 else "fqf_duel"
spec = recipe.SPECS[name]; B = 16
hp = Hyper(double_q=True, n_step=3)
L = NetLayout.from_spec(spec)
sd_o, sd_t = recipe.make_state_dict(spec, 11), recipe.make_state_dict(spec, 12)
ora = olearner.OracleLearner(spec, sd_o, sd_t, hp, batch_size=B, target_update_freq=2)
dev = DeviceLearner(hip, L, B, n_step=3, double_q=True, target_update_freq=2)
dev.online.load_state_dict(sd_o); dev.target.load_state_dict(sd_t)
F, A = L.F, L.A
for s in range(2):
    frames = recipe.make_frames(B, 61 + s, spec.obs_shape); a, r, d, w = recipe.make_transitions(B, spec.action_dim, 62 + s)
    pre = {k: v.detach().clone() for k, v in ora.po.items()}
    nets.TAU_LOG = []
    res = ora.train(frames.reshape(B, -1), a, r, d.astype(np.float32), w, np.arange(B))
    log = nets.TAU_LOG; nets.TAU_LOG = None
    rand = [torch.from_numpy(np.ascontiguousarray(x.numpy()).reshape(-1).copy()).cuda() for pair in log for x in pair]
    D = lambda x: torch.from_numpy(x).cuda()
    loss, frac = dev.update(D(frames).reshape(-1), None, 2 * 28224, D(a.astype(np.int32)), D(r), D(d.astype(np.float32)), D(w), rand=rand)
    print("step", s, "q_loss", float((loss[:B].cpu() - res["q_loss"]).abs().max()), "frac", float((frac[:B].cpu() - res["fraction_loss"]).abs().max()))
    from util import golden
    gx = golden(f"g6_{name}_b16_dq1_n3")
    fx = torch.from_numpy(gx[f"s{s}::fraction_loss"])
    print("   vs fixture: oracle", float((res["fraction_loss"] - fx).abs().max()), "device", float((frac[:B].cpu() - fx).abs().max()), "threads", torch.get_num_threads())
    # the oracle's q at the interior fractions and at tau-hat, from the pre-step parameters
    ft = nets.normalize(torch.from_numpy(frames)); obs, nxt = torch.split(ft, 4, 1)
    taus, tau_hat = log[0]
    with torch.no_grad():
        feat = nets.encoder(pre, obs)
        q_hat = nets.head_iqn(pre, spec, feat, tau_hat.reshape(B, F, 1))[torch.arange(B), :, torch.from_numpy(a)]
        q_in = nets.head_iqn(pre, spec, feat, taus.reshape(B, F + 1)[:, 1:-1].reshape(B, F - 1, 1))[torch.arange(B), :, torch.from_numpy(a)]
    ar = torch.arange(B)
    qi_d = dev.ws_f.q[: B * (F - 1) * A].view(B, F - 1, A).cpu()[ar, :, torch.from_numpy(a)]
    qh_d = dev.ws_o.q[: B * F * A].view(B, F, A).cpu()[ar, :, torch.from_numpy(a)]
    print("   q interior |dev - oracle| max", float((qi_d - q_in).abs().max()), " q_hat", float((qh_d - q_hat).abs().max()), " scale", float(q_in.abs().max()))
    for tag, qi, qh in (("oracle", q_in, q_hat), ("device", qi_d, qh_d)):
        c1 = qi - torch.cat((qh[:, :1], qi[:, :-1]), 1)
        c2 = torch.cat((qi[:, 1:], qh[:, -1:]), 1) - qi
        print("  ", tag, "comparisons with |difference| < 1e-6:", int((c1.abs() < 1e-6).sum()), int((c2.abs() < 1e-6).sum()), "of", c1.numel(),
              " smallest |difference|", float(torch.minimum(c1.abs(), c2.abs()).min()))
    s1_o = q_in > torch.cat((q_hat[:, :1], q_in[:, :-1]), 1); s1_d = qi_d > torch.cat((qh_d[:, :1], qi_d[:, :-1]), 1)
    s2_o = q_in < torch.cat((q_in[:, 1:], q_hat[:, -1:]), 1); s2_d = qi_d < torch.cat((qi_d[:, 1:], qh_d[:, -1:]), 1)
    print("   decisions that differ:", int((s1_o != s1_d).sum()), int((s2_o != s2_d).sum()))
    L.pack({k: v.detach() for k, v in ora.po.items()}, dev.online.flat); L.pack({k: v.detach() for k, v in ora.pt.items()}, dev.target.flat)
    dev.online.refresh_wt(); dev.target.refresh_wt()
