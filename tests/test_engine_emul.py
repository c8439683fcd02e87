"""CPU: agent0_amd.deepq.engine (the product's layer composition) driven by the emulation backend, against the oracle.

What this covers without a GPU: the packed parameter layout and its round trip to the reference state_dict, the
gather tables / phase decomposition / split-K and slab logic of the shared C++ orchestration (net_impl.h via
tests/host_emul.cpp), forward/backward wiring of every head (dueling, NoisyNet, IQN, FQF), loss-gradient formulas,
Adam / RMSprop / NaN-skip / target-sync sequencing.  The MFMA tile code itself is only reachable on the GPU.
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import recipe
from recipe import NetSpec, SPECS
from agent0_amd.deepq.engine import DeviceLearner, DeviceNet, Workspace
from agent0_amd.deepq.layout import NetLayout
from cpu_ops import CpuOps
from oracle import learner as olearner, nets
from oracle.losses import Hyper
from util import assert_close

TINY = (4, 36, 36)
CASES = {
    "dqn": NetSpec("dqn", 4, obs_shape=TINY),
    "dqn_duel": NetSpec("dqn", 5, dueling=True, obs_shape=TINY),
    "c51": NetSpec("c51", 3, num_atoms=51, obs_shape=TINY),
    "c51_duel_noisy": NetSpec("c51", 6, dueling=True, noisy=True, obs_shape=TINY),
    "qr": NetSpec("qr", 3, num_atoms=200, obs_shape=TINY),
    "qr_duel_noisy": NetSpec("qr", 2, dueling=True, noisy=True, num_atoms=200, obs_shape=TINY),
    "iqn": NetSpec("iqn", 9, obs_shape=TINY),
    "iqn_duel": NetSpec("iqn", 3, dueling=True, obs_shape=TINY),
    "fqf": NetSpec("fqf", 4, obs_shape=TINY),
    "dqn_noisy": NetSpec("dqn", 4, noisy=True, obs_shape=TINY),
    "mdqn": NetSpec("mdqn", 5, obs_shape=TINY),
    # the reference's suite configuration (README.md:62-112: fqf + double-Q + dueling) and the 18-action games (Seaquest)
    "fqf_duel": NetSpec("fqf", 9, dueling=True, obs_shape=TINY),
    "dqn_duel_a18": NetSpec("dqn", 18, dueling=True, obs_shape=TINY),
    "dqn_a24": NetSpec("dqn", 24, obs_shape=TINY),             # upper end of the head/loss kernel's A + dueling <= 24 range
    "dqn_duel_a24": NetSpec("dqn", 24, dueling=True, obs_shape=TINY),      # just beyond it: the generic path
    "fqf_duel_a18": NetSpec("fqf", 18, dueling=True, obs_shape=TINY),
    "c51_duel_a18": NetSpec("c51", 18, dueling=True, num_atoms=51, obs_shape=TINY),
    "dqn_odd": NetSpec("dqn", 4, obs_shape=(4, 44, 52)),   # non-square, odd conv1 output (10x12 -> 4x5 -> 2x3)
}


@pytest.fixture(scope="module")
def ops():
    return CpuOps()


@pytest.mark.parametrize("name", list(CASES))
def test_layout_roundtrip(name):
    spec = CASES[name]
    L = NetLayout.from_spec(spec)
    sd = {k: torch.from_numpy(v) for k, v in recipe.make_state_dict(spec, 5).items()}
    flat = torch.zeros(L.n_params_padded)
    L.pack(sd, flat)
    back = L.unpack(flat)
    train = [k for k in sd if not recipe.is_buffer(k)]
    assert set(back) == set(train)
    for k in train:
        assert back[k].shape == sd[k].shape and torch.equal(back[k], sd[k]), k
    # every float of the flat buffer is either a mapped parameter or zero padding
    assert int((flat != 0).sum()) == sum(int((sd[k] != 0).sum()) for k in train)


def noise_draws(spec, seed):
    g = recipe.gen(seed)
    L = NetLayout.from_spec(spec)
    out = []
    for prefix, block, r0, r1, in_f in L.noise_modules:
        out += [(g.standard_normal(in_f) * 0.1).astype(np.float32), (g.standard_normal(r1 - r0) * 0.1).astype(np.float32),
                (g.standard_normal(r1 - r0) * 0.1).astype(np.float32)]
    return out


def install_noise(net: DeviceNet, draws):
    it = iter(draws)
    for prefix, *_ in net.L.noise_modules:
        net.set_noise(prefix, next(it), next(it), next(it))


def check_forward(ops, name):
    spec = CASES[name]
    L = NetLayout.from_spec(spec)
    B = 5
    sd = recipe.make_state_dict(spec, 11)
    p = olearner.to_params(sd)
    net = DeviceNet(ops, L, ops.net(*spec.obs_shape))
    net.load_state_dict(sd)
    if spec.noisy:
        dr = noise_draws(spec, 3)
        install_noise(net, dr)
        net.compose_noise()
        it = iter(dr)
        for prefix in nets.dense_prefixes(spec):
            for leaf in ("noise_in", "noise_out_weight", "noise_out_bias"):
                p[f"{prefix}.{leaf}"] = torch.from_numpy(next(it))
            nets.compose_noise(p, prefix)
    frames = torch.from_numpy(recipe.make_frames(B, 21, spec.obs_shape))
    obs_bytes = int(np.prod(spec.obs_shape))
    n_tau = 7 if L.quantile else 1
    ws = Workspace(ops, L, B, n_tau if spec.algo != "fqf" else L.F)
    net.encode(ws, frames.reshape(-1).to(ops.device), None, 2 * obs_bytes, obs_bytes, B)      # st_next half
    x = nets.normalize(frames[:, spec.obs_shape[0]:])
    with torch.no_grad():
        feat, (a1, a2, a3) = nets.encoder(p, x, return_all=True)
    assert_close(ws.act1.view(B, L.H1, L.W1, 32).permute(0, 3, 1, 2), a1, 1e-5, 1e-6, "conv1")
    assert_close(ws.act2.view(B, L.H2, L.W2, 64).permute(0, 3, 1, 2), a2, 1e-5, 1e-6, "conv2")
    assert_close(ws.act3.view(B, L.H3, L.W3, 64).permute(0, 3, 1, 2), a3, 1e-5, 1e-6, "conv3")
    a_star = ops.zeros(B, dtype=torch.int32)
    qsel = ops.zeros(B * L.A)
    with torch.no_grad():
        if spec.algo == "iqn":
            taus = torch.from_numpy(recipe.gen(9).random((B, n_tau, 1), dtype=np.float32))
            q = net.head(ws, B, taus.reshape(-1).contiguous().to(ops.device), n_tau)
            assert_close(q[: B * n_tau * L.A].view(B, n_tau, L.A), nets.head_iqn(p, spec, feat, taus), 2e-5, 2e-6, "iqn q")
            net.select(ws, B, n_tau, a_star, qsel)
            want = nets.head_iqn(p, spec, feat, taus).mean(1)
        elif spec.algo == "fqf":
            net.fqf_taus(ws, B)
            t, th, _ = nets.fqf_prop_taus(p, spec, feat)
            assert_close(ws.tau_all.view(B, L.F + 1), t[:, :, 0], 1e-5, 1e-6, "taus")
            assert_close(ws.tau_hat.view(B, L.F), th[:, :, 0], 1e-5, 1e-6, "tau_hat")
            q = net.head(ws, B, ws.tau_hat, L.F)
            # cos(pi*64*tau) amplifies the 1e-7 differences of the two tau computations ~200x: compare at the SAME taus tightly
            q2 = net.head(ws, B, th.reshape(-1).contiguous().to(ops.device), L.F).clone()
            assert_close(q2[: B * L.F * L.A].view(B, L.F, L.A), nets.head_iqn(p, spec, feat, th), 2e-5, 2e-6, "fqf q_hat @ oracle taus")
            q = net.head(ws, B, ws.tau_hat, L.F)
            assert_close(q[: B * L.F * L.A].view(B, L.F, L.A), nets.head_iqn(p, spec, feat, th), 5e-4, 5e-5, "fqf q_hat")
            net.select(ws, B, L.F, a_star, qsel)
            want = nets.qval_from_feat(p, spec, feat)
        else:
            q = net.head(ws, B)
            assert_close(q[: B * L.A * L.T].view(B, L.A, L.T).squeeze(-1) if L.T == 1 else q[: B * L.A * L.T].view(B, L.A, L.T),
                         nets.forward(p, spec, x), 2e-5, 2e-6, "head out")
            atoms = nets.c51_atoms(spec).to(ops.device) if spec.algo == "c51" else None
            net.select(ws, B, 1, a_star, qsel, atoms=atoms)
            want = nets.qval(p, spec, x)
    assert_close(qsel.view(B, L.A), want, *((5e-4, 5e-5) if spec.algo == "fqf" else (2e-5, 2e-6)), "qval")
    assert torch.equal(a_star.long().cpu(), want.argmax(-1))


@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_oracle(ops, name):
    check_forward(ops, name)


def run_both(ops, name, B, double_q, n_step, steps=2, target_freq=2, resync=True, inject_taus=True, spec=None, hp=None, arbiter=False):
    spec = spec or CASES[name]
    L = NetLayout.from_spec(spec)
    hp = hp or Hyper(double_q=double_q, n_step=n_step, K=6, N=8, N_dash=5)
    sd_o, sd_t = recipe.make_state_dict(spec, 11), recipe.make_state_dict(spec, 12)
    ora = olearner.OracleLearner(spec, sd_o, sd_t, hp, batch_size=B, target_update_freq=target_freq)
    dev = DeviceLearner(ops, L, B, n_step=n_step, double_q=double_q, target_update_freq=target_freq, K=hp.K, N=hp.N, N_dash=hp.N_dash)
    dev.online.load_state_dict(sd_o)
    dev.target.load_state_dict(sd_t)
    obs_bytes = int(np.prod(spec.obs_shape))
    results = []
    for s in range(steps):
        frames = recipe.make_frames(B, 61 + s, spec.obs_shape)
        a, r, d, w = recipe.make_transitions(B, spec.action_dim, 62 + s)
        rand_np = None
        if spec.algo == "iqn":
            g = recipe.gen(70 + s)
            rand_np = [g.random((B, n, 1), dtype=np.float32) for n in (hp.K, hp.N_dash, hp.N)]
        no = nt = None
        if spec.noisy:
            no, nt = noise_draws(spec, 80 + s), noise_draws(spec, 90 + s)
            install_noise(dev.online, no)
            install_noise(dev.target, nt)
        nets.TAU_LOG = [] if (spec.algo == "fqf" and inject_taus) else None
        before = ({k: v.detach().clone() for k, v in ora.po.items()}, {k: v.detach().clone() for k, v in ora.pt.items()}) if arbiter else None
        res_o = ora.train(frames.reshape(B, -1), a, r, d.astype(np.float32), w, np.arange(B), rand=rand_np, noise_online=no, noise_target=nt)
        D = lambda t: t.to(ops.device)
        if nets.TAU_LOG is not None:
            # FQF: both sides evaluate q(tau) at the ORACLE's fractions (online net on obs, then the action-selection pass): cos(pi*64*tau)
            # amplifies the ulp-level differences of two softmax/cumsum evaluations ~200x, which would otherwise need a loose tolerance
            assert len(nets.TAU_LOG) == 2
            rand_np = [x.numpy() for pair in nets.TAU_LOG for x in pair]
        g64 = None
        if arbiter:      # the same step in float64 on the same fp32 inputs (oracle.learner.exact_gradients)
            g64, _ = olearner.exact_gradients(spec, hp, before[0], before[1], frames.reshape(B, -1), a, r, d.astype(np.float32), w,
                                              rand=rand_np if spec.algo == "iqn" else None, taus=nets.TAU_LOG)
        nets.TAU_LOG = None
        out = dev.update(D(torch.from_numpy(frames).reshape(-1)), None, 2 * obs_bytes, D(torch.from_numpy(a.astype(np.int32))), D(torch.from_numpy(r)),
                         D(torch.from_numpy(d.astype(np.float32))), D(torch.from_numpy(w)),
                         rand=None if rand_np is None else [D(torch.from_numpy(np.ascontiguousarray(x).reshape(-1).copy())) for x in rand_np])
        out = tuple(o.clone() for o in out) if isinstance(out, tuple) else out.clone()      # device buffers are reused by the next update
        got_p, got_t = dev.online.state_dict(), dev.target.state_dict()
        results.append((res_o, out, {k: v.clone() for k, v in ora.last_grads.items()}, dev.grads.clone(), got_p, got_t,
                        {k: v.detach().clone() for k, v in ora.po.items()}, {k: v.detach().clone() for k, v in ora.pt.items()}, g64))
        if resync:   # remove accumulated ulp-level drift so that every step is compared from identical parameters
            dev.online.L.pack({k: v.detach() for k, v in ora.po.items()}, dev.online.flat)
            dev.target.L.pack({k: v.detach() for k, v in ora.pt.items()}, dev.target.flat)
    return spec, L, ora, dev, results


TRAIN = [("dqn", 8, False, 1), ("dqn_duel", 8, True, 3), ("c51", 8, False, 1), ("c51_duel_noisy", 8, True, 3), ("qr", 6, False, 1),
         ("qr_duel_noisy", 6, True, 1), ("iqn", 6, False, 1), ("iqn_duel", 6, True, 3), ("fqf", 6, False, 1), ("fqf", 6, True, 3),
         ("dqn_noisy", 8, True, 1), ("dqn_odd", 8, True, 1), ("mdqn", 8, False, 3),
         ("fqf_duel", 6, True, 3), ("dqn_duel_a18", 8, True, 1), ("dqn_a24", 8, False, 1), ("dqn_duel_a24", 8, True, 3), ("fqf_duel_a18", 6, True, 3),
         ("c51_duel_a18", 6, True, 3)]


def check_update(ops, name, B, dq, n, **kw):
    """Losses, every gradient tensor, parameters and target after the optimizer step, at the same tolerances for all six learners (FQF
    included: the fractions are injected, see run_both)."""
    spec, L, ora, dev, results = run_both(ops, name, B, dq, n, **kw)
    steps = len(results)
    for s, (res_o, out, g_o, g_d, got, tgt, want_p, want_t, g64) in enumerate(results):
        loss_d, frac_d = (out if isinstance(out, tuple) else (out, None))
        assert_close(loss_d[:B], res_o["q_loss"], 5e-5, 5e-6, f"step {s} q_loss")
        if frac_d is not None:
            # the fraction loss sums 31 signed differences of quantile values: absolute error ~ a few ulp of |q| x 31
            assert_close(frac_d[:B], res_o["fraction_loss"], 5e-5, 2e-5, f"step {s} fraction_loss")
        # gradients, tensor by tensor, in the reference layout
        g_ref = L.unpack(g_d)
        for k, g in g_o.items():
            if g is None:
                continue
            scale = float(g.abs().max()) + 1e-12
            if g64 is None:
                assert_close(g_ref[k] / scale, g / scale, 0, 3e-5, f"step {s} grad {k}")
            else:
                # long reductions (``arbiter``): within 3e-5 of the tensor's max of the oracle's fp32 value, or — where two correct fp32
                # summation orders differ by more than that — at least as close to the float64 evaluation as twice the oracle's own error
                e_dev = float((g_ref[k].double().cpu() - g64[k]).abs().max())
                e_ora = float((g.double() - g64[k]).abs().max())
                e_rel = float((g_ref[k].cpu() - g).abs().max())
                assert e_rel <= 3e-5 * scale or e_dev <= 2.0 * e_ora, f"step {s} grad {k}: |hip - fp64| {e_dev:.3e}, |torch fp32 - fp64| {e_ora:.3e}, scale {scale:.3e}"
        # parameters after this step's optimizer update (Adam normalises the step size: compare absolutely)
        for src, ref, tag in ((got, want_p, "param"), (tgt, want_t, "target param")):
            for k in nets.trainable_keys(ref):
                assert_close(src[k], ref[k], 0, 2e-5, f"step {s} {tag} {k}")
    assert int(dev.state[1]) == ora.update_steps == steps


@pytest.mark.parametrize("name,B,dq,n", TRAIN)
def test_update_matches_oracle(ops, name, B, dq, n):
    check_update(ops, name, B, dq, n)


def device_relu_masks(dev: DeviceLearner, L: NetLayout, B: int):
    """The 0/1 ReLU decisions of the device's differentiated (online, current-observation) pass, in the oracle's layouts."""
    ws, R = dev.ws_o, dev.ws_o.R
    nchw = lambda t, h, w, c: (t[: B * h * w * c].view(B, h, w, c).permute(0, 3, 1, 2) > 0).float().cpu()
    m = {"conv1": nchw(ws.act1, L.H1, L.W1, 32), "conv2": nchw(ws.act2, L.H2, L.W2, 64), "conv3": nchw(ws.act3, L.H3, L.W3, 64),
         "fc1": (ws.h[: R * 512].view(R, 512) > 0).float().cpu()}
    if L.quantile:      # embedding columns are (h, w, c) on the device, (c, h, w) in the reference
        m["cos"] = (ws.emb[: R * L.feat].view(R, L.H3, L.W3, 64).permute(0, 3, 1, 2).reshape(R, L.feat) > 0).float().cpu()
    return m


def check_update_full_size(ops, spec, hp, B, seed=61, lr=5e-4):
    """One whole update at a batch size where reductions are long (B = 512: 204 800 products per conv1 weight, 32 768 rows per fc1 weight
    of the quantile networks), HIP against the oracle.  Two effects that the small cases do not show are separated out instead of being
    absorbed into a loose tolerance:

      * ReLU decisions.  With ~10^7 ReLUs in the differentiated pass a handful of pre-activations sit within fp32 rounding of zero, where
        two correct evaluations disagree on the 0/1 decision; each disagreement moves a weight gradient by one whole term.  The test (a)
        asserts that the device's decisions differ from the oracle's only at pre-activations below 1e-5 of the layer's largest, and at
        most for 1e-4 of them, and (b) has the oracle back-propagate through the device's decisions (oracle.nets.RELU_MASKS; forward
        values untouched).  Then losses agree to rtol 5e-5 and EVERY gradient tensor to 3e-5 of its largest element — the small cases'
        tolerances.
      * Adam's first step is lr * s(g), s(g) = g / (|g| + eps) with eps = 1e-2 / B = 2e-5: where |g| is below the gradient tolerance the
        step's sign is undetermined.  Parameters are compared at 2e-5 absolute plus what the two (already compared) gradients imply
        through that formula, lr * |s(g_hip) - s(g_oracle)|; the test asserts that this term exceeds 1e-4 (a fifth of a step) for fewer
        than 2 % of a tensor's elements.
    """
    L = NetLayout.from_spec(spec)
    sd_o, sd_t = recipe.make_state_dict(spec, 11), recipe.make_state_dict(spec, 12)
    dev = DeviceLearner(ops, L, B, n_step=hp.n_step, double_q=hp.double_q, target_update_freq=1, K=hp.K, N=hp.N, N_dash=hp.N_dash, lr=lr)
    dev.online.load_state_dict(sd_o)
    dev.target.load_state_dict(sd_t)
    frames = recipe.make_frames(B, seed, spec.obs_shape)
    a, r, d, w = recipe.make_transitions(B, spec.action_dim, seed + 1)
    rand_np = None
    if spec.algo == "iqn":
        g = recipe.gen(seed + 9)
        rand_np = [g.random((B, n, 1), dtype=np.float32) for n in (hp.K, hp.N_dash, hp.N)]
    args = (frames.reshape(B, -1), a, r, d.astype(np.float32), w, np.arange(B))
    taus = None
    if spec.algo == "fqf":      # the fractions both sides use (see run_both)
        nets.TAU_LOG = []
        olearner.OracleLearner(spec, sd_o, sd_t, hp, batch_size=B, target_update_freq=1).train(*args)
        taus, nets.TAU_LOG = nets.TAU_LOG, None
        rand_np = [x.numpy() for pair in taus for x in pair]
    D = lambda t: t.to(ops.device)
    no = nt = None
    if spec.noisy:          # NoisyNet: both sides get the same injected N(0, 0.1^2) draws (agent.py:125-127 resets both networks' noise per train call)
        no, nt = noise_draws(spec, 80), noise_draws(spec, 90)
        install_noise(dev.online, no)
        install_noise(dev.target, nt)
    out = dev.update(D(torch.from_numpy(frames).reshape(-1)), None, 2 * int(np.prod(spec.obs_shape)), D(torch.from_numpy(a.astype(np.int32))), D(torch.from_numpy(r)),
                     D(torch.from_numpy(d.astype(np.float32))), D(torch.from_numpy(w)),
                     rand=None if rand_np is None else [D(torch.from_numpy(np.ascontiguousarray(x).reshape(-1).copy())) for x in rand_np])
    loss_d, frac_d = (out if isinstance(out, tuple) else (out, None))
    ora = olearner.OracleLearner(spec, sd_o, sd_t, hp, batch_size=B, lr=lr, target_update_freq=1)
    nets.RELU_MASKS, nets.RELU_STATS = device_relu_masks(dev, L, B), {}
    try:
        res = ora.train(*args, rand=rand_np if spec.algo == "iqn" else None, noise_online=no, noise_target=nt)
    finally:
        nets.RELU_MASKS = None
    stats = dict(nets.RELU_STATS)
    assert set(stats) == set(["conv1", "conv2", "conv3", "fc1"] + (["cos"] if L.quantile else []))
    for name, (n_bad, n, worst) in stats.items():
        assert worst <= 1e-5 and n_bad <= 1e-4 * n, f"ReLU decisions of {name}: {n_bad}/{n} differ, largest |pre-activation| {worst:.2e} of the layer's largest"
    assert_close(loss_d[:B], res["q_loss"], 5e-5, 5e-6, "q_loss")
    if frac_d is not None:
        assert_close(frac_d[:B], res["fraction_loss"], 5e-5, 2e-5, "fraction_loss")
    g_dev = L.unpack(dev.grads)
    got, tgt = dev.online.state_dict(), dev.target.state_dict()
    eps = 1e-2 / B
    for k, g in ora.last_grads.items():
        if g is None:
            continue
        scale = float(g.abs().max()) + 1e-12
        assert_close(g_dev[k] / scale, g / scale, 0, 3e-5, f"grad {k}")
        step = lambda x: x / (x.abs() + eps)              # Adam's first step in units of lr (m = 0.1 g, v = 0.001 g^2, both bias-corrected)
        tol = 2e-5 + (0.0 if "fraction" in k else lr) * (step(g_dev[k].cpu()) - step(g)).abs()
        for src, ref, tag in ((got, ora.po, "param"), (tgt, ora.pt, "target param")):
            err = (src[k].cpu() - ref[k].detach()).abs()
            assert bool((err <= tol).all()), f"{tag} {k}: {int((err > tol).sum())}/{err.numel()} outside the Adam-sensitivity tolerance, worst {float((err - tol).max()):.3e}"
        assert float((tol > 1e-4).float().mean()) <= 2e-2, f"{k}: a tolerance above 1e-4 (a fifth of one Adam step) applies to more than 2 % of the elements"
    assert int(dev.state[1]) == ora.update_steps == 1
    return stats


def check_update_fqf_own_fractions(ops, spec, hp, B, seed=61, bounds=None):
    """One FQF update in which the DEVICE proposes its own fractions (a0_fqf_taus -> cosine embedding -> head -> losses -> gradients), against
    the oracle proposing its own.  The injected-fraction tests above compare q(tau) at bit-identical fractions and therefore cannot see a
    wiring error between a0_fqf_taus and the rest of the update; this one can.  The two softmax / cumsum evaluations differ by a few ulp
    (asserted: fractions to 1e-5 absolute) and cos(pi * i * tau), i <= 64, amplifies that ~200x into the quantile values, so the comparison
    is a bounded-outlier one: the median and the 99th percentile of the per-sample loss error (relative to the batch's mean |loss|), the worst
    element of every gradient tensor relative to the tensor's max, the share of parameters further than 2e-5 + a fifth of an Adam step from the
    oracle's.  A mis-wired fraction buffer moves all of these by O(1).  -> the measured metrics."""
    L = NetLayout.from_spec(spec)
    sd_o, sd_t = recipe.make_state_dict(spec, 11), recipe.make_state_dict(spec, 12)
    dev = DeviceLearner(ops, L, B, n_step=hp.n_step, double_q=hp.double_q, target_update_freq=1)
    dev.online.load_state_dict(sd_o)
    dev.target.load_state_dict(sd_t)
    frames = recipe.make_frames(B, seed, spec.obs_shape)
    a, r, d, w = recipe.make_transitions(B, spec.action_dim, seed + 1)
    D = lambda t: t.to(ops.device)
    out = dev.update(D(torch.from_numpy(frames).reshape(-1)), None, 2 * int(np.prod(spec.obs_shape)), D(torch.from_numpy(a.astype(np.int32))), D(torch.from_numpy(r)),
                     D(torch.from_numpy(d.astype(np.float32))), D(torch.from_numpy(w)), rand=None)
    loss_d, frac_d = out
    F = L.F
    dev_taus = [(dev.ws_o.tau_all[: B * (F + 1)].view(B, F + 1).cpu().clone(), dev.ws_o.tau_hat[: B * F].view(B, F).cpu().clone())]
    ws_sel = dev.ws_s if hp.double_q else dev.ws_t
    dev_taus.append((ws_sel.tau_all[: B * (F + 1)].view(B, F + 1).cpu().clone(), ws_sel.tau_hat[: B * F].view(B, F).cpu().clone()))
    ora = olearner.OracleLearner(spec, sd_o, sd_t, hp, batch_size=B, target_update_freq=1)
    nets.TAU_LOG = []
    try:
        res = ora.train(frames.reshape(B, -1), a, r, d.astype(np.float32), w, np.arange(B))
        log = nets.TAU_LOG
    finally:
        nets.TAU_LOG = None
    assert len(log) == 2
    m = {}
    for k, ((t_d, th_d), (t_o, th_o)) in enumerate(zip(dev_taus, log)):
        m[f"taus_{k}"] = max(float((t_d - t_o.reshape(B, F + 1)).abs().max()), float((th_d - th_o.reshape(B, F)).abs().max()))
        assert m[f"taus_{k}"] <= 1e-5, f"fractions of pass {k}: {m[f'taus_{k}']:.2e}"
        assert float(t_d[:, 0].abs().max()) == 0.0 and float((t_d[:, -1] - 1).abs().max()) <= 1e-5 and bool((t_d[:, 1:] >= t_d[:, :-1]).all())
    for tag, got, want in (("q_loss", loss_d[:B].cpu(), res["q_loss"]), ("fraction_loss", frac_d[:B].cpu(), res["fraction_loss"])):
        err = (got - want).abs() / (float(want.abs().mean()) + 1e-12)
        m[f"{tag}_median"], m[f"{tag}_p99"], m[f"{tag}_max"] = float(err.median()), float(err.kthvalue(max(1, int(0.99 * B))).values), float(err.max())
    g_dev = L.unpack(dev.grads)
    worst = ("", 0.0)
    for k, g in ora.last_grads.items():
        if g is None:
            continue
        e = float((g_dev[k].cpu() - g).abs().max()) / (float(g.abs().max()) + 1e-12)
        if e > worst[1]:
            worst = (k, e)
    m["grad_worst"], m["grad_worst_tensor"] = worst[1], worst[0]
    got = dev.online.state_dict()
    far = tot = 0
    for k in nets.trainable_keys(ora.po):
        err = (got[k].cpu() - ora.po[k].detach()).abs()
        lr_k = 0.0 if "fraction" in k else 5e-4
        far += int((err > 2e-5 + 0.2 * lr_k).sum()); tot += err.numel()
    m["params_far_share"] = far / tot
    # observed (MI355X, B = 512, A = 9 / 18: profiles/r03_test_stats.json; CPU backend, B = 16): q_loss median 3.5e-7, p99 3.7e-6, max 5.9e-6; fraction
    # loss median 3.3e-5, p99 1.5e-4; worst gradient tensor 8.4e-3 of its max (single ReLU / indicator flips); parameters far: 9.2e-5
    b = dict(q_loss_median=5e-6, q_loss_p99=5e-5, q_loss_max=5e-4, fraction_loss_median=3e-4, fraction_loss_p99=2e-3, grad_worst=3e-2, params_far_share=2e-3)
    b.update(bounds or {})
    for k, v in b.items():
        assert m[k] <= v, f"{k} = {m[k]:.3e} exceeds {v:.1e} ({m})"
    return m


@pytest.mark.parametrize("name", ["fqf", "fqf_duel", "fqf_duel_a18"])
def test_fqf_update_with_device_fractions(ops, name):
    print(check_update_fqf_own_fractions(ops, CASES[name], Hyper(double_q=(name != "fqf"), n_step=3), 16))


def check_nan_skip(ops):
    spec = CASES["dqn"]
    L = NetLayout.from_spec(spec)
    B = 4
    dev = DeviceLearner(ops, L, B, target_update_freq=500)
    sd = recipe.make_state_dict(spec, 11)
    dev.online.load_state_dict(sd)
    dev.target.load_state_dict(sd)
    before = dev.online.flat.clone()
    D = lambda t: t.to(ops.device)
    frames = D(torch.from_numpy(recipe.make_frames(B, 1, spec.obs_shape)).reshape(-1))
    a, r, d, w = recipe.make_transitions(B, 4, 2)
    r = r.copy(); r[1] = np.nan
    dev.update(frames, None, 2 * int(np.prod(spec.obs_shape)), D(torch.from_numpy(a.astype(np.int32))), D(torch.from_numpy(r)), D(torch.from_numpy(d.astype(np.float32))), D(torch.from_numpy(w)))
    assert torch.equal(before, dev.online.flat)            # parameters untouched (agent.py:152-158)
    assert int(dev.state[1]) == 0 and int(dev.state[2]) == 1 and int(dev.state[0]) == 0
    # quirk: update_steps % freq == 0 still triggers the target copy after a skipped step (agent.py:160-161)
    assert torch.equal(dev.target.flat, dev.online.flat)


def test_nan_loss_skips_the_step(ops):
    check_nan_skip(ops)


def test_gather_row_division_by_multiply_high_is_exact():
    """a0_udiv replaces the two integer divisions of the im2col row descriptor (row -> sample, oh, ow) by a multiply-high with
    magic = ceil(2^32 / d) and one correction: exact for every 31-bit row index, for the divisors the network geometries produce."""
    import ctypes as C
    import cpu_ops
    lib = C.CDLL(cpu_ops.build_emul())
    lib.emul_udiv_mismatches.restype = C.c_longlong
    lib.emul_udiv_mismatches.argtypes = [C.c_uint] * 4
    for d in (1, 2, 3, 7, 9, 10, 20, 49, 81, 100, 400, 1024, 6561, 65535):
        assert lib.emul_udiv_mismatches(d, 0, 3_000_000, 1) == 0, d                       # every row index of a 512- to 7 000-sample batch
        assert lib.emul_udiv_mismatches(d, 0, 2**31 - 1, 104_729) == 0, d                  # strided up to the largest int
        assert lib.emul_udiv_mismatches(d, 2**31 - 1 - 2_000_000, 2**31 - 1, 1) == 0, d   # and the top of the range


@pytest.mark.parametrize("name", ["dqn_duel", "iqn", "fqf"])
def test_full_size_check_machinery_on_the_small_geometry(ops, name):
    """check_update_full_size (ReLU decisions injected into the oracle's backward pass, Adam-sensitivity tolerance) runs on the GPU at
    B = 512 and 84x84; here the same function on the emulation backend at the tiny geometry, so that the machinery itself is tested on CPU."""
    spec = CASES[name]
    stats = check_update_full_size(ops, spec, Hyper(double_q=(name != "fqf"), n_step=3, K=6, N=8, N_dash=5), 16)
    assert all(n > 0 for _, n, _ in stats.values())


@pytest.mark.parametrize("name,dq,n", [("iqn", False, 1), ("fqf_duel", True, 3)])
def test_quantile_update_through_the_kept_embedding_launch(name, dq, n):
    """The differentiated pass of the quantile heads through dense_fwd_mul_keep (the embedding kept for the backward pass and its product with
    the features written by one launch; on the GPU only for the short-reduction kernel's shapes): forced at the emulation's tiny shapes, the
    update must still match the oracle — the host-side wiring of engine.DeviceNet.head."""
    import cpu_ops
    o = cpu_ops.CpuOps()
    rows = []

    def keep(X, ldx, W, b, M, group, E, Y, R, N, K, relu):
        rows.append(R)
        o.dense_fwd(X, ldx, W, b, E, R, N, K, relu, o.empty(max(o.dense_fwd_scratch(R, N, K), 1)))
        Y[: R * N] = (E[: R * N].view(R // group, group, N) * M[: (R // group) * N].view(R // group, 1, N)).reshape(-1)

    o.dense_fwd_mul_keep_ok = lambda R, N, K, ldx: True
    o.dense_fwd_mul_keep = keep
    check_update(o, name, 6, dq, n)
    assert rows
