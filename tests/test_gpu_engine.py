"""GPU (MI355X): the HIP kernels, called through the C-ABI, against the CPU oracle and the reference goldens.

Same checks as tests/test_engine_emul.py but on the real library (fp32 MFMA tile kernel included), plus full-size
cases pinned by fixtures generated from the reference itself (tests/golden/g3_*, g6_*).
Tolerances: the MFMA accumulates a k-ordered fp32 fmaf chain, torch CPU uses blocked/vectorised sums, so results agree
to a few fp32 ulp of the accumulated magnitude: rtol 5e-5 / atol 5e-6 on losses and Q-values (stated per assert).
"""
import os

import numpy as np
import pytest
import torch

import recipe
from recipe import SPECS
import test_engine_emul as E
from util import assert_close, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from agent0_amd.ops import HipOps
    ops = HipOps()
    cu, mem, arch = ops.device_info()
    assert "gfx950" in arch, f"these kernels are built for gfx950 only, found {arch}"
    return ops


@pytest.mark.parametrize("name", list(E.CASES))
def test_forward(hip, name):
    E.check_forward(hip, name)


@pytest.mark.parametrize("name,B,dq,n", E.TRAIN)
def test_update(hip, name, B, dq, n):
    E.check_update(hip, name, B, dq, n)


def test_nan_skip(hip):
    E.check_nan_skip(hip)


# ----------------------------------------------------------------------------- full geometry, reference goldens
def _device_learner(hip, spec, B, dq, n, **kw):
    from agent0_amd.deepq.engine import DeviceLearner
    from agent0_amd.deepq.layout import NetLayout
    L = NetLayout.from_spec(spec)
    dev = DeviceLearner(hip, L, B, n_step=n, double_q=dq, **kw)
    dev.online.load_state_dict(recipe.make_state_dict(spec, 11))
    dev.target.load_state_dict(recipe.make_state_dict(spec, 12))
    return L, dev


def _batch(hip, spec, B, seed_f, seed_t):
    frames = recipe.make_frames(B, seed_f, spec.obs_shape)
    a, r, d, w = recipe.make_transitions(B, spec.action_dim, seed_t)
    D = lambda x: torch.from_numpy(x).to(hip.device)
    return D(frames.reshape(-1)), D(a.astype(np.int32)), D(r), D(d.astype(np.float32)), D(w)


def _golden_noise_g3(dev, L, g):
    """Install the NoisyNet draws the reference made for this fixture (keys noise::<online|target>::<module>.<vector>)."""
    for tag, net in (("online", dev.online), ("target", dev.target)):
        for prefix, *_ in L.noise_modules:
            net.set_noise(prefix, *[g[f"noise::{tag}::{prefix}.{leaf}"] for leaf in ("noise_in", "noise_out_weight", "noise_out_bias")])


def _oracle_fqf_taus(spec, po, pt, hp, frames, a, r, d):
    """The fractions the oracle (== the reference: same torch ops) proposes for this batch: (online net on obs, selection pass on next obs).
    Both sides then evaluate q(tau) at bit-identical fractions — cos(pi*64*tau) amplifies ulp-level differences of two softmax/cumsum
    evaluations ~200x; the fraction net itself is compared in test_forward[fqf]."""
    from oracle import losses, nets
    ft = nets.normalize(torch.from_numpy(frames))
    obs, nxt = torch.split(ft, spec.obs_shape[0], 1)
    nets.TAU_LOG = []
    try:
        with torch.no_grad():
            losses.train_step(po, pt, spec, hp, obs, torch.from_numpy(a), torch.from_numpy(r), torch.from_numpy(d).float(), nxt, None)
        log = nets.TAU_LOG
    finally:
        nets.TAU_LOG = None
    assert len(log) == 2
    return [x for pair in log for x in pair]


def _case(case):
    name, b, dq, n = case.rsplit("_", 3)
    return name, int(b[1:]), bool(int(dq[2:])), int(n[1:])


G3_GPU = ["dqn_b8_dq0_n1", "dqn_b8_dq1_n3", "dqn_duel_b8_dq1_n1", "mdqn_b8_dq0_n1", "c51_b8_dq0_n1", "c51_b8_dq1_n3", "c51_duel_noisy_b8_dq1_n3",
          "qr_b8_dq0_n1", "qr_duel_b8_dq1_n3", "iqn_b8_dq0_n1", "iqn_duel_b8_dq1_n3", "fqf_b8_dq0_n1", "fqf_b8_dq1_n3", "dqn_tiny_b32_dq1_n1",
          "c51_tiny_b32_dq1_n3", "dqn_b512_dq0_n1", "c51_b512_dq1_n3", "fqf_duel_b8_dq1_n3", "dqn_duel_a18_b8_dq1_n1", "fqf_duel_a18_b8_dq1_n3"]


def test_every_committed_g3_g6_fixture_reaches_the_gpu():
    import glob, os
    from util import GOLDEN
    have3 = sorted(os.path.basename(f)[3:-4] for f in glob.glob(os.path.join(GOLDEN, "g3_*.npz")))
    have6 = sorted(os.path.basename(f)[3:-4] for f in glob.glob(os.path.join(GOLDEN, "g6_*.npz")))
    assert have3 == sorted(G3_GPU) and have6 == sorted(G6_GPU)


@pytest.mark.parametrize("case", G3_GPU)
def test_golden_losses(hip, case):
    """Per-sample TD losses of the reference learners (fixture group G3, all 17 cases: six algorithms, dueling, NoisyNet, double-Q, n-step,
    B up to 512) reproduced by the HIP path, with the reference's own random draws (IQN taus, NoisyNet noise) injected."""
    from oracle import learner as olearner
    from oracle.losses import Hyper
    name, B, dq, n = _case(case)
    spec = SPECS[name]
    g = golden(f"g3_{case}")
    L, dev = _device_learner(hip, spec, B, dq, n)
    if spec.noisy:
        _golden_noise_g3(dev, L, g)
    frames_np = recipe.make_frames(B, 31, spec.obs_shape)
    a_np, r_np, d_np, _ = recipe.make_transitions(B, spec.action_dim, 32)
    frames, a, r, d, w = _batch(hip, spec, B, 31, 32)
    w = torch.ones_like(w)
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x).reshape(-1).copy()).to(hip.device)
    rand = None
    if spec.algo == "iqn":
        rand = [D(g[f"rand_{i}"]) for i in range(3)]
    elif spec.algo == "fqf":
        po, pt = olearner.to_params(recipe.make_state_dict(spec, 11)), olearner.to_params(recipe.make_state_dict(spec, 12))
        rand = [D(x.numpy()) for x in _oracle_fqf_taus(spec, po, pt, Hyper(double_q=dq, n_step=n), frames_np, a_np, r_np, d_np)]
    out = dev.update(frames, None, 2 * int(np.prod(spec.obs_shape)), a, r, d, w, rand=rand)
    loss, frac = out if isinstance(out, tuple) else (out, None)
    assert_close(loss[:B], g["loss"], 5e-5, 5e-6, "per-sample loss vs reference")
    if frac is not None:
        assert_close(frac[:B], g["fraction_loss"], 5e-5, 2e-5, "fraction loss vs reference")
    assert int(dev.state[2]) == 0


G6_GPU = ["dqn_b16_dq0_n1", "dqn_duel_b16_dq1_n3", "c51_duel_noisy_b16_dq1_n3", "c51_b16_dq0_n1", "qr_b16_dq0_n1", "iqn_b16_dq0_n1", "fqf_b16_dq0_n1",
          "mdqn_b16_dq0_n1", "dqn_tiny_b32_dq1_n1", "c51_tiny_b32_dq1_n3", "dqn_b512_dq0_n1", "fqf_duel_b16_dq1_n3", "dqn_duel_a18_b16_dq1_n1",
          "fqf_duel_a18_b16_dq1_n3"]


@pytest.mark.parametrize("case", G6_GPU)
def test_golden_train_steps(hip, case):
    """Full reference ``learner.train()`` calls (fixture group G6, all 11 cases; two consecutive steps with a target sync for B <= 32):
    per-sample losses, gradient and post-optimizer parameter fingerprints (sum-of-squares norm + first eight values) of every tensor of
    the online net, the target net's fingerprints, the update counter — Adam eps = 1e-2/B, FQF's RMSprop on the fraction net included."""
    from oracle import learner as olearner, nets
    from oracle.losses import Hyper
    name, B, dq, n = _case(case)
    spec = SPECS[name]
    g = golden(f"g6_{case}")
    L, dev = _device_learner(hip, spec, B, dq, n, target_update_freq=2)
    ora = None
    if spec.algo == "fqf":      # the oracle steps alongside only to supply the fractions (see _oracle_fqf_taus)
        ora = olearner.OracleLearner(spec, recipe.make_state_dict(spec, 11), recipe.make_state_dict(spec, 12), Hyper(double_q=dq, n_step=n), batch_size=B, target_update_freq=2)
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x).reshape(-1).copy()).to(hip.device)
    steps = 2 if B <= 32 else 1
    for s in range(steps):
        frames_np = recipe.make_frames(B, 61 + s, spec.obs_shape)
        a_np, r_np, d_np, w_np = recipe.make_transitions(B, spec.action_dim, 62 + s)
        frames, a, r, d, w = _batch(hip, spec, B, 61 + s, 62 + s)
        rand = None
        if spec.algo == "iqn":
            rand = [D(g[f"s{s}::rand_{i}"]) for i in range(3)]
        elif spec.algo == "fqf":
            nets.TAU_LOG = []
            res_o = ora.train(frames_np.reshape(B, -1), a_np, r_np, d_np.astype(np.float32), w_np, np.arange(B))
            rand = [D(x.numpy()) for pair in nets.TAU_LOG for x in pair]
            nets.TAU_LOG = None
        if spec.noisy:
            nd = len(L.noise_modules) * 3
            draws = [g[f"s{s}::normal_{i}"] for i in range(2 * nd)]
            E.install_noise(dev.online, draws[:nd])
            E.install_noise(dev.target, draws[nd:])
        out = dev.update(frames, None, 2 * int(np.prod(spec.obs_shape)), a, r, d, w, rand=rand)
        loss, frac = out if isinstance(out, tuple) else (out, None)
        # What the step is compared with.  Normally the reference's fixture.  FQF from the SECOND step on: the oracle's result of the same
        # step on THIS machine, and the fixture only loosely — the reference's own numbers are not reproducible across CPUs there: torch's
        # CPU kernels round differently on the GPU box's host than on the build container's (measured for fqf_duel, oracle vs fixture: 1.3e-5
        # of the fraction loss at step 0 — the ulp-level difference of the proposed fractions times cos(pi i tau)'s ~200x — and, after one Adam
        # step at eps = 1e-2/16 has turned that into ~1e-4 of some parameters, 6.5e-4 at step 1; tools/debug/fqf_duel_s1.py), while HIP and
        # the oracle agree to 2e-6 on the same machine.  The oracle is pinned to the fixture on the build CPU by tests/test_oracle_golden.py.
        vs_oracle = ora is not None and s >= 1
        want = {}
        if vs_oracle:
            want[f"s{s}::q_loss"], want[f"s{s}::fraction_loss"] = res_o["q_loss"].numpy(), res_o["fraction_loss"].numpy()
            for k_, v_ in ora.last_grads.items():
                if v_ is not None:
                    want[f"s{s}::grad::{k_}"] = recipe.checksum(v_.numpy())
            for k_, v_ in ora.po.items():
                want[f"s{s}::param::{k_}"] = recipe.checksum(v_.detach().numpy())
            for k_, v_ in ora.pt.items():
                want[f"s{s}::target::{k_}"] = recipe.checksum(v_.detach().numpy())
            assert_close(loss[:B], g[f"s{s}::q_loss"], 1e-3, 1e-4, f"s{s} q_loss vs the fixture (loose: see above)")
            assert_close(frac[:B], g[f"s{s}::fraction_loss"], 5e-3, 5e-4, f"s{s} fraction_loss vs the fixture (loose: see above)")
        ref = lambda key: want[key] if vs_oracle else g[key]
        assert_close(loss[:B], ref(f"s{s}::q_loss"), 5e-5, 5e-6, f"s{s} q_loss")
        if frac is not None:
            assert_close(frac[:B], ref(f"s{s}::fraction_loss"), 5e-5, 2e-5, f"s{s} fraction_loss")
        assert int(dev.state[1]) == int(g[f"s{s}::update_steps"])
        grads = L.unpack(dev.grads)
        params, target = dev.online.state_dict(), dev.target.state_dict()
        for k in (want if vs_oracle else g.files):
            parts = k.split("::")
            if parts[0] != f"s{s}" or len(parts) < 3:
                continue
            want_k = ref(k)
            if parts[2] not in grads and parts[1] == "grad":
                continue
            if parts[1] == "grad":
                got = recipe.checksum(grads[parts[2]].cpu().numpy())
                scale = max(want_k[1], 1e-6)
                assert abs(got[1] - want_k[1]) <= 3e-4 * scale, (k, got[1], want_k[1])
                # At B = 512 a handful of the batch's ~10^7 ReLU decisions fall on pre-activations within fp32 rounding of zero and come out
                # differently than in torch; each moves a convolution weight gradient by one whole term of its sum (measured: up to 1.2e-3
                # of the tensor's max, profiles/r02_grad_accuracy.txt).  The fixture cannot be re-evaluated with the device's decisions, so
                # its convolution fingerprints get that width here; test_update_full_size[dqn] makes the same comparison against the
                # oracle WITH the decisions injected, at 3e-5.
                flip = float(os.environ.get("A0_TEST_FLIP", "2e-4")) * float(grads[parts[2]].abs().max()) if (B >= 256 and "convs" in k) else 0.0
                assert np.all(np.abs(got[2:] - want_k[2:]) <= 3e-4 * scale / np.sqrt(max(grads[parts[2]].numel(), 1)) + 2e-4 * np.abs(want_k[2:]) + 1e-7 + flip), (k, got, want_k)
            elif parts[1] in ("param", "target"):
                src = params if parts[1] == "param" else target
                if parts[2] not in src:
                    continue
                got = recipe.checksum(src[parts[2]].cpu().numpy())
                assert abs(got[1] - want_k[1]) <= 1e-5 * max(want_k[1], 1e-6), (k, got[1], want_k[1])
                assert np.all(np.abs(got[2:] - want_k[2:]) <= 5e-5 + 1e-4 * np.abs(want_k[2:])), (k, got[2:], want_k[2:])
        if ora is not None and s + 1 < steps:
            # FQF, before the second step: continue from the oracle's post-step state, so that the step is compared from a common state (one
            # Adam step at eps = 1e-2/16 leaves two correct evaluations up to ~1e-4 apart where |g| ~ eps, and the fraction loss, a sum of
            # DIFFERENCES of neighbouring quantile values, amplifies that to ~1e-3).  Drift over consecutive un-synchronised updates is what
            # tests/test_gpu_trace.py::test_free_running_trace... records and bounds.
            L.pack({k_: v.detach() for k_, v in ora.po.items()}, dev.online.flat)
            L.pack({k_: v.detach() for k_, v in ora.pt.items()}, dev.target.flat)
            dev.online.refresh_wt(); dev.target.refresh_wt()
            zeros = {k_: torch.zeros_like(ora.po[k_]) for k_ in ora.f_keys}
            L.pack({**ora.adam.m, **zeros}, dev.adam_m)
            L.pack({**ora.adam.v, **zeros}, dev.adam_v)


@pytest.mark.parametrize("n", [1, 3])
def test_golden_c51_projection_edge_cases(hip, n):
    """Fixture group G4: the reference's categorical projection (agent.py:236-264) on the known-answer rows — integer b (lo == up, both
    fix-ups), b = 0 and b = 50, rewards that clamp to either end, terminals — through a0_loss_c51.  (i) The kernel alone, fed the
    reference's selected target distribution as log-probabilities: projected distribution m against the reference's ``target_prob``.
    (ii) The whole learner on the fixture's frames: m and the per-sample loss."""
    g = golden(f"g4_c51_projection_n{n}")
    spec = SPECS["c51"]
    A, T = spec.action_dim, spec.num_atoms
    rew, done, act_np = g["rewards"], g["terminals"], g["actions"]
    B = len(rew)
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(hip.device)
    atoms = torch.linspace(-10.0, 10.0, T).to(hip.device)
    gamma_n = float(0.99 ** n)
    # (i) kernel alone: a_star = 1 everywhere, its row holds log p (softmax(log p) == p to an ulp), the other rows are arbitrary
    tgt = torch.full((B, A, T), -3.0)
    tgt[:, 1, :] = torch.from_numpy(g["prob_next_sel"]).log()
    a_star = torch.ones(B, dtype=torch.int32)
    loss, dq, m, state = hip.empty(B), hip.zeros(B * A * T), hip.empty(B * T), hip.zeros(8, dtype=torch.int32)
    hip.loss_c51(hip.zeros(B * A * T), D(tgt.reshape(-1).numpy()), A, T, D(act_np.astype(np.int32)), D(a_star.numpy()), D(rew), D(done), D(np.ones(B, np.float32)), atoms,
                 gamma_n, -10.0, 10.0, B, loss, dq, m, state)
    assert_close(m.view(B, T), g["target_prob"], 2e-6, 2e-7, "projected distribution vs the reference's target_prob")
    assert_close(m.view(B, T).sum(-1), np.ones(B), 1e-5, 0, "mass conserved")
    # every edge row puts its mass exactly where the reference does (same support, zero elsewhere)
    assert np.array_equal(m.view(B, T).cpu().numpy() > 0, g["target_prob"] > 0)
    # (ii) whole learner
    L, dev = _device_learner(hip, spec, B, False, n)
    frames = D(recipe.make_frames(B, 41).reshape(-1))
    out = dev.update(frames, None, 2 * 4 * 84 * 84, D(act_np.astype(np.int32)), D(rew), D(done), D(np.ones(B, np.float32)))
    assert_close(out[:B], g["loss"], 5e-5, 5e-6, "c51 loss on the edge-case rows")
    assert_close(dev.m_proj.view(B, T), g["target_prob"], 5e-5, 1e-6, "projected distribution, end to end")


@pytest.mark.parametrize("tag", ["4x200x200", "8x64x64", "8x32x32", "3x8x5"])
def test_golden_quantile_huber(hip, tag):
    """Fixture group G5: the reference's huber_qr_loss (agent.py:110-114) and its gradient on random (q, T, tau) including exact ties
    (T == q: indicator 0, zero Huber slope) and |q - T| == 1 (the quadratic/linear seam), shared QR midpoints (4x200x200) and per-sample
    taus — through a0_loss_quantile_huber: loss [B] and d(sum_b w_b loss_b)/dq."""
    g = golden("g5_huber_qr")
    q, t, tau, w = g[f"q_{tag}"], g[f"t_{tag}"], g[f"tau_{tag}"], g[f"w_{tag}"]
    B, N = q.shape
    Nd = t.shape[1]
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x).reshape(-1)).to(hip.device)
    loss, dq, state = hip.empty(B), hip.zeros(B * N), hip.zeros(8, dtype=torch.int32)
    hip.loss_quantile_huber(D(q), N, 1, N, D(t), D(tau), (N if tau.shape[0] == B and tau.shape[0] > 1 else 0), hip.zeros(B, dtype=torch.int32), D(w), B, N, Nd, loss, dq, state)
    assert_close(loss, g[f"loss_{tag}"], 1e-5, 1e-6, "loss")
    assert_close(dq.view(B, N), g[f"dq_{tag}"], 1e-5, 1e-7, "dloss/dq")
    assert int(state[0]) == 0


def _record(name, payload):
    """Measured statistics of a GPU test, for profiles/ (gpurun merges gpurun_out/ back): one JSON file per statistic and case."""
    import json, os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "test_stats")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as f:
            json.dump(payload, f, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.mark.parametrize("algo,A,dq,n,duel,noisy", [("dqn", 4, False, 1, False, False), ("c51", 4, True, 3, False, False), ("iqn", 9, True, 3, False, False),
                                                    ("fqf", 9, False, 1, False, False), ("dqn", 18, True, 1, True, False), ("fqf", 18, True, 3, True, False),
                                                    ("fqf", 9, True, 3, True, False),
                                                    # BASELINE configs[2] as configured: rainbow-lite = c51 + double-Q + dueling + NoisyNet, n = 3 (model.py:137-177,28-87)
                                                    ("c51", 4, True, 3, True, True), ("c51", 4, False, 1, True, False), ("c51", 18, True, 3, True, True),
                                                    # QR at full size: 512 x 200 x 200 quantile pairs, the largest quantile-Huber instance (agent.py:272-293)
                                                    ("qr", 4, True, 3, False, False), ("qr", 4, False, 1, True, True)])
def test_update_full_size(hip, algo, A, dq, n, duel, noisy):
    """BASELINE configs[1..4] at their real geometry — 84x84 observations, B = 512; Breakout's A = 4 for dqn / c51, Asterix's A = 9 with
    IQN N = N' = 64, K = 32 and FQF F = 32 — one whole update against the oracle: per-sample losses, every gradient tensor (the
    cosine-embedding weight gradient, Hadamard backward and the fc1 GEMMs over B*64 = 32 768 rows included), parameters and target after
    Adam / RMSprop, at the small cases' tolerances (losses rtol 5e-5, gradients 3e-5 of the tensor's max, parameters 2e-5 absolute), with
    the two effects of scale — ReLU decisions at rounding-level pre-activations, Adam's sign-like first step at eps = 2e-5 — separated
    out and bounded (tests/test_engine_emul.py::check_update_full_size)."""
    from oracle.losses import Hyper
    # the last three: the reference's own suite configuration (README.md:62-112, atari8_double_duel_prior: fqf + double-Q + dueling) at
    # Asterix's A = 9 and Seaquest's A = 18, and the 18-action dueling dqn head (upper range of the head/loss kernel's A + dueling <= 24 path)
    atoms = {"c51": {"num_atoms": 51}, "qr": {"num_atoms": 200}}.get(algo, {})
    stats = E.check_update_full_size(hip, recipe.NetSpec(algo, A, dueling=duel, noisy=noisy, **atoms), Hyper(double_q=dq, n_step=n, K=32, N=64, N_dash=64), 512)
    print({k: (v[0], v[1], f"{v[2]:.1e}") for k, v in stats.items()})
    _record(f"relu_decisions_{algo}_a{A}_duel{int(duel)}_dq{int(dq)}_n{n}{'_noisy' if noisy else ''}_b512",
            {k: {"differ": int(v[0]), "of": int(v[1]), "largest_preactivation_rel": float(v[2])} for k, v in stats.items()})


@pytest.mark.parametrize("name", ["fqf", "fqf_duel", "fqf_duel_a18"])
def test_fqf_update_with_device_fractions(hip, name):
    """The un-injected FQF update at the small geometry: the device's own a0_fqf_taus -> cos -> head -> loss -> gradient chain against
    the oracle's own fractions (tests/test_engine_emul.py::check_update_fqf_own_fractions, bounded-outlier metric)."""
    from oracle.losses import Hyper
    m = E.check_update_fqf_own_fractions(hip, E.CASES[name], Hyper(double_q=(name != "fqf"), n_step=3), 16)
    print(m)


@pytest.mark.parametrize("A,duel,dq", [(9, False, False), (18, True, True)])
def test_fqf_update_with_device_fractions_full_size(hip, A, duel, dq):
    """... and at BASELINE configs[4]'s geometry (84x84, B = 512, F = 32; Asterix A = 9 plain, Seaquest A = 18 dueling + double-Q)."""
    from oracle.losses import Hyper
    m = E.check_update_fqf_own_fractions(hip, recipe.NetSpec("fqf", A, dueling=duel), Hyper(double_q=dq, n_step=3), 512)
    print(m)
    _record(f"fqf_device_fractions_a{A}_duel{int(duel)}_b512", m)


def test_gather_fused_equals_dense_batch(hip):
    """Reading the batch through replay slot indices (gather fused into conv1) == running on the gathered copy."""
    spec = SPECS["dqn"]
    B, cap = 16, 40
    L, dev = _device_learner(hip, spec, B, False, 1)
    L2, dev2 = _device_learner(hip, spec, B, False, 1)
    ring = torch.from_numpy(recipe.make_frames(cap, 5, spec.obs_shape)).to(hip.device).reshape(cap, -1).contiguous()
    slot = torch.from_numpy(recipe.gen(6).permutation(cap)[:B].astype(np.int32)).to(hip.device)
    _, a, r, d, w = _batch(hip, spec, B, 1, 2)
    row = ring.shape[1]
    dense = torch.empty(B * row, dtype=torch.uint8, device=hip.device)
    hip.replay_gather(ring.reshape(-1), row, slot, B, dense, cap)
    assert torch.equal(dense.view(B, row), ring[slot.long()])
    l1 = dev.update(ring.reshape(-1), slot, row, a, r, d, w).clone()
    l2 = dev2.update(dense, None, row, a, r, d, w).clone()
    assert torch.equal(l1, l2) and torch.equal(dev.grads, dev2.grads) and torch.equal(dev.online.flat, dev2.online.flat)


def test_update_is_deterministic(hip):
    spec = SPECS["c51_duel_noisy"]
    B = 32
    outs = []
    for _ in range(2):
        L, dev = _device_learner(hip, spec, B, True, 3)
        frames, a, r, d, w = _batch(hip, spec, B, 3, 4)
        for net, seed in ((dev.online, 1), (dev.target, 2)):
            gg = recipe.gen(seed)
            for prefix, block, r0, r1, in_f in L.noise_modules:
                net.set_noise(prefix, gg.standard_normal(in_f).astype(np.float32) * 0.1, gg.standard_normal(r1 - r0).astype(np.float32) * 0.1,
                              gg.standard_normal(r1 - r0).astype(np.float32) * 0.1)
        dev.update(frames, None, 2 * int(np.prod(spec.obs_shape)), a, r, d, w)
        outs.append((dev.loss.clone(), dev.grads.clone(), dev.online.flat.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y), "bit-identical results across runs (no atomics anywhere on the path)"


@pytest.mark.parametrize("shape,B", [((4, 84, 84), 37), ((4, 36, 36), 5), ((4, 44, 52), 9), ((4, 84, 84), 300)])
def test_fused_encoder_matches_unfused(hip, shape, B):
    """encoder_fused.hip (one workgroup per observation, activations in LDS) vs the three implicit GEMMs, including the replay-slot
    gather, the st_next half and the optional act1/act2 outputs.  The fused conv1 multiplies raw bytes by w/255 instead of
    (x/255) by w — same two roundings per term — so the comparison is to fp32 rounding (rtol 2e-6 of the activation scale)."""
    from agent0_amd.deepq.engine import DeviceNet, Workspace
    from agent0_amd.deepq.layout import NetLayout
    spec = recipe.NetSpec("dqn", 4, obs_shape=shape)
    L = NetLayout.from_spec(spec)
    assert hip.fused_supported(*shape)
    net = DeviceNet(hip, L, hip.net(*shape))
    net.load_state_dict(recipe.make_state_dict(spec, 11))
    cap = B + 11
    ring = torch.from_numpy(recipe.make_frames(cap, 5, shape)).to(hip.device).reshape(-1).contiguous()
    slot = torch.from_numpy(recipe.gen(6).permutation(cap)[:B].astype(np.int32)).to(hip.device)
    ob = int(np.prod(shape))
    a, b = Workspace(hip, L, B), Workspace(hip, L, B)
    for w in (a, b):
        for t in (w.act1, w.act2, w.act3):
            t.fill_(float("nan"))
    hip.encoder_fwd(net.net, net.encoder_weights(), ring, slot, 2 * ob, ob, B, a.act1, a.act2, a.act3)
    hip.encoder_fwd_fused(net.net, net.wt, net.encoder_weights(), ring, slot, 2 * ob, ob, B, b.act1, b.act2, b.act3)
    for x, y, name in ((a.act1, b.act1, "act1"), (a.act2, b.act2, "act2"), (a.act3, b.act3, "act3")):
        assert torch.isfinite(y).all()
        assert_close(y, x, 2e-6, 2e-6 * float(x.abs().max()), name)
        assert torch.equal(x == 0, y == 0) or float(((x == 0) != (y == 0)).float().mean()) < 1e-4, f"{name}: ReLU pattern"
    c = Workspace(hip, L, B)
    c.act3.fill_(float("nan"))
    hip.encoder_fwd_fused(net.net, net.wt, net.encoder_weights(), ring, slot, 2 * ob, ob, B, None, None, c.act3)
    assert torch.equal(b.act3, c.act3)


def test_fused_dgrad_matches_unfused():
    """a0_net_encoder_dgrad_fused (both conv data gradients per observation, LDS-resident d2) against the implicit-GEMM path of
    a0_net_encoder_bwd on the same inputs: same taps and masks, different summation order -> fp32 rounding (rtol 2e-5 of the scale)."""
    from agent0_amd.ops import HipOps
    from agent0_amd.deepq.engine import DeviceNet, Workspace
    from agent0_amd.deepq.layout import NetLayout
    hip = HipOps()
    spec = recipe.NetSpec("dqn", 4)
    L = NetLayout.from_spec(spec)
    net = DeviceNet(hip, L, hip.net(4, 84, 84))
    net.load_state_dict(recipe.make_state_dict(spec, 5))
    assert net.fused_dgrad
    for B in (1, 3, 64, 300, 512):      # 300: more observations than workgroups (the looping path), unevenly divided
        g = recipe.gen(B)
        frames = torch.from_numpy(g.integers(0, 256, B * 28224, dtype=np.uint8)).cuda()
        ws = Workspace(hip, L, B, grads=True)
        net.encode(ws, frames, None, 28224, 0, B, keep=True)
        d3 = torch.from_numpy(g.standard_normal(B * L.feat).astype(np.float32)).cuda() * (ws.act3 > 0)
        ws.d3.copy_(d3)
        grads = hip.zeros(L.n_params_padded)
        slabs = hip.empty(max(hip.encoder_bwd_scratch(net.net, B), 4))
        g1, g2, g3 = grads[L.blocks["conv1"].all], grads[L.blocks["conv2"].all], grads[L.blocks["conv3"].all]
        hip.encoder_bwd(net.net, net.encoder_weights(), frames, None, 28224, 0, B, ws.act1, ws.act2, ws.d3, ws.d2, ws.d1, g1, g2, g3, slabs)
        ref2, ref1, refg = ws.d2.clone(), ws.d1.clone(), grads.clone()
        ws.d2.fill_(7.0); ws.d1.fill_(7.0); grads.zero_()
        hip.encoder_dgrad_fused(net.net, net.wt, ws.d3, ws.act1, ws.act2, B, ws.d2, ws.d1)
        hip.encoder_wgrad(net.net, net.encoder_weights(), frames, None, 28224, 0, B, ws.act1, ws.act2, ws.d3, ws.d2, ws.d1, g1, g2, g3, slabs)
        torch.cuda.synchronize()
        checks = [("d2", ws.d2, ref2), ("d1", ws.d1, ref1), ("grads", grads, refg)]
        # ... and every convolution block on its own scale, weights and bias separately: a0_net_encoder_wgrad computes conv2 / conv3 with the
        # per-observation kernel (conv23_wgrad.hip: LDS-resident operands, transposed fragment reads, groups of observations per slab), conv1
        # with conv1_wgrad.hip; the reference side is the implicit-GEMM path
        for name in ("conv1", "conv2", "conv3"):
            blk = L.blocks[name]
            checks += [(name + ".weight", grads[blk.w], refg[blk.w]), (name + ".bias", grads[blk.b], refg[blk.b])]
        for name, got, ref in checks:
            scale = float(ref.abs().max())
            err = float((got - ref).abs().max())
            assert err <= 2e-5 * scale, (name, B, err, scale)
        assert torch.equal(ws.d2 == 0, ref2 == 0) or float(((ws.d2 == 0) != (ref2 == 0)).float().mean()) < 1e-4


def test_conv1_bf16_split_is_exact_and_fp32_accurate():
    """conv1 of the fused encoder runs on the bf16 matrix pipe.  (1) The three bf16 terms a0_conv_wt_kernel stores for every weight add up
    to fl(w/255) EXACTLY (bit for bit), so with bytes exact in bf16 every product is exact in fp32.  (2) The layer's output is as close
    to an fp64 evaluation as the fp32 fmaf chain of the unfused kernel (both within 2e-6 of the scale)."""
    from agent0_amd.ops import HipOps
    from agent0_amd.deepq.engine import DeviceNet, Workspace
    from agent0_amd.deepq.layout import NetLayout
    hip = HipOps()
    spec = recipe.NetSpec("dqn", 4)
    L = NetLayout.from_spec(spec)
    net = DeviceNet(hip, L, hip.net(4, 84, 84))
    sd = recipe.make_state_dict(spec, 9)
    net.load_state_dict(sd)
    # ---- (1) exact split: uint4 fragment ((t*32 + n)*4 + q)*3 + s holds k = 32t + 8q .. +7 of channel n, term s
    K1 = 256
    raw = net.wt[:48 * K1].view(torch.int32).cpu().numpy().view(np.uint16).reshape(8, 32, 4, 3, 8)       # [t][n][q][s][8]
    terms = (raw.astype(np.uint32) << 16).view(np.float32)                                                # bf16 -> fp32, exact
    total = (terms[:, :, :, 0].astype(np.float64) + terms[:, :, :, 1].astype(np.float64) + terms[:, :, :, 2].astype(np.float64))
    got = np.transpose(total, (1, 0, 2, 3)).reshape(32, K1)                                               # [n][k = 32t + 8q + j]
    w1 = net.encoder_weights()["w1"].cpu().numpy().reshape(32, K1)
    want = (w1 / np.float32(255.0)).astype(np.float32)
    assert np.array_equal(got.astype(np.float32), want) and np.array_equal(got, want.astype(np.float64))
    # conv2 / conv3 (split-operand path): same exactness for the weight terms, fragment layout [t][n][q][s][8] with 64 channels
    off = 48 * K1 + 64 * 512 + 64 * 576 + 64 * 576 + 4 * 32 * 256
    for name, K, length in (("w2", 512, 96 * 512), ("w3", 576, 96 * 576)):
        raw = net.wt[off:off + length].view(torch.int32).cpu().numpy().view(np.uint16).reshape(K // 32, 64, 4, 3, 8)
        terms = (raw.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        got = np.transpose(terms[:, :, :, 0] + terms[:, :, :, 1] + terms[:, :, :, 2], (1, 0, 2, 3)).reshape(64, K)
        want = net.encoder_weights()[name].cpu().numpy().reshape(64, K)
        assert np.array_equal(got, want.astype(np.float64)), name
        off += length
    # ---- (2) accuracy against fp64
    B = 2
    g = recipe.gen(77)
    frames_np = g.integers(0, 256, (B, 4, 84, 84), dtype=np.uint8)
    frames = torch.from_numpy(frames_np.reshape(-1)).cuda()
    ws_f, ws_u = Workspace(hip, L, B), Workspace(hip, L, B)
    hip.encoder_fwd_fused(net.net, net.wt, net.encoder_weights(), frames, None, 28224, 0, B, ws_f.act1, ws_f.act2, ws_f.act3)
    hip.encoder_fwd(net.net, net.encoder_weights(), frames, None, 28224, 0, B, ws_u.act1, ws_u.act2, ws_u.act3)
    w = np.asarray(sd["encoder.convs.0.weight"], dtype=np.float64)        # [32][4][8][8]
    bias = np.asarray(sd["encoder.convs.0.bias"], dtype=np.float64)
    x = frames_np.astype(np.float64) / 255.0
    ref = np.zeros((B, 20, 20, 32))
    for oh in range(20):
        for ow in range(20):
            patch = x[:, :, 4 * oh:4 * oh + 8, 4 * ow:4 * ow + 8].reshape(B, -1)
            ref[:, oh, ow, :] = patch @ w.reshape(32, -1).T + bias
    ref = np.maximum(ref, 0.0).reshape(-1)
    scale = np.abs(ref).max()
    e_f = np.abs(ws_f.act1.cpu().numpy().astype(np.float64) - ref).max()
    e_u = np.abs(ws_u.act1.cpu().numpy().astype(np.float64) - ref).max()
    assert e_f <= 2e-6 * scale and e_u <= 2e-6 * scale, (e_f, e_u, scale)
    assert e_f <= 2.0 * e_u + 1e-7 * scale, (e_f, e_u)
    # the whole encoder (conv2 / conv3 with both operands split) against an fp64 evaluation of the three layers
    def conv64(x, w, bias, stride):      # x [B][C][H][W] float64, w [N][C][kh][kw]
        Bn, C_, H_, W_ = x.shape
        N_, _, kh, kw = w.shape
        Ho, Wo = (H_ - kh) // stride + 1, (W_ - kw) // stride + 1
        out = np.zeros((Bn, N_, Ho, Wo))
        wm = w.reshape(N_, -1).T
        for i in range(Ho):
            for j in range(Wo):
                out[:, :, i, j] = x[:, :, i * stride:i * stride + kh, j * stride:j * stride + kw].reshape(Bn, -1) @ wm + bias
        return np.maximum(out, 0.0)
    w2, b2 = (np.asarray(sd[f"encoder.convs.2.{k}"], dtype=np.float64) for k in ("weight", "bias"))
    w3, b3 = (np.asarray(sd[f"encoder.convs.4.{k}"], dtype=np.float64) for k in ("weight", "bias"))
    y3 = conv64(conv64(conv64(x, w, bias, 4), w2, b2, 2), w3, b3, 1)                   # [B][64][7][7]
    ref3 = np.transpose(y3, (0, 2, 3, 1)).reshape(-1)                                   # NHWC like act3
    s3 = np.abs(ref3).max()
    e3_f = np.abs(ws_f.act3.cpu().numpy().astype(np.float64) - ref3).max()
    e3_u = np.abs(ws_u.act3.cpu().numpy().astype(np.float64) - ref3).max()
    assert e3_f <= 3e-6 * s3 and e3_u <= 3e-6 * s3, (e3_f, e3_u, s3)
    assert e3_f <= 3.0 * e3_u + 2e-7 * s3, (e3_f, e3_u)


@pytest.mark.parametrize("R,N,group,relu", [(8192, 3136, 32, True), (2048, 3136, 64, True), (300, 96, 5, True), (257, 64, 32, False), (1000, 200, 1, True),
                                            (992, 3136, 31, True), (15872, 128, 31, True), (1056, 192, 33, True), (1280, 64, 40, False), (512, 64, 16, True)])
def test_short_reduction_forward_kernel(hip, R, N, group, relu):
    """short_k_fwd.h (K = 64: the quantile networks' cosine embedding, model.py:244-247) in its three modes against float64, ragged rows /
    columns and groups that do not divide a 32-row block included (31 rows per group — fqf's F - 1 interior fractions — and 33 / 40 take the
    straight-line kernel with two M rows per block, 5 / 16 / 1 the guarded one); the kept embedding and the product must agree with the separate launches."""
    K = 64
    g = torch.Generator().manual_seed(R + N)
    X = torch.randn(R, K, generator=g) * 0.7
    W = torch.randn(N, K, generator=g) * 0.2
    b = torch.randn(N, generator=g) * 0.1
    G = (R + group - 1) // group
    M = torch.randn(G, N, generator=g).abs()
    assert hip.dense_fwd_mul_keep_ok(R, N, K, K)
    ref = X.double() @ W.double().t() + b.double()
    if relu:
        ref = ref.clamp_min(0.0)
    ref_mul = ref * M.double().repeat_interleave(group, 0)[:R]
    dev = lambda t: t.contiguous().view(-1).to(hip.device)
    Xd, Wd, bd, Md = dev(X), dev(W), dev(b), dev(M)
    Y0, Y1, E, Y2 = (torch.full((R * N,), float("nan"), device=hip.device) for _ in range(4))
    hip.dense_fwd(Xd, K, Wd, bd, Y0, R, N, K, relu, None)
    hip.dense_fwd_mul(Xd, K, Wd, bd, Md, group, Y1, R, N, K, relu)
    hip.dense_fwd_mul_keep(Xd, K, Wd, bd, Md, group, E, Y2, R, N, K, relu)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((Y0.cpu().double().view(R, N) - ref).abs().max()) <= 2e-6 * scale
    assert float((Y1.cpu().double().view(R, N) - ref_mul).abs().max()) <= 2e-6 * float(ref_mul.abs().max())
    assert torch.equal(E, Y0) and torch.equal(Y2, Y1)


@pytest.mark.parametrize("A,T,dueling,double_q,B", [(4, 51, True, True, 512), (4, 51, False, False, 37), (18, 51, True, True, 65), (6, 11, True, False, 9), (3, 64, False, True, 130)])
def test_c51_head_loss_from_slabs_equals_the_separate_kernels(hip, A, T, dueling, double_q, B):
    """a0_c51_head_loss_slabs (round 4: one launch from the head GEMMs' split-K slabs to loss + head gradient) against the launches it replaces — slab
    reduction, a0_dueling_fwd x3, a0_select_action, a0_loss_c51, a0_dueling_bwd — on the same slabs: the same arithmetic statement for statement, so every
    output must be BIT-identical (torch.equal), ragged last workgroup and padded head columns included; and a0_reduce_bias_act_multi against a0_dense_fwd's
    own reduction."""
    g = recipe.gen(A * 1000 + T + B)
    NQ = A + (1 if dueling else 0)
    ld = (NQ * T + 31) // 32 * 32
    R_on = 2 * B if double_q else B
    ns_on, ns_tg = 5, 3
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(hip.device)
    s_on = D(g.standard_normal((ns_on, R_on, ld)).astype(np.float32))
    s_tg = D(g.standard_normal((ns_tg, B, ld)).astype(np.float32))
    b_on, b_tg = D(g.standard_normal(ld).astype(np.float32)), D(g.standard_normal(ld).astype(np.float32))
    a, r, d, w = recipe.make_transitions(B, A, 5)
    act, rew, done, wgt = D(a.astype(np.int32)), D(r), D(d.astype(np.float32)), D(w)
    atoms = torch.linspace(-10.0, 10.0, T).to(hip.device)
    state = hip.zeros(8, dtype=torch.int32)
    loss, draw = hip.empty(B), hip.empty(B * ld).fill_(7.0)
    q_on, q_tg, m_out, a_star = hip.empty(B * A * T), hip.empty(B * A * T), hip.empty(B * T), hip.zeros(B, dtype=torch.int32)
    hip.c51_head_loss_slabs(s_on.reshape(-1), ns_on, R_on, s_tg.reshape(-1), ns_tg, B if double_q else -1, b_on, b_tg, ld, A, T, dueling, act, rew, done, wgt, atoms,
                            0.97, -10.0, 10.0, B, loss, draw, state, q_on=q_on, q_tg=q_tg, m_out=m_out, a_star=a_star)
    # the separate launches on the same inputs
    def reduce(slabs, bias):
        acc = torch.zeros_like(slabs[0])
        for z in range(slabs.shape[0]):
            acc = acc + slabs[z]                # slab order
        return (acc + bias).contiguous()
    raw_on, raw_tg = reduce(s_on, b_on), reduce(s_tg, b_tg)
    q1, q2, q3 = hip.empty(B * A * T), hip.empty(B * A * T), hip.empty(B * A * T)
    hip.dueling_fwd(raw_on[:B].reshape(-1), ld, q1, B, A, T, dueling)
    hip.dueling_fwd(raw_tg.reshape(-1), ld, q2, B, A, T, dueling)
    a2 = hip.zeros(B, dtype=torch.int32)
    if double_q:
        hip.dueling_fwd(raw_on[B:].reshape(-1).contiguous(), ld, q3, B, A, T, dueling)
        hip.select_action(q3, A * T, T, 1, B, A, T, 2, atoms, a2, None, None)
    else:
        hip.select_action(q2, A * T, T, 1, B, A, T, 2, atoms, a2, None, None)
    loss2, dq2, m2, draw2 = hip.empty(B), hip.zeros(B * A * T), hip.empty(B * T), hip.empty(B * ld)
    hip.loss_c51(q1, q2, A, T, act, a2, rew, done, wgt, atoms, 0.97, -10.0, 10.0, B, loss2, dq2, m2, state)
    hip.dueling_bwd(dq2, draw2, ld, B, A, T, dueling)
    torch.cuda.synchronize()
    assert torch.equal(q_on, q1) and torch.equal(q_tg, q2), "combined logits"
    assert torch.equal(a_star, a2), "greedy next action"
    assert torch.equal(m_out, m2), "projected target distribution"
    assert torch.equal(loss, loss2), "per-sample loss"
    assert torch.equal(draw, draw2), "gradient w.r.t. the raw head output"
    assert int(state[0]) == 0
    # the multi-layer reduction: three layers of different slab counts in one launch == per-layer sums in slab order + bias + ReLU
    N = 512
    layers, want = [], []
    for i, (ns, rows) in enumerate([(8, B), (3, B), (5, max(B // 2, 1))]):
        sl = D(g.standard_normal((ns, rows, N)).astype(np.float32))
        bias = D(g.standard_normal(N).astype(np.float32))
        out = hip.empty(rows * N)
        layers.append((sl.reshape(-1), ns, bias, out, rows))
        want.append(torch.relu(reduce(sl, bias)).reshape(-1))
    hip.reduce_bias_act_multi(layers, N, True)
    torch.cuda.synchronize()
    for (_, _, _, out, _), wv in zip(layers, want):
        assert torch.equal(out, wv)


@pytest.mark.parametrize("A,T,dueling,double_q,B", [(4, 200, True, True, 512), (4, 200, False, False, 37), (18, 200, True, True, 33), (6, 11, True, False, 9), (3, 64, False, True, 130),
                                                    (2, 300, True, False, 5)])
def test_qr_head_loss_from_slabs_equals_the_separate_kernels(hip, A, T, dueling, double_q, B):
    """a0_qr_head_loss_slabs (round 5: one launch from the head GEMMs' split-K slabs to the quantile Huber loss + head gradient; reference agent.py:272-293) against the
    launches it replaces — slab reduction, a0_dueling_fwd x3, a0_select_action (mean over quantiles), a0_quantile_target, dq memset + a0_loss_quantile_huber,
    a0_dueling_bwd — on the same slabs: the same arithmetic statement for statement, so every output must be BIT-identical (torch.equal); T > 256 (strided
    quantile ownership), T not a multiple of four (scalar tail of the target sweep) and padded head columns included."""
    g = recipe.gen(A * 1000 + T + B)
    NQ = A + (1 if dueling else 0)
    ld = (NQ * T + 31) // 32 * 32
    R_on = 2 * B if double_q else B
    ns_on, ns_tg = 5, 3
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(hip.device)
    s_on = D((g.standard_normal((ns_on, R_on, ld)) * 0.7).astype(np.float32))
    s_tg = D((g.standard_normal((ns_tg, B, ld)) * 0.7).astype(np.float32))
    b_on, b_tg = D(g.standard_normal(ld).astype(np.float32)), D(g.standard_normal(ld).astype(np.float32))
    a, r, d, w = recipe.make_transitions(B, A, 5)
    act, rew, done, wgt = D(a.astype(np.int32)), D(r), D(d.astype(np.float32)), D(w)
    taus = ((2 * torch.arange(T, dtype=torch.float32) + 1) / (2.0 * T)).to(hip.device)
    state = hip.zeros(8, dtype=torch.int32)
    loss, draw = hip.empty(B), hip.empty(B * ld).fill_(7.0)
    q_on, q_tg, a_star = hip.empty(B * A * T), hip.empty(B * A * T), hip.zeros(B, dtype=torch.int32)
    hip.qr_head_loss_slabs(s_on.reshape(-1), ns_on, R_on, s_tg.reshape(-1), ns_tg, B if double_q else -1, b_on, b_tg, ld, A, T, dueling, act, rew, done, wgt, taus,
                           0.97, B, loss, draw, state, q_on=q_on, q_tg=q_tg, a_star=a_star)
    def reduce(slabs, bias):
        acc = torch.zeros_like(slabs[0])
        for z in range(slabs.shape[0]):
            acc = acc + slabs[z]                # slab order
        return (acc + bias).contiguous()
    raw_on, raw_tg = reduce(s_on, b_on), reduce(s_tg, b_tg)
    q1, q2, q3 = hip.empty(B * A * T), hip.empty(B * A * T), hip.empty(B * A * T)
    hip.dueling_fwd(raw_on[:B].reshape(-1), ld, q1, B, A, T, dueling)
    hip.dueling_fwd(raw_tg.reshape(-1), ld, q2, B, A, T, dueling)
    a2 = hip.zeros(B, dtype=torch.int32)
    if double_q:
        hip.dueling_fwd(raw_on[B:].reshape(-1).contiguous(), ld, q3, B, A, T, dueling)
        hip.select_action(q3, A * T, T, 1, B, A, T, 1, None, a2, None, None)
    else:
        hip.select_action(q2, A * T, T, 1, B, A, T, 1, None, a2, None, None)
    y = hip.empty(B * T)
    hip.quantile_target(q2, A * T, 1, T, a2, rew, done, 0.97, B, T, y)
    loss2, dq2, draw2 = hip.empty(B), hip.zeros(B * A * T), hip.empty(B * ld)
    hip.loss_quantile_huber(q1, A * T, 1, T, y, taus, 0, act, wgt, B, T, T, loss2, dq2, state)
    hip.dueling_bwd(dq2, draw2, ld, B, A, T, dueling)
    torch.cuda.synchronize()
    assert torch.equal(q_on, q1) and torch.equal(q_tg, q2), "combined quantile values"
    assert torch.equal(a_star, a2), "greedy next action"
    assert torch.equal(loss, loss2), "per-sample loss"
    assert torch.equal(draw, draw2), "gradient w.r.t. the raw head output"
    assert int(state[0]) == 0
    # and against a float64 evaluation of agent.py:110-114 on the device's own q values (the pair tensor materialised)
    qv = q1.view(B, A, T).double().cpu()
    qsel = qv[torch.arange(B), torch.from_numpy(a.astype(np.int64))]                                            # [B][T] online quantiles at the taken action
    tgt = y.view(B, T).double().cpu()
    dlt = qsel[:, None, :] - tgt[:, :, None]                                                                    # [B][N' = j][N = i]
    hub = torch.where(dlt.abs() < 1, 0.5 * dlt * dlt, dlt.abs() - 0.5)
    wq = (taus.double().cpu()[None, None, :] - (tgt[:, :, None] < qsel[:, None, :]).double()).abs()
    want = (hub * wq).sum(-1).mean(-1)
    assert float((loss.double().cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("A,dueling,B", [(4, False, 512), (4, True, 37), (18, True, 65), (23, True, 9), (6, False, 130)])
def test_mdqn_head_loss_from_slabs_equals_the_separate_kernels(hip, A, dueling, B):
    """a0_mdqn_head_loss_slabs (round 5: MDQNLearner.train_step, agent.py:193-215, from the three fc1 GEMMs' split-K slabs in one launch) against the separate
    kernels: fc1 = slab sums + bias + ReLU, the heads as a0_dense_fwd GEMMs + a0_dueling_fwd (another association of the 512-term dot products: compared at
    2e-6 of the scale), and — on the q values the fused kernel itself reports — a0_loss_mdqn + a0_dueling_bwd + a0_dense_dgrad bit for bit (loss, draw) / to
    rounding (dh).  The third pass must use the TARGET network's fc1 bias and head rows."""
    g = recipe.gen(A * 977 + B)
    NQ = A + (1 if dueling else 0)
    ld = (NQ + 31) // 32 * 32
    ns = 4
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(hip.device)
    s_on, s_tg, s_cur = (D((g.standard_normal((ns, B, 512)) * 0.5).astype(np.float32)) for _ in range(3))
    b1_on, b1_tg = D((g.standard_normal(512) * 0.3).astype(np.float32)), D((g.standard_normal(512) * 0.3).astype(np.float32))
    W_on, W_tg = D((g.standard_normal((ld, 512)) * 0.05).astype(np.float32)), D((g.standard_normal((ld, 512)) * 0.05).astype(np.float32))
    b_on, b_tg = D((g.standard_normal(ld) * 0.1).astype(np.float32)), D((g.standard_normal(ld) * 0.1).astype(np.float32))
    a, r, d, w = recipe.make_transitions(B, A, 5)
    act, rew, done, wgt = D(a.astype(np.int32)), D(r), D(d.astype(np.float32)), D(w)
    state = hip.zeros(8, dtype=torch.int32)
    loss, draw, dh, h_on = hip.empty(B), hip.empty(B * ld).fill_(7.0), hip.empty(B * 512), hip.empty(B * 512)
    q_on, q_tg, q_cur = hip.empty(B * A), hip.empty(B * A), hip.empty(B * A)
    tau, lo, gam = 0.03, -1.0, 0.97
    hip.mdqn_head_loss_slabs(s_on.reshape(-1), s_tg.reshape(-1), s_cur.reshape(-1), ns, b1_on, b1_tg, h_on, W_on.reshape(-1), b_on, W_tg.reshape(-1), b_tg, A, dueling, ld,
                             act, rew, done, wgt, gam, tau, lo, B, loss, q_on, q_tg, q_cur, draw, state, dh)
    def fc1(slabs, bias):
        acc = torch.zeros_like(slabs[0])
        for z in range(ns):
            acc = acc + slabs[z]
        return torch.relu(acc + bias).reshape(-1).contiguous()
    def head(h, W, b):
        raw, q = hip.empty(B * ld), hip.empty(B * A)
        hip.dense_fwd(h, 512, W.reshape(-1), b, raw, B, ld, 512, False, hip.empty(max(hip.dense_fwd_scratch(B, ld, 512), 4)))
        hip.dueling_fwd(raw, ld, q, B, A, 1, dueling)
        return q
    h1 = fc1(s_on, b1_on)
    q1, q2, q3 = head(h1, W_on, b_on), head(fc1(s_tg, b1_tg), W_tg, b_tg), head(fc1(s_cur, b1_tg), W_tg, b_tg)
    torch.cuda.synchronize()
    assert torch.equal(h_on, h1), "online fc1 activations"
    scale = float(q1.abs().max())
    for got, want, what in ((q_on, q1, "online(s)"), (q_tg, q2, "target(s')"), (q_cur, q3, "target(s)")):
        assert float((got - want).abs().max()) <= 2e-6 * max(scale, 1.0), what
    loss2, dq2, draw2, dh2 = hip.empty(B), hip.zeros(B * A), hip.empty(B * ld), hip.empty(B * 512)
    hip.loss_mdqn(q_on, q_tg, q_cur, A, act, rew, done, wgt, gam, tau, lo, B, loss2, dq2, state)
    hip.dueling_bwd(dq2, draw2, ld, B, A, 1, dueling)
    hip.dense_dgrad(draw2, W_on.reshape(-1), h1, dh2, B, ld, 512)
    torch.cuda.synchronize()
    assert torch.equal(loss, loss2), "per-sample Munchausen loss"
    assert torch.equal(draw, draw2), "gradient w.r.t. the raw head output"
    assert float((dh - dh2).abs().max()) <= 2e-6 * max(float(dh2.abs().max()), 1e-6)
    assert int(state[0]) == 0


G6P_GPU = ["fqf_b16_dq0_n1", "fqf_duel_b16_dq1_n3", "fqf_duel_a18_b16_dq1_n3"]


@pytest.mark.parametrize("case", G6P_GPU)
def test_golden_fqf_train_steps_pinned(hip, case):
    """Round 4: THREE consecutive reference ``FQFLearner.train()`` calls against the REFERENCE's numbers at the tight tolerances every other
    learner is held to.  The ordinary G6 fixtures cannot serve from the second step on — torch's CPU kernels round differently on this host than on the build
    container that recorded them, and FQF amplifies that (test_golden_train_steps) — so these fixtures (g6p_*) were recorded with torch on a CPU-independent code
    path (tests/golden/pinned.py), and the oracle, run the same way in a child process, must first reproduce them ON THIS MACHINE; then the HIP path, fed the
    fractions of that oracle run (cos(pi i tau) amplifies an ulp of tau ~200x: both sides evaluate q at the same fractions, as in every FQF parity test) and
    continued from its post-step state after each step, is compared with the FIXTURE: losses rtol 5e-5, gradient and parameter fingerprints as for G6."""
    from util import pinned_oracle
    name, B, dq, n = _case(case)
    spec = SPECS[name]
    g = golden(f"g6p_{case}")
    o = pinned_oracle(case)
    # Does the oracle, run in the pinned mode on THIS host, reproduce the reference's recorded numbers?  On the build container it does, bit for bit, over all
    # three steps (tests/test_oracle_golden.py::test_g6p_pinned_fqf_train_steps).  Measured on the GPU box's host (round 4): NOT quite — 5e-6 of the loss already
    # at step 0, i.e. MKL's reproducibility mode does not extend across CPU vendors — so there the tight comparison is against this machine's pinned oracle and
    # the fixture is held loosely, exactly like test_golden_train_steps does from the second step on; where the host does reproduce it, the fixture itself is
    # the tight reference for all three steps.
    dev_q = max(float(np.abs(o[f"s{s}::q_loss"] - g[f"s{s}::q_loss"]).max() / np.abs(g[f"s{s}::q_loss"]).mean()) for s in range(3))
    dev_f = max(float(np.abs(o[f"s{s}::fraction_loss"] - g[f"s{s}::fraction_loss"]).max() / np.abs(g[f"s{s}::fraction_loss"]).mean()) for s in range(3))
    pinned_here = dev_q <= 1e-6 and dev_f <= 1e-6
    _record(f"fqf_pinned_oracle_vs_reference_{case}", {"q_loss_max_rel_dev": dev_q, "fraction_loss_max_rel_dev": dev_f, "host_reproduces_fixture": pinned_here})
    ref = g if pinned_here else o
    L, dev = _device_learner(hip, spec, B, dq, n, target_update_freq=2)
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x).reshape(-1).copy()).to(hip.device)
    T = lambda prefix, s: {k.split("::")[2]: torch.from_numpy(v) for k, v in o.items() if k.startswith(f"s{s}::{prefix}::")}
    for s in range(3):
        frames, a, r, d, w = _batch(hip, spec, B, 61 + s, 62 + s)
        rand = [D(o[f"s{s}::{nm}_{i}"]) for i in range(2) for nm in ("taus", "tau_hats")]
        loss, frac = dev.update(frames, None, 2 * int(np.prod(spec.obs_shape)), a, r, d, w, rand=rand)
        assert_close(loss[:B], ref[f"s{s}::q_loss"], 5e-5, 5e-6, f"s{s} q_loss")
        assert_close(frac[:B], ref[f"s{s}::fraction_loss"], 5e-5, 2e-5, f"s{s} fraction_loss")
        if s == 0:      # from the common initial state the reference's numbers hold on any host (loosely where the host does not reproduce them: measured
            # on the GPU box, the pinned oracle itself is 5e-6 off the fixture at step 0, 1.5e-3 / 4.4e-2 of the two losses by step 2 — FQF's amplification)
            assert_close(loss[:B], g[f"s{s}::q_loss"], 1e-3, 1e-4, f"s{s} q_loss vs the reference's fixture")
            assert_close(frac[:B], g[f"s{s}::fraction_loss"], 5e-3, 5e-4, f"s{s} fraction_loss vs the reference's fixture")
        assert int(dev.state[1]) == int(g[f"s{s}::update_steps"])
        grads = L.unpack(dev.grads)
        params, target = dev.online.state_dict(), dev.target.state_dict()
        n_cmp = 0
        for k in g.files:
            parts = k.split("::")
            if parts[0] != f"s{s}" or len(parts) < 3 or k not in ref:
                continue
            want_k = ref[k]
            if parts[1] == "grad" and parts[2] in grads:
                got = recipe.checksum(grads[parts[2]].cpu().numpy())
                scale = max(want_k[1], 1e-6)
                assert abs(got[1] - want_k[1]) <= 3e-4 * scale, (k, got[1], want_k[1])
                assert np.all(np.abs(got[2:] - want_k[2:]) <= 3e-4 * scale / np.sqrt(max(grads[parts[2]].numel(), 1)) + 2e-4 * np.abs(want_k[2:]) + 1e-7), (k, got, want_k)
                n_cmp += 1
            elif parts[1] in ("param", "target"):
                src = params if parts[1] == "param" else target
                if parts[2] not in src:
                    continue
                got = recipe.checksum(src[parts[2]].cpu().numpy())
                assert abs(got[1] - want_k[1]) <= 1e-5 * max(want_k[1], 1e-6), (k, got[1], want_k[1])
                assert np.all(np.abs(got[2:] - want_k[2:]) <= 5e-5 + 1e-4 * np.abs(want_k[2:])), (k, got[2:], want_k[2:])
                n_cmp += 1
        assert n_cmp > 30
        if s < 2:      # continue from the pinned oracle's post-step state (== the reference's, by the check above): every step from a common state
            po, pt = T("po", s), T("pt", s)
            L.pack(po, dev.online.flat)
            L.pack(pt, dev.target.flat)
            dev.online.refresh_wt(); dev.target.refresh_wt()
            fk = [k for k in po if "fraction" in k]
            zeros = {k: torch.zeros_like(po[k]) for k in fk}
            L.pack({**T("adam_m", s), **zeros}, dev.adam_m)
            L.pack({**T("adam_v", s), **zeros}, dev.adam_v)


@pytest.mark.parametrize("Bs", [(512, 512), (512, 512, 512), (100, 37, 512), (3, 1), (700, 64, 2)])
def test_encoder_passes_in_one_launch_equal_separate_launches(hip, Bs):
    """a0_net_encoder_fwd_fused_multi (round 4): up to three forward passes — own weights, own frames / slot gather / row half, own outputs — share one launch of at
    most 256 looping workgroups.  Each pass must produce exactly the bytes of its own a0_net_encoder_fwd_fused launch (torch.equal on act1 / act2 / act3), for
    equal and very unequal pass sizes (every pass keeps at least one workgroup) and for passes that do not store act1 / act2."""
    from agent0_amd.deepq.engine import DeviceNet, Workspace
    from agent0_amd.deepq.layout import NetLayout
    spec = recipe.NetSpec("dqn", 4)
    L = NetLayout.from_spec(spec)
    nets_ = []
    for seed in (11, 12):
        n = DeviceNet(hip, L, hip.net(4, 84, 84))
        n.load_state_dict(recipe.make_state_dict(spec, seed))
        nets_.append(n)
    cap = max(Bs) + 5
    ring = torch.from_numpy(recipe.make_frames(cap, 5, spec.obs_shape)).to(hip.device).reshape(-1).contiguous()
    passes, want = [], []
    for i, B in enumerate(Bs):
        net = nets_[i % 2]
        slot = None if i == 1 else torch.from_numpy(recipe.gen(6 + i).permutation(cap)[:B].astype(np.int32)).to(hip.device)
        chan_off = 28224 if i != 2 else 0
        keep = i == len(Bs) - 1
        ws, ws2 = Workspace(hip, L, B), Workspace(hip, L, B)
        for w_ in (ws, ws2):
            w_.act1.fill_(-7.0); w_.act2.fill_(-7.0); w_.act3.fill_(-7.0)
        hip.encoder_fwd_fused(net.net, net.wt, net.encoder_weights(), ring, slot, 2 * 28224, chan_off, B, ws.act1 if keep else None, ws.act2 if keep else None, ws.act3)
        passes.append((net.wt, net.encoder_weights(), ring, slot, 2 * 28224, chan_off, B, ws2.act1 if keep else None, ws2.act2 if keep else None, ws2.act3))
        want.append((ws, ws2, keep))
    hip.encoder_fwd_fused_multi(nets_[0].net, passes)
    torch.cuda.synchronize()
    for i, (ws, ws2, keep) in enumerate(want):
        assert torch.equal(ws.act3, ws2.act3), f"pass {i}: features"
        assert torch.equal(ws.act1, ws2.act1) and torch.equal(ws.act2, ws2.act2), f"pass {i}: stored activations (or untouched buffers)"
        assert float(ws2.act3.min()) >= 0.0


@pytest.mark.parametrize("A,dueling,double_q,n_step,gather", [(4, False, False, 1, True), (4, True, True, 3, True), (18, True, False, 1, False), (6, False, True, 1, True)])
def test_native_learner_handle_equals_the_per_kernel_composition(hip, A, dueling, double_q, n_step, gather):
    """``a0_learner_*`` (round 4; SURVEY §8(b): opaque handle, library-owned HBM, BaseLearner.train as ONE C call): the handle's update must leave exactly the
    parameters, target parameters, Adam moments, status words and per-sample losses that agent0_amd/deepq/engine.py's composition of the per-kernel entry points
    leaves — torch.equal after each of five updates from the same packed parameters on the same batches (ring-slot gather and dense batch), across a target sync
    (target_update_freq = 3), for plain / dueling / double-Q heads and n-step discounting."""
    from agent0_amd.deepq.engine import DeviceLearner
    from agent0_amd.deepq.layout import NetLayout
    spec = recipe.NetSpec("dqn", A, dueling=dueling)
    L = NetLayout.from_spec(spec)
    B, cap = 64, 200
    dev = DeviceLearner(hip, L, B, n_step=n_step, double_q=double_q, target_update_freq=3)
    dev.online.load_state_dict(recipe.make_state_dict(spec, 11))
    dev.target.load_state_dict(recipe.make_state_dict(spec, 12))
    nat = hip.native_learner(A=A, dueling=dueling, double_q=double_q, B=B, n_step=n_step, discount=0.99, lr=5e-4, target_update_freq=3)
    assert nat.n == L.n_params_padded
    nat.set_params(dev.online.flat, dev.target.flat)
    ring = torch.from_numpy(recipe.make_frames(cap, 5, spec.obs_shape)).to(hip.device).reshape(-1).contiguous()
    loss_n = hip.empty(B)
    for s in range(5):
        slot = torch.from_numpy(recipe.gen(40 + s).permutation(cap)[:B].astype(np.int32)).to(hip.device) if gather else None
        a_np, r_np, d_np, w_np = recipe.make_transitions(B, A, 70 + s)
        a, r, d, w = (torch.from_numpy(x).to(hip.device) for x in (a_np.astype(np.int32), r_np, d_np.astype(np.float32), w_np))
        loss_e = dev.update(ring, slot, 2 * 28224, a, r, d, w).clone()
        nat.update(ring, slot, 2 * 28224, a, r, d, w, loss_out=loss_n)
        torch.cuda.synchronize()
        on, tg, m, v, st = nat.get()
        torch.cuda.synchronize()
        assert torch.equal(loss_n, loss_e[:B]), f"update {s}: per-sample losses"
        assert torch.equal(on, dev.online.flat) and torch.equal(tg, dev.target.flat), f"update {s}: parameters / target"
        assert torch.equal(m, dev.adam_m) and torch.equal(v, dev.adam_v), f"update {s}: Adam moments"
        assert torch.equal(st, dev.state), f"update {s}: status words {st.tolist()} vs {dev.state.tolist()}"
    assert int(st[1]) == 5 and not torch.equal(on, tg)
    nat.close()


@pytest.mark.parametrize("A,dueling,double_q,n_step,noisy,B", [(4, False, False, 1, False, 64), (4, True, True, 3, True, 64), (18, True, True, 3, True, 32), (6, False, True, 1, True, 64),
                                                              (4, True, True, 3, True, 512)], ids=["c51", "rainbow-lite", "rainbow-lite-a18", "c51-noisy-dq", "rainbow-lite-b512"])
def test_native_c51_learner_handle_equals_the_per_kernel_composition(hip, A, dueling, double_q, n_step, noisy, B):
    """The a0_learner handle with algo = A0_ALGO_C51 (BASELINE configs[2]: c51 + NoisyNet + dueling + double-Q + n-step): one a0_learner_update call per update — the
    joint Philox noise fill of both networks (agent.py:125-127), the composed weights, the three encoder passes, the 2B-row online fc1 / head GEMMs, projection +
    cross entropy, backward, the sigma gradients, Adam with the target sync — must leave exactly what the Python classes leave: BaseLearner's noise draws
    (DeviceRng, stream 4) + DeviceLearner.update.  torch.equal on losses, parameters, target, Adam moments and status words after each of five updates across a
    target sync; the default support equals torch.linspace's."""
    from agent0_amd.common.utils import DeviceRng
    from agent0_amd.deepq.engine import DeviceLearner
    from agent0_amd.deepq.layout import NetLayout
    spec = recipe.NetSpec("c51", A, dueling=dueling, noisy=noisy)
    L = NetLayout.from_spec(spec)
    cap = max(200, B + 40)
    dev = DeviceLearner(hip, L, B, n_step=n_step, double_q=double_q, target_update_freq=3)
    dev.online.load_state_dict(recipe.make_state_dict(spec, 11))
    dev.target.load_state_dict(recipe.make_state_dict(spec, 12))
    seed = 42 + 15485863
    rng = DeviceRng(hip, seed)
    if noisy:
        rng.reserve(rng.STREAM_NOISE, dev.online.noise_len); rng.reserve(rng.STREAM_NOISE, dev.target.noise_len)      # BaseLearner.__init__: both networks' first noise
    nat = hip.native_learner(A=A, dueling=dueling, double_q=double_q, B=B, n_step=n_step, discount=0.99, lr=5e-4, target_update_freq=3, algo="c51", num_atoms=L.T, vmin=dev.vmin,
                             vmax=dev.vmax, noisy=noisy, seed=seed)
    assert nat.n == L.n_params_padded
    nat.set_params(dev.online.flat, dev.target.flat)
    if A == 6:
        nat.set_support(dev.atoms.cpu().tolist())          # the caller's own support values (the other cases run on the handle's default: torch.linspace's)
    ring = torch.from_numpy(recipe.make_frames(cap, 5, spec.obs_shape)).to(hip.device).reshape(-1).contiguous()
    loss_n = hip.empty(B)
    for s in range(5):
        slot = torch.from_numpy(recipe.gen(40 + s).permutation(cap)[:B].astype(np.int32)).to(hip.device)
        a_np, r_np, d_np, w_np = recipe.make_transitions(B, A, 70 + s)
        a, r, d, w = (torch.from_numpy(x).to(hip.device) for x in (a_np.astype(np.int32), r_np, d_np.astype(np.float32), w_np))
        if noisy:
            assert dev.noise_joint is not None
            rng.normal(rng.STREAM_NOISE, 0.1, dev.noise_joint, dev.noise_joint.numel())         # BaseLearner.train_batch
        loss_e = dev.update(ring, slot, 2 * 28224, a, r, d, w).clone()
        nat.update(ring, slot, 2 * 28224, a, r, d, w, loss_out=loss_n)
        torch.cuda.synchronize()
        on, tg, m, v, st = nat.get()
        torch.cuda.synchronize()
        assert torch.equal(loss_n, loss_e[:B]), f"update {s}: per-sample losses (max diff {float((loss_n - loss_e[:B]).abs().max())})"
        assert torch.equal(on, dev.online.flat) and torch.equal(tg, dev.target.flat), f"update {s}: parameters / target"
        assert torch.equal(m, dev.adam_m) and torch.equal(v, dev.adam_v), f"update {s}: Adam moments"
        assert torch.equal(st, dev.state), f"update {s}: status words {st.tolist()} vs {dev.state.tolist()}"
    assert int(st[1]) == 5 and not torch.equal(on, tg) and float(loss_n.min()) > 0.0
    if noisy:
        sg = L.blocks["fc1.sigma"]
        assert float(m[sg.all].abs().max()) > 0.0, "the sigma blocks are trained"
    nat.close()


@pytest.mark.parametrize("A,dueling,double_q,n_step,B,KNN", [(9, False, False, 1, 32, (32, 64, 64)), (9, True, True, 3, 16, (32, 64, 64)), (4, False, True, 1, 8, (8, 16, 24)),
                                                              (9, False, False, 1, 512, (32, 64, 64))], ids=["iqn", "iqn-duel-double-n3", "iqn-small-taus", "iqn-b512"])
def test_native_iqn_learner_handle_equals_the_per_kernel_composition(hip, A, dueling, double_q, n_step, B, KNN):
    """The a0_learner handle with algo = A0_ALGO_IQN (BASELINE configs[3]): the three tau draws of an update (Philox stream 3 of the learner's seed, in BaseLearner's
    order K, N', N), the cosine-embedding heads, the greedy next action, the quantile target, the quantile Huber loss, the backward pass over B * N rows with the
    embedding's weight gradient, Adam with the target sync — one a0_learner_update call, torch.equal to DeviceRng + DeviceLearner.update on losses, parameters, target,
    Adam moments and status words after each of four updates across a target sync."""
    from agent0_amd.common.utils import DeviceRng
    from agent0_amd.deepq.engine import DeviceLearner
    from agent0_amd.deepq.layout import NetLayout
    K, N, Nd = KNN
    spec = recipe.NetSpec("iqn", A, dueling=dueling)
    L = NetLayout.from_spec(spec)
    cap = max(200, B + 40)
    dev = DeviceLearner(hip, L, B, n_step=n_step, double_q=double_q, target_update_freq=3, K=K, N=N, N_dash=Nd)
    dev.online.load_state_dict(recipe.make_state_dict(spec, 11))
    dev.target.load_state_dict(recipe.make_state_dict(spec, 12))
    seed = 42 + 15485863
    rng = DeviceRng(hip, seed)
    nat = hip.native_learner(A=A, dueling=dueling, double_q=double_q, B=B, n_step=n_step, discount=0.99, lr=5e-4, target_update_freq=3, algo="iqn", seed=seed, K=K, N=N, N_dash=Nd)
    assert nat.n == L.n_params_padded
    nat.set_params(dev.online.flat, dev.target.flat)
    ring = torch.from_numpy(recipe.make_frames(cap, 5, spec.obs_shape)).to(hip.device).reshape(-1).contiguous()
    loss_n = hip.empty(B)
    taus = [hip.empty(B * n) for n in (K, Nd, N)]
    for s in range(4):
        slot = torch.from_numpy(recipe.gen(40 + s).permutation(cap)[:B].astype(np.int32)).to(hip.device)
        a_np, r_np, d_np, w_np = recipe.make_transitions(B, A, 70 + s)
        a, r, d, w = (torch.from_numpy(x).to(hip.device) for x in (a_np.astype(np.int32), r_np, d_np.astype(np.float32), w_np))
        for t in taus:
            rng.uniform(rng.STREAM_TAUS, t, t.numel())                     # BaseLearner.train_batch
        loss_e = dev.update(ring, slot, 2 * 28224, a, r, d, w, rand=taus).clone()
        nat.update(ring, slot, 2 * 28224, a, r, d, w, loss_out=loss_n)
        torch.cuda.synchronize()
        on, tg, m, v, st = nat.get()
        torch.cuda.synchronize()
        assert torch.equal(loss_n, loss_e[:B]), f"update {s}: per-sample losses (max diff {float((loss_n - loss_e[:B]).abs().max())})"
        assert torch.equal(on, dev.online.flat) and torch.equal(tg, dev.target.flat), f"update {s}: parameters / target"
        assert torch.equal(m, dev.adam_m) and torch.equal(v, dev.adam_v), f"update {s}: Adam moments"
        assert torch.equal(st, dev.state), f"update {s}: status words {st.tolist()} vs {dev.state.tolist()}"
    assert int(st[1]) == 4 and not torch.equal(on, tg) and float(loss_n.min()) >= 0.0
    cb = L.blocks["cos"]
    assert float(m[cb.all].abs().max()) > 0.0, "the cosine embedding is trained"
    nat.close()


@pytest.mark.parametrize("A,dueling,double_q,n_step,B", [(9, False, False, 1, 32), (18, True, True, 3, 16), (9, False, False, 1, 512)], ids=["fqf", "fqf-a18-duel-double-n3", "fqf-b512"])
def test_native_fqf_learner_handle_equals_the_per_kernel_composition(hip, A, dueling, double_q, n_step, B):
    """The a0_learner handle with algo = A0_ALGO_FQF (BASELINE configs[4]): fractions proposed by the fraction net on the detached features, quantile values at the
    tau-hats (the target at the ONLINE tau-hats, quirk Q16), quantile Huber loss, the fraction loss from the values at the interior fractions, backward over B * F rows,
    the fraction net's own RMSprop step in front of Adam — one a0_learner_update call, torch.equal to DeviceLearner.update on the losses, the fraction losses, parameters
    (fraction net included), target, Adam moments and status words after each of four updates across a target sync."""
    from agent0_amd.deepq.engine import DeviceLearner
    from agent0_amd.deepq.layout import NetLayout
    spec = recipe.NetSpec("fqf", A, dueling=dueling)
    L = NetLayout.from_spec(spec)
    cap = max(200, B + 40)
    dev = DeviceLearner(hip, L, B, n_step=n_step, double_q=double_q, target_update_freq=3)
    dev.online.load_state_dict(recipe.make_state_dict(spec, 11))
    dev.target.load_state_dict(recipe.make_state_dict(spec, 12))
    nat = hip.native_learner(A=A, dueling=dueling, double_q=double_q, B=B, n_step=n_step, discount=0.99, lr=5e-4, target_update_freq=3, algo="fqf", seed=1, F=L.F)
    assert nat.n == L.n_params_padded
    nat.set_params(dev.online.flat, dev.target.flat)
    ring = torch.from_numpy(recipe.make_frames(cap, 5, spec.obs_shape)).to(hip.device).reshape(-1).contiguous()
    loss_n, fl_n = hip.empty(B), hip.empty(B)
    fb = L.blocks["frac"]
    frac0 = dev.online.flat[fb.all].clone()
    for s in range(4):
        slot = torch.from_numpy(recipe.gen(40 + s).permutation(cap)[:B].astype(np.int32)).to(hip.device)
        a_np, r_np, d_np, w_np = recipe.make_transitions(B, A, 70 + s)
        a, r, d, w = (torch.from_numpy(x).to(hip.device) for x in (a_np.astype(np.int32), r_np, d_np.astype(np.float32), w_np))
        loss_e, fl_e = dev.update(ring, slot, 2 * 28224, a, r, d, w)
        loss_e, fl_e = loss_e.clone(), fl_e.clone()
        nat.update(ring, slot, 2 * 28224, a, r, d, w, loss_out=loss_n)
        nat.frac_loss(fl_n)
        torch.cuda.synchronize()
        on, tg, m, v, st = nat.get()
        torch.cuda.synchronize()
        assert torch.equal(loss_n, loss_e[:B]), f"update {s}: per-sample losses (max diff {float((loss_n - loss_e[:B]).abs().max())})"
        assert torch.equal(fl_n, fl_e[:B]), f"update {s}: fraction losses"
        assert torch.equal(on, dev.online.flat) and torch.equal(tg, dev.target.flat), f"update {s}: parameters / target"
        assert torch.equal(m, dev.adam_m) and torch.equal(v, dev.adam_v), f"update {s}: Adam moments"
        assert torch.equal(st, dev.state), f"update {s}: status words {st.tolist()} vs {dev.state.tolist()}"
    assert int(st[1]) == 4 and not torch.equal(on, tg)
    assert not torch.equal(on[fb.all], frac0), "the fraction net moved (its RMSprop step)"
    nat.close()


@pytest.mark.parametrize("algo,A,dueling,double_q,n_step,noisy,B", [("qr", 4, False, False, 1, False, 32), ("qr", 6, True, True, 3, True, 32), ("qr", 4, False, True, 1, False, 512),
                                                                   ("mdqn", 4, False, False, 1, False, 64), ("mdqn", 18, True, False, 3, True, 32)],
                         ids=["qr", "qr-duel-double-n3-noisy", "qr-b512", "mdqn", "mdqn-a18-duel-n3-noisy"])
def test_native_qr_and_mdqn_learner_handles_equal_the_per_kernel_composition(hip, algo, A, dueling, double_q, n_step, noisy, B):
    """The reference's remaining learners behind the handle: A0_ALGO_QR (200 fixed quantiles per action, quantile Huber at the midpoints, agent.py:272-293) and
    A0_ALGO_MDQN (Munchausen DQN, agent.py:194-215: the target network also on the current observation), with NoisyNet, dueling, double-Q (qr) and n-step —
    torch.equal to DeviceRng + DeviceLearner.update on losses, parameters, target, Adam moments and status words after each of four updates across a target sync."""
    from agent0_amd.common.utils import DeviceRng
    from agent0_amd.deepq.engine import DeviceLearner
    from agent0_amd.deepq.layout import NetLayout
    spec = recipe.NetSpec(algo, A, dueling=dueling, noisy=noisy, **({"num_atoms": 200} if algo == "qr" else {}))
    L = NetLayout.from_spec(spec)
    cap = max(200, B + 40)
    dev = DeviceLearner(hip, L, B, n_step=n_step, double_q=double_q, target_update_freq=3)
    dev.online.load_state_dict(recipe.make_state_dict(spec, 11))
    dev.target.load_state_dict(recipe.make_state_dict(spec, 12))
    seed = 42 + 15485863
    rng = DeviceRng(hip, seed)
    if noisy:
        rng.reserve(rng.STREAM_NOISE, dev.online.noise_len); rng.reserve(rng.STREAM_NOISE, dev.target.noise_len)      # BaseLearner.__init__: both networks' first noise
    nat = hip.native_learner(A=A, dueling=dueling, double_q=double_q, B=B, n_step=n_step, discount=0.99, lr=5e-4, target_update_freq=3, algo=algo, num_atoms=L.T, noisy=noisy,
                             seed=seed, mdqn_tau=dev.mdqn_tau, mdqn_lo=dev.mdqn_lo)
    assert nat.n == L.n_params_padded
    nat.set_params(dev.online.flat, dev.target.flat)
    ring = torch.from_numpy(recipe.make_frames(cap, 5, spec.obs_shape)).to(hip.device).reshape(-1).contiguous()
    loss_n = hip.empty(B)
    for s in range(4):
        slot = torch.from_numpy(recipe.gen(40 + s).permutation(cap)[:B].astype(np.int32)).to(hip.device)
        a_np, r_np, d_np, w_np = recipe.make_transitions(B, A, 70 + s)
        a, r, d, w = (torch.from_numpy(x).to(hip.device) for x in (a_np.astype(np.int32), r_np, d_np.astype(np.float32), w_np))
        if noisy:
            rng.normal(rng.STREAM_NOISE, 0.1, dev.noise_joint, dev.noise_joint.numel())
        loss_e = dev.update(ring, slot, 2 * 28224, a, r, d, w).clone()
        nat.update(ring, slot, 2 * 28224, a, r, d, w, loss_out=loss_n)
        torch.cuda.synchronize()
        on, tg, m, v, st = nat.get()
        torch.cuda.synchronize()
        assert torch.equal(loss_n, loss_e[:B]), f"update {s}: per-sample losses (max diff {float((loss_n - loss_e[:B]).abs().max())})"
        assert torch.equal(on, dev.online.flat) and torch.equal(tg, dev.target.flat), f"update {s}: parameters / target"
        assert torch.equal(m, dev.adam_m) and torch.equal(v, dev.adam_v), f"update {s}: Adam moments"
        assert torch.equal(st, dev.state), f"update {s}: status words {st.tolist()} vs {dev.state.tolist()}"
    assert int(st[1]) == 4 and not torch.equal(on, tg)
    nat.close()


def test_plain_c_host_drives_a_learner_through_the_c_abi(tmp_path):
    """The drop-in boundary is a C-ABI: tests/c_host_demo.c — plain C, include/agent0_hip.h and the HIP runtime, no Python, no torch — creates an a0_learner, loads
    parameters, runs three dueling double-Q n-step updates (one a0_learner_update call each) and reads the state back.  Compiled here with gcc (the HIP runtime's C API) against the in-tree
    library and run as a child process."""
    import json, os, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no C compiler / ROCm headers on this box")
    exe = str(tmp_path / "c_host_demo")
    lib = os.path.join(root, "agent0_amd", "lib")
    r = subprocess.run([gcc, "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "tests", "c_host_demo.c"), "-I/opt/rocm/include", "-I", os.path.join(root, "include"), "-L", lib,
                        "-lagent0_hip", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["update_steps"] == 3 and out["nan_skipped"] == 0 and out["finite"] == 1 and out["param_floats"] > 1_600_000
    assert out["moved_l1"] > 0 and out["mean_loss"] > 0
    assert out["online_minus_target_l1"] > 0, "the target was synced after update 2 and the online network moved once more"
