"""CPU: the oracle restatement vs golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  This is what "pins" the oracle.

Tolerances: the oracle uses the same torch CPU kernels as the reference but in a
different op order in places (e.g. scatter_add_ vs index_add_, hand-written Adam),
so fp32 results agree to a few ulp — checked with rtol 2e-5 / atol 2e-6 unless noted.
Integer / byte / boolean outputs are compared exactly.
"""
import hashlib
from collections import OrderedDict

import numpy as np
import pytest
import torch

import recipe
from recipe import SPECS
from oracle import actor as oactor
from oracle import learner as olearner
from oracle import losses, nets, replay as oreplay, schedules
from oracle.losses import Hyper
from util import assert_close, golden

RT, AT = 2e-5, 2e-6


def params_for(spec, seed):
    p = olearner.to_params(recipe.make_state_dict(spec, seed))
    return p


def set_noise_from(p, spec, buf):
    """buf: dict key -> array for noise_in/out vectors."""
    for prefix in nets.dense_prefixes(spec):
        for leaf in ("noise_in", "noise_out_weight", "noise_out_bias"):
            p[f"{prefix}.{leaf}"] = torch.from_numpy(np.array(buf[f"{prefix}.{leaf}"]))
        nets.compose_noise(p, prefix)


@pytest.mark.parametrize("name", list(SPECS))
def test_g1_forward(name):
    spec = SPECS[name]
    g = golden(f"g1_{name}")
    p = params_for(spec, 11)
    if spec.noisy:
        set_noise_from(p, spec, {k[len("buf::"):]: g[k] for k in g.files if k.startswith("buf::")})
        for k in g.files:
            if k.startswith("bufsum::"):
                assert_close(recipe.checksum(p[k[len("bufsum::"):]].numpy()), g[k], 1e-6, 1e-7, k)
    frames = recipe.make_frames(8, seed=21, obs_shape=spec.obs_shape)
    x = nets.normalize(torch.from_numpy(frames[:, : spec.obs_shape[0]].copy()))
    with torch.no_grad():
        if spec.algo == "iqn":
            feat = nets.encoder(p, x)
            assert_close(feat, g["features"], RT, AT, "features")
            assert_close(nets.head_iqn(p, spec, feat, torch.from_numpy(g["taus_16"])), g["q_16"], RT, AT, "q_16")
            assert_close(nets.qval(p, spec, x, torch.from_numpy(g["taus_qval"])), g["qval"], RT, AT, "qval")
        elif spec.algo == "fqf":
            feat = nets.encoder(p, x)
            t, th, ent = nets.fqf_prop_taus(p, spec, feat)
            assert_close(t, g["taus"], RT, AT, "taus")
            assert_close(th, g["taus_hat"], RT, AT, "taus_hat")
            assert_close(ent, g["entropies"], RT, AT, "ent")
            assert_close(nets.head_iqn(p, spec, feat, th), g["q_hat"], RT, AT, "q_hat")
            assert_close(nets.qval(p, spec, x), g["qval"], RT, AT, "qval")
        else:
            assert_close(nets.forward(p, spec, x), g["out"], RT, AT, "out")
            assert_close(nets.qval(p, spec, x), g["qval"], RT, AT, "qval")


def test_g2_layers():
    spec = SPECS["dqn"]
    g = golden("g2_layers")
    p = params_for(spec, 11)
    x = nets.normalize(torch.from_numpy(recipe.make_frames(2, seed=22)[:, :4].copy()))
    with torch.no_grad():
        feat, (a1, a2, a3) = nets.encoder(p, x, return_all=True)
    assert_close(a1, g["convs_1"], RT, AT, "conv1+relu")
    assert_close(a2, g["convs_3"], RT, AT, "conv2+relu")
    assert_close(a3, g["convs_5"], RT, AT, "conv3+relu")
    assert_close(feat, g["convs_6"], RT, AT, "flatten (C,H,W)")


G3 = ["dqn_b8_dq0_n1", "dqn_b8_dq1_n3", "dqn_duel_b8_dq1_n1", "mdqn_b8_dq0_n1", "c51_b8_dq0_n1", "c51_b8_dq1_n3",
      "c51_duel_noisy_b8_dq1_n3", "qr_b8_dq0_n1", "qr_duel_b8_dq1_n3", "iqn_b8_dq0_n1", "iqn_duel_b8_dq1_n3",
      "fqf_b8_dq0_n1", "fqf_b8_dq1_n3", "dqn_tiny_b32_dq1_n1", "c51_tiny_b32_dq1_n3", "dqn_b512_dq0_n1", "c51_b512_dq1_n3",
      # the reference's own suite configuration (README.md:62-112: fqf + double-Q + dueling) and the 18-action games
      "fqf_duel_b8_dq1_n3", "dqn_duel_a18_b8_dq1_n1", "fqf_duel_a18_b8_dq1_n3"]


def parse_case(case):
    name, b, dq, n = case.rsplit("_", 3)
    return name, int(b[1:]), bool(int(dq[2:])), int(n[1:])


def batch_inputs(spec, B, seed_f, seed_t):
    frames = recipe.make_frames(B, seed=seed_f, obs_shape=spec.obs_shape)
    a, r, d, w = recipe.make_transitions(B, spec.action_dim, seed=seed_t)
    return frames, a, r, d, w


@pytest.mark.parametrize("case", G3)
def test_g3_train_step_losses(case):
    name, B, dq, n = parse_case(case)
    spec = SPECS[name]
    g = golden(f"g3_{case}")
    po, pt = params_for(spec, 11), params_for(spec, 12)
    if spec.noisy:
        for tag, p in (("online", po), ("target", pt)):
            set_noise_from(p, spec, {k.split("::")[2]: g[k] for k in g.files if k.startswith(f"noise::{tag}::")})
    frames, a, r, d, w = batch_inputs(spec, B, 31, 32)
    ft = nets.normalize(torch.from_numpy(frames))
    obs, nxt = torch.split(ft, spec.obs_shape[0], 1)
    rand = [torch.from_numpy(g[f"rand_{i}"]) for i in range(3)] if spec.algo == "iqn" else None
    hp = Hyper(double_q=dq, n_step=n)
    with torch.no_grad():
        loss, floss = losses.train_step(po, pt, spec, hp, obs, torch.from_numpy(a), torch.from_numpy(r),
                                        torch.from_numpy(d).float(), nxt, rand)
    assert_close(loss, g["loss"], 5e-5, 5e-6, "loss")
    if spec.algo == "fqf":
        assert_close(floss, g["fraction_loss"], 5e-5, 5e-6, "fraction_loss")


@pytest.mark.parametrize("n", [1, 3])
def test_g4_c51_projection(n):
    g = golden(f"g4_c51_projection_n{n}")
    spec = SPECS["c51"]
    hp = Hyper(n_step=n)
    atoms = nets.c51_atoms(spec)
    m = losses.c51_project(torch.from_numpy(g["prob_next_sel"]), torch.from_numpy(g["rewards"]),
                           torch.from_numpy(g["terminals"]), atoms, hp, delta=(20.0 / 50))
    assert_close(m, g["target_prob"], 1e-6, 1e-7, "projected target distribution")
    assert_close(m.sum(-1), np.ones(m.shape[0]), 1e-5, 0, "mass conserved")
    # and the full loss through the oracle path
    po, pt = params_for(spec, 11), params_for(spec, 12)
    ft = nets.normalize(torch.from_numpy(recipe.make_frames(len(g["rewards"]), seed=41)))
    obs, nxt = torch.split(ft, 4, 1)
    with torch.no_grad():
        loss = losses.c51_loss(po, pt, spec, hp, obs, torch.from_numpy(g["actions"]), torch.from_numpy(g["rewards"]),
                               torch.from_numpy(g["terminals"]), nxt)
    assert_close(loss, g["loss"], 5e-5, 5e-6, "c51 loss")


@pytest.mark.parametrize("tag", ["4x200x200", "8x64x64", "8x32x32", "3x8x5"])
def test_g5_quantile_huber(tag):
    g = golden("g5_huber_qr")
    q = torch.from_numpy(g[f"q_{tag}"]).requires_grad_(True)
    loss = losses.huber_quantile(q, torch.from_numpy(g[f"t_{tag}"]), torch.from_numpy(g[f"tau_{tag}"]))
    (loss * torch.from_numpy(g[f"w_{tag}"])).sum().backward()
    assert_close(loss, g[f"loss_{tag}"], 1e-5, 1e-6, "loss")
    assert_close(q.grad, g[f"dq_{tag}"], 1e-5, 1e-7, "dloss/dq")


G6 = ["dqn_b16_dq0_n1", "dqn_duel_b16_dq1_n3", "c51_duel_noisy_b16_dq1_n3", "c51_b16_dq0_n1", "qr_b16_dq0_n1",
      "iqn_b16_dq0_n1", "fqf_b16_dq0_n1", "mdqn_b16_dq0_n1", "dqn_tiny_b32_dq1_n1", "c51_tiny_b32_dq1_n3", "dqn_b512_dq0_n1",
      "fqf_duel_b16_dq1_n3", "dqn_duel_a18_b16_dq1_n1", "fqf_duel_a18_b16_dq1_n3"]


def run_oracle_g6(case, g):
    name, B, dq, n = parse_case(case)
    spec = SPECS[name]
    L = olearner.OracleLearner(spec, recipe.make_state_dict(spec, 11), recipe.make_state_dict(spec, 12),
                               Hyper(double_q=dq, n_step=n), batch_size=B, target_update_freq=2)
    steps = 2 if B <= 32 else 1
    out = []
    for s in range(steps):
        frames, a, r, d, w = batch_inputs(spec, B, 61 + s, 62 + s)
        rand = [g[f"s{s}::rand_{i}"] for i in range(3)] if spec.algo == "iqn" else None
        no = nt = None
        if spec.noisy:
            nd = len(nets.dense_prefixes(spec)) * 3
            draws = [g[f"s{s}::normal_{i}"] for i in range(2 * nd)]
            no, nt = draws[:nd], draws[nd:]
        res = L.train(frames.reshape(B, -1), a, r, d.astype(np.float32), w, np.arange(B), rand=rand, noise_online=no, noise_target=nt)
        out.append((res, {k: v.detach().clone() for k, v in L.po.items()}, {k: v.detach().clone() for k, v in L.pt.items()},
                    dict(L.last_grads), L.update_steps))
    return spec, out


@pytest.mark.parametrize("case", G6)
def test_g6_full_train_step(case):
    g = golden(f"g6_{case}")
    spec, out = run_oracle_g6(case, g)
    for s, (res, po, pt, grads, upd) in enumerate(out):
        assert upd == int(g[f"s{s}::update_steps"])
        assert_close(res["q_loss"], g[f"s{s}::q_loss"], 5e-5, 5e-6, f"s{s} q_loss")
        if spec.algo == "fqf":
            assert_close(res["fraction_loss"], g[f"s{s}::fraction_loss"], 5e-5, 5e-6, f"s{s} fraction_loss")
        for k in g.files:
            if not k.startswith(f"s{s}::"):
                continue
            parts = k.split("::")
            if parts[1] == "grad":
                # checksum = (sum, l2, first 8): sums of ~1e6 fp32 grads cancel, so tolerance is absolute on l2 scale
                want = g[k]
                got = recipe.checksum(grads[parts[2]].numpy())
                scale = max(want[1], 1e-6)
                assert abs(got[1] - want[1]) <= 2e-4 * scale, (k, got[1], want[1])
                assert np.all(np.abs(got[2:] - want[2:]) <= 2e-4 * scale / np.sqrt(max(grads[parts[2]].numel(), 1)) + 1e-4 * np.abs(want[2:]) + 1e-7), (k, got, want)
            elif parts[1] in ("param", "target"):
                src = po if parts[1] == "param" else pt
                want = g[k]
                got = recipe.checksum(src[parts[2]].numpy())
                assert abs(got[1] - want[1]) <= 1e-5 * max(want[1], 1e-6), (k, got[1], want[1])
                # Adam's first step moves every weight by ~lr regardless of gradient scale: compare heads tightly
                assert np.all(np.abs(got[2:] - want[2:]) <= 2e-5 + 1e-4 * np.abs(want[2:])), (k, got[2:], want[2:])


# ----------------------------------------------------------------------------- G7 actor
def _script(E, T, seed, with_life):
    gg = recipe.gen(seed)
    obs = gg.integers(0, 256, size=(T + 1, E, 4, 84, 84), dtype=np.uint8)
    rew = gg.choice(np.array([-1.0, 0.0, 0.0, 1.0]), size=(T, E)).astype(np.float64)
    term = gg.random((T, E)) < 0.15
    trunc = gg.random((T, E)) < 0.08
    life = (gg.random((T, E)) < 0.15) if with_life else None
    fmask = term | trunc
    fret = gg.integers(0, 50, size=(T, E)).astype(np.float32)
    return obs, rew, term, trunc, life, fmask, fret


class FakeEnv:
    def __init__(self, E, script):
        self.E = E
        self.obs, self.rew, self.term, self.trunc, self.life, self.fmask, self.fret = script
        self.t = 0
        self.actions = []

    def reset(self):
        self.t = 0
        return self.obs[0].copy(), {}

    def step(self, a):
        t = self.t
        self.actions.append(np.asarray(a).copy())
        info = {}
        if self.life is not None:
            info["life_loss"] = self.life[t].copy()
        if self.fmask[t].any():
            fi = np.empty(self.E, dtype=object)
            for i in range(self.E):
                fi[i] = {"episode": {"r": np.array([self.fret[t, i]], dtype=np.float32)}} if self.fmask[t, i] else None
            info["final_info"] = fi
            info["_final_info"] = self.fmask[t].copy()
        self.t += 1
        return self.obs[t + 1].copy(), self.rew[t].copy(), self.term[t].copy(), self.trunc[t].copy(), info


@pytest.mark.parametrize("n_step,with_life", [(1, True), (3, True), (3, False)])
def test_g7_actor_sample(n_step, with_life):
    g = golden(f"g7_actor_n{n_step}_life{int(with_life)}")
    spec = SPECS["dqn"]
    E, T = 4, 12
    env = FakeEnv(E, _script(E, T, 70 + n_step, with_life))
    it = iter(zip(g["draws_int"], g["draws_u"]))
    act = oactor.OracleActor(env, params_for(spec, 11), spec, n_step=n_step, sample_steps=6, draw=lambda E_: next(it))
    a, r, d, h, rs, qs = [], [], [], [], [], []
    blobs = []
    for call in range(2):
        data, rs_c, qs_c = act.sample(float(g["eps"]))
        for (frames, at, rt, dt) in data:
            blob = frames.tobytes()
            a.append(int(at)); r.append(float(rt)); d.append(bool(dt))
            h.append(np.frombuffer(hashlib.sha256(blob).digest()[:8], dtype=np.uint64)[0])
            blobs.append(frames)
        rs.append(np.array(rs_c, dtype=np.float64)); qs.append(np.array(qs_c))
    assert np.array_equal(np.stack(env.actions), g["actions_env"])
    assert np.array_equal(np.array(a), g["a"])
    assert np.array_equal(np.array(r), g["r"])  # float64 n-step sums: exact
    assert np.array_equal(np.array(d), g["d"])
    assert np.array_equal(np.array(h, dtype=np.uint64), g["blob_hash"])
    assert np.array_equal(blobs[0].ravel(), g["first_blob"]) and np.array_equal(blobs[1].ravel(), g["second_blob"])
    assert np.array_equal(rs[0], g["rs0"]) and np.array_equal(rs[1], g["rs1"])
    assert_close(qs[0], g["qs0"], RT, AT, "qmax call 0")
    assert_close(qs[1], g["qs1"], RT, AT, "qmax call 1")


# ----------------------------------------------------------------------------- G8 replay trace
@pytest.mark.parametrize("policy", ["uniform", "prioritize"])
def test_g8_replay_trace(policy):
    g = golden(f"g8_replay_{policy}")
    rp = oreplay.ReferenceReplay(24, policy == "prioritize", total_steps=1000)
    uid = 0
    snap = 0
    gg = recipe.gen(81)
    for op, n in g["op_log"]:
        if op == 0:
            trans = []
            for _ in range(n):
                trans.append((uid, uid % 4, float(uid % 3 - 1), bool(uid % 5 == 0)))
                uid += 1
            rp.extend(trans)
            tag = "extend"
        elif op == 1:
            ids = g[f"{snap:02d}::update_in::ids"]
            rp.update_priority(ids, g[f"{snap:02d}::update_in::losses"])
            tag = "update"
        else:
            idxs = g[f"{snap:02d}::get::idx_in"]
            got = [rp[int(i)] for i in idxs]
            assert np.array_equal(np.array([x[5] for x in got]), g[f"{snap:02d}::get::idx_out"])
            assert np.array_equal(np.array([x[0] for x in got]), g[f"{snap:02d}::get::payload_id"])
            assert np.array_equal(np.array([x[4] for x in got], dtype=np.float32), g[f"{snap:02d}::get::prio"])
            if policy == "prioritize":
                w = oreplay.is_weights(np.array([x[4] for x in got]), float(torch.from_numpy(rp.priority).sum().item()), rp.top, rp.beta)
                assert_close(w, g[f"{snap:02d}::get::is_weights"], 1e-6, 1e-7, "IS weights")
            tag = "get"
        assert np.array_equal(rp.priority, g[f"{snap:02d}::{tag}::priority"]), f"priority after op {snap} ({tag})"
        assert rp.top == int(g[f"{snap:02d}::{tag}::top"]) and len(rp) == int(g[f"{snap:02d}::{tag}::len"])
        ids_now = [rp.slots[rp.slot_of(i)][0] for i in range(rp.count)]
        assert np.array_equal(np.array(ids_now), g[f"{snap:02d}::{tag}::ids"])
        if policy == "prioritize":
            assert rp.beta == float(g[f"{snap:02d}::{tag}::beta"])
            assert abs(rp.max_p - float(g[f"{snap:02d}::{tag}::max_p"])) <= 1e-7 * abs(rp.max_p)
        snap += 1


def test_g9_schedules():
    g = golden("g9_schedules")
    s = schedules.LinearSchedule(0.4, 1.0, 1e7)
    assert np.array_equal(np.array([s(1280) for _ in range(6)]), g["lin_a"])
    s = schedules.LinearSchedule(0.4, 1.0, 1000)
    assert np.array_equal(np.array([s(300) for _ in range(6)]), g["lin_b"])
    s = schedules.LinearSchedule(1.0, 0.1, 10)
    assert np.array_equal(np.array([s() for _ in range(14)]), g["lin_c"])
    s = schedules.LinearSchedule(0.7)
    assert np.array_equal(np.array([s(5) for _ in range(3)]), g["lin_d"])
    assert np.array_equal(np.array([schedules.epsilon(int(t)) for t in g["eps_steps"]]), g["eps"])
    assert schedules.epsilon(0) == 1.01  # quirk Q15


def test_g10_noisy_linear():
    g = golden("g10_noisy")
    p = {f"l.{k}": torch.from_numpy(g[k]) for k in ("weight_mu", "weight_sigma", "bias_mu", "bias_sigma", "noise_in", "noise_out_weight", "noise_out_bias")}
    nets.compose_noise(p, "l")
    assert_close(p["l.weight_epsilon"], g["weight_epsilon"], 1e-6, 1e-8, "weight_epsilon")
    assert_close(p["l.bias_epsilon"], g["bias_epsilon"], 1e-6, 1e-8, "bias_epsilon")
    assert_close(nets.dense(p, "l", torch.from_numpy(g["x"]), True), g["y"], 1e-5, 1e-6, "noisy linear out")


# ----------------------------------------------------------------------------- G6P: FQF train steps in the CPU-independent mode
G6P = ["fqf_b16_dq0_n1", "fqf_duel_b16_dq1_n3", "fqf_duel_a18_b16_dq1_n3"]


@pytest.mark.parametrize("case", G6P)
def test_g6p_pinned_fqf_train_steps(case):
    """Three consecutive reference ``FQFLearner.train()`` calls recorded in the CPU-independent mode of tests/golden/pinned.py (scalar ATen kernels, MKL's
    reproducibility mode, one thread, no oneDNN) against the oracle run in the same mode in a child process: in that mode the reference's numbers do not depend on
    the host CPU — the ordinary G6 fixtures' FQF numbers do from the second step on — and the oracle reproduces them to the last bit."""
    from util import pinned_oracle
    g = golden(f"g6p_{case}")
    o = pinned_oracle(case)
    assert str(o["cpu_capability"]).upper().startswith("DEFAULT") and "ATEN_CPU_CAPABILITY=default" in str(g["pinned_env"])
    for s in range(3):
        assert int(o[f"s{s}::update_steps"]) == int(g[f"s{s}::update_steps"])
        assert_close(o[f"s{s}::q_loss"], g[f"s{s}::q_loss"], 1e-6, 1e-7, f"s{s} q_loss")
        assert_close(o[f"s{s}::fraction_loss"], g[f"s{s}::fraction_loss"], 1e-6, 1e-7, f"s{s} fraction_loss")
        n = 0
        for k in g.files:
            parts = k.split("::")
            if parts[0] == f"s{s}" and len(parts) == 3 and parts[1] in ("grad", "param", "target") and k in o:
                assert_close(o[k], g[k], 1e-6, 1e-7 * max(float(np.abs(g[k]).max()), 1.0), k)
                n += 1
        assert n > 30
