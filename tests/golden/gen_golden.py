#!/usr/bin/env python3
"""Generate golden fixtures by importing the reference (build container only).

Usage (from the repo root, with /root/reference present):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Writes ``tests/golden/*.npz``.  Inputs come from ``recipe.py``; the arrays
stored are the reference's outputs plus the small random draws we injected
(IQN taus, NoisyNet noise, epsilon-greedy draws).  Nothing of the reference's
source is copied: it is imported, called and its results recorded.  The GPU box
never runs this script (there is no /root/reference there).

Third-party modules the reference imports but which are absent from this image
(lz4, prefetch_generator, gymnasium, ale_py) are replaced by inert stand-ins in
``sys.modules`` *for the import only* — none of them does arithmetic on the
path (lz4 is lossless; the env is replaced by a scripted fake).
"""
from __future__ import annotations

import os
import sys
import types
from collections import OrderedDict
from contextlib import contextmanager
from unittest import mock

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import pinned
import recipe

PINNED = os.environ.get("A0_PINNED") == "1"      # python tests/golden/pinned.py gen: the CPU-independent code path (fixture group G6P)
if PINNED:
    pinned.configure()
from recipe import NetSpec, SPECS

REF = os.environ.get("A0_REFERENCE", "/root/reference")


# --------------------------------------------------------------------------- stubs
def _install_stubs():
    lz4 = types.ModuleType("lz4")
    lz4b = types.ModuleType("lz4.block")
    lz4b.compress = lambda b: bytes(np.ascontiguousarray(b).tobytes()) if not isinstance(b, (bytes, bytearray)) else bytes(b)
    lz4b.decompress = lambda b: bytes(b)
    lz4.block = lz4b
    sys.modules["lz4"] = lz4
    sys.modules["lz4.block"] = lz4b

    pg = types.ModuleType("prefetch_generator")

    class BackgroundGenerator:  # never used by the fixtures
        def __init__(self, it, max_prefetch=1):
            self.it = it

        def __iter__(self):
            return iter(self.it)

    pg.BackgroundGenerator = BackgroundGenerator
    sys.modules["prefetch_generator"] = pg

    gym = types.ModuleType("gymnasium")

    class _W:
        def __init__(self, env=None, *a, **k):
            self.env = env

    class Wrapper:
        """gymnasium.Wrapper as far as the reference's wrappers use it (gymnasium 0.28.1 core.py): holds ``env``, forwards step / reset / unwrapped and unknown
        attributes.  FUNCTIONAL since fixture group G12: the reference's FireResetEnv / EpisodicLifeEnv / ClipRewardEnv run on it."""

        def __init__(self, env):
            self.env = env

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self.env, name)

        def step(self, action):
            return self.env.step(action)

        def reset(self, **kwargs):
            return self.env.reset(**kwargs)

    class RewardWrapper(Wrapper):
        """gymnasium.RewardWrapper: step() passes the reward through ``self.reward`` (gymnasium 0.28.1 core.py)."""

        def step(self, action):
            obs, reward, terminated, truncated, info = self.env.step(action)
            return obs, self.reward(reward), terminated, truncated, info

    gym.RewardWrapper = RewardWrapper
    gym.Wrapper = Wrapper
    gym.Env = object
    gym.make_vec = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no gymnasium here"))
    core = types.ModuleType("gymnasium.core")
    core.Env = object
    wr = types.ModuleType("gymnasium.wrappers")
    wr.AtariPreprocessing = _W
    wr.FrameStack = _W
    wr.RecordEpisodeStatistics = _W
    gym.core = core
    gym.wrappers = wr
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.core"] = core
    sys.modules["gymnasium.wrappers"] = wr
    sys.modules["ale_py"] = types.ModuleType("ale_py")


def import_reference():
    _install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import agent0.deepq.config as rcfg
    import agent0.deepq.model as rmodel
    import agent0.deepq.agent as ragent
    import agent0.deepq.replay as rreplay
    import agent0.common.utils as rutils
    import agent0.common.atari_wrappers as rwrap

    globals()["rwrap"] = rwrap
    assert os.path.realpath(rcfg.__file__).startswith(os.path.realpath(REF)), rcfg.__file__
    return rcfg, rmodel, ragent, rreplay, rutils


rcfg, rmodel, ragent, rreplay, rutils = import_reference()


# --------------------------------------------------------------------------- helpers
def ref_cfg(spec: NetSpec, batch: int, *, double_q=False, n_step=1, prioritize=False, num_envs=4,
            replay_size=64, total_steps=10_000_000):
    cfg = rcfg.ExpConfig()
    cfg.device = rcfg.DeviceEnum.cpu
    cfg.obs_shape = tuple(spec.obs_shape)
    cfg.action_dim = spec.action_dim
    cfg.learner.algo = rcfg.AlgoEnum[spec.algo]
    cfg.learner.batch_size = batch
    cfg.learner.double_q = double_q
    cfg.learner.dueling_head = spec.dueling
    cfg.learner.noisy_net = spec.noisy
    cfg.learner.n_step_q = n_step
    cfg.actor.num_envs = num_envs
    cfg.replay.size = replay_size
    cfg.trainer.total_steps = total_steps
    if prioritize:
        cfg.replay.policy = rcfg.ReplayEnum.prioritize
    return cfg


def load_recipe_weights(model, spec: NetSpec, seed: int):
    sd_ref = model.state_dict()
    shapes = recipe.state_dict_shapes(spec)
    got = OrderedDict((k, tuple(v.shape)) for k, v in sd_ref.items())
    assert list(got.items()) == list(shapes.items()), (
        f"state_dict inventory mismatch for {spec}:\n ref={list(got.items())}\n mine={list(shapes.items())}"
    )
    sd = recipe.make_state_dict(spec, seed)
    new = OrderedDict()
    for k, v in sd.items():
        if k in ("head.atoms", "head.cumulative_density"):
            new[k] = sd_ref[k].clone()  # keep the reference's own buffer values
        else:
            new[k] = torch.from_numpy(v.copy())
    model.load_state_dict(new)
    return model


class Injector:
    """Replaces torch.rand / Tensor.normal_ by a recorded PCG64 stream."""

    def __init__(self, seed: int):
        self.g = recipe.gen(seed)
        self.rand_log = []
        self.normal_log = []

    def rand(self, *size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list)):
            size = tuple(size[0])
        x = self.g.random(size, dtype=np.float32)
        self.rand_log.append(x.copy())
        return torch.from_numpy(x)

    def normal_(self, tensor, mean=0.0, std=1.0):
        x = (self.g.standard_normal(tuple(tensor.shape)).astype(np.float32) * np.float32(std)) + np.float32(mean)
        self.normal_log.append(x.copy())
        with torch.no_grad():
            tensor.copy_(torch.from_numpy(x))
        return tensor

    @contextmanager
    def active(self):
        inj = self

        def _normal_(self_t, mean=0.0, std=1.0, *, generator=None):
            return inj.normal_(self_t, mean, std)

        with mock.patch.object(torch, "rand", self.rand), mock.patch.object(torch.Tensor, "normal_", _normal_):
            yield self


def t2n(x):
    return x.detach().cpu().numpy().copy()


def batch_tuple(frames_u8, actions, rewards, terminals, weights, indices):
    """What Trainer.step hands to learner.train (trainer.py:88-97): all .float()."""
    B = frames_u8.shape[0]
    return (
        torch.from_numpy(frames_u8.reshape(B, -1).copy()).float(),
        torch.from_numpy(actions.copy()).float(),
        torch.from_numpy(rewards.copy()).float(),
        torch.from_numpy(terminals.copy()).float(),
        torch.from_numpy(weights.copy()).float(),
        torch.from_numpy(indices.copy()).float(),
    )


OUT = {}


ONLY = [n for n in os.environ.get("A0_GOLDEN_ONLY", "").split(",") if n]      # spec names: (re)generate just their G1 / G3 / G6 fixtures


def wanted(name: str) -> bool:
    return not ONLY or name in ONLY


def save(name, **arrays):
    path = os.path.join(HERE, f"{name}.npz")
    meta = dict(torch_version=np.array(torch.__version__), numpy_version=np.array(np.__version__))
    if PINNED:
        meta["pinned_env"] = np.array(" ".join(f"{k}={v}" for k, v in sorted(pinned.ENV.items())) + " torch.set_num_threads(1) torch.backends.mkldnn.enabled=False")
    np.savez_compressed(path, **meta, **arrays)
    OUT[name] = sum(np.asarray(a).nbytes for a in arrays.values())
    print(f"  wrote {name}.npz  ({os.path.getsize(path)/1024:.1f} KiB, {len(arrays)} arrays)")


# --------------------------------------------------------------------------- G1 / G2
def g1_forward():
    print("G1 forward/qval per algo")
    for name, spec in SPECS.items():
        if not wanted(name):
            continue
        B = 8
        cfg = ref_cfg(spec, B)
        inj = Injector(1000 + len(name))
        with inj.active():
            model = rmodel.DeepQNet(cfg)
            load_recipe_weights(model, spec, seed=11)
            if spec.noisy:
                model.reset_noise()
            frames = recipe.make_frames(B, seed=21, obs_shape=spec.obs_shape)
            x = torch.from_numpy(frames[:, : spec.obs_shape[0]].copy()).float().div(255.0)
            arrays = {}
            n_noise = len(inj.normal_log)
            with torch.no_grad():
                if spec.algo in ("iqn", "fqf"):
                    feats = model.encoder(x)
                    arrays["features"] = t2n(feats)
                    if spec.algo == "iqn":
                        q, taus = model.head(feats, n=16)
                        arrays["taus_16"] = t2n(taus)
                        arrays["q_16"] = t2n(q)
                        r0 = len(inj.rand_log)
                        arrays["qval"] = t2n(model.qval(x))  # draws K=32 taus
                        arrays["taus_qval"] = inj.rand_log[r0]
                    else:
                        taus, taus_hat, ent = model.head.prop_taus(feats)
                        arrays["taus"] = t2n(taus)
                        arrays["taus_hat"] = t2n(taus_hat)
                        arrays["entropies"] = t2n(ent)
                        q, _ = model.head(feats, taus=taus_hat)
                        arrays["q_hat"] = t2n(q)
                        arrays["qval"] = t2n(model.qval(x))
                else:
                    arrays["out"] = t2n(model(x))
                    arrays["qval"] = t2n(model.qval(x))
            if spec.noisy:
                sd = model.state_dict()
                for k, v in sd.items():
                    if k.endswith(("noise_in", "noise_out_weight", "noise_out_bias")):
                        arrays["buf::" + k] = t2n(v)
                    elif k.endswith(("weight_epsilon", "bias_epsilon")):
                        arrays["bufsum::" + k] = recipe.checksum(t2n(v))
            assert n_noise == len(inj.normal_log) or not spec.noisy
            save(f"g1_{name}", **arrays)


def g2_layers():
    print("G2 per-layer activations")
    spec = SPECS["dqn"]
    cfg = ref_cfg(spec, 2)
    model = rmodel.DeepQNet(cfg)
    load_recipe_weights(model, spec, seed=11)
    frames = recipe.make_frames(2, seed=22)
    x = torch.from_numpy(frames[:, :4].copy()).float().div(255.0)
    acts = {}
    h = x
    with torch.no_grad():
        for i, layer in enumerate(model.encoder.convs):
            h = layer(h)
            acts[f"convs_{i}"] = t2n(h)
        hd = torch.relu(model.head.first_dense(h))
        acts["fc1_relu"] = t2n(hd)
        acts["q"] = t2n(model.head.q_head(hd))
    save("g2_layers", **acts)


# --------------------------------------------------------------------------- G3 train_step losses
def run_train_step(name, spec, B, *, double_q, n_step, seed):
    cfg = ref_cfg(spec, B, double_q=double_q, n_step=n_step)
    inj = Injector(seed)
    with inj.active():
        Learner = getattr(ragent, f"{spec.algo.upper()}Learner")
        learner = Learner(cfg)
        load_recipe_weights(learner.model, spec, seed=11)
        load_recipe_weights(learner.model_target, spec, seed=12)
        if spec.noisy:
            learner.model.reset_noise()
            learner.model_target.reset_noise()
        frames = recipe.make_frames(B, seed=31, obs_shape=spec.obs_shape)
        actions, rewards, terminals, weights = recipe.make_transitions(B, spec.action_dim, seed=32)
        ft = torch.from_numpy(frames.copy()).float().div(255.0)
        obs, next_obs = torch.split(ft, spec.obs_shape[0], 1)
        r0 = len(inj.rand_log)
        loss = learner.train_step(
            obs,
            torch.from_numpy(actions.copy()),
            torch.from_numpy(rewards.copy()),
            torch.from_numpy(terminals.copy()).float(),
            next_obs,
        )
        arrays = {}
        if spec.algo == "fqf":
            q_loss, f_loss = loss
            arrays["loss"] = t2n(q_loss)
            arrays["fraction_loss"] = t2n(f_loss)
        else:
            arrays["loss"] = t2n(loss)
        for i, r in enumerate(inj.rand_log[r0:]):
            arrays[f"rand_{i}"] = r
        if spec.noisy:
            for tag, m in (("online", learner.model), ("target", learner.model_target)):
                for k, v in m.state_dict().items():
                    if k.endswith(("noise_in", "noise_out_weight", "noise_out_bias")):
                        arrays[f"noise::{tag}::{k}"] = t2n(v)
    return arrays


def g3_losses():
    print("G3 train_step per-sample losses")
    cases = [
        ("dqn", 8, False, 1), ("dqn", 8, True, 3), ("dqn_duel", 8, True, 1), ("mdqn", 8, False, 1),
        ("c51", 8, False, 1), ("c51", 8, True, 3), ("c51_duel_noisy", 8, True, 3),
        ("qr", 8, False, 1), ("qr_duel", 8, True, 3),
        ("iqn", 8, False, 1), ("iqn_duel", 8, True, 3),
        ("fqf", 8, False, 1), ("fqf", 8, True, 3),
        ("dqn_tiny", 32, True, 1), ("c51_tiny", 32, True, 3),
        ("dqn", 512, False, 1), ("c51", 512, True, 3),
        ("fqf_duel", 8, True, 3), ("dqn_duel_a18", 8, True, 1), ("fqf_duel_a18", 8, True, 3),
    ]
    for name, B, dq, ns in cases:
        if not wanted(name):
            continue
        arrays = run_train_step(name, SPECS[name], B, double_q=dq, n_step=ns, seed=3000 + B + ns)
        save(f"g3_{name}_b{B}_dq{int(dq)}_n{ns}", **arrays)


# --------------------------------------------------------------------------- G4 C51 projection known answers
def g4_c51_projection():
    print("G4 C51 projection known-answer set")
    spec = SPECS["c51"]
    # (reward, terminal) rows chosen for the edge cases of agent.py:236-264
    rows = [(0.0, 1), (1.0, 1), (1.0, 0), (-1.0, 0), (25.0, 0), (-25.0, 0), (-10.0, 1), (10.0, 1),
            (0.4, 1), (-0.4, 1), (0.0, 0), (3.0, 0), (-3.0, 1), (9.6, 1), (-9.6, 1), (0.2, 1)]
    B = len(rows)
    for n_step in (1, 3):
        cfg = ref_cfg(spec, B, double_q=False, n_step=n_step)
        learner = ragent.C51Learner(cfg)
        load_recipe_weights(learner.model, spec, seed=11)
        load_recipe_weights(learner.model_target, spec, seed=12)
        frames = recipe.make_frames(B, seed=41)
        actions = np.arange(B, dtype=np.int64) % spec.action_dim
        rewards = np.array([r for r, _ in rows], dtype=np.float32)
        terminals = np.array([d for _, d in rows], dtype=np.float32)
        ft = torch.from_numpy(frames.copy()).float().div(255.0)
        obs, next_obs = torch.split(ft, 4, 1)
        # capture target_prob by recording the second operand of the final mul
        captured = {}
        orig_mul = torch.Tensor.mul

        def spy_mul(self_t, other):
            if isinstance(other, torch.Tensor) and self_t.dim() == 2 and self_t.shape == (B, 51) and other.shape == (B, 51) and other.requires_grad:
                captured["target_prob"] = self_t.detach().clone()
            return orig_mul(self_t, other)

        with mock.patch.object(torch.Tensor, "mul", spy_mul):
            loss = learner.train_step(obs, torch.from_numpy(actions), torch.from_numpy(rewards),
                                      torch.from_numpy(terminals), next_obs)
        assert "target_prob" in captured
        with torch.no_grad():
            prob_next = learner.model_target(next_obs).softmax(dim=-1)
            a_next = prob_next.mul(learner.model.head.atoms).sum(-1).argmax(-1)
            prob_sel = prob_next[torch.arange(B), a_next]
        save(f"g4_c51_projection_n{n_step}", rewards=rewards, terminals=terminals, actions=actions,
             prob_next_sel=t2n(prob_sel), target_prob=t2n(captured["target_prob"]), loss=t2n(loss))


# --------------------------------------------------------------------------- G5 quantile huber
def g5_huber():
    print("G5 huber_qr_loss + grad")
    g = recipe.gen(51)
    arrays = {}
    for (B, N, Nd, per_sample_tau) in [(4, 200, 200, False), (8, 64, 64, True), (8, 32, 32, True), (3, 8, 5, True)]:
        q = (g.standard_normal((B, N)) * 1.5).astype(np.float32)
        tgt = (g.standard_normal((B, Nd)) * 1.5).astype(np.float32)
        # force some exact ties and |d|==1 boundary cases
        tgt[0, 0] = q[0, 0]
        tgt[1, 1] = q[1, 1] + 1.0
        tgt[2, 2] = q[2, 2] - 1.0
        if per_sample_tau:
            taus = g.random((B, N)).astype(np.float32)
        else:
            taus = np.broadcast_to(((2 * np.arange(N) + 1) / (2.0 * N)).astype(np.float32), (1, N)).copy()
        w = g.uniform(0.2, 1.0, B).astype(np.float32)
        qt = torch.from_numpy(q.copy()).requires_grad_(True)
        loss = ragent.BaseLearner.huber_qr_loss(qt.view(B, 1, N), torch.from_numpy(tgt.copy()).view(B, Nd, 1),
                                                torch.from_numpy(taus.copy()).view(-1, 1, N))
        loss.mul(torch.from_numpy(w)).sum().backward()
        tag = f"{B}x{N}x{Nd}"
        arrays.update({f"q_{tag}": q, f"t_{tag}": tgt, f"tau_{tag}": taus, f"w_{tag}": w,
                       f"loss_{tag}": t2n(loss), f"dq_{tag}": t2n(qt.grad)})
    save("g5_huber_qr", **arrays)


# --------------------------------------------------------------------------- G6 full train() step
def g6_train(pinned_fqf: bool = False):
    """``pinned_fqf`` (group G6P): the FQF cases only, three consecutive steps, in the CPU-independent mode of tests/golden/pinned.py, as g6p_*.npz."""
    print("G6 full learner.train(): post-step parameter fingerprints" + (" — pinned FQF cases" if pinned_fqf else ""))
    if pinned_fqf and not PINNED:
        raise SystemExit("g6p must run in pinned mode: python tests/golden/pinned.py gen")
    cases = [("dqn", 16, False, 1), ("dqn_duel", 16, True, 3), ("c51_duel_noisy", 16, True, 3), ("c51", 16, False, 1),
             ("qr", 16, False, 1), ("iqn", 16, False, 1), ("fqf", 16, False, 1), ("mdqn", 16, False, 1),
             ("dqn_tiny", 32, True, 1), ("c51_tiny", 32, True, 3), ("dqn", 512, False, 1),
             ("fqf_duel", 16, True, 3), ("dqn_duel_a18", 16, True, 1), ("fqf_duel_a18", 16, True, 3)]
    for name, B, dq, ns in cases:
        if not wanted(name) or (pinned_fqf and SPECS[name].algo != "fqf"):
            continue
        spec = SPECS[name]
        cfg = ref_cfg(spec, B, double_q=dq, n_step=ns)
        cfg.learner.target_update_freq = 2
        inj = Injector(6000 + B)
        arrays = {}
        with inj.active():
            Learner = getattr(ragent, f"{spec.algo.upper()}Learner")
            learner = Learner(cfg)
            load_recipe_weights(learner.model, spec, seed=11)
            load_recipe_weights(learner.model_target, spec, seed=12)
            n_steps = 3 if pinned_fqf else (2 if B <= 32 else 1)
            for step in range(n_steps):
                frames = recipe.make_frames(B, seed=61 + step, obs_shape=spec.obs_shape)
                actions, rewards, terminals, weights = recipe.make_transitions(B, spec.action_dim, seed=62 + step)
                indices = np.arange(B, dtype=np.int64)
                r0, z0 = len(inj.rand_log), len(inj.normal_log)
                grads_before = None
                res = learner.train(batch_tuple(frames, actions, rewards, terminals, weights, indices))
                arrays[f"s{step}::q_loss"] = t2n(res["q_loss"])
                if res["fraction_loss"] is not None:
                    arrays[f"s{step}::fraction_loss"] = t2n(res["fraction_loss"])
                for i, r in enumerate(inj.rand_log[r0:]):
                    arrays[f"s{step}::rand_{i}"] = r
                for i, r in enumerate(inj.normal_log[z0:]):
                    arrays[f"s{step}::normal_{i}"] = r
                for k, p in learner.model.named_parameters():
                    arrays[f"s{step}::param::{k}"] = recipe.checksum(t2n(p))
                    if p.grad is not None:
                        arrays[f"s{step}::grad::{k}"] = recipe.checksum(t2n(p.grad))
                for k, p in learner.model_target.named_parameters():
                    arrays[f"s{step}::target::{k}"] = recipe.checksum(t2n(p))
                arrays[f"s{step}::update_steps"] = np.array(learner.update_steps)
        save(f"g6{'p' if pinned_fqf else ''}_{name}_b{B}_dq{int(dq)}_n{ns}", **arrays)


# --------------------------------------------------------------------------- G7 actor
class FakeVecEnv:
    """Scripted vector env honouring the gymnasium 0.28 tuple/info contract the
    Actor consumes (agent.py:55-62,85-88).  Pure data: the script is stored."""

    def __init__(self, num_envs, obs_script, rew, term, trunc, life, final_mask, final_ret):
        self.n = num_envs
        self.obs_script = obs_script  # [T+1, E, 4, 84, 84] u8
        self.rew, self.term, self.trunc, self.life = rew, term, trunc, life
        self.final_mask, self.final_ret = final_mask, final_ret
        self.t = 0
        self.actions = []

    def reset(self):
        self.t = 0
        return self.obs_script[0].copy(), {}

    def step(self, action):
        t = self.t
        self.actions.append(np.asarray(action).copy())
        info = {}
        if self.life is not None:
            info["life_loss"] = self.life[t].copy()
        if self.final_mask[t].any():
            fi = np.empty(self.n, dtype=object)
            for i in range(self.n):
                fi[i] = {"episode": {"r": np.array([self.final_ret[t, i]], dtype=np.float32)}} if self.final_mask[t, i] else None
            info["final_info"] = fi
            info["_final_info"] = self.final_mask[t].copy()
        self.t += 1
        return (self.obs_script[t + 1].copy(), self.rew[t].copy(), self.term[t].copy(), self.trunc[t].copy(), info)

    def close(self):
        pass


def make_script(E, T, seed, with_life=True, obs_shape=(4, 84, 84)):
    g = recipe.gen(seed)
    obs = g.integers(0, 256, size=(T + 1, E) + tuple(obs_shape), dtype=np.uint8)
    rew = g.choice(np.array([-1.0, 0.0, 0.0, 1.0]), size=(T, E)).astype(np.float64)
    term = g.random((T, E)) < 0.15
    trunc = g.random((T, E)) < 0.08
    life = (g.random((T, E)) < 0.15) if with_life else None
    final_mask = term | trunc
    final_ret = g.integers(0, 50, size=(T, E)).astype(np.float32)
    return obs, rew, term, trunc, life, final_mask, final_ret


def g7_actor():
    print("G7 Actor.sample under a scripted vector env")
    import hashlib

    spec = SPECS["dqn"]
    for n_step, with_life in ((1, True), (3, True), (3, False)):
        E, T = 4, 12
        cfg = ref_cfg(spec, 8, n_step=n_step, num_envs=E)
        cfg.actor.sample_steps = 6  # two sample() calls cover T=12 and the persisting tracker (Q9)
        script = make_script(E, T, seed=70 + n_step, with_life=with_life)
        env = FakeVecEnv(E, *script)
        with mock.patch.object(ragent, "make_atari", lambda env_id, n: env):
            model = rmodel.DeepQNet(cfg)
            load_recipe_weights(model, spec, seed=11)
            actor = ragent.Actor(cfg, model)
        np.random.seed(1234)
        # record the numpy global stream the actor will consume: randint(E) then rand(E) per step
        st = np.random.get_state()
        draws_int, draws_u = [], []
        for _ in range(T):
            draws_int.append(np.random.randint(0, cfg.action_dim, E))
            draws_u.append(np.random.rand(E))
        np.random.set_state(st)
        out_a, out_r, out_d, out_hash, out_rs, out_qs = [], [], [], [], [], []
        first_frames = []
        for call in range(2):
            data, rs, qs = actor.sample(0.5)
            for (blob, at, rt, dt) in data:
                out_a.append(int(at)); out_r.append(float(rt)); out_d.append(bool(dt))
                out_hash.append(np.frombuffer(hashlib.sha256(blob).digest()[:8], dtype=np.uint64)[0])
                if len(first_frames) < 2:
                    first_frames.append(np.frombuffer(blob, dtype=np.uint8).copy())
            out_rs.append(np.array(rs, dtype=np.float64)); out_qs.append(np.array(qs, dtype=np.float64))
        save(f"g7_actor_n{n_step}_life{int(with_life)}",
             actions_env=np.stack(env.actions), draws_int=np.stack(draws_int), draws_u=np.stack(draws_u),
             a=np.array(out_a, dtype=np.int64), r=np.array(out_r, dtype=np.float64), d=np.array(out_d),
             blob_hash=np.array(out_hash, dtype=np.uint64), rs0=out_rs[0], rs1=out_rs[1], qs0=out_qs[0], qs1=out_qs[1],
             first_blob=first_frames[0], second_blob=first_frames[1], eps=np.array(0.5))


# --------------------------------------------------------------------------- G8 replay trace
def g8_replay():
    print("G8 ReplayDataset op-sequence trace + IS weights")
    spec = SPECS["dqn"]
    for policy in ("uniform", "prioritize"):
        cfg = ref_cfg(spec, 4, prioritize=(policy == "prioritize"), replay_size=24, total_steps=1000)
        rp = rreplay.ReplayDataset(cfg)
        g = recipe.gen(81)
        arrays = {}
        uid = 0
        snap = 0

        def snapshot(tag):
            nonlocal snap
            arrays[f"{snap:02d}::{tag}::priority"] = t2n(rp.priority)
            arrays[f"{snap:02d}::{tag}::top"] = np.array(rp.top)
            arrays[f"{snap:02d}::{tag}::len"] = np.array(len(rp))
            arrays[f"{snap:02d}::{tag}::ids"] = np.array([int(np.frombuffer(x[0][:8], dtype=np.int64)[0]) for x in rp.data], dtype=np.int64)
            if policy == "prioritize":
                arrays[f"{snap:02d}::{tag}::beta"] = np.array(rp.beta, dtype=np.float64)
                arrays[f"{snap:02d}::{tag}::max_p"] = np.array(rp.max_p, dtype=np.float64)
            snap += 1

        ops = [("extend", 10), ("update", 4), ("extend", 10), ("get", 5), ("update", 4), ("extend", 10), ("update", 4), ("get", 5), ("extend", 3)]
        op_log = []
        for op, n in ops:
            if op == "extend":
                trans = []
                for _ in range(n):
                    blob = np.full(16, uid, dtype=np.int64).tobytes()  # stand-in payload: lz4 stub is identity
                    trans.append((blob, uid % 4, float(uid % 3 - 1), bool(uid % 5 == 0)))
                    uid += 1
                rp.extend(trans)
                op_log.append((0, n))
            elif op == "update" and policy == "prioritize":
                ids = g.integers(0, rp.top, size=n)
                losses = g.uniform(0.0, 3.0, size=n).astype(np.float32)
                rp.update_priority(torch.from_numpy(ids), torch.from_numpy(losses))
                arrays[f"{snap:02d}::update_in::ids"] = ids
                arrays[f"{snap:02d}::update_in::losses"] = losses
                op_log.append((1, n))
            elif op == "get":
                idxs = g.integers(0, 100, size=n)
                got = [rp[int(i)] for i in idxs]
                arrays[f"{snap:02d}::get::idx_in"] = idxs
                arrays[f"{snap:02d}::get::idx_out"] = np.array([x[5] for x in got], dtype=np.int64)
                arrays[f"{snap:02d}::get::payload_id"] = np.array([int(x[0].view(np.int64)[0]) for x in got], dtype=np.int64)
                arrays[f"{snap:02d}::get::prio"] = np.array([float(x[4]) for x in got], dtype=np.float32)
                if policy == "prioritize":
                    # IS weights exactly as trainer.py:91-94 on the captured tensors
                    priorities = torch.tensor([float(x[4]) for x in got]).float()
                    probs = priorities / rp.priority.sum().item()
                    weights = (rp.top * probs).pow(-rp.beta)
                    weights = weights / weights.max().add(1e-8)
                    arrays[f"{snap:02d}::get::is_weights"] = t2n(weights)
                op_log.append((2, n))
            else:
                continue
            snapshot(op)
        arrays["op_log"] = np.array(op_log, dtype=np.int64)
        save(f"g8_replay_{policy}", **arrays)


# --------------------------------------------------------------------------- G9 schedules
def g9_schedules():
    print("G9 schedules")
    s = rutils.LinearSchedule(0.4, 1.0, 1e7)
    vals = [s(1280) for _ in range(6)]
    s2 = rutils.LinearSchedule(0.4, 1.0, 1000)
    vals2 = [s2(300) for _ in range(6)]
    s3 = rutils.LinearSchedule(1.0, 0.1, 10)
    vals3 = [s3() for _ in range(14)]
    s4 = rutils.LinearSchedule(0.7)
    vals4 = [s4(5) for _ in range(3)]
    # epsilon_fn is a lambda inside Trainer.__init__ (trainer.py:46-50); Trainer is not
    # constructible here (wandb/tensorboard absent), so evaluate the same closed form
    # the reference code spells, with the reference's default config values.
    cfg = rcfg.ExpConfig()
    steps = np.array([0, 1, 500_000, 999_999, 1_000_000, 1_000_001, 5_000_000], dtype=np.int64)
    eps = [cfg.actor.min_eps if st > cfg.trainer.exploration_steps else (1.0 - st / cfg.trainer.exploration_steps) + cfg.actor.min_eps for st in steps]
    save("g9_schedules", lin_a=np.array(vals), lin_b=np.array(vals2), lin_c=np.array(vals3), lin_d=np.array(vals4),
         eps_steps=steps, eps=np.array(eps, dtype=np.float64))


# --------------------------------------------------------------------------- G10 noisy linear
def g10_noisy():
    print("G10 NoisyLinear")
    inj = Injector(101)
    with inj.active():
        layer = rmodel.NoisyLinear(40, 24)
        g = recipe.gen(102)
        with torch.no_grad():
            layer.weight_mu.copy_(torch.from_numpy(g.standard_normal((24, 40)).astype(np.float32) * 0.2))
            layer.weight_sigma.copy_(torch.from_numpy(g.uniform(0.01, 0.1, (24, 40)).astype(np.float32)))
            layer.bias_mu.copy_(torch.from_numpy(g.standard_normal(24).astype(np.float32) * 0.1))
            layer.bias_sigma.copy_(torch.from_numpy(g.uniform(0.01, 0.1, 24).astype(np.float32)))
        layer.reset_noise()
        x = torch.from_numpy(g.standard_normal((5, 40)).astype(np.float32))
        y = layer(x)
        mu_range = 1 / np.sqrt(40)
        save("g10_noisy", weight_mu=t2n(layer.weight_mu), weight_sigma=t2n(layer.weight_sigma), bias_mu=t2n(layer.bias_mu),
             bias_sigma=t2n(layer.bias_sigma), noise_in=t2n(layer.noise_in), noise_out_weight=t2n(layer.noise_out_weight),
             noise_out_bias=t2n(layer.noise_out_bias), weight_epsilon=t2n(layer.weight_epsilon), bias_epsilon=t2n(layer.bias_epsilon),
             x=t2n(x), y=t2n(y), std=np.array(layer.noisy_layer_std), std_init=np.array(layer.std_init))
    # init statistics (G11): fresh layer
    fresh = rmodel.NoisyLinear(3136, 512)
    save("g11_init_stats", noisy_mu_absmax=np.array(float(fresh.weight_mu.abs().max())),
         noisy_sigma_w=np.array(float(fresh.weight_sigma[0, 0])), noisy_sigma_b=np.array(float(fresh.bias_sigma[0])))


# --------------------------------------------------------------------------- G12 Atari wrapper semantics
def g12_wrappers():
    """The reference's single-env wrappers (atari_wrappers.py:11-56) in make_atari's order — EpisodicLifeEnv, FireResetEnv, [RecordEpisodeStatistics: gymnasium's, absent],
    ClipRewardEnv — over the scripted emulator of fake_ale.py, driven under the vector env's autoreset rule.  Stored per case: what every agent step returned (obs id,
    clipped and raw reward, terminated, truncated, life_loss, the info's emulator counter), the observation of every reset, and the emulator's complete action log."""
    import fake_ale
    print("G12 atari wrappers (ClipRewardEnv, FireResetEnv, EpisodicLifeEnv) on a scripted emulator")
    for name, (kw, needs_fire, n) in fake_ale.CASES.items():
        actions = fake_ale.actions_for(name, n)
        arrays = {"actions": actions}
        for tag, clip in (("clip", True), ("raw", False)):
            ale = fake_ale.ScriptedAle(fake_ale.make_script(**kw), needs_fire=needs_fire)
            env = rwrap.EpisodicLifeEnv(ale)
            if needs_fire:                       # FireResetEnv asserts a FIRE action (atari_wrappers.py:23): games without one cannot take it
                env = rwrap.FireResetEnv(env)
            if clip:
                env = rwrap.ClipRewardEnv(env)
            got = fake_ale.drive(env, actions)
            if clip:
                arrays.update({k: v for k, v in got.items()})
                arrays["emulator_log"] = np.array(ale.log, dtype=np.int64)
            else:
                arrays["raw_reward"] = got["reward"]
                assert np.array_equal(np.array(ale.log, dtype=np.int64), arrays["emulator_log"]), "clipping changes no emulator call"
        save(f"g12_wrappers_{name}", **arrays)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1 if PINNED else 8)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g12"]
    if PINNED and which != ["g6p"]:
        raise SystemExit("pinned mode generates group g6p only")
    table = dict(g1=g1_forward, g2=g2_layers, g3=g3_losses, g4=g4_c51_projection, g5=g5_huber, g6=g6_train,
                 g7=g7_actor, g8=g8_replay, g9=g9_schedules, g10=g10_noisy, g12=g12_wrappers, g6p=lambda: g6_train(pinned_fqf=True))
    for w in which:
        table[w]()
    print("done; total payload bytes:", sum(OUT.values()))


if __name__ == "__main__":
    main()
