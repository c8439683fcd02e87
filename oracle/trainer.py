"""Oracle: the composed actor -> replay -> sampler -> learner -> priority loop (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates /root/reference agent0/deepq/trainer.py:
  Trainer.run loop body   trainer.py:171-184  eps = epsilon_fn(frame_count); actors[1].sample(eps); step(...)
  Trainer.step            trainer.py:74-119   Qs/Rs bookkeeping, replay.extend, frame_count += sample_steps * num_envs, then — once
                                              len(replay) > training_start_steps — learner_steps times: batch, importance weights
                                              (91-96), learner.train, update_priority(indices, q_loss) (103-104), loss means; result dict
and the launch schedule with one actor, agent0/deepq/launch.py:30-63: a rollout is issued with the weights and epsilon of the moment it
is issued, the NEXT rollout is issued before the update block that consumes the previous one, so rollout k+1 acts with the weights after
update block k-1.

Every piece is the oracle's own (actor.OracleActor, replay.ReferenceReplay / UniformPermutationSampler / is_weights, learner.OracleLearner,
core.SumTree, schedules); this module only wires them the way the reference's Trainer does.  What has no reference counterpart is a
CONTRACT of this build, defined here and in oracle/*.c, which the device must reproduce bit for bit:
  * random streams (the reference uses numpy's and torch's host generators): Philox4x32-10 keyed by (seed, rank), one stream id per
    consumer, a running offset per stream that advances by the number of draws rounded up to a multiple of four;
  * uniform sampling: the reference's RandomSampler + DataPrefetcher semantics (a permutation of range(top) frozen when the fetcher is
    created, batches of B, the last batch never returned, utils.py:51-56) with the permutation drawn as oracle/sumtree.c's Feistel
    bijection, seeded per epoch from the sampler's stream; indices are resolved against the LIVE ring (idx %= top, replay.py:32-37);
  * prioritized sampling from the sum-tree (replay.sumtree=true; the reference only weights, it never samples by priority — quirk Q2):
    leaves addressed by ring slot, new transitions enter at max_p^alpha, stratified draws u_k = (k + xi_k) * total / B.
The class is step-able (rollout / begin_step / sample_batch / train_batch / update_priority / end_step) so that a test can walk the
product's Trainer through the same sequence and compare after every link.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import core
from . import replay as oreplay
from .actor import OracleActor
from .learner import OracleLearner
from .losses import Hyper
from .schedules import LinearSchedule, epsilon

# stream ids and seed offsets of the build's random-stream contract
STREAM_EGREEDY_U, STREAM_EGREEDY_A, STREAM_TAUS, STREAM_NOISE, STREAM_SUMTREE, STREAM_PERM = 1, 2, 3, 4, 5, 6
SAMPLER_SEED_OFFSET = 104729
LEARNER_SEED_OFFSET = 15485863


def seed64(seed: int, rank: int = 0) -> int:
    return (int(seed) & 0xFFFFFFFF) | ((int(rank) & 0xFFFF) << 32)


class Stream:
    """One consumer's cursor into its Philox streams."""

    def __init__(self, seed: int, rank: int = 0):
        self.seed = seed64(seed, rank)
        self.off = {}

    def advance(self, stream: int, n: int) -> int:
        o = self.off.get(stream, 0)
        self.off[stream] = o + ((n + 3) // 4) * 4
        return o

    def uniform(self, stream: int, n: int) -> np.ndarray:
        return core.rng_uniform(self.seed, stream, self.advance(stream, n), n)

    def u32(self, stream: int, n: int) -> np.ndarray:
        return core.rng_u32(self.seed, stream, self.advance(stream, n), n)

    def seed32(self, stream: int) -> int:
        off = self.advance(stream, 4)
        return (self.seed * 0x9E3779B1 + off * 0x85EBCA77 + stream) & 0xFFFFFFFF


class SumTreeReplay:
    """Ring storage + fp32 sum-tree over ring slots (contract: oracle/sumtree.c).  Same bookkeeping as replay.py:45-59 — new entries at
    max_p^alpha, beta = beta_schedule(n), update: (loss + eps)^alpha and max_p = max(max_p, max loss) — with the priorities living in the
    tree's leaves, so that (unlike the reference's flat vector, quirk Q1) a transition's priority stays attached to its slot."""

    def __init__(self, size: int, beta0=0.4, alpha=0.5, eps=0.01, total_steps=int(1e7)):
        self.size, self.alpha, self.eps = size, alpha, eps
        self.slots = [None] * size
        self.written = 0
        self.top = 0
        self.tree = core.SumTree(size)
        self.beta_schedule = LinearSchedule(beta0, 1.0, total_steps)
        self.beta = beta0
        self.max_p = np.float32(1.0)

    def __len__(self):
        return self.top

    def extend(self, transitions):
        n = len(transitions)
        idx = (self.written + np.arange(n)) % self.size
        for i, t in zip(idx, transitions):
            self.slots[i] = t
        self.written += n
        self.top = min(self.top + n, self.size)
        val = np.float32(float(self.max_p) ** self.alpha)
        k = min(n, self.size)
        self.tree.set(idx[-k:], np.full(k, val, np.float32))
        self.beta = self.beta_schedule(n)

    def update_priority(self, ids, losses):
        losses = np.asarray(losses, dtype=np.float32)
        x = losses + np.float32(self.eps)
        val = np.sqrt(x) if self.alpha == 0.5 else np.power(x, np.float32(self.alpha))
        self.tree.set(np.asarray(ids, dtype=np.int64), val.astype(np.float32))
        self.max_p = np.float32(max(float(self.max_p), float(losses.max())))


@dataclass
class BatchRecord:
    idx: np.ndarray            # what update_priority is called with (logical deque index, or ring slot in sum-tree mode)
    slot: np.ndarray           # ring slot of every sampled transition
    frames: np.ndarray         # u8 [B, 2C, H, W]
    act: np.ndarray
    rew: np.ndarray
    done: np.ndarray
    prio: np.ndarray
    weights: np.ndarray
    q_loss: Optional[torch.Tensor] = None
    fraction_loss: Optional[torch.Tensor] = None


class OracleTrainer:
    def __init__(self, spec, state_dict, *, num_envs: int, sample_steps: int, batch_size: int, replay_size: int, learner_steps: int,
                 training_start_steps: int, policy: str = "uniform", sumtree: bool = True, n_step: int = 1, double_q: bool = False, seed: int = 42,
                 rank: int = 0, discount: float = 0.99, lr: float = 5e-4, target_update_freq: int = 500, alpha: float = 0.5, prio_eps: float = 0.01,
                 beta0: float = 0.4, total_steps: int = int(1e7), exploration_steps: int = int(1e6), min_eps: float = 0.01, launch: bool = False,
                 reset_noise_freq: int = 4, actor_noise=None, learner_noise=None, env_task: str = "stream"):
        """NoisyNet (spec.noisy): the N(0, 0.1^2) draws are supplied by the caller, like every other random number — ``actor_noise()`` returns the
        draws of the actor's next ``reset_noise`` (agent.py:49-50: every reset_noise_freq steps), ``learner_noise()`` the (online, target) draws
        of the next ``train`` (agent.py:125-127); each a list [noise_in, noise_out_weight, noise_out_bias] per NoisyLinear in module order."""
        self.spec, self.E, self.T, self.B = spec, num_envs, sample_steps, batch_size
        self.learner_steps, self.start = learner_steps, training_start_steps
        self.prioritize, self.sumtree = policy == "prioritize", (policy == "prioritize" and sumtree)
        self.exploration_steps, self.min_eps = exploration_steps, min_eps
        self.launch = launch
        self.actor_noise, self.learner_noise = actor_noise, learner_noise
        if spec.noisy and (actor_noise is None or learner_noise is None):
            raise ValueError("a NoisyNet spec needs actor_noise and learner_noise draw sources")
        hp = Hyper(discount=discount, n_step=n_step, double_q=double_q)
        self.learner = OracleLearner(spec, state_dict, state_dict, hp, batch_size=batch_size, lr=lr, target_update_freq=target_update_freq)   # target = deepcopy(model), agent.py:100
        self.actor_rng = Stream(seed, rank)
        self.sampler_rng = Stream(seed + SAMPLER_SEED_OFFSET)
        self.learner_rng = Stream(seed + LEARNER_SEED_OFFSET)
        self.hp = hp
        A = spec.action_dim

        def draw(E_):   # agent.py:29-36: randint(0, A, E) first, then rand(E) — each from its own stream
            a = (self.actor_rng.u32(STREAM_EGREEDY_A, E_) % A).astype(np.int64)
            return a, self.actor_rng.uniform(STREAM_EGREEDY_U, E_)

        env = core.SynthVecEnv(num_envs, seed=seed, rank=rank, action_dim=A, task=env_task)
        # main schedule: the actor shares the learner's model (trainer.py:41-44); launch schedule: it owns a copy, refreshed per rollout
        def noisy_reset(p):      # NoisyLinear.reset_noise on the actor's model (agent.py:49-50, model.py:73-83)
            from . import nets
            it = iter(self.actor_noise())
            with torch.no_grad():
                for prefix in nets.dense_prefixes(spec):
                    for leaf in ("noise_in", "noise_out_weight", "noise_out_bias"):
                        p[f"{prefix}.{leaf}"] = torch.from_numpy(np.array(next(it), dtype=np.float32))
                    nets.compose_noise(p, prefix)

        def actor_taus(E_):      # IQN: K fresh taus per env for every act (model.py:238, agent.py:25-28), from the actor's tau stream
            return torch.from_numpy(self.actor_rng.uniform(STREAM_TAUS, E_ * hp.K).reshape(E_, hp.K, 1))

        self.actor = OracleActor(env, self.learner.po if not launch else self._snapshot(), spec, n_step=n_step, discount=discount, sample_steps=sample_steps, draw=draw,
                                 noisy_reset=noisy_reset if spec.noisy else None, reset_noise_freq=reset_noise_freq, taus_fn=actor_taus if spec.algo == "iqn" else None)
        if self.sumtree:
            self.replay = SumTreeReplay(replay_size, beta0, alpha, prio_eps, total_steps)
        else:
            self.replay = oreplay.ReferenceReplay(replay_size, self.prioritize, beta0, alpha, prio_eps, total_steps)
        self.fetcher = None
        self.frame_count = 0
        self.num_transitions = sample_steps * num_envs
        self.Ls, self.Rs, self.Qs, self.FLs = [], [], [], []
        self.pending = None

    # ------------------------------------------------------------------ rollout
    def _snapshot(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.learner.po.items())

    def rollout(self):
        """One ``Actor.sample`` with the epsilon (and, on the launch schedule, the weights) of this moment."""
        if self.launch:
            self.actor.p = self._snapshot()
        return self.actor.sample(epsilon(self.frame_count, self.exploration_steps, self.min_eps))

    def next_transitions(self):
        """The (transitions, returns, qmax) the next ``step`` consumes, per schedule (trainer.py:179-180 / launch.py:32-37,47-62)."""
        if not self.launch:
            return self.rollout()
        if self.pending is None:
            self.pending = self.rollout()
        out, self.pending = self.pending, self.rollout()
        return out

    # ------------------------------------------------------------------ trainer.py:74-80
    def begin_step(self, transitions, returns, qmax) -> bool:
        self.Qs.extend(qmax)
        self.Rs.extend(returns)
        self.replay.extend(transitions)
        self.frame_count += self.num_transitions
        return len(self.replay) > self.start

    # ------------------------------------------------------------------ trainer.py:83-96 (+ the sampler contracts in the module docstring)
    def sample_batch(self) -> BatchRecord:
        rp, B = self.replay, self.B
        if self.sumtree:
            xi = self.sampler_rng.uniform(STREAM_SUMTREE, B)
            idx, prio = rp.tree.sample(xi)
            slot = idx % rp.size
            items = [rp.slots[s] for s in slot]
            w = oreplay.is_weights(prio, float(rp.tree.total), rp.top, rp.beta)
        else:
            f = self.fetcher
            if f is None or f["pos"] + 1 >= f["nb"]:                 # StopIteration / first use: a new fetcher over range(top) as of now
                f = self.fetcher = {"top": rp.top, "nb": (rp.top + B - 1) // B, "pos": 0, "seed": self.sampler_rng.seed32(STREAM_PERM)}
                assert f["nb"] >= 2
            perm = core.perm_batch(f["pos"] * B, B, f["top"], f["seed"])
            f["pos"] += 1
            got = [rp[int(i)] for i in perm]
            idx = np.array([g[5] for g in got], dtype=np.int64)
            slot = np.array([rp.slot_of(int(i)) for i in idx], dtype=np.int64)
            items = [(g[0], g[1], g[2], g[3]) for g in got]
            prio = np.array([g[4] for g in got], dtype=np.float32)
            if self.prioritize:
                w = oreplay.is_weights(prio, float(torch.from_numpy(rp.priority).sum().item()), rp.top, rp.beta)
            else:
                w = prio.copy()                                        # trainer.py:96: weights = priorities (all ones)
        return BatchRecord(idx=idx, slot=slot, frames=np.stack([it[0] for it in items]), act=np.array([it[1] for it in items], dtype=np.int64),
                           rew=np.array([it[2] for it in items], dtype=np.float32), done=np.array([it[3] for it in items], dtype=np.float32),
                           prio=np.asarray(prio, dtype=np.float32), weights=np.asarray(w, dtype=np.float32))

    # ------------------------------------------------------------------ trainer.py:97-110
    def train_batch(self, rec: BatchRecord) -> BatchRecord:
        no = nt = None
        if self.spec.noisy:
            no, nt = self.learner_noise()
        rand = None
        if self.spec.algo == "iqn":      # agent.py:297-327 draws action-selection taus (K), target taus (N') and online taus (N), in this order
            rand = [self.learner_rng.uniform(STREAM_TAUS, self.B * n).reshape(self.B, n, 1) for n in (self.hp.K, self.hp.N_dash, self.hp.N)]
        res = self.learner.train(rec.frames.reshape(self.B, -1), rec.act, rec.rew, rec.done, rec.weights, rec.idx, rand=rand, noise_online=no, noise_target=nt)
        rec.q_loss, rec.fraction_loss = res["q_loss"], res["fraction_loss"]
        if rec.q_loss is not None:
            self.Ls.append(float(rec.q_loss.mean()))
        if rec.fraction_loss is not None:
            self.FLs.append(float(rec.fraction_loss.mean()))
        return rec

    def update_priority(self, rec: BatchRecord):
        if self.prioritize and rec.q_loss is not None:
            self.replay.update_priority(rec.idx, rec.q_loss.numpy())

    def end_step(self) -> dict:
        m = lambda xs, k: float(np.mean(xs[-k:])) if len(xs) > 0 else None
        return dict(frames=self.frame_count, fraction_loss=m(self.FLs, 20), loss=m(self.Ls, 20), return_train=m(self.Rs, 20),
                    return_train_max=float(np.max(self.Rs)) if len(self.Rs) > 0 else None, qmax=m(self.Qs, 100))

    # ------------------------------------------------------------------ the whole iteration
    def iteration(self) -> dict:
        records: List[BatchRecord] = []
        if self.begin_step(*self.next_transitions()):
            for _ in range(self.learner_steps):
                rec = self.train_batch(self.sample_batch())
                self.update_priority(rec)
                records.append(rec)
        out = self.end_step()
        out["records"] = records
        return out
