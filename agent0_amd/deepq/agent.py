"""``Actor`` and the six ``<Algo>Learner`` classes with the reference's names and call signatures, running on the device.

Mirrors /root/reference agent0/deepq/agent.py: ``Actor`` (16-93: ``act``, ``reset``, ``sample``, ``close``),
``BaseLearner`` (96-169: Adam(lr, eps=1e-2/B), ``train`` -> {"q_loss","fraction_loss","indices"}), and
``DQNLearner`` / ``MDQNLearner`` / ``C51Learner`` / ``QRLearner`` / ``IQNLearner`` / ``FQFLearner`` (172-388), which
``Trainer`` resolves by name (trainer.py:31-34).  Differences that follow from keeping everything in HBM:
  * ``Actor.sample`` returns a ``TransitionBlock`` (slots already written into the replay ring) instead of a Python
    list of lz4 blobs; episode returns and per-step mean max-Q come back as Python lists after ONE device->host copy;
  * ``Learner.train`` accepts the reference's 6-tuple of tensors, or — on the hot path — ``train_batch`` takes ring
    slots and never touches the host; losses are returned as device tensors (the reference's ``.cpu()`` copies and its
    ``isnan().any()`` sync are gone, quirk Q14; the NaN-skip itself is kept, on the device).
"""
from __future__ import annotations

import os

from typing import Optional

import numpy as np
import torch

from agent0_amd.common.atari_wrappers import make_atari
from agent0_amd.common.utils import DeviceRng
from .config import AlgoEnum, ExpConfig
from .dist import graph_capture_kwargs
from .engine import DeviceLearner, Workspace
from .model import DeepQNet, layout_from_cfg
from .replay import ReplayDataset, StageRing, TransitionBlock


def _ops_from(cfg, ops=None):
    if cfg.device.value != "cuda":
        raise RuntimeError("agent0_amd runs on MI355X only: set device=cuda (no CPU fallback; see oracle/ for the CPU restatement used in tests)")
    if ops is None:
        from agent0_amd.ops import HipOps
        ops = HipOps()
    return ops


class Actor:
    def __init__(self, cfg: ExpConfig, model: Optional[DeepQNet] = None, replay: Optional[ReplayDataset] = None, ops=None, rank: int = 0, envs=None):
        self.cfg = cfg
        self.ops = ops = model.ops if model is not None else _ops_from(cfg, ops)
        self.rng = DeviceRng(ops, cfg.seed, rank)
        self.envs = envs if envs is not None else make_atari(cfg.env_id, cfg.actor.num_envs, seed=cfg.seed, rank=rank, ops=ops, task=cfg.env_task)
        self.obs, _ = self.envs.reset()             # a list of per-group observations for a grouped host env (env_pool.HostEnvGroups)
        self.model = model if model is not None else DeepQNet(cfg, ops=ops)
        self.replay = replay
        self.L = self.model.L
        E = self.E = int(cfg.actor.num_envs)
        self.n = int(cfg.learner.n_step_q)
        self.steps = 0
        self.obs_bytes = self.L.C * self.L.H * self.L.W
        n_tau = self.L.F if self.L.algo == "fqf" else (cfg.learner.iqn.K if self.L.algo == "iqn" else 1)
        self.n_tau = n_tau
        self.ws = Workspace(ops, self.L, E, n_tau)
        self.taus = ops.empty(E * n_tau) if self.L.algo == "iqn" else None
        self.greedy, self.rand_a, self.action = ops.zeros(E, dtype=torch.int32), ops.zeros(E, dtype=torch.int32), ops.zeros(E, dtype=torch.int32)
        self.u, self.qmax = ops.zeros(E), ops.zeros(E)
        T = max(int(cfg.actor.sample_steps), int(cfg.actor.test_steps) if False else int(cfg.actor.sample_steps))
        self.qs = ops.zeros(T)
        # scalar heads take the fused tail (fc1 slabs -> q -> dueling -> argmax -> epsilon-greedy in one kernel, a0_actor_qhead)
        self.fused_tail = self.L.algo in ("dqn", "mdqn") and self.L.feat % 4 == 0 and self.L.A + (1 if self.L.dueling else 0) <= 24
        # distributional heads (c51, qr): head GEMM slabs -> one tail kernel (bias, dueling, expectation, argmax, epsilon-greedy)
        self.dist_tail = (not self.fused_tail) and self.L.algo in ("c51", "qr") and 4 * (self.L.A * self.L.T + self.L.T) * 4 <= 160 * 1024
        self._head_slabs = ops.empty(ops.dense_fwd_partial_slabs(E, self.L.Npad, 512) * E * self.L.Npad) if self.dist_tail else None
        # quantile heads (iqn, fqf) on the device env: head GEMM slabs -> one kernel for the tail AND the env step (a0_actor_quantile_tail_env_step)
        self.quant_tail = (self.L.algo in ("iqn", "fqf") and hasattr(self.envs, "act_step_commit") and self.obs_bytes == 4 * 84 * 84
                           and ops.dense_fwd_scratch(E * n_tau, self.L.feat, self.L.num_cosines) == 0 and os.environ.get("A0_QUANT_TAIL", "1") != "0")     # 0: tuning aid (same bytes)
        if self.quant_tail:
            self._head_slabs = ops.empty(ops.dense_fwd_partial_slabs(E * n_tau, self.L.Npad, 512) * E * n_tau * self.L.Npad)
        self.qmax_all = ops.zeros(T * E) if (self.fused_tail or self.dist_tail or self.quant_tail) else None
        self._qh_scratch = ops.empty(ops.actor_qhead_scratch(E, self.L.feat)) if self.fused_tail else None
        self.stat_mask, self.stat_ret = ops.zeros(T * E), ops.zeros(T * E)
        self.ring_act, self.ring_rew, self.ring_done = ops.zeros(self.n * E, dtype=torch.int32), ops.zeros(self.n * E), ops.zeros(self.n * E)
        # observation ring for n-step: holds the last n observations; its length divides sample_steps when possible so that the
        # slot pattern of a rollout repeats (required for hipGraph replay)
        T_ = int(cfg.actor.sample_steps)
        self.ring_len = next((r for r in range(self.n, 2 * self.n + 1) if T_ % r == 0), self.n)
        # a device env that keeps its own observation history needs no copy at all: n + 1 (or the next divisor of the rollout length) buffers
        self.env_history = self.n > 1 and hasattr(self.envs, "set_history")
        if self.env_history:
            self.envs.set_history(next((r for r in range(self.n + 1, 2 * self.n + 3) if T_ % r == 0), self.n + 1))
        self.ring_obs = ops.zeros(self.ring_len * E * self.obs_bytes, dtype=torch.uint8) if (self.n > 1 and not self.env_history) else None
        self.use_graph = True
        self._graph, self._graph_warm = None, 0
        self.ctrl = ops.zeros(8, dtype=torch.int64)
        self.eps_dev = ops.zeros(1)
        self._ctrl_host = torch.zeros(8, dtype=torch.int64).pin_memory()
        self._eps_host = torch.zeros(1).pin_memory()
        self.out_act, self.out_rew, self.out_done = ops.zeros(E, dtype=torch.int32), ops.zeros(E), ops.zeros(E)
        self.atoms = self.model.head.atoms.reshape(-1).contiguous() if self.L.algo == "c51" else None
        self._stage = None
        self.fused_commit = hasattr(self.envs, "step_commit") and (self.obs_bytes == 4 * 84 * 84)
        # scalar heads on the synthetic env: the tail and the env step share one launch (a0_actor_qhead_env_step)
        self.tail_env = (self.fused_tail or self.dist_tail) and self.fused_commit and hasattr(self.envs, "act_step_commit") and os.environ.get("A0_TAIL_ENV", "1") != "0"      # 0: tuning aid (same bytes)
        # a host env split into groups (env_pool.HostEnvGroups): per-group workspaces and n-step state; the CPU steps one group while the GPU infers the other
        self.groups = None
        if hasattr(self.envs, "pools"):
            if not (self.fused_tail or self.dist_tail) or cfg.learner.noisy_net:
                raise ValueError("a grouped host env needs a scalar or distributional head without NoisyNet (per-step noise resets and the quantile heads' tau draws are "
                                 "ordered over the whole env batch); use one group")
            self.groups = []
            for pool, off in zip(self.envs.pools, self.envs.offsets):
                k = pool.E
                g = dict(pool=pool, off=off, E=k, ws=Workspace(ops, self.L, k, 1), action=self.action[off:off + k],
                         ring_act=ops.zeros(self.n * k, dtype=torch.int32), ring_rew=ops.zeros(self.n * k), ring_done=ops.zeros(self.n * k),
                         out_act=ops.zeros(k, dtype=torch.int32), out_rew=ops.zeros(k), out_done=ops.zeros(k),
                         scratch=ops.empty(ops.actor_qhead_scratch(k, self.L.feat)) if self.fused_tail else None,
                         slabs=ops.empty(ops.dense_fwd_partial_slabs(k, self.L.Npad, 512) * k * self.L.Npad) if self.dist_tail else None,
                         ring_obs=ops.zeros(self.ring_len * k * self.obs_bytes, dtype=torch.uint8) if self.n > 1 else None)
                self.groups.append(g)

    # ------------------------------------------------------------------ agent.py:25-39
    def _qhead_args(self, epsilon, ctrl, eps_ptr, t):
        L, E, dev, rng = self.L, self.E, self.model._dev, self.rng
        (W1, b1), (W2, b2) = dev.wb("fc1"), dev.wb("head")
        return (self.ws.act3, E, L.feat, W1, b1, W2, b2, L.A, L.dueling, self._qh_scratch, rng.seed, rng.STREAM_EGREEDY_A, rng.STREAM_EGREEDY_U,
                rng.reserve(rng.STREAM_EGREEDY_A, E), rng.reserve(rng.STREAM_EGREEDY_U, E), float(epsilon), self.action, self.qmax_all[t * E:(t + 1) * E], ctrl, eps_ptr)

    def _dist_tail_args(self, epsilon, ctrl, eps_ptr, t):
        """fc1 and the head GEMM's slabs (enqueued here), then the arguments of ``ops.actor_dist_tail``."""
        L, ops, E, dev, rng = self.L, self.ops, self.E, self.model._dev, self.rng
        dev._dense(self.ws.act3, L.feat, "fc1", self.ws.h, E, True)
        Wh, bh = dev.wb("head")
        ns = ops.dense_fwd_partial(self.ws.h, 512, Wh, E, L.Npad, 512, self._head_slabs)
        return (self._head_slabs, ns, bh, L.Npad, L.A, L.T, L.dueling, 2 if L.algo == "c51" else 1, self.atoms, E, rng.seed, rng.STREAM_EGREEDY_A, rng.STREAM_EGREEDY_U,
                rng.reserve(rng.STREAM_EGREEDY_A, E), rng.reserve(rng.STREAM_EGREEDY_U, E), float(epsilon), self.action, self.qmax_all[t * E:(t + 1) * E], ctrl, eps_ptr)

    def _quant_tail_args(self, epsilon, ctrl, eps_ptr, t):
        """The quantile head up to the head GEMM's slabs (enqueued here), then the arguments of ``ops.actor_quantile_tail_env_step``."""
        L, ops, E, dev, rng = self.L, self.ops, self.E, self.model._dev, self.rng
        fused_cos = hasattr(ops, "tau_cos_features") and os.environ.get("A0_TAU_COS", "1") != "0"      # round 6: the fractions and their cosine features in one launch (0: tuning aid, same bits)
        if L.algo == "fqf":
            dev.fqf_taus(self.ws, E, with_cos=fused_cos)
            taus, aux, mode = self.ws.tau_hat, self.ws.tau_all, 3
        else:
            if fused_cos:
                rng.uniform_cos(rng.STREAM_TAUS, self.taus, self.ws.cosx, E * self.n_tau, L.num_cosines)
            else:
                rng.uniform(rng.STREAM_TAUS, self.taus, E * self.n_tau)
            taus, aux, mode = self.taus, None, 1
        ns = dev.head_slabs(self.ws, E, taus, self.n_tau, self._head_slabs, cos_ready=fused_cos, w_planes=getattr(self, "_planes_on", False))
        _, bh = dev.wb("head")
        return (self._head_slabs, ns, bh, L.Npad, L.A, self.n_tau, L.dueling, mode, aux, E, rng.seed, rng.STREAM_EGREEDY_A, rng.STREAM_EGREEDY_U,
                rng.reserve(rng.STREAM_EGREEDY_A, E), rng.reserve(rng.STREAM_EGREEDY_U, E), float(epsilon), self.action, self.qmax_all[t * E:(t + 1) * E], ctrl, eps_ptr)

    def _act_device(self, epsilon: float, qs_slot: Optional[torch.Tensor], ctrl=None, eps_ptr=None, t: int = 0, tail: bool = True):
        """``tail=False``: the encoder only — the caller runs the tail together with the env step (``_qhead_args`` / ``_dist_tail_args``)."""
        L, ops, E, dev = self.L, self.ops, self.E, self.model._dev
        dev.encode(self.ws, self.obs, None, self.obs_bytes, 0, E, keep=False)
        if not tail:
            return
        if self.fused_tail:
            ops.actor_qhead(*self._qhead_args(epsilon, ctrl, eps_ptr, t))
            return
        if self.dist_tail:
            ops.actor_dist_tail(*self._dist_tail_args(epsilon, ctrl, eps_ptr, t))
            return
        if L.algo == "fqf":
            dev.fqf_taus(self.ws, E)
            dev.head(self.ws, E, self.ws.tau_hat, L.F)
        elif L.algo == "iqn":
            self.rng.uniform(self.rng.STREAM_TAUS, self.taus, E * self.n_tau)
            dev.head(self.ws, E, self.taus, self.n_tau)
        else:
            dev.head(self.ws, E)
        dev.select(self.ws, E, self.n_tau, self.greedy, qmax=self.qmax, atoms=self.atoms)
        # the reference draws randint(0, A, E) and then rand(E) (agent.py:29-36): both come from their own Philox stream, generated
        # inside the selection kernel
        rng = self.rng
        ops.actor_egreedy_rng(self.greedy, rng.seed, rng.STREAM_EGREEDY_A, rng.STREAM_EGREEDY_U, rng.reserve(rng.STREAM_EGREEDY_A, E),
                              rng.reserve(rng.STREAM_EGREEDY_U, E), L.A, float(epsilon), E, self.action, self.qmax, qs_slot, ctrl, eps_ptr)

    def act(self, epsilon):
        one = self.ops.zeros(1)
        self._act_device(epsilon, one)
        if self.fused_tail or self.dist_tail:
            self.ops.mean_rows(self.qmax_all, 1, self.E, one)
        return self.action.cpu().numpy().astype(np.int64), float(one[0])

    def reset(self):
        self.obs, _ = self.envs.reset()

    # ------------------------------------------------------------------ agent.py:44-90
    def _rollout(self, epsilon, T, start, bound, test, stage, frames_out, ctrl=None, eps_ptr=None):
        """The body of Actor.sample's loop (agent.py:48-88), every step enqueued on the stream without touching the host."""
        cfg, ops, E = self.cfg, self.ops, self.E
        R = self.ring_len
        if cfg.learner.noisy_net and self.steps % cfg.learner.reset_noise_freq != 0:
            # NoisyLinear.forward composes mu + sigma * epsilon on every call (model.py:54-62), i.e. with the parameters as they are NOW; the
            # composed copies the device keeps were last written before the learner's latest Adam step (or come from a weight snapshot).
            # A rollout that does not start on a noise reset recomposes them once (with the default sample_steps = 80 it always does).
            self.model._dev.compose_noise()
        # round 5: for scalar and distributional heads the tail + env-step launch of step t also ENCODES the env's new observation (a0_actor_qhead_env_step_enc /
        # a0_actor_dist_tail_env_step_enc) — the next step starts with its features in place, and a scalar-head step is two launches (fc1 GEMM | tail + env step + next
        # encoder) instead of three.  The convolution weights do not change inside a rollout (NoisyNet touches the dense layers only), the last step has no next one.
        dev = self.model._dev
        step_enc = ((self.tail_env and (self.fused_tail or self.dist_tail) or (self.quant_tail and self.fused_commit and hasattr(ops, "actor_quantile_tail_env_step_enc")))
                    and bound and not test and dev.fused and (self.L.C, self.L.H, self.L.W) == (4, 84, 84) and hasattr(ops, "actor_qhead_env_step_enc")
                    and os.environ.get("A0_NO_X9") is None and os.environ.get("A0_STEP_ENC", "1") != "0"       # 0: tuning aid (same bytes, three launches per step)
                    # not on the launch schedule (the rollout into a stage ring): there the rollout runs BESIDE the update block, which is the critical path, and a
                    # workgroup that holds a CU's LDS from the tail to the end of the encoder takes more from the block than the saved boundary gives (9.43 -> 9.75 ms)
                    and not isinstance(self.replay, StageRing))
        # quantile actors (round 6): fc1's weight operand as bf16 term planes for the rollout's T GEMMs of E * K rows, split once here and after every noise reset
        # (a0_split_planes; the same exact terms the GEMM forms per tile, hence the same bits — A0_NO_WPLANES: tuning aid)
        self._planes_on = bool(self.quant_tail and bound and not test and self.fused_commit and hasattr(ops, "dense_fwd_wplanes") and os.environ.get("A0_NO_WPLANES") is None
                               and ops.dense_fwd_wplanes_ok(E * self.n_tau, 512, self.L.feat))
        if self._planes_on and not (cfg.learner.noisy_net and self.steps % cfg.learner.reset_noise_freq == 0):
            dev.refresh_fc1_planes()
        feat_ready = False
        for t in range(T):
            if cfg.learner.noisy_net and self.steps % cfg.learner.reset_noise_freq == 0:
                self.model.reset_noise(rng=self.rng)
                if self._planes_on:
                    dev.refresh_fc1_planes()
            merged = bound and not test and (self.tail_env or (self.quant_tail and self.fused_commit))
            if not feat_ready:
                self._act_device(epsilon, self.qs[t:t + 1], ctrl, eps_ptr, t, tail=not merged)
            cur_obs = self.obs
            if self.n > 1 and self.env_history:
                obs0 = self.envs.history(min(self.steps + 1, self.n) - 1)         # first observation of the emitted n-step transition
            elif self.n > 1:
                slot = self.steps % R
                self.ring_obs[slot * E * self.obs_bytes:(slot + 1) * E * self.obs_bytes].copy_(cur_obs)
                count = min(self.steps + 1, self.n)
                oldest = (self.steps - (count - 1)) % R
                obs0 = self.ring_obs[oldest * E * self.obs_bytes:(oldest + 1) * E * self.obs_bytes]
            else:
                obs0 = cur_obs
            if merged:
                # fc1's tail (head, argmax, epsilon-greedy) and the env step + n-step bookkeeping + replay row in ONE launch
                rp = self.replay
                kind = "qhead" if self.fused_tail else ("dist" if self.dist_tail else "quantile")
                targs = {"qhead": self._qhead_args, "dist": self._dist_tail_args, "quantile": self._quant_tail_args}[kind](epsilon, ctrl, eps_ptr, t)
                enc = (dev.wt, dev.encoder_weights(), self.ws.act3) if (step_enc and t + 1 < T) else None
                self.obs = self.envs.act_step_commit(targs, self.stat_mask[t * E:(t + 1) * E], self.stat_ret[t * E:(t + 1) * E], self.n, self.steps,
                                                     float(cfg.learner.discount), self.ring_act, self.ring_rew, self.ring_done, obs0, rp, (start + t * E) % rp.size,
                                                     kind=kind, enc=enc)
                feat_ready = enc is not None
                self.steps += 1
                continue
            if bound and not test and self.fused_commit:
                # env step, n-step bookkeeping and the replay row in one launch (synthetic env)
                rp = self.replay
                obs_next = self.envs.step_commit(self.action, self.stat_mask[t * E:(t + 1) * E], self.stat_ret[t * E:(t + 1) * E], self.n, self.steps,
                                                 float(cfg.learner.discount), self.ring_act, self.ring_rew, self.ring_done, obs0, rp, (start + t * E) % rp.size, ctrl)
                self.steps += 1
                self.obs = obs_next
                continue
            obs_next, reward, terminal, truncated, info = self.envs.step(self.action, final_mask=self.stat_mask[t * E:(t + 1) * E],
                                                                         final_ret=self.stat_ret[t * E:(t + 1) * E], ctrl=ctrl)
            ops.actor_nstep(E, self.n, self.steps, float(cfg.learner.discount), self.action, reward, terminal, truncated, info.get("life_loss"),
                            self.ring_act, self.ring_rew, self.ring_done, self.out_act, self.out_rew, self.out_done, ctrl)
            self.steps += 1
            if test:
                frames_out.append(obs_next.view(E, self.L.C, self.L.H, self.L.W)[:4, -1:].cpu().numpy())
            elif bound:
                rp = self.replay
                ops.replay_insert(rp.frames, rp.size, self.obs_bytes, (start + t * E) % rp.size, E, obs0, obs_next, self.out_act, self.out_rew, self.out_done,
                                  rp.act, rp.rew, rp.done, ctrl)
            else:
                sl = slice(t * E, (t + 1) * E)
                stage["obs"][sl].copy_(obs0.view(E, -1)); stage["obs_next"][sl].copy_(obs_next.view(E, -1))
                stage["act"][sl].copy_(self.out_act); stage["rew"][sl].copy_(self.out_rew); stage["done"][sl].copy_(self.out_done)
            self.obs = obs_next
        if self.fused_tail or self.dist_tail or (self.quant_tail and bound and not test and self.fused_commit):
            ops.mean_rows(self.qmax_all, T, E, self.qs)          # per-step mean max-Q (agent.py:38,88), all steps at once

    # ------------------------------------------------------------------ host envs (env_pool.HostEnvPool): nothing but inference between two env steps
    def _rollout_host(self, epsilon, T, start):
        """Actor.sample's loop (agent.py:48-88) over a host env that takes its actions without waiting (``step_send`` / ``step_recv``).  What lies between "the
        workers have finished step t" and "the workers see the actions of step t + 1" is the upload and Actor.act alone: the bookkeeping of step t (n-step window,
        replay row, episode statistics: agent.py:63-88) is enqueued AFTER the actions of step t + 1 have been sent and runs while the workers step.  Same kernels on
        the same inputs in an order that respects every dependency — the bytes of ``_rollout`` (tests/test_gpu_trainer.py::test_host_env_pool_matches_device_env);
        the action buffer alternates between two tensors because step t's n-step update reads its actions after step t + 1's have been chosen."""
        cfg, ops, E, rp = self.cfg, self.ops, self.E, self.replay
        R, ob, gamma = self.ring_len, self.obs_bytes, float(cfg.learner.discount)
        if getattr(self, "_host_actions", None) is None:
            self._host_actions = (self.action, ops.zeros(E, dtype=torch.int32))

        def bookkeeping(t, steps, action, cur_obs, obs_next, reward, terminal, truncated, info):
            sl = slice(t * E, (t + 1) * E)
            self.stat_mask[sl].copy_(info["final_mask"], non_blocking=True)
            self.stat_ret[sl].copy_(info["final_ret"], non_blocking=True)
            if self.n > 1:
                slot = steps % R
                self.ring_obs[slot * E * ob:(slot + 1) * E * ob].copy_(cur_obs)
                oldest = (steps - (min(steps + 1, self.n) - 1)) % R
                obs0 = self.ring_obs[oldest * E * ob:(oldest + 1) * E * ob]
            else:
                obs0 = cur_obs
            ops.actor_nstep(E, self.n, steps, gamma, action, reward, terminal, truncated, info.get("life_loss"), self.ring_act, self.ring_rew, self.ring_done,
                            self.out_act, self.out_rew, self.out_done, None)
            ops.replay_insert(rp.frames, rp.size, ob, (start + t * E) % rp.size, E, obs0, obs_next, self.out_act, self.out_rew, self.out_done, rp.act, rp.rew, rp.done, None)

        if cfg.learner.noisy_net and self.steps % cfg.learner.reset_noise_freq != 0:
            self.model._dev.compose_noise()                     # as in _rollout
        pending = None
        for t in range(T):
            if cfg.learner.noisy_net and self.steps % cfg.learner.reset_noise_freq == 0:
                self.model.reset_noise(rng=self.rng)
            self.action = self._host_actions[t & 1]
            self._act_device(epsilon, self.qs[t:t + 1], None, None, t)
            self.envs.step_send(self.action)
            if pending is not None:
                bookkeeping(*pending)
            obs_next, reward, terminal, truncated, info = self.envs.step_recv()
            pending = (t, self.steps, self.action, self.obs, obs_next, reward, terminal, truncated, info)
            self.steps += 1
            self.obs = obs_next
        bookkeeping(*pending)
        self.action = self._host_actions[0]
        if self.fused_tail or self.dist_tail:
            ops.mean_rows(self.qmax_all, T, E, self.qs)

    # ------------------------------------------------------------------ grouped host envs: CPU stepping of one group beside the GPU's inference of the other
    def _group_infer_send(self, g, obs, epsilon, t, offs):
        """Actor.act for group ``g`` on its observations (agent.py:25-39), then the actions go to the group's workers without waiting for them."""
        L, ops, dev, rng, k, off = self.L, self.ops, self.model._dev, self.rng, g["E"], g["off"]
        E = self.E
        dev.encode(g["ws"], obs, None, self.obs_bytes, 0, k, keep=False)
        qmax = self.qmax_all[t * E + off:t * E + off + k]
        if self.fused_tail:
            (W1, b1), (W2, b2) = dev.wb("fc1"), dev.wb("head")
            ops.actor_qhead(g["ws"].act3, k, L.feat, W1, b1, W2, b2, L.A, L.dueling, g["scratch"], rng.seed, rng.STREAM_EGREEDY_A, rng.STREAM_EGREEDY_U,
                            offs[0] + off, offs[1] + off, float(epsilon), g["action"], qmax)
        else:
            dev._dense(g["ws"].act3, L.feat, "fc1", g["ws"].h, k, True)
            Wh, bh = dev.wb("head")
            ns = ops.dense_fwd_partial(g["ws"].h, 512, Wh, k, L.Npad, 512, g["slabs"])
            ops.actor_dist_tail(g["slabs"], ns, bh, L.Npad, L.A, L.T, L.dueling, 2 if L.algo == "c51" else 1, self.atoms, k, rng.seed, rng.STREAM_EGREEDY_A,
                                rng.STREAM_EGREEDY_U, offs[0] + off, offs[1] + off, float(epsilon), g["action"], qmax)
        g["pool"].step_send(g["action"])

    def _rollout_groups(self, epsilon, T, start):
        """Actor.sample's loop (agent.py:48-88) over a grouped host env: while group A's worker processes step their envs, the GPU runs Actor.act for group B,
        and vice versa (the overlap the reference gets from ``num_actors`` actor processes, launch.py:30-61).  Every env sees exactly what it would see in
        a one-group rollout — its own observations, the epsilon-greedy draws at its own offset of the step's Philox block, its own n-step window — and
        group g's transitions of step t land in the ring slots [start + t E + off_g, ...): the same bytes in the same order (tests/test_gpu_trainer.py)."""
        cfg, ops, E, rng, rp = self.cfg, self.ops, self.E, self.rng, self.replay
        R, gamma = self.ring_len, float(cfg.learner.discount)
        reserve = lambda: (rng.reserve(rng.STREAM_EGREEDY_A, E), rng.reserve(rng.STREAM_EGREEDY_U, E))
        offs = reserve()
        for gi, g in enumerate(self.groups):                      # step 0's actions: nothing to overlap with yet
            self._group_infer_send(g, self.obs[gi], epsilon, 0, offs)
        for t in range(T):
            nxt_offs = reserve() if t + 1 < T else None
            for gi, g in enumerate(self.groups):
                k, off = g["E"], g["off"]
                cur_obs = self.obs[gi]
                if self.n > 1:
                    slot = self.steps % R
                    g["ring_obs"][slot * k * self.obs_bytes:(slot + 1) * k * self.obs_bytes].copy_(cur_obs)
                    oldest = (self.steps - (min(self.steps + 1, self.n) - 1)) % R
                    obs0 = g["ring_obs"][oldest * k * self.obs_bytes:(oldest + 1) * k * self.obs_bytes]
                else:
                    obs0 = cur_obs
                sl = slice(t * E + off, t * E + off + k)
                obs_next, reward, terminal, truncated, info = g["pool"].step_recv(self.stat_mask[sl], self.stat_ret[sl])      # waits for THIS group's workers only
                ops.actor_nstep(k, self.n, self.steps, gamma, g["action"], reward, terminal, truncated, info.get("life_loss"), g["ring_act"], g["ring_rew"], g["ring_done"],
                                g["out_act"], g["out_rew"], g["out_done"], None)
                ops.replay_insert(rp.frames, rp.size, self.obs_bytes, (start + t * E + off) % rp.size, k, obs0, obs_next, g["out_act"], g["out_rew"], g["out_done"],
                                  rp.act, rp.rew, rp.done, None)
                self.obs[gi] = obs_next
                if nxt_offs is not None:                            # the next step's actions for this group, while the other group's workers are stepping
                    self._group_infer_send(g, obs_next, epsilon, t + 1, nxt_offs)
            self.steps += 1
        ops.mean_rows(self.qmax_all, T, E, self.qs)

    def _graph_eligible(self, T, bound, test, state_dict) -> bool:
        cfg = self.cfg
        return (self.use_graph and bound and not test and state_dict is None and hasattr(self.envs, "_cur") and T % len(self.envs._obs) == 0
                and (not cfg.learner.noisy_net or T % cfg.learner.reset_noise_freq == 0) and (self.n == 1 or self.env_history or T % self.ring_len == 0))

    def _snapshot(self):
        rng = self.rng
        return {"g": self.envs.g, "steps": self.steps, "slot": self.replay.write_cursor(), "rng": dict(rng.offsets), "cur": self.envs._cur}

    def _rollout_graphed(self, epsilon, T, start):
        """One hipGraph launch per rollout: the 80 x ~10 launches are captured once; counters that keep advancing (env step, n-step
        index, Philox offsets, replay cursor) reach the kernels through a device control block, epsilon through a device scalar."""
        rng, rp = self.rng, self.replay
        if self._graph is None:
            if self._graph_warm < 2:                     # eager first: lazy allocations (workspaces, scratch) happen here
                self._graph_warm += 1
                self._rollout(epsilon, T, start, True, False, None, None)
                return
            self._ctrl_host.zero_(); self.ctrl.zero_(); self._eps_host[0] = float(epsilon); self.eps_dev.copy_(self._eps_host)
            rng.ctrl = self.ctrl
            base = self._snapshot()
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, **graph_capture_kwargs()):
                self._rollout(epsilon, T, start, True, False, None, None, ctrl=self.ctrl, eps_ptr=self.eps_dev)
            rng.ctrl = None                              # the captured kernels hold the pointer; eager rollouts must not add stale deltas
            after = self._snapshot()
            assert after["cur"] == base["cur"]
            self._graph = (graph, base, {k: after["rng"].get(k, 0) - base["rng"].get(k, 0) for k in after["rng"]})
            graph.replay()                               # capture only records; this is the rollout itself
            return
        graph, base, rng_adv = self._graph
        assert self.envs._cur == base["cur"]
        h = self._ctrl_host
        h[0] = self.envs.g - base["g"]
        h[1] = self.steps - base["steps"]
        h[2] = rng.offsets.get(rng.STREAM_EGREEDY_A, 0) - base["rng"].get(rng.STREAM_EGREEDY_A, 0)
        h[3] = rng.offsets.get(rng.STREAM_EGREEDY_U, 0) - base["rng"].get(rng.STREAM_EGREEDY_U, 0)
        h[4] = (start - base["slot"]) % rp.size
        h[5] = rng.offsets.get(rng.STREAM_TAUS, 0) - base["rng"].get(rng.STREAM_TAUS, 0)
        h[6] = rng.offsets.get(rng.STREAM_NOISE, 0) - base["rng"].get(rng.STREAM_NOISE, 0)
        self._eps_host[0] = float(epsilon)
        self.ctrl.copy_(h, non_blocking=True)
        self.eps_dev.copy_(self._eps_host, non_blocking=True)
        graph.replay()
        self.envs.g += T
        self.steps += T
        for k, v in rng_adv.items():
            rng.offsets[k] = rng.offsets.get(k, 0) + v

    def sample(self, epsilon, state_dict=None, test: bool = False):
        return self.sample_finish(self.sample_async(epsilon, state_dict, test))

    def sample_async(self, epsilon, state_dict=None, test: bool = False):
        """Enqueue one rollout on the current stream and return without waiting for it (the device counterpart of
        ``actor.futures.sample(...)``, launch.py:34-36); ``sample_finish`` collects the result."""
        cfg, ops, E = self.cfg, self.ops, self.E
        if isinstance(state_dict, DeepQNet):
            self.model._dev.copy_from(state_dict._dev)          # device-to-device weight snapshot
            state_dict = None
        elif state_dict is not None:
            self.model.load_state_dict(state_dict)
        T = int(cfg.actor.sample_steps)
        bound = self.replay is not None and not test
        st = None
        if not bound and not test:
            st = self._stage
            if st is None or st["obs"].shape[0] != T * E:
                st = self._stage = {"obs": ops.zeros(T * E, self.obs_bytes, dtype=torch.uint8), "obs_next": ops.zeros(T * E, self.obs_bytes, dtype=torch.uint8),
                                    "act": ops.zeros(T * E, dtype=torch.int32), "rew": ops.zeros(T * E), "done": ops.zeros(T * E)}
        start = self.replay.write_cursor() if bound else 0
        frames_out = []
        if self.groups is not None:
            if not bound:
                raise NotImplementedError("a grouped host env serves training rollouts into a replay ring (test / staged rollouts: use one group)")
            self._rollout_groups(epsilon, T, start)
        elif self._graph_eligible(T, bound, test, state_dict):
            self._rollout_graphed(epsilon, T, start)
        elif bound and hasattr(self.envs, "step_send") and os.environ.get("A0_HOST_ROLLOUT", "1") != "0":      # 0: the step-by-step order of _rollout (same bytes; a tuning aid)
            self._rollout_host(epsilon, T, start)
        else:
            self._rollout(epsilon, T, start, bound, test, st, frames_out)
        done_ev = torch.cuda.Event()
        done_ev.record()
        return (T, bound, test, st, start, frames_out, done_ev)

    def block_of(self, pending) -> TransitionBlock:
        """The TransitionBlock of a rollout that has been issued (``sample_async``) but not necessarily finished: what ``replay.extend`` needs is
        known on the host from the start (row count, ring position); the rows themselves are ordered before any later kernel on the stream."""
        T, bound, test, st, start, frames_out, done_ev = pending
        if test:
            raise ValueError("a test rollout produces frames, not transitions")
        if bound:
            return TransitionBlock(T * self.E, start=start, source=self.replay if isinstance(self.replay, StageRing) else None)
        return TransitionBlock(T * self.E, staged=st)

    def sample_finish(self, pending):
        T, bound, test, st, start, frames_out, done_ev = pending
        E = self.E
        done_ev.synchronize()
        # one device->host copy per rollout: mean max-Q per step and finished-episode returns, in the reference's order
        qs = self.qs[:T].cpu().tolist()
        mask = self.stat_mask[:T * E].cpu().numpy() != 0
        rs = self.stat_ret[:T * E].cpu().numpy()[mask].tolist()
        if test:
            return frames_out, rs, qs
        return self.block_of(pending), rs, qs

    def stats_async(self, pending):
        """The statistics of an issued rollout on their way to page-locked host buffers, stream-ordered behind it (and ahead of the next rollout, which reuses the
        device buffers); ``stats_finish`` waits for exactly these copies, not for whatever was enqueued afterwards."""
        T = pending[0]
        n = T * self.E
        host = self.__dict__.setdefault("_stat_host", {})
        out = []
        for name, src in (("qs", self.qs[:T]), ("mask", self.stat_mask[:n]), ("ret", self.stat_ret[:n])):
            key = (name, src.numel())
            if key not in host:
                host[key] = torch.empty(src.numel(), dtype=src.dtype).pin_memory()
            host[key].copy_(src, non_blocking=True)
            out.append(host[key])
        ev = torch.cuda.Event()
        ev.record()
        return (ev, out)

    def stats_finish(self, handle):
        """(returns, per-step mean max-Q) as ``sample_finish`` reports them."""
        ev, (qs, mask, ret) = handle
        ev.synchronize()
        return ret.numpy()[mask.numpy() != 0].tolist(), qs.tolist()

    def close(self):
        self.envs.close()


class BaseLearner:
    def __init__(self, cfg: ExpConfig, ops=None):
        self.cfg = cfg
        self.ops = ops = _ops_from(cfg, ops)
        lc = cfg.learner
        L = layout_from_cfg(cfg)
        self.engine = DeviceLearner(ops, L, int(lc.batch_size), discount=lc.discount, n_step=lc.n_step_q, double_q=lc.double_q, lr=lc.learning_rate,
                                    target_update_freq=lc.target_update_freq, vmin=lc.c51.vmin, vmax=lc.c51.vmax, K=lc.iqn.K, N=lc.iqn.N, N_dash=lc.iqn.N_dash,
                                    max_grad_norm=lc.max_grad_norm, mdqn_tau=lc.mdqn.tau, mdqn_lo=lc.mdqn.lo)
        self.rng = DeviceRng(ops, cfg.seed + 15485863)
        self.model = DeepQNet(cfg, ops=ops, dev_net=self.engine.online, rng=self.rng)
        self.model_target = DeepQNet(cfg, ops=ops, dev_net=self.engine.target, rng=self.rng)
        self.engine.sync_target(force=True)                 # model_target = deepcopy(model), agent.py:100
        self.batch_indices = torch.arange(lc.batch_size, device=ops.device)
        self.optimizer = self.engine                        # Adam state lives in the engine's flat buffers
        B = int(lc.batch_size)
        self._taus = [ops.empty(B * n) for n in (lc.iqn.K, lc.iqn.N_dash, lc.iqn.N)] if L.algo == "iqn" else None
        self._ones = torch.ones(B, device=ops.device)
        self.use_graph = True
        self._graphs, self._graph_warm = {}, {}
        # updates issued through this learner (eager or replayed): the device keeps the same count in state[6] and files each update's mean loss under it
        # (DeviceLearner.loss_ring), so the Trainer can read a block's means back in one copy
        self.updates_issued = 0

    @property
    def update_steps(self) -> int:
        return int(self.engine.state[1])

    # ------------------------------------------------------------------ hot path: batch addressed by ring slots
    def target_stage_batch(self, frames: torch.Tensor, slot: torch.Tensor, row_bytes: int, parity: int):
        """The target network's pass of the update that will consume this batch (``train_batch(..., tstage=parity)``), enqueued on the CURRENT stream —
        the Trainer's pipelined update block calls it on a second stream while the previous update is in flight.  Replayed from a hipGraph per parity."""
        eng = self.engine
        if not self.use_graph:
            return eng.target_stage(frames, slot, row_bytes, parity)
        key = ("tstage", frames.data_ptr(), slot.data_ptr(), row_bytes, parity)
        g = self._graphs.get(key)
        if g is None:
            if self._graph_warm.get(key, 0) < 2:
                self._graph_warm[key] = self._graph_warm.get(key, 0) + 1
                return eng.target_stage(frames, slot, row_bytes, parity)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, **graph_capture_kwargs()):
                eng.target_stage(frames, slot, row_bytes, parity)
            self._graphs[key] = g
        g.replay()

    def train_batch(self, frames: torch.Tensor, slot: Optional[torch.Tensor], row_bytes: int, act, rew, done, weights, rand=None, tstage=None):
        """``rand`` (parity tests): the update's random / proposed fractions handed in instead of drawn here — see DeviceLearner.forward_dense;
        the tensors must be persistent device buffers (the update is replayed from a hipGraph that holds their addresses)."""
        cfg = self.cfg
        if cfg.learner.noisy_net:
            eng = self.engine
            if eng.noise_joint is not None:
                # the two resets of agent.py:125-127 (online, then target) as ONE fill of the joint buffer: the same Philox draws as two fills
                self.rng.normal(self.rng.STREAM_NOISE, 0.1, eng.noise_joint, eng.noise_joint.numel())
            else:
                self.model.reset_noise(compose=False)            # DeviceLearner.forward_dense composes both nets' effective weights
                self.model_target.reset_noise(compose=False)
        if rand is None and self._taus is not None:
            for t in self._taus:
                self.rng.uniform(self.rng.STREAM_TAUS, t, t.numel())
            rand = self._taus
        out = self._update(frames, slot, row_bytes, act, rew, done, weights, rand, tstage)
        return out if isinstance(out, tuple) else (out, None)

    def _update(self, frames, slot, row_bytes, act, rew, done, weights, rand, tstage=None):
        """engine.update, replayed from hipGraphs when the caller keeps handing in the same device buffers (the Trainer's hot loop
        does: the replay's persistent batch tensors).  The whole update is ONE graph — under data parallelism too: the gradient exchange
        (dist.RcclGradAllReduce -> a0_dp_allreduce) is a stream-ordered launch like any other and is captured with it, the dense bucket's
        all-reduce as a parallel branch beside the encoder backward.  Only a hook that must run on the host (a plain callable, or the
        torch.distributed fallback) splits the update into three graphs — forward + dense backward | encoder backward | optimizer step —
        around its eager calls."""
        eng = self.engine
        self.updates_issued += 1
        if not self.use_graph:
            return eng.update(frames, slot, row_bytes, act, rew, done, weights, rand=rand, tstage=tstage)
        hooked = eng.grad_hook is not None and not getattr(eng.grad_hook, "in_graph", False)
        key = (frames.data_ptr(), None if slot is None else slot.data_ptr(), row_bytes, act.data_ptr(), rew.data_ptr(), done.data_ptr(), weights.data_ptr(), hooked, tstage)
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 8 or self._graph_warm.get(key, 0) < 2:       # two eager runs first: every lazy allocation has happened
                self._graph_warm[key] = self._graph_warm.get(key, 0) + 1
                return eng.update(frames, slot, row_bytes, act, rew, done, weights, rand=rand, tstage=tstage)
            mode = graph_capture_kwargs()
            g_f, g_e, g_apply = torch.cuda.CUDAGraph(), (torch.cuda.CUDAGraph() if hooked else None), (torch.cuda.CUDAGraph() if hooked else None)
            torch.cuda.synchronize()
            try:
                with torch.cuda.graph(g_f, **mode):
                    out = eng.forward_dense(frames, slot, row_bytes, act, rew, done, weights, rand, tstage=tstage)
                    if not hooked:                                # the whole update is one graph, in-graph exchange included
                        eng.exchange_begin()
                        eng.backward_encoder()
                        eng.exchange_end()
                        eng.apply()
            except Exception as e:      # noqa: BLE001
                if eng.grad_hook is None or hooked:
                    raise
                # an RCCL build that cannot be captured: keep the same exchange, issued eagerly between three graphs from now on
                import sys
                print(f"agent0_amd: capturing the gradient exchange into the update's hipGraph failed ({e}); splitting the update around it", file=sys.stderr)
                eng.grad_hook.in_graph = False
                torch.cuda.synchronize()
                return eng.update(frames, slot, row_bytes, act, rew, done, weights, rand=rand, tstage=tstage)
            if hooked:
                with torch.cuda.graph(g_e, **mode):
                    eng.backward_encoder()
                with torch.cuda.graph(g_apply, **mode):
                    eng.apply()
            g = self._graphs[key] = (g_f, g_e, g_apply, out)      # capturing records the launches without running them
        g[0].replay()
        if g[1] is not None:
            eng.exchange_begin()
            g[1].replay()
            eng.exchange_end()
            g[2].replay()
        return g[3]

    # ------------------------------------------------------------------ reference signature (agent.py:124-169)
    def train(self, data):
        frames, actions, rewards, terminals, weights, indices = data
        dev = self.ops.device
        B = frames.shape[0]
        u8 = frames.to(dev)
        u8 = (u8 if u8.dtype == torch.uint8 else u8.round().clamp_(0, 255).to(torch.uint8)).reshape(-1).contiguous()
        q_loss, f_loss = self.train_batch(u8, None, u8.numel() // B, actions.to(dev).to(torch.int32).contiguous(), rewards.to(dev).float().contiguous(),
                                          terminals.to(dev).float().contiguous(), weights.to(dev).float().contiguous())
        skipped = bool(self.engine.state[3])
        # CPU copies, like the reference's `.detach().cpu()` (agent.py:163-169); the Trainer's hot loop uses train_batch and stays on the device
        return {"q_loss": None if skipped else q_loss[:B].cpu(), "fraction_loss": None if f_loss is None else f_loss[:B].cpu(),
                "indices": indices.long().cpu()}


class DQNLearner(BaseLearner):
    pass


class MDQNLearner(BaseLearner):
    pass


class C51Learner(BaseLearner):
    pass


class QRLearner(BaseLearner):
    pass


class IQNLearner(BaseLearner):
    pass


class FQFLearner(BaseLearner):
    pass
