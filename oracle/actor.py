"""Oracle: vectorized rollout — epsilon-greedy action selection and n-step packing.

Restates /root/reference agent0/deepq/agent.py:16-93 (class Actor):
  act     agent.py:25-39   — draw order: randint(0,A,E) FIRST, then rand(E)
  sample  agent.py:44-90   — done logic 57-62, n-step reverse scan 64-73 over a
          deque(maxlen=n) that is never cleared (quirk Q9), pack (st || st_next)
          78-81, episode returns 85-88.
The two random draws per step are supplied by the caller (``draw(E)`` returns
``(action_random int[E], u float[E])``) so that a GPU run can be compared on
identical draws.
"""
from __future__ import annotations

from collections import deque
from typing import Callable, Optional

import numpy as np
import torch

from . import nets


def numpy_global_draw(action_dim: int):
    """The reference's own stream: numpy global RNG, randint then rand."""

    def draw(E):
        a = np.random.randint(0, action_dim, E)
        u = np.random.rand(E)
        return a, u

    return draw


def egreedy(q: np.ndarray, action_random: np.ndarray, u: np.ndarray, epsilon: float):
    """-> (action[E], mean_e max_a Q).  ``u > epsilon`` picks greedy."""
    greedy = q.argmax(axis=-1)
    return np.where(u > epsilon, greedy, action_random), float(q.max(axis=-1).mean())


def nstep_scan(tracker, discount: float):
    """Reverse scan over the tracker entries (agent.py:64-69) -> (R float64[E], D bool[E])."""
    reward0 = tracker[-1][2]
    R = np.zeros_like(reward0)
    D = np.zeros_like(reward0, dtype=np.bool_)
    for _, _, rt, dt in reversed(tracker):
        D = np.logical_or(D, dt)
        R = R * discount * (1 - dt) + rt
    return R, D


class OracleActor:
    def __init__(self, envs, params, spec, n_step: int = 1, discount: float = 0.99, sample_steps: int = 80,
                 draw: Optional[Callable] = None, taus_fn: Optional[Callable] = None,
                 noisy_reset: Optional[Callable] = None, reset_noise_freq: int = 4):
        self.envs, self.p, self.spec = envs, params, spec
        self.discount, self.sample_steps = discount, sample_steps
        self.obs, _ = envs.reset()
        self.tracker = deque(maxlen=n_step)
        self.steps = 0
        self.draw = draw or numpy_global_draw(spec.action_dim)
        self.taus_fn = taus_fn
        self.noisy_reset, self.reset_noise_freq = noisy_reset, reset_noise_freq

    @torch.no_grad()
    def qvalues(self, obs_u8: np.ndarray) -> np.ndarray:
        x = nets.normalize(torch.from_numpy(obs_u8))
        taus = self.taus_fn(obs_u8.shape[0]) if self.taus_fn is not None else None
        return nets.qval(self.p, self.spec, x, taus).numpy()

    def act(self, epsilon: float):
        q = self.qvalues(self.obs)
        a_rand, u = self.draw(self.obs.shape[0])
        return egreedy(q, a_rand, u, epsilon)

    def sample(self, epsilon: float):
        rs, qs, data = [], [], []
        for _ in range(self.sample_steps):
            if self.spec.noisy and self.noisy_reset is not None and self.steps % self.reset_noise_freq == 0:
                self.noisy_reset(self.p)
            action, qmax = self.act(epsilon)
            obs_next, reward, terminal, truncated, info = self.envs.step(action)
            self.steps += 1
            done = np.logical_or(terminal, info["life_loss"]) if "life_loss" in info else terminal
            done = np.logical_and(done, np.logical_not(truncated))
            self.tracker.append((self.obs, action, reward, done))
            R, D = nstep_scan(self.tracker, self.discount)
            obs0, act0 = self.tracker[0][0], self.tracker[0][1]
            for st, at, rt, dt, st_next in zip(obs0, act0, R, D, obs_next):
                data.append((np.concatenate((st, st_next), axis=0), at, rt, dt))
            self.obs = obs_next
            qs.append(qmax)
            if "final_info" in info:
                for stat in info["final_info"][info["_final_info"]]:
                    rs.append(stat["episode"]["r"][0])
        return data, rs, qs
