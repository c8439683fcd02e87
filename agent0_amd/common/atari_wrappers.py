"""Environment factory with the reference's ``make_atari`` signature (agent0/common/atari_wrappers.py:59-69).

The reference builds a gymnasium AsyncVectorEnv of ALE emulators on the host; neither gymnasium nor ale-py exists in
this image, and the hot path this build accelerates starts at the (E,4,84,84) uint8 observation batch.  ``make_atari``
therefore returns a DEVICE-RESIDENT synthetic vector env (agent0_amd/csrc/synth_env.hip, defined in
oracle/synth_env.c) that honours the tuple/info contract Actor.sample consumes (agent.py:55-62,85-88).  It is not
Atari — it exists so that throughput is measured on inputs of the right shape; a host ALE front-end
feeding the same device buffers is the next-row item N1.
"""
from __future__ import annotations

from typing import Optional

import torch

# minimal ALE action-set sizes of the README's eight games (README.md:65-112); others default to the full 18
ACTION_DIMS = {"Asterix": 9, "BeamRider": 9, "Breakout": 4, "Enduro": 9, "MsPacman": 9, "Qbert": 6, "Seaquest": 18, "SpaceInvaders": 6, "Pong": 6}


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n

    def __getitem__(self, i):
        return self


class DeviceSynthVecEnv:
    """obs (E,4,84,84) u8 on the GPU; ``step`` takes a device int32 action tensor and returns device tensors."""

    H = W = 84

    def __init__(self, env_id: str, num_envs: int, seed: int = 42, rank: int = 0, ops=None):
        if ops is None:
            from agent0_amd.ops import HipOps
            ops = HipOps()
        self.ops, self.E, self.seed, self.rank = ops, num_envs, seed, rank
        self.env_id = env_id
        self.action_dim = ACTION_DIMS.get(env_id, 18)
        self.observation_space = _Space(shape=(num_envs, 4, self.H, self.W))
        self.action_space = _Space(n=self.action_dim)
        n = num_envs * 4 * self.H * self.W
        self._obs = [ops.zeros(n, dtype=torch.uint8), ops.zeros(n, dtype=torch.uint8)]
        self._cur = 0
        self.ep_ret = ops.zeros(num_envs)
        self.reward, self.terminal, self.truncated = ops.zeros(num_envs), ops.zeros(num_envs), ops.zeros(num_envs)
        self.life_loss, self.final_mask, self.final_ret = ops.zeros(num_envs), ops.zeros(num_envs), ops.zeros(num_envs)
        self.g = 0

    @property
    def obs(self) -> torch.Tensor:
        return self._obs[self._cur]

    def reset(self, **kwargs):
        self.g = 0
        self._cur = 0
        self.ops.env_reset(self.seed, self.rank, self.E, self._obs[0], self.ep_ret)
        return self._obs[0], {}

    def step(self, action: torch.Tensor, final_mask: Optional[torch.Tensor] = None, final_ret: Optional[torch.Tensor] = None, ctrl: Optional[torch.Tensor] = None):
        """``final_mask`` / ``final_ret``: optional caller buffers for the finished-episode record of this step (the gymnasium
        ``info["final_info"]`` / ``info["_final_info"]`` pair, agent.py:85-88)."""
        self.g += 1
        nxt = 1 - self._cur
        fm = self.final_mask if final_mask is None else final_mask
        fr = self.final_ret if final_ret is None else final_ret
        self.ops.env_step(self.seed, self.rank, self.E, self.g, self._obs[self._cur], self._obs[nxt], self.ep_ret, self.reward,
                          self.terminal, self.truncated, self.life_loss, fm, fr, ctrl)
        self._cur = nxt
        info = {"life_loss": self.life_loss, "final_mask": fm, "final_ret": fr}
        return self._obs[nxt], self.reward, self.terminal, self.truncated, info

    def close(self):
        pass


def make_atari(env_id: str, num_envs: int, episode_life: bool = True, seed: int = 42, rank: int = 0, ops=None):
    return DeviceSynthVecEnv(env_id, num_envs, seed=seed, rank=rank, ops=ops)
