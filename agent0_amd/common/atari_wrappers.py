"""Environment factory with the reference's ``make_atari`` signature (agent0/common/atari_wrappers.py:59-69).

The reference builds a gymnasium AsyncVectorEnv of ALE emulators on the host; neither gymnasium nor ale-py exists in
this image, and the hot path this build accelerates starts at the (E,4,84,84) uint8 observation batch.  ``make_atari``
therefore returns a DEVICE-RESIDENT synthetic vector env (agent0_amd/csrc/synth_env.hip, defined in
oracle/synth_env.c) that honours the tuple/info contract Actor.sample consumes (agent.py:55-62,85-88).  It is not
Atari — it exists so that throughput is measured on inputs of the right shape.

Host environments go through ``env_pool.HostEnvPool``: worker processes step slices of the vector env into a
double-buffered page-locked uint8 ring, observations are uploaded on a copy stream, actions reach the workers by DMA.  With gymnasium +
ale-py installed ``make_atari`` builds the real Atari pipeline that way (``AtariSlice``: per-env AtariPreprocessing + FrameStack from
gymnasium, the life-loss / FIRE handling below, episode statistics and reward clipping in ``env_pool.VectorizedSingles``).
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
    # A0_ENV_TASK_* (include/agent0_hip.h): action-independent reward stream / the learnable block-quadrant task (a contextual bandit) / the chase task (temporal credit:
    # the action moves the block, +1 on arrival at the target cell)
    TASKS = {"stream": 0, "block": 1, "chase": 2}

    def __init__(self, env_id: str, num_envs: int, seed: int = 42, rank: int = 0, ops=None, task: str = "stream"):
        if task not in self.TASKS:
            raise ValueError(f"env_task={task!r}: expected one of {sorted(self.TASKS)}")
        self.task = self.TASKS[task]
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

    def set_history(self, k: int):
        """Keep the last ``k`` observations addressable (k >= 2 buffers, cycled): an n-step actor reads the observation of n-1 steps ago
        straight from here (``history``) instead of copying every observation into a ring of its own."""
        k = max(int(k), 2)
        if k != len(self._obs):
            n = self._obs[0].numel()
            cur = self._obs[self._cur]
            self._obs = [cur] + [self.ops.zeros(n, dtype=torch.uint8) for _ in range(k - 1)]
            self._cur = 0

    def history(self, j: int) -> torch.Tensor:
        """The observation of ``j`` steps ago (0 = current); valid for j < number of buffers - 1 while the next step is being written."""
        return self._obs[(self._cur - j) % len(self._obs)]

    def reset(self, **kwargs):
        self.g = 0
        self._cur = 0
        self.ops.env_reset(self.seed, self.rank, self.E, self._obs[0], self.ep_ret, task=self.task)
        return self._obs[0], {}

    def step(self, action: torch.Tensor, final_mask: Optional[torch.Tensor] = None, final_ret: Optional[torch.Tensor] = None, ctrl: Optional[torch.Tensor] = None):
        """``final_mask`` / ``final_ret``: optional caller buffers for the finished-episode record of this step (the gymnasium
        ``info["final_info"]`` / ``info["_final_info"]`` pair, agent.py:85-88)."""
        self.g += 1
        nxt = (self._cur + 1) % len(self._obs)
        fm = self.final_mask if final_mask is None else final_mask
        fr = self.final_ret if final_ret is None else final_ret
        self.ops.env_step(self.seed, self.rank, self.E, self.g, self._obs[self._cur], self._obs[nxt], self.ep_ret, self.reward,
                          self.terminal, self.truncated, self.life_loss, fm, fr, ctrl, action=action, A=self.action_dim, task=self.task)
        self._cur = nxt
        info = {"life_loss": self.life_loss, "final_mask": fm, "final_ret": fr}
        return self._obs[nxt], self.reward, self.terminal, self.truncated, info

    def step_commit(self, action, final_mask, final_ret, n, steps, gamma, ring_act, ring_rew, ring_done, obs0, replay, start_slot, ctrl=None):
        """``step`` + the actor's n-step bookkeeping + the replay row commit in one launch (a0_env_synth_step_commit); ``obs0`` is the first
        observation of the emitted transition, ``replay`` anything with frames / size / act / rew / done (ReplayDataset, StageRing)."""
        self.g += 1
        nxt = (self._cur + 1) % len(self._obs)
        self.ops.env_step_commit(self.seed, self.rank, self.E, self.g, self._obs[self._cur], self._obs[nxt], self.ep_ret, final_mask, final_ret, n, steps, gamma,
                                 action, ring_act, ring_rew, ring_done, obs0, replay.frames, replay.size, start_slot, replay.act, replay.rew, replay.done, ctrl, A=self.action_dim, task=self.task)
        self._cur = nxt
        return self._obs[nxt]

    def act_step_commit(self, tail_args, final_mask, final_ret, n, steps, gamma, ring_act, ring_rew, ring_done, obs0, replay, start_slot, kind: str = "qhead", enc=None):
        """``step_commit`` with the actor's tail in the same launch: ``tail_args`` are the arguments of ``ops.actor_qhead`` (``kind="qhead"``: scalar heads,
        a0_actor_qhead_env_step), of ``ops.actor_dist_tail`` (``"dist"``: c51 / qr, a0_actor_dist_tail_env_step) or of the quantile tail (``"quantile"``:
        iqn / fqf, a0_actor_quantile_tail_env_step), all ending in action, qmax, ctrl, eps_ptr; the action chosen there is the one this step takes."""
        self.g += 1
        nxt = (self._cur + 1) % len(self._obs)
        common = (self.seed, self.rank, self.g, self._obs[self._cur], self._obs[nxt], self.ep_ret, final_mask, final_ret, n, steps, gamma, ring_act, ring_rew, ring_done, obs0,
                  replay.frames, replay.size, start_slot, replay.act, replay.rew, replay.done)
        if enc is not None:      # (wt, encoder weights, act3): the same launch goes on to encode the new observation — the next step's features (a0_actor_*_env_step_enc)
            {"qhead": self.ops.actor_qhead_env_step_enc, "dist": self.ops.actor_dist_tail_env_step_enc, "quantile": self.ops.actor_quantile_tail_env_step_enc}[kind](
                *tail_args, *common, task=self.task, wt=enc[0], enc_w=enc[1], act3_next=enc[2])
        else:
            {"qhead": self.ops.actor_qhead_env_step, "dist": self.ops.actor_dist_tail_env_step, "quantile": self.ops.actor_quantile_tail_env_step}[kind](*tail_args, *common, task=self.task)
        self._cur = nxt
        return self._obs[nxt]

    def close(self):
        pass


from .host_envs import AtariSlice, FireOnReset, LifeLossInfo, PRESS_AFTER_RESET  # noqa: E402,F401  (numpy-only module: env workers import it without torch)


def real_atari_available() -> bool:
    try:
        import gymnasium, ale_py  # noqa: F401,E401
    except ImportError:
        return False
    return True


def make_atari(env_id: str, num_envs: int, episode_life: bool = True, seed: int = 42, rank: int = 0, ops=None, synthetic=None, num_workers=None, task: str = "stream"):
    """``task``: reward task of the synthetic env (``cfg.env_task``: "stream" = the bench workload, "block" = learnable; ignored by the real Atari env).
    ``synthetic=None``: the real Atari env (through the host env pool) when gymnasium + ale-py are importable, else the device-resident
    synthetic env.  ``num_workers``: env worker processes (default: one per 16 envs, at most 16; 0 = step in this process)."""
    if synthetic is None:
        synthetic = not real_atari_available()
    if synthetic:
        return DeviceSynthVecEnv(env_id, num_envs, seed=seed, rank=rank, ops=ops, task=task)
    from .env_pool import HostEnvPool

    if num_workers is None:
        num_workers = 0 if num_envs < 4 else min(16, max(1, num_envs // 16))
    return HostEnvPool(AtariSlice(env_id, episode_life, seed + 1000003 * rank), num_envs, obs_shape=(4, 84, 84), action_dim=ACTION_DIMS.get(env_id, 18),
                       num_workers=num_workers, ops=ops)
