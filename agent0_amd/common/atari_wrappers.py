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
        self.ops.env_reset(self.seed, self.rank, self.E, self._obs[0], self.ep_ret)
        return self._obs[0], {}

    def step(self, action: torch.Tensor, final_mask: Optional[torch.Tensor] = None, final_ret: Optional[torch.Tensor] = None, ctrl: Optional[torch.Tensor] = None):
        """``final_mask`` / ``final_ret``: optional caller buffers for the finished-episode record of this step (the gymnasium
        ``info["final_info"]`` / ``info["_final_info"]`` pair, agent.py:85-88)."""
        self.g += 1
        nxt = (self._cur + 1) % len(self._obs)
        fm = self.final_mask if final_mask is None else final_mask
        fr = self.final_ret if final_ret is None else final_ret
        self.ops.env_step(self.seed, self.rank, self.E, self.g, self._obs[self._cur], self._obs[nxt], self.ep_ret, self.reward,
                          self.terminal, self.truncated, self.life_loss, fm, fr, ctrl)
        self._cur = nxt
        info = {"life_loss": self.life_loss, "final_mask": fm, "final_ret": fr}
        return self._obs[nxt], self.reward, self.terminal, self.truncated, info

    def step_commit(self, action, final_mask, final_ret, n, steps, gamma, ring_act, ring_rew, ring_done, obs0, replay, start_slot, ctrl=None):
        """``step`` + the actor's n-step bookkeeping + the replay row commit in one launch (a0_env_synth_step_commit); ``obs0`` is the first
        observation of the emitted transition, ``replay`` anything with frames / size / act / rew / done (ReplayDataset, StageRing)."""
        self.g += 1
        nxt = (self._cur + 1) % len(self._obs)
        self.ops.env_step_commit(self.seed, self.rank, self.E, self.g, self._obs[self._cur], self._obs[nxt], self.ep_ret, final_mask, final_ret, n, steps, gamma,
                                 action, ring_act, ring_rew, ring_done, obs0, replay.frames, replay.size, start_slot, replay.act, replay.rew, replay.done, ctrl)
        self._cur = nxt
        return self._obs[nxt]

    def close(self):
        pass


class HostVecEnvAdapter:
    """Runs any host vector env with the gymnasium step/reset contract the reference consumes (agent0/deepq/agent.py:42,55-62,85-88:
    ``reset() -> (obs, info)``, ``step(a) -> (obs, reward, terminated, truncated, info)`` with ``info["life_loss"]``,
    ``info["final_info"]`` / ``info["_final_info"]``) behind the device interface of DeviceSynthVecEnv: actions come down with one
    D2H copy, the (E,4,84,84) uint8 observation batch goes up through a pinned buffer (7 KB per env per step).  This is the N1
    front-end of SURVEY.md §8(f): with gymnasium + ale-py installed, ``make_atari`` wraps the real Atari env in it."""

    def __init__(self, env, num_envs: int, ops=None, action_dim=None):
        if ops is None:
            from agent0_amd.ops import HipOps
            ops = HipOps()
        self.env, self.E, self.ops = env, num_envs, ops
        obs_shape = tuple(env.observation_space.shape[1:]) if hasattr(env, "observation_space") else (4, 84, 84)
        self.observation_space = _Space(shape=(num_envs,) + obs_shape)
        n = action_dim if action_dim is not None else int(env.action_space[0].n)
        self.action_dim = n
        self.action_space = _Space(n=n)
        numel = num_envs * int(torch.tensor(obs_shape).prod())
        self._obs = [ops.zeros(numel, dtype=torch.uint8), ops.zeros(numel, dtype=torch.uint8)]
        self._pin = torch.zeros(numel, dtype=torch.uint8).pin_memory()
        self._cur_i = 0
        self.reward, self.terminal, self.truncated = ops.zeros(num_envs), ops.zeros(num_envs), ops.zeros(num_envs)
        self.life_loss, self.final_mask, self.final_ret = ops.zeros(num_envs), ops.zeros(num_envs), ops.zeros(num_envs)
        self._scal = torch.zeros(6, num_envs).pin_memory()
        self.g = 0

    def _upload(self, obs_np, dst):
        import numpy as np
        self._pin.copy_(torch.from_numpy(np.ascontiguousarray(obs_np, dtype=np.uint8).reshape(-1)))
        dst.copy_(self._pin, non_blocking=True)

    def reset(self, **kwargs):
        obs, info = self.env.reset(**kwargs)
        self._cur_i = 0
        self._upload(obs, self._obs[0])
        self.g = 0
        return self._obs[0], info

    def step(self, action: torch.Tensor, final_mask=None, final_ret=None, ctrl=None):
        import numpy as np
        obs, reward, terminated, truncated, info = self.env.step(action.cpu().numpy().astype(np.int64))
        self.g += 1
        nxt = 1 - self._cur_i
        torch.cuda.current_stream().synchronize()          # the pinned staging buffers are reused every step
        self._upload(obs, self._obs[nxt])
        E = self.E
        sc = self._scal
        sc.zero_()
        sc[0] = torch.from_numpy(np.asarray(reward, dtype=np.float32))
        sc[1] = torch.from_numpy(np.asarray(terminated, dtype=np.float32))
        sc[2] = torch.from_numpy(np.asarray(truncated, dtype=np.float32))
        has_life = "life_loss" in info
        if has_life:
            sc[3] = torch.from_numpy(np.asarray(info["life_loss"], dtype=np.float32))
        if "final_info" in info:
            mask = np.asarray(info["_final_info"], dtype=bool)
            sc[4] = torch.from_numpy(mask.astype(np.float32))
            for i in np.nonzero(mask)[0]:
                sc[5, i] = float(info["final_info"][i]["episode"]["r"][0])
        fm = self.final_mask if final_mask is None else final_mask
        fr = self.final_ret if final_ret is None else final_ret
        for k, dst in enumerate((self.reward, self.terminal, self.truncated, self.life_loss, fm, fr)):
            dst.copy_(sc[k], non_blocking=True)
        self._cur_i = nxt
        out_info = {"final_mask": fm, "final_ret": fr}
        if has_life:
            out_info["life_loss"] = self.life_loss
        return self._obs[nxt], self.reward, self.terminal, self.truncated, out_info

    def close(self):
        self.env.close()


def _make_gymnasium_atari(env_id: str, num_envs: int, episode_life: bool):
    """The reference's wrapper stack (atari_wrappers.py:11-69) on a real gymnasium/ALE install."""
    import gymnasium as gym
    import numpy as np
    from gymnasium.wrappers import AtariPreprocessing, FrameStack, RecordEpisodeStatistics
    import ale_py  # noqa: F401

    class ClipRewardEnv(gym.RewardWrapper):
        def reward(self, reward):
            return np.sign(reward)

    class FireResetEnv(gym.Wrapper):
        def reset(self, **kwargs):
            self.env.reset(**kwargs)
            for a in range(3):
                obs, _, terminated, _, info = self.env.step(a)
                if terminated:
                    obs, info = self.env.reset(**kwargs)
            return obs, info

    class EpisodicLifeEnv(gym.Wrapper):
        def step(self, action):
            old = self.env.unwrapped.ale.lives()
            obs, reward, done, trunc, info = super().step(action)
            new = self.env.unwrapped.ale.lives()
            life_loss = old > new > 0
            info["life_loss"] = life_loss
            if life_loss and self.env.unwrapped.get_action_meanings()[1] == "FIRE":
                for a in range(3):
                    obs, _, _, _, step_info = self.env.step(a)
                info.update(step_info)
            return obs, reward, done, trunc, info

    wrappers = [lambda x: AtariPreprocessing(x, terminal_on_life_loss=False), lambda x: FrameStack(x, 4),
                (lambda x: EpisodicLifeEnv(x)) if episode_life else (lambda x: x), FireResetEnv, RecordEpisodeStatistics, ClipRewardEnv]
    return gym.make_vec(f"{env_id}NoFrameskip-v4", num_envs, wrappers=wrappers)


def make_atari(env_id: str, num_envs: int, episode_life: bool = True, seed: int = 42, rank: int = 0, ops=None, synthetic=None):
    """``synthetic=None``: use the real Atari env when gymnasium + ale-py are importable, else the device-resident synthetic env."""
    if synthetic is None:
        try:
            import gymnasium, ale_py  # noqa: F401,E401
            synthetic = False
        except ImportError:
            synthetic = True
    if synthetic:
        return DeviceSynthVecEnv(env_id, num_envs, seed=seed, rank=rank, ops=ops)
    return HostVecEnvAdapter(_make_gymnasium_atari(env_id, num_envs, episode_life), num_envs, ops=ops)
