"""Host-side environment code of the N1 front-end — everything an env WORKER PROCESS imports: numpy only, no torch, so
that a worker never opens the GPU.  The device / DMA side is env_pool.HostEnvPool.

  VectorizedSingles   k single envs behind the gymnasium vector contract (autoreset, final_info, episode statistics, reward clipping)
  FireOnReset, LifeLossInfo, AtariSlice   the real Atari pipeline (gymnasium + ale-py when installed); semantics of the reference's
                      agent0/common/atari_wrappers.py:20-51,59-69
  HostSynthSlice      a host synthetic env of the right shapes for PCIe-inclusive throughput measurements
  worker_main         the worker loop: poll the command sequence number in the shared block, step / reset the slice, write the results
"""
from __future__ import annotations

import os
import time
from multiprocessing import shared_memory

import numpy as np

CMD_NONE, CMD_STEP, CMD_RESET, CMD_CLOSE = 0, 1, 2, 3
# control block (int64): [0] command WORD = (command << 56) | sequence number — one 8-byte word, so a worker can never see a new sequence
# number with the previous command (the word arrives by one aligned DMA / one store); [1] argument of a reset (seed, -1 = none), written
# by the parent before the word; [2 + w] sequence number worker w has completed
CTL_WORD, CTL_ARG, CTL_DONE0 = 0, 1, 2
CTL_SEQ = CTL_WORD
CMD_SHIFT = 56
SEQ_MASK = (1 << CMD_SHIFT) - 1


def ctl_word(cmd: int, seq: int) -> int:
    return (int(cmd) << CMD_SHIFT) | (int(seq) & SEQ_MASK)


def accept_word(word: int, seen: int):
    """A worker's view of the control word: (cmd, seq) if it carries the NEXT sequence number, else None (keep polling).

    Step words arrive by a device-to-host DMA copy of 8 bytes, and nothing guarantees that a blit kernel stores them as ONE 8-byte write: a
    reader can see some bytes of the new word beside bytes of the old one.  Two rules make every such mixture harmless: (1) only
    ``seen + 1`` is accepted — a torn sequence number is anything but that (or, when the bytes that changed have all arrived, already the
    right value); (2) the parent re-writes the word as CMD_STEP itself once a reset has been acknowledged (HostEnvPool.reset), so the
    command byte never changes under a DMA write and a torn word cannot pair the new number with a stale CMD_RESET."""
    seq = int(word) & SEQ_MASK
    if seq != ((int(seen) + 1) & SEQ_MASK):
        return None
    return int(word) >> CMD_SHIFT, seq


N_SCAL = 7      # scalar rows per env and step: reward, terminated, truncated, life_loss, final mask, final return, advance (see record)


def block_layout(E: int, obs_bytes: int, workers: int, frame_bytes: int = 0):
    """Byte offsets of the shared block: observations [2][E][obs_bytes] u8 | scalars [2][N_SCAL][E] f32 | actions [E] i32 | control i64 |
    newest frames [2][E][frame_bytes] u8 (device frame-stack mode; frame_bytes = 0: absent)."""
    a16 = lambda n: (n + 63) // 64 * 64
    off_obs = 0
    off_scal = a16(off_obs + 2 * E * obs_bytes)
    off_act = a16(off_scal + 2 * N_SCAL * E * 4)
    off_ctl = a16(off_act + E * 4)
    off_new = a16(off_ctl + 8 * (CTL_DONE0 + max(workers, 1)))
    total = a16(off_new + 2 * E * frame_bytes)
    return off_obs, off_scal, off_act, off_ctl, off_new, total


def block_views(buf, E: int, obs_bytes: int, workers: int, frame_bytes: int = 0):
    o_obs, o_scal, o_act, o_ctl, o_new, _ = block_layout(E, obs_bytes, workers, frame_bytes)
    return {"obs": np.ndarray((2, E, obs_bytes), dtype=np.uint8, buffer=buf, offset=o_obs),
            "scal": np.ndarray((2, N_SCAL, E), dtype=np.float32, buffer=buf, offset=o_scal),
            "act": np.ndarray((E,), dtype=np.int32, buffer=buf, offset=o_act),
            "ctl": np.ndarray((CTL_DONE0 + max(workers, 1),), dtype=np.int64, buffer=buf, offset=o_ctl),
            "new": np.ndarray((2, E, frame_bytes), dtype=np.uint8, buffer=buf, offset=o_new)}


class VectorizedSingles:
    """k single environments behind the gymnasium vector contract: autoreset (the observation returned at the end of an episode is the
    first one of the next; the finished episode's return goes to ``info["final_info"][i]["episode"]["r"]``), episode statistics over the
    UNCLIPPED rewards and sign-clipped rewards out — the order of the reference's wrapper list (atari_wrappers.py:61-68: statistics inside,
    clipping outside) — plus the OR of the per-env ``life_loss`` flags the reference's EpisodicLifeEnv reports (atari_wrappers.py:35-51)."""

    def __init__(self, envs, clip_reward: bool = True, seeds=None):
        """``seeds``: one seed per env, used by the FIRST reset that is not given one (every env gets its own emulator stream and a run is
        reproducible; later resets continue the streams, like gymnasium's)."""
        self.envs, self.clip = list(envs), clip_reward
        self.returns = np.zeros(len(self.envs), dtype=np.float64)
        self.seeds = None if seeds is None else [int(s) for s in seeds]

    def reset(self, seed=None, **kw):
        """``seed``: None, an int (env i gets seed + i, gymnasium's vector convention) or one per env."""
        k = len(self.envs)
        if seed is None and self.seeds is not None:
            seed, self.seeds = self.seeds, None
        seeds = [None] * k if seed is None else [int(seed) + i for i in range(k)] if np.isscalar(seed) else [None if s is None else int(s) for s in seed]
        obs = [e.reset(**kw)[0] if sd is None else e.reset(seed=sd, **kw)[0] for e, sd in zip(self.envs, seeds)]
        self.returns[:] = 0
        return np.stack(obs), {}

    def step(self, actions):
        k = len(self.envs)
        obs, rew, term, trunc, life = [None] * k, np.zeros(k, np.float64), np.zeros(k, bool), np.zeros(k, bool), np.zeros(k, bool)
        final = [None] * k
        for i, (env, a) in enumerate(zip(self.envs, actions)):
            o, r, te, tr, info = env.step(int(a))
            self.returns[i] += r
            rew[i], term[i], trunc[i], life[i] = r, te, tr, bool(info.get("life_loss", False))
            if te or tr:
                final[i] = {"episode": {"r": np.array([self.returns[i]], dtype=np.float32)}}
                self.returns[i] = 0
                o, _ = env.reset()
            obs[i] = o
        info = {"life_loss": life}
        if any(f is not None for f in final):
            info["final_info"] = np.array(final, dtype=object)
            info["_final_info"] = np.array([f is not None for f in final])
        return np.stack(obs), (np.sign(rew) if self.clip else rew), term, trunc, info

    def close(self):
        for e in self.envs:
            if hasattr(e, "close"):
                e.close()


class HostSynthSlice:
    """``make_slice`` of a HOST synthetic env with the shapes and rates of the device one (84x84 frame stack, rewards P = 0.05 / 0.05 / 0.9,
    terminal 1/500, life loss 1/200; frames drawn from a small pre-generated bank): stands in for ALE when measuring the front-end's
    PCIe-inclusive throughput (tools/bench_host_env.py).  Not byte-compatible with the device env — parity tests use the oracle's twin."""

    def __init__(self, seed: int = 42, bank: int = 32):
        self.seed, self.bank = seed, bank

    def __call__(self, e0: int, k: int):
        return _HostSynthEnv(e0, k, self.seed, self.bank)


class _HostSynthEnv:
    def __init__(self, e0, k, seed, bank):
        self.k = k
        self.rng = np.random.default_rng([seed, e0])
        self.frames = self.rng.integers(0, 256, (bank, 84, 84), dtype=np.uint8) * (self.rng.random((bank, 84, 84)) < 0.25)
        self.obs = np.zeros((k, 4, 84, 84), dtype=np.uint8)
        self.ret = np.zeros(k, dtype=np.float32)

    def reset(self, **kw):
        self.obs[:] = self.frames[self.rng.integers(0, len(self.frames), self.k)][:, None]
        self.ret[:] = 0
        return self.obs.copy(), {}

    def step(self, action):
        k, rng = self.k, self.rng
        new = self.frames[rng.integers(0, len(self.frames), k)]
        u = rng.random(k)
        rew = np.where(u < 0.05, -1.0, np.where(u < 0.10, 1.0, 0.0))
        term = rng.random(k) < 1 / 500
        life = (~term) & (rng.random(k) < 1 / 200)
        self.ret += rew
        self.obs[:, :3] = self.obs[:, 1:]
        self.obs[:, 3] = new
        self.obs[term] = new[term][:, None]
        info = {"life_loss": life}
        if term.any():
            fi = np.empty(k, dtype=object)
            for i in np.nonzero(term)[0]:
                fi[i] = {"episode": {"r": np.array([self.ret[i]], dtype=np.float32)}}
            info["final_info"], info["_final_info"] = fi, term.copy()
            self.ret[term] = 0
        return self.obs.copy(), rew, term, np.zeros(k, bool), info

    def close(self):
        pass


def record(buf, half, lo, k, obs, reward, terminated, truncated, info):
    """One slice's step result into the shared buffers (views into shared memory).

    Device frame-stack mode (``buf["new"]`` has a frame size): besides the whole stack, the newest frame of every env goes into its own
    compact array, and row 6 of the scalars says whether the stack merely ADVANCED — its older frames are byte-for-byte the previous
    observation's newer ones (the other half of the ring: what this slice wrote one command ago).  Then the device can rebuild the stack
    from what it already holds plus the newest frame, and only that frame has to cross PCIe; otherwise (episode boundaries, the extra
    presses after a lost life, anything a wrapper did to the stack) the whole stack is uploaded.  The test is on the bytes, so it is exact
    for any wrapper order."""
    obs = np.asarray(obs, dtype=np.uint8).reshape(k, -1)
    buf["obs"][half, lo:lo + k] = obs
    sc = buf["scal"][half]
    fb = buf["new"].shape[2]
    if fb:
        prev = buf["obs"][half ^ 1, lo:lo + k]
        buf["new"][half, lo:lo + k] = obs[:, -fb:]
        sc[6, lo:lo + k] = (obs[:, :-fb] == prev[:, fb:]).all(axis=1).astype(np.float32)
    else:
        sc[6, lo:lo + k] = 0.0
    sc[0, lo:lo + k] = np.asarray(reward, dtype=np.float32)
    sc[1, lo:lo + k] = np.asarray(terminated, dtype=np.float32)
    sc[2, lo:lo + k] = np.asarray(truncated, dtype=np.float32)
    sc[3, lo:lo + k] = np.asarray(info["life_loss"], dtype=np.float32) if "life_loss" in info else 0.0
    sc[4, lo:lo + k] = 0.0
    sc[5, lo:lo + k] = 0.0
    if "final_info" in info:
        mask = np.asarray(info["_final_info"], dtype=bool)
        sc[4, lo:lo + k] = mask.astype(np.float32)
        for i in np.nonzero(mask)[0]:
            sc[5, lo + i] = float(info["final_info"][i]["episode"]["r"][0])


def worker_main(w, make_slice, lo, k, shm_name, E, obs_bytes, workers, spin_us, frame_bytes=0, busy_us=500.0):
    shm = shared_memory.SharedMemory(name=shm_name)
    buf = block_views(shm.buf, E, obs_bytes, workers, frame_bytes)
    ctl = buf["ctl"]
    env = make_slice(lo, k)
    seen = 0
    parent, spins, busy_until = os.getppid(), 0, 0.0
    try:
        while True:
            # the sequence number arrives by DMA (steps) or from the parent (reset / close).  Inside a rollout the next command follows
            # within a few hundred microseconds (one actor step on the GPU + the upload): poll without sleeping for busy_us after each
            # command — a sleep costs ~60 us of timer slack per step, a sixth of the step — and fall back to sleeping between rollouts.
            while True:
                got = accept_word(int(ctl[CTL_WORD]), seen)          # ONE read: command and sequence number belong together
                if got is not None:
                    break
                if time.perf_counter() < busy_until:
                    continue
                time.sleep(spin_us * 1e-6)
                spins += 1
                if (spins & 0x3FFF) == 0 and os.getppid() != parent:
                    return                                   # the parent is gone (killed): do not poll a dead ring forever
            cmd, seen = got
            if cmd == CMD_CLOSE:
                break
            half = seen & 1
            if cmd == CMD_RESET:
                seed = int(ctl[CTL_ARG])                 # written by the parent before the word
                obs, _ = env.reset(seed=seed + lo) if seed >= 0 else env.reset()
                buf["obs"][half, lo:lo + k] = np.asarray(obs, dtype=np.uint8).reshape(k, -1)
            else:
                record(buf, half, lo, k, *env.step(buf["act"][lo:lo + k].copy()))
            ctl[CTL_DONE0 + w] = seen
            busy_until = time.perf_counter() + busy_us * 1e-6
    finally:
        env.close()
        ctl[CTL_DONE0 + w] = -1
        del buf, ctl
        shm.close()


# ------------------------------------------------------------------------------------------------ the real Atari pipeline
class _Delegate:
    """Minimal single-env wrapper base (no gymnasium dependency): everything not overridden goes to the wrapped env."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        return self.env.step(action)


PRESS_AFTER_RESET = (0, 1, 2)      # NOOP, FIRE, and the action after it: what the reference presses to get a game going (atari_wrappers.py:24-27,47-48)


def _press_start(env, fallback_reset):
    """Press the start sequence; a game that ends during it is reset again.  -> (obs, info) of the last press."""
    obs = info = None
    for a in PRESS_AFTER_RESET:
        obs, _, over, _, info = env.step(a)
        if over:
            obs, info = fallback_reset()
    return obs, info


class FireOnReset(_Delegate):
    """Some Atari games idle until FIRE is pressed: a reset is followed by the start sequence (semantics of atari_wrappers.py:20-32)."""

    def reset(self, **kw):
        self.env.reset(**kw)
        return _press_start(self.env, lambda: self.env.reset(**kw))


class LifeLossInfo(_Delegate):
    """Reports a lost life as ``info["life_loss"]`` — the flag Actor.sample ORs into ``done`` (agent.py:57-60) — without ending the
    episode, and restarts games that wait for FIRE after a lost life (semantics of atari_wrappers.py:35-51).  ``lives`` is read through
    the ALE handle of the unwrapped env."""

    def _lives(self) -> int:
        return int(self.env.unwrapped.ale.lives())

    def step(self, action):
        before = self._lives()
        obs, reward, terminated, truncated, info = self.env.step(action)
        after = self._lives()
        lost = before > after > 0
        info = dict(info, life_loss=lost)
        if lost and self.env.unwrapped.get_action_meanings()[1] == "FIRE":
            # the reference presses the three actions and ignores what they return except the last observation and info — a game that
            # ends during the presses is NOT reset here, the next step reports it (atari_wrappers.py:52-55)
            extra = None
            for a in PRESS_AFTER_RESET:
                obs, _, _, _, extra = self.env.step(a)
            info.update(extra or {})
        return obs, reward, terminated, truncated, info


class AtariSlice:
    """``make_slice`` for the env pool (picklable): k real Atari envs behind the vector contract.  Per env: gymnasium's AtariPreprocessing
    (grey 84x84, frame-skip 4, max-pool) and FrameStack(4) — library code, as in atari_wrappers.py:61-63 — then LifeLossInfo (when
    ``episode_life``) and FireOnReset; autoreset, episode statistics and sign-clipping are done by ``VectorizedSingles``."""

    def __init__(self, env_id: str, episode_life: bool = True, seed: int = 42):
        self.env_id, self.episode_life, self.seed = env_id, episode_life, seed

    def single(self, index: int):
        import gymnasium as gym
        from gymnasium.wrappers import AtariPreprocessing, FrameStack
        import ale_py  # noqa: F401  (registers the ALE namespace)

        env = FrameStack(AtariPreprocessing(gym.make(f"{self.env_id}NoFrameskip-v4"), terminal_on_life_loss=False), 4)
        env = LifeLossInfo(env) if self.episode_life else env
        return FireOnReset(env)

    def __call__(self, e0: int, k: int):
        # env e0 + i of the vector gets emulator seed `seed + e0 + i` on its first reset (make_atari adds 1000003 * rank to `seed`)
        return VectorizedSingles([self.single(e0 + i) for i in range(k)], clip_reward=True, seeds=[self.seed + e0 + i for i in range(k)])




class OffsetSlices:
    """``make_slice`` of one group of a larger vector env (env_pool.HostEnvGroups): local env range [lo, lo + k) -> global range [e0 + lo, e0 + lo + k).
    Lives here, in the numpy-only module, because worker processes unpickle it."""

    def __init__(self, make_slice, e0: int):
        self.make_slice, self.e0 = make_slice, int(e0)

    def __call__(self, lo: int, k: int):
        return self.make_slice(self.e0 + lo, k)
