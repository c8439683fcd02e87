"""Picklable ``make_slice`` callables for the env-pool tests (worker processes import this module by name)."""
import functools

from oracle import core


def _synth(seed, rank, e0, k):
    return core.SynthVecEnv(k, seed=seed, rank=rank, env_offset=e0)


def synth_slice(seed=42, rank=0):
    """envs [e0, e0 + k) of the oracle's CPU twin of the synthetic vector env."""
    return functools.partial(_synth, seed, rank)


class ScriptedStack:
    """A vector env over envs [e0, e0 + k) whose (4, 84, 84) frame stacks change in every way a wrapper stack can produce: most steps
    append one frame, some append two (the stack jumps), some replace the whole stack (episode boundary).  Frames are a pure function
    of (seed, env, step), so any process can rebuild the expected observations."""

    def __init__(self, seed, e0, k):
        import numpy as np
        self.np, self.seed, self.e0, self.k, self.t = np, seed, e0, k, 0
        self.obs = np.zeros((k, 4, 84, 84), dtype=np.uint8)

    def frame(self, e, t, j=0):
        return self.np.random.default_rng([self.seed, e, t, j]).integers(0, 256, (84, 84), dtype=self.np.uint8)

    @staticmethod
    def mode(e, t):
        return (t + 3 * e) % 7        # 0: whole new stack, 3: two new frames, else one

    def reset(self, **kw):
        self.t = 0
        for i in range(self.k):
            self.obs[i] = self.frame(self.e0 + i, 0)[None]
        return self.obs.copy(), {}

    def step(self, action):
        np = self.np
        self.t += 1
        for i in range(self.k):
            e = self.e0 + i
            m = self.mode(e, self.t)
            if m == 0:
                self.obs[i] = np.stack([self.frame(e, self.t, j) for j in range(4)])
            elif m == 3:
                self.obs[i, :2] = self.obs[i, 2:].copy()
                self.obs[i, 2], self.obs[i, 3] = self.frame(e, self.t, 0), self.frame(e, self.t, 1)
            else:
                self.obs[i, :3] = self.obs[i, 1:].copy()
                self.obs[i, 3] = self.frame(e, self.t)
        z = np.zeros(self.k)
        return self.obs.copy(), z, z.astype(bool), z.astype(bool), {"life_loss": z.astype(bool)}

    def close(self):
        pass


def scripted_slice(seed=5):
    return functools.partial(ScriptedStack, seed)


# ------------------------------------------------------------------------------------------------ fixture group G12 behind the env pool
class _AsFrames:
    """Turns the scripted emulator's (counter, base step) observation into a (4, 84, 84) uint8 stack that names it: every byte is counter % 251, bytes 0..3 of
    frame 0 spell the counter itself (little endian)."""

    def __init__(self, env):
        self.env = env

    def _frames(self, obs):
        import numpy as np
        c = int(obs[0])
        f = np.full((4, 84, 84), c % 251, dtype=np.uint8)
        f[0, 0, :4] = np.frombuffer(np.uint32(c).tobytes(), np.uint8)
        return f

    def reset(self, **kw):
        o, info = self.env.reset(**kw)
        return self._frames(o), info

    def step(self, a):
        o, r, te, tr, info = self.env.step(a)
        return self._frames(o), r, te, tr, info

    def close(self):
        pass


G12_NAMES = ("fire_lives", "fire_busy", "fire_ends_in_presses", "nofire")


def _g12(e0, k):
    import os, sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import fake_ale
    from agent0_amd.common.host_envs import FireOnReset, LifeLossInfo, VectorizedSingles
    envs = []
    for i in range(e0, e0 + k):
        kw, needs_fire, _ = fake_ale.CASES[G12_NAMES[i % len(G12_NAMES)]]
        env = LifeLossInfo(fake_ale.ScriptedAle(fake_ale.make_script(**kw), needs_fire=needs_fire))
        envs.append(_AsFrames(FireOnReset(env) if needs_fire else env))
    return VectorizedSingles(envs, clip_reward=True)


def g12_slice():
    """env i of the vector = the product's wrapper chain (what AtariSlice builds, minus gymnasium's preprocessing) over G12 case i % 4's scripted emulator."""
    return _g12
