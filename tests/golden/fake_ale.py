"""A scripted stand-in for one ALE emulator, shared by the fixture generator (gen_golden.py group G12, where the REFERENCE's wrappers run on it) and the tests
(where the product's wrappers run on the same scripts).  Own test code — nothing of the reference is in here.

The outcome of base step number t (counted over the env's whole life, resets included) is fixed by the script, whatever the action: reward, lives after the step,
terminated, truncated.  Observations carry the running counter ``c`` of resets + steps, so a recorded ``obs id`` names exactly which emulator frame a wrapper
returned; every action the emulator receives is logged (``-1`` = reset).  Two chains that make the same emulator calls in the same order therefore see the same
outcomes, and any difference in the calls shows up in the log, the obs ids and everything after."""
from __future__ import annotations

import numpy as np

LIVES0 = 3
ACTIONS_FIRE = ["NOOP", "FIRE", "RIGHT", "LEFT"]
ACTIONS_NOFIRE = ["NOOP", "UP", "RIGHT", "LEFT"]


def make_script(seed: int, steps: int, *, p_life=0.10, p_term=0.01, p_trunc=0.01, qbert_delay=2, early_end=()):
    """Per base step: reward (raw, magnitudes up to 7), lives_after, terminated, truncated.  A lost last life ends the game ``qbert_delay`` steps LATER (the frames
    with lives == 0 that the reference's comment at atari_wrappers.py:44-46 is about).  ``early_end``: base-step numbers that terminate regardless (used to end a
    game during the start presses).  Generated sequentially: after a terminated / truncated step the lives start again at LIVES0 (a reset follows in every chain
    that honours the flags; a chain that ignores them sees lives jump back up, which no wrapper reads as a loss)."""
    g = np.random.Generator(np.random.PCG64(seed))
    rew = g.choice(np.array([0.0, 0.0, 0.0, 1.0, -0.5, 7.0, -7.0, 0.25]), size=steps)
    lives_after = np.zeros(steps, np.int64)
    term, trunc = np.zeros(steps, bool), np.zeros(steps, bool)
    lives, pending = LIVES0, -1
    for t in range(steps):
        u = g.random(3)
        if pending >= 0:                      # lives == 0, game not over yet
            pending -= 1
            if pending < 0:
                term[t] = True
        elif t in early_end or u[1] < p_term:
            term[t] = True
        elif u[2] < p_trunc:
            trunc[t] = True
        elif u[0] < p_life:
            lives -= 1
            if lives == 0:
                if qbert_delay > 0:
                    pending = qbert_delay - 1
                else:
                    term[t] = True
        lives_after[t] = lives
        if term[t] or trunc[t]:
            lives, pending = LIVES0, -1
    return {"reward": rew, "lives_after": lives_after, "terminated": term, "truncated": trunc}


class ScriptedAle:
    """Single env with the attribute surface the wrappers touch: ``unwrapped``, ``unwrapped.ale.lives()``, ``get_action_meanings()``, ``reset``, ``step``."""

    def __init__(self, script, needs_fire: bool = True):
        self.script, self.needs_fire = script, needs_fire
        self.t = 0                    # base steps taken
        self.c = 0                    # resets + steps: the observation id
        self._lives = LIVES0
        self.log = []
        self.ale = self

    @property
    def unwrapped(self):
        return self

    def lives(self):
        return int(self._lives)

    def get_action_meanings(self):
        return list(ACTIONS_FIRE if self.needs_fire else ACTIONS_NOFIRE)

    def _obs(self):
        return np.array([self.c, self.t], dtype=np.int64)

    def reset(self, **kw):
        self.log.append(-1)
        self.c += 1
        self._lives = LIVES0
        return self._obs(), {"c": self.c}

    def step(self, action):
        t = self.t
        if t >= len(self.script["reward"]):
            raise IndexError("script exhausted")
        self.log.append(int(action))
        self.t += 1
        self.c += 1
        self._lives = int(self.script["lives_after"][t])
        return self._obs(), float(self.script["reward"][t]), bool(self.script["terminated"][t]), bool(self.script["truncated"][t]), {"c": self.c, "base_t": t}

    def close(self):
        pass


# the cases of fixture group G12: name -> (script kwargs, needs_fire, agent steps driven)
CASES = {
    "fire_lives": (dict(seed=1201, steps=900, p_life=0.12, p_term=0.004, p_trunc=0.004, qbert_delay=2), True, 240),
    "fire_busy": (dict(seed=1202, steps=1200, p_life=0.25, p_term=0.03, p_trunc=0.03, qbert_delay=0), True, 240),
    # the game ends during the start presses after the first reset (base steps 0..2) and again during the presses that follow a lost life
    "fire_ends_in_presses": (dict(seed=1203, steps=900, p_life=0.2, p_term=0.0, p_trunc=0.0, qbert_delay=1, early_end=(1, 9, 10, 30, 31, 32)), True, 160),
    "nofire": (dict(seed=1204, steps=600, p_life=0.15, p_term=0.01, p_trunc=0.01, qbert_delay=2), False, 240),
}


def actions_for(name: str, n: int):
    g = np.random.Generator(np.random.PCG64(9000 + len(name)))
    return g.integers(0, 4, size=n)


def drive(env, actions):
    """One env under the vector env's autoreset rule (reset as soon as a step reports terminated or truncated): -> dict of per-step arrays.  ``env`` is any chain of
    single-env wrappers over a ScriptedAle; the emulator's action log is read by the caller."""
    out = {k: [] for k in ("obs_c", "obs_t", "reward", "terminated", "truncated", "life_loss", "info_c", "reset_obs_c")}
    obs, info = env.reset()
    out["reset_obs_c"].append(int(obs[0]))
    for a in actions:
        obs, r, te, tr, info = env.step(int(a))
        out["obs_c"].append(int(obs[0])); out["obs_t"].append(int(obs[1]))
        out["reward"].append(float(r)); out["terminated"].append(bool(te)); out["truncated"].append(bool(tr))
        out["life_loss"].append(bool(info.get("life_loss", False))); out["info_c"].append(int(info.get("c", -1)))
        if te or tr:
            obs, info = env.reset()
            out["reset_obs_c"].append(int(obs[0]))
    return {"obs_c": np.array(out["obs_c"], np.int64), "obs_t": np.array(out["obs_t"], np.int64), "reward": np.array(out["reward"], np.float64),
            "terminated": np.array(out["terminated"]), "truncated": np.array(out["truncated"]), "life_loss": np.array(out["life_loss"]),
            "info_c": np.array(out["info_c"], np.int64), "reset_obs_c": np.array(out["reset_obs_c"], np.int64)}
