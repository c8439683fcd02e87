"""CPU: the host-environment front-end's own logic — the single-env wrappers (life-loss flag, FIRE start sequence;
semantics of the reference's atari_wrappers.py:20-51) and the vectorisation layer (autoreset, final_info, episode statistics over unclipped
rewards, sign clipping; atari_wrappers.py:61-68 + gymnasium 0.28 vector semantics) — against a scripted fake ALE.  The pool itself (shared
page-locked ring, DMA) needs a GPU: tests/test_gpu_trainer.py::test_host_env_pool_matches_device_env."""
import numpy as np

from agent0_amd.common.atari_wrappers import FireOnReset, LifeLossInfo
from agent0_amd.common.env_pool import VectorizedSingles


class FakeAle:
    """A tiny deterministic 'Atari': obs = step counter; a life is lost every ``life_every`` steps, the game ends when lives reach 0;
    rewards are 0 / +7 / -3 by step; the game does not start (no reward, counter frozen) until FIRE (action 1) has been pressed."""

    def __init__(self, lives=3, life_every=4, needs_fire=True):
        self.lives0, self.life_every, self.needs_fire = lives, life_every, needs_fire
        self.unwrapped = self
        self.ale = self
        self.log = []

    def lives(self):
        return self._lives

    def get_action_meanings(self):
        return ["NOOP", "FIRE" if self.needs_fire else "UP", "RIGHT", "LEFT"]

    def reset(self, **kw):
        self._lives, self.t, self.started = self.lives0, 0, not self.needs_fire
        self.log.append("reset")
        return np.full((2, 2), 0, np.uint8), {}

    def step(self, a):
        self.log.append(int(a))
        if a == 1:
            self.started = True
        if not self.started:
            return np.full((2, 2), self.t, np.uint8), 0.0, False, False, {}
        self.t += 1
        r = 7.0 if self.t % 3 == 0 else (-3.0 if self.t % 5 == 0 else 0.0)
        if self.t % self.life_every == 0:
            self._lives -= 1
            self.started = not self.needs_fire       # the game waits for FIRE again after a lost life
        return np.full((2, 2), self.t, np.uint8), r, self._lives == 0, False, {"t": self.t}


def test_fire_on_reset_presses_the_start_sequence():
    env = FakeAle()
    obs, info = FireOnReset(env).reset()
    assert env.log == ["reset", 0, 1, 2] and env.started and int(obs[0, 0]) == 2      # NOOP (frozen), FIRE (t=1), action 2 (t=2)


def test_life_loss_is_reported_not_terminal_and_fire_restarts():
    env = FakeAle(lives=3, life_every=4)
    w = LifeLossInfo(env)
    w.reset()
    env.step(1)                                     # start the game: t = 1
    flags, terms = [], []
    for _ in range(9):
        obs, r, term, trunc, info = w.step(3)
        flags.append(bool(info["life_loss"])); terms.append(bool(term))
        if term:
            break
    # t = 2, 3, 4 (life 3 -> 2: reported, FIRE sequence pressed -> game runs again), ...; the LAST life ends the episode and is NOT a life_loss
    assert flags[2] is True and terms[2] is False
    assert env.log.count(1) >= 2, "FIRE pressed again after the lost life"
    assert terms[-1] is True and flags[-1] is False, "lives hitting 0 is terminal, not a life loss (old > new > 0, atari_wrappers.py:42)"
    # a game without FIRE is left alone
    env2 = FakeAle(lives=2, life_every=2, needs_fire=False)
    w2 = LifeLossInfo(env2)
    w2.reset()
    n0 = len(env2.log)
    w2.step(3); _, _, _, _, info = w2.step(3)
    assert info["life_loss"] is True and len(env2.log) == n0 + 2


def test_vectorised_singles_autoreset_statistics_and_clipping():
    envs = [FakeAle(lives=1, life_every=6, needs_fire=False), FakeAle(lives=1, life_every=4, needs_fire=False)]
    v = VectorizedSingles(envs)
    obs, _ = v.reset()
    assert obs.shape == (2, 2, 2)
    raw = [0.0, 0.0]
    for step in range(1, 7):
        obs, rew, term, trunc, info = v.step(np.array([3, 3]))
        for i, e in enumerate(envs):
            pass
        assert set(np.unique(rew)) <= {-1.0, 0.0, 1.0}, "sign-clipped rewards out (ClipRewardEnv is the outermost wrapper)"
        if step == 4:          # env 1 (life_every 4, one life) ends: final_info carries the UNCLIPPED return 7 (t=3), obs is the reset one
            assert term.tolist() == [False, True] and info["_final_info"].tolist() == [False, True]
            assert float(info["final_info"][1]["episode"]["r"][0]) == 7.0 and info["final_info"][0] is None
            assert int(obs[1, 0, 0]) == 0 and int(obs[0, 0, 0]) == 4, "autoreset: the first observation of the next episode is returned"
        if step == 6:          # env 0 ends at t = 6: rewards 7 (t=3) - 3 (t=5) + 7 (t=6) = 11
            assert term.tolist() == [True, False] and float(info["final_info"][0]["episode"]["r"][0]) == 11.0
    assert "life_loss" in info and info["life_loss"].dtype == bool


def test_atari_slice_builds_the_reference_pipeline_from_gymnasium(monkeypatch):
    """``AtariSlice`` (what ``make_atari`` hands to the env pool when gymnasium + ale-py are installed): per env
    ``gym.make("<Game>NoFrameskip-v4")`` -> AtariPreprocessing(terminal_on_life_loss=False) -> FrameStack(4) (the library wrappers of
    atari_wrappers.py:61-63) -> LifeLossInfo (only with episode_life) -> FireOnReset, vectorised with statistics and clipping.  gymnasium and
    ale-py are not in this image, so stand-in modules record how they are called."""
    import sys
    import types

    from agent0_amd.common.host_envs import AtariSlice

    calls = []

    class Pre:
        def __init__(self, env, terminal_on_life_loss=None):
            calls.append(("AtariPreprocessing", terminal_on_life_loss))
            self.env, self.unwrapped = env, env

        def __getattr__(self, n):
            return getattr(self.env, n)

    class Stack(Pre):
        def __init__(self, env, k):
            calls.append(("FrameStack", k))
            self.env, self.unwrapped = env, env.unwrapped

    gym = types.ModuleType("gymnasium")
    gym.make = lambda name: (calls.append(("make", name)), FakeAle(lives=2, life_every=3))[1]
    wr = types.ModuleType("gymnasium.wrappers")
    wr.AtariPreprocessing, wr.FrameStack = Pre, Stack
    gym.wrappers = wr
    for name, mod in (("gymnasium", gym), ("gymnasium.wrappers", wr), ("ale_py", types.ModuleType("ale_py"))):
        monkeypatch.setitem(sys.modules, name, mod)
    from agent0_amd.common.atari_wrappers import real_atari_available
    assert real_atari_available()
    vec = AtariSlice("Breakout", episode_life=True)(4, 2)              # envs 4 and 5 of the vector env
    assert [c for c in calls if c[0] == "make"] == [("make", "BreakoutNoFrameskip-v4")] * 2
    assert ("AtariPreprocessing", False) in calls and ("FrameStack", 4) in calls
    assert type(vec).__name__ == "VectorizedSingles" and len(vec.envs) == 2
    assert type(vec.envs[0]).__name__ == "FireOnReset" and type(vec.envs[0].env).__name__ == "LifeLossInfo"
    obs, _ = vec.reset()
    assert obs.shape[0] == 2
    seen_life = False
    for _ in range(8):
        obs, rew, term, trunc, info = vec.step(np.array([3, 3]))
        seen_life |= bool(info["life_loss"].any())
    assert seen_life and set(np.unique(rew)) <= {-1.0, 0.0, 1.0}
    no_life = AtariSlice("Breakout", episode_life=False)(0, 1)
    assert type(no_life.envs[0].env).__name__ == "Stack", "without episode_life the life-loss wrapper is left out (atari_wrappers.py:64)"


def test_record_marks_envs_whose_stack_merely_advanced():
    """Device frame-stack mode of the shared block: the newest frame of every env is written to its own array and scalar row 6 is 1 exactly
    for envs whose older frames equal the previous observation's newer ones (byte test — independent of the wrapper order)."""
    import host_slices
    from agent0_amd.common.host_envs import N_SCAL, block_layout, block_views, record
    E, ob, fb = 5, 4 * 84 * 84, 84 * 84
    total = block_layout(E, ob, 1, fb)[-1]
    buf = block_views(bytearray(total), E, ob, 1, fb)
    assert buf["scal"].shape == (2, N_SCAL, E) and buf["new"].shape == (2, E, fb)
    env = host_slices.ScriptedStack(5, 0, E)
    obs, _ = env.reset()
    buf["obs"][0] = obs.reshape(E, -1)                  # what a worker writes on CMD_RESET (sequence number 0 -> half 0)
    for t in range(1, 9):
        half = t & 1
        out = env.step(None)
        record(buf, half, 0, E, *out)
        assert np.array_equal(buf["obs"][half].reshape(E, 4, 84, 84), out[0])
        assert np.array_equal(buf["new"][half].reshape(E, 84, 84), out[0][:, 3])
        want = [0.0 if host_slices.ScriptedStack.mode(e, t) in (0, 3) else 1.0 for e in range(E)]
        assert buf["scal"][half, 6].tolist() == want
    # without the newest-frame array the flag is never set (whole stacks are uploaded)
    plain = block_views(bytearray(block_layout(E, ob, 1)[-1]), E, ob, 1)
    record(plain, 1, 0, E, *env.step(None))
    assert plain["new"].shape[2] == 0 and not plain["scal"][1, 6].any()


def test_game_over_during_the_post_life_loss_presses_is_left_to_the_next_step():
    """atari_wrappers.py:52-55: after a lost life the reference presses actions 0, 1, 2, ignores their termination flags and returns the LAST
    press's observation and info — it does not reset there."""
    class EndsDuringPresses(FakeAle):
        def step(self, a):
            out = super().step(a)
            if self.t == 5:                           # the game "ends" on a press (t: 4 = lost life, then presses NOOP (frozen), FIRE t=5, 2 t=6)
                return out[0], out[1], True, False, {"t": self.t, "over": True}
            return out
    env = EndsDuringPresses(lives=3, life_every=4)
    w = LifeLossInfo(env)
    w.reset(); env.step(1)
    for _ in range(2):
        w.step(3)
    n_resets = env.log.count("reset")
    obs, r, term, trunc, info = w.step(3)             # t = 4: life lost, three presses follow
    assert info["life_loss"] is True and env.log[-3:] == [0, 1, 2]
    assert env.log.count("reset") == n_resets, "no reset inside the wrapper"
    assert int(obs[0, 0]) == 6 and info["t"] == 6, "observation and info of the LAST press"
    assert term is False, "termination is the lost-life step's own flag; the presses' flags are ignored"


def test_vectorised_singles_seed_every_env_on_the_first_reset():
    class Rec(FakeAle):
        def reset(self, **kw):
            self.seeds = getattr(self, "seeds", []) + [kw.get("seed")]
            return super().reset()
    envs = [Rec(needs_fire=False) for _ in range(3)]
    v = VectorizedSingles(envs, seeds=[1042, 1043, 1044])
    v.reset(); v.reset()
    assert [e.seeds for e in envs] == [[1042, None], [1043, None], [1044, None]], "own streams on the first reset, continued afterwards"
    v.reset(seed=7)
    assert [e.seeds[-1] for e in envs] == [7, 8, 9], "an int seed spreads as seed + i (gymnasium's vector convention)"
    from agent0_amd.common.host_envs import AtariSlice
    import inspect
    src = inspect.getsource(AtariSlice.__call__)
    assert "seeds=" in src and "self.seed + e0 + i" in src


def test_worker_handshake_is_one_word():
    """command and sequence number travel in ONE 8-byte word, so a worker cannot pair a new sequence number with the previous
    command (a step after a reset being executed as a second reset).  worker_main runs on a thread against a plain bytearray block."""
    import threading
    import time
    import host_slices
    from agent0_amd.common.host_envs import (CMD_CLOSE, CMD_RESET, CMD_STEP, CTL_ARG, CTL_DONE0, CTL_WORD, block_layout, block_views, ctl_word,
                                             worker_main)
    from multiprocessing import shared_memory
    assert ctl_word(CMD_STEP, 5) >> 56 == CMD_STEP and ctl_word(CMD_STEP, 5) & ((1 << 56) - 1) == 5
    E, ob = 3, 4 * 84 * 84
    total = block_layout(E, ob, 1)[-1]
    shm = shared_memory.SharedMemory(create=True, size=total)
    try:
        buf = block_views(shm.buf, E, ob, 1)
        for v in buf.values():
            v[...] = 0
        resets = []

        class Slice(host_slices.ScriptedStack):
            def reset(self, **kw):
                resets.append(kw)
                return super().reset()
        th = threading.Thread(target=worker_main, args=(0, lambda lo, k: Slice(5, lo, k), 0, E, shm.name, E, ob, 1, 20.0), daemon=True)
        th.start()

        def post(cmd, seq, arg=-1):
            buf["ctl"][CTL_ARG] = arg
            buf["ctl"][CTL_WORD] = ctl_word(cmd, seq)
            t0 = time.time()
            while int(buf["ctl"][CTL_DONE0]) != seq:
                assert time.time() - t0 < 20
                time.sleep(1e-4)
        post(CMD_RESET, 1, arg=1234)
        assert resets == [{"seed": 1234}], "the reset's seed reaches the worker (slice offset 0)"
        twin = host_slices.ScriptedStack(5, 0, E)
        twin.reset()
        for seq in (2, 3, 4):
            post(CMD_STEP, seq)              # the stale CMD_RESET is gone with the old word: these are steps
            assert np.array_equal(buf["obs"][seq & 1].reshape(E, 4, 84, 84), twin.step(None)[0])
        assert len(resets) == 1
        buf["ctl"][CTL_WORD] = ctl_word(CMD_CLOSE, 5)
        th.join(10)
        assert not th.is_alive() and int(buf["ctl"][CTL_DONE0]) == -1
        del buf
    finally:
        shm.close(); shm.unlink()


def test_worker_accepts_only_the_next_sequence_number():
    """A step word is an 8-byte device-to-host DMA copy that may land in pieces.  accept_word takes a word only if it carries seen + 1: every
    mixture of old and new bytes of the sequence number is either rejected or already the complete new value; and because HostEnvPool.reset
    re-writes the acknowledged word as CMD_STEP from the host, the command byte never changes under a DMA write (the hazard: the next number
    beside a stale CMD_RESET, i.e. a second reset instead of the first step)."""
    import inspect
    import itertools
    from agent0_amd.common import env_pool
    from agent0_amd.common.host_envs import CMD_RESET, CMD_STEP, accept_word, ctl_word
    for seen in (0, 1, 254, 255, 256, 65535, (1 << 32) - 1, (1 << 40) + 255):
        old, new = ctl_word(CMD_STEP, seen), ctl_word(CMD_STEP, seen + 1)
        ob, nb = old.to_bytes(8, "little"), new.to_bytes(8, "little")
        for pick in itertools.product((0, 1), repeat=8):                      # every byte-wise mixture of the old and the new word
            w = int.from_bytes(bytes(nb[i] if pick[i] else ob[i] for i in range(8)), "little")
            got = accept_word(w, seen)
            assert got is None or got == (CMD_STEP, seen + 1), (seen, pick, got)
        assert accept_word(new, seen) == (CMD_STEP, seen + 1)
        assert accept_word(old, seen) is None and accept_word(ctl_word(CMD_STEP, seen + 2), seen) is None     # replayed / skipped numbers are not commands
    assert accept_word(ctl_word(CMD_RESET, 8), 7) == (CMD_RESET, 8)
    src = inspect.getsource(env_pool.HostEnvPool.reset)
    assert src.index("self._wait_workers()") < src.index('self._np["ctl"][CTL_WORD] = ctl_word(CMD_STEP, self.seq)'), "reset() re-arms the word as CMD_STEP after the acknowledgement"


# ------------------------------------------------------------------------------------------------ fixture group G12: the reference's own wrappers
import pytest

import fake_ale
from util import golden


def _product_chain(name):
    kw, needs_fire, n = fake_ale.CASES[name]
    ale = fake_ale.ScriptedAle(fake_ale.make_script(**kw), needs_fire=needs_fire)
    env = LifeLossInfo(ale)
    if needs_fire:
        env = FireOnReset(env)
    return ale, env, n


@pytest.mark.parametrize("name", list(fake_ale.CASES))
def test_g12_single_env_wrappers_equal_the_reference(name):
    """FireOnReset / LifeLossInfo against what the reference's FireResetEnv / EpisodicLifeEnv (atari_wrappers.py:20-56) returned on the same scripted emulator
    (tests/golden/gen_golden.py g12): every emulator call (action log, resets included), every returned observation, the life_loss flags — including the frames
    with lives == 0 before the game ends, games that end during the start presses (reset again after a reset, ignored after a lost life) and games without FIRE."""
    g = golden(f"g12_wrappers_{name}")
    ale, env, n = _product_chain(name)
    got = fake_ale.drive(env, g["actions"])
    assert np.array_equal(np.array(ale.log), g["emulator_log"]), "the same emulator calls in the same order"
    for k in ("obs_c", "obs_t", "terminated", "truncated", "life_loss", "info_c", "reset_obs_c"):
        assert np.array_equal(got[k], g[k]), k
    assert np.array_equal(got["reward"], g["raw_reward"]), "rewards reach the vectorisation layer unclipped (statistics first, clipping last: atari_wrappers.py:66-67)"
    assert np.array_equal(np.sign(got["reward"]), g["reward"]), "ClipRewardEnv: the sign (atari_wrappers.py:15-17)"


@pytest.mark.parametrize("name", list(fake_ale.CASES))
def test_g12_vectorised_singles_equal_the_reference(name):
    """The vectorisation layer over the product chain (autoreset, sign clipping, life_loss vector, final_info with the UNCLIPPED episode return) against the same
    fixture: two copies of the case side by side, the second one step behind, so that the per-env bookkeeping cannot leak between envs."""
    g = golden(f"g12_wrappers_{name}")
    (ale0, e0, n), (ale1, e1, _) = _product_chain(name), _product_chain(name)
    v = VectorizedSingles([e0, e1])
    obs, _ = v.reset()
    assert obs[:, 0].tolist() == [int(g["reset_obs_c"][0])] * 2
    acts = g["actions"]
    resets = [1, 1]                                         # index of the next reset observation per env
    ret = [0.0, 0.0]
    for t in range(n + 1):
        a = np.array([acts[t] if t < n else 0, acts[t - 1] if t >= 1 else 0])
        if t == 0:                                          # env 1 starts one step later: step env 0 alone through a one-env view
            o, r, te, tr, info = VectorizedSingles.step(_OneOf(v, 0), a[:1])
            idx = [(0, 0)]
        elif t == n:
            o, r, te, tr, info = VectorizedSingles.step(_OneOf(v, 1), a[1:])
            idx = [(1, 0)]
        else:
            o, r, te, tr, info = v.step(a)
            idx = [(0, 0), (1, 1)]
        for env_i, col in idx:
            s = t if env_i == 0 else t - 1
            assert float(r[col]) == float(g["reward"][s]) and bool(te[col]) == bool(g["terminated"][s]) and bool(tr[col]) == bool(g["truncated"][s]), (t, env_i)
            assert bool(info["life_loss"][col]) == bool(g["life_loss"][s])
            ret[env_i] += float(g["raw_reward"][s])
            if g["terminated"][s] or g["truncated"][s]:
                assert int(o[col, 0]) == int(g["reset_obs_c"][resets[env_i]]), "autoreset: the next episode's first observation (after the start presses) is returned"
                resets[env_i] += 1
                assert bool(info["_final_info"][col]) and float(info["final_info"][col]["episode"]["r"][0]) == np.float32(ret[env_i]), "episode return over the unclipped rewards"
                ret[env_i] = 0.0
            else:
                assert int(o[col, 0]) == int(g["obs_c"][s])
                assert "_final_info" not in info or not bool(info["_final_info"][col])
    assert np.array_equal(np.array(ale0.log), g["emulator_log"]) and np.array_equal(np.array(ale1.log), g["emulator_log"])


class _OneOf:
    """A one-env view of a VectorizedSingles (shares its envs and return accumulators)."""

    def __init__(self, v, i):
        self.envs, self.clip, self.returns = v.envs[i:i + 1], v.clip, v.returns[i:i + 1]
