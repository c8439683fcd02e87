"""GPU (MI355X): the actor -> replay -> learner loop LEARNS.

The reference validates its loop by learning curves and a score table only (README.md:62-112); parity tests pin single links of the loop,
this file pins what they compose to: on the synthetic env's learnable task (``env_task=block``: +1 for naming the quadrant of the bright
block in the newest frame, -1 for the next class; chance 0, optimum +1 per step; oracle/synth_env.c) the reward per step must rise from
chance to the optimum under the reference's own schedule — epsilon from 1 + min_eps down to min_eps over ``exploration_steps``
(trainer.py:46-50), training from ``training_start_steps`` on (trainer.py:83), 20 updates per 80-step rollout, target sync every 500
updates (agent.py:160-161) — for every algorithm family of BASELINE configs[1..4], on the ``main`` and the ``launch`` schedule, with uniform
and prioritized replay; and deliberately broken runs must not.  Curves: profiles/r04_learning.json (``python tests/learning_runs.py``).
"""
import pytest

import learning_runs as LR

pytestmark = pytest.mark.gpu


def check_learned(r):
    optimum = 1.0 - 0.01                            # the default min_eps; a random action earns 0 on average
    first, last = r["curve"][0][1], r["final_reward_per_step"]
    assert first < 0.35, f"{r['name']}: the first window should still be near chance while epsilon is high, got {first}"
    assert last >= 0.9 * optimum, f"{r['name']} ({r['schedule']}): reward per step {last} after {r['frames']} frames, optimum {optimum:.3f}"
    ups = [b[1] - a[1] for a, b in zip(r["curve"], r["curve"][1:])]
    assert min(ups) > -0.05, f"{r['name']}: the curve should not collapse once learned: {r['curve']}"
    # the value estimate bootstraps through the target network: one more reward per sync (every 25 iterations), not stuck at the one-step reward
    assert r["qmax"] > 3.0, f"{r['name']}: max-Q {r['qmax']} — the target network does not seem to propagate value"


@pytest.mark.parametrize("launch", [False, True], ids=["main", "launch"])
@pytest.mark.parametrize("name,algo,extra,frames,env_id", [(n, a, x, min(f, 2_600_000), e) for n, a, x, f, e in LR.FAMILIES], ids=[f[0] for f in LR.FAMILIES])
def test_every_family_learns_the_block_task(name, algo, extra, frames, env_id, launch):
    r = LR.run(algo, extra, frames, launch, env_id=env_id)
    r["name"] = name
    check_learned(r)


@pytest.mark.parametrize("name,algo,extra,env_id", [("dqn", "dqn", {}, "Breakout"), ("c51_rainbow_lite", "c51", LR.RAINBOW, "Breakout"), ("fqf", "fqf", {}, "Asterix")])
def test_the_native_loop_learns_the_block_task(name, algo, extra, env_id, monkeypatch):
    """The same criterion with the loop issued by the library's own handles (agent0_amd/deepq/native_loop.py: the production default for these configurations)."""
    monkeypatch.setenv("A0_NATIVE_LOOP", "1")
    r = LR.run(algo, extra, 2_600_000, env_id=env_id)
    r["name"] = name
    assert r["host_loop"] == "native handles"
    check_learned(r)


def test_prioritized_replay_learns():
    for extra in ({"replay.policy": "prioritize"}, {"replay.policy": "prioritize", "replay.sumtree": "false"}):
        r = LR.run("dqn", extra, 2_600_000)
        r["name"] = f"dqn {extra}"
        check_learned(r)


@pytest.mark.parametrize("sabotage", ["eps_one", "lr_zero", "no_target_sync"])
def test_broken_loops_do_not_pass(sabotage):
    """The same run with one link cut must fail the criterion above: an actor that never exploits, a learner whose optimizer does not move
    the weights, and a target network that is never refreshed (the task is a contextual bandit — the policy still forms — but the value
    estimate stays at the one-step reward instead of growing by ~one reward per sync)."""
    r = LR.run("dqn", {}, 2_600_000, sabotage=sabotage)
    r["name"] = f"dqn_{sabotage}"
    with pytest.raises(AssertionError):
        check_learned(r)
    if sabotage == "no_target_sync":
        assert r["qmax"] < 2.0 and r["final_reward_per_step"] > 0.9
    else:
        assert abs(r["final_reward_per_step"]) < 0.2
