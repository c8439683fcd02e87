"""GPU (MI355X): the actor -> replay -> learner loop LEARNS — on a task that needs temporal credit.

The reference validates its loop by learning curves and a score table only (README.md:62-112); parity tests pin single links of the loop, this file pins what
they compose to.  Two learnable tasks of the synthetic env (oracle/synth_env.c, csrc/synth_env.h):

  * ``env_task=chase`` (round 5): the ACTION MOVES the bright block on a 4 x 4 lattice and the only reward is +1 on arrival at the target cell, three to six moves
    after the respawn — no single move pays by itself, so the policy exists only if the bootstrap term r + gamma^n (1 - d) max Q_target(s') (agent.py:64-73,183-190)
    carries value backwards through the target network.  Chance ~0.02 per step, optimum 0.25.  Every algorithm family of BASELINE configs[1..4] must reach 0.9 of
    the optimum under the reference's own schedule — epsilon from 1 + min_eps to min_eps over ``exploration_steps`` (trainer.py:46-50), training from
    ``training_start_steps`` (trainer.py:83), 20 updates per 80-step rollout, target sync every 500 updates (agent.py:160-161) — on the ``main`` AND the ``launch``
    schedule, with uniform, flat-priority and sum-tree replay; and runs with ONE link of the credit path cut must NOT: gamma = 0, the n-step window mis-indexed by
    one step, the bootstrap taken from s instead of s', a target network that is never refreshed — none of which the bandit task below can see — plus an actor
    that never exploits and an optimizer that does not move.
  * ``env_task=block`` (round 4; a contextual bandit: +1 for naming the block's quadrant): kept as a second, independent task (one run).

Curves: profiles/r06_learning.json (``python tests/learning_runs.py``).
"""
import pytest

import learning_runs as LR

pytestmark = pytest.mark.gpu

CHASE_OPTIMUM = 0.25 * (1.0 - 0.01 * 0.75)          # one reward per 4.0 steps (the mean spawn distance); min_eps = 0.01 of the moves are random, a quarter of them right anyway
# the launch schedule runs the quantile networks with 16 fractions instead of 64 / 32 (a quarter of the update's rows): same code paths, a third of the wall-clock
SMALL_Q = {"learner.iqn.K": 16, "learner.iqn.N": 16, "learner.iqn.N_dash": 16, "learner.iqn.F": 16}
# (round 6: 4.7 M frames for the dqn runs, one 25-iteration window more than round 5's 4.2 M.  The curves are chaotic in the low bits — the knee between 3 M and 4.2 M frames
# moves by a window with anything that changes a rounding: the six-product kernels, even the BLAS thread count behind the orthogonal initialisation — and at 4.2 M one of
# eleven runs ended its last window at 0.221, under the 0.223 bar, on its way to the same 0.247 plateau: gpurun_out/r06/learn_*.log)
FAMILIES = [("dqn", "dqn", {}, 4_700_000, "Breakout"), ("c51_rainbow_lite", "c51", LR.RAINBOW, 2_600_000, "Breakout"), ("iqn", "iqn", {}, 3_800_000, "Asterix"),
            ("fqf", "fqf", {}, 3_800_000, "Asterix")]


def check_chase(r):
    first, last = r["curve"][0][1], r["final_reward_per_step"]
    assert first < 0.08, f"{r['name']}: the first window should still be near chance (0.02) while epsilon is high, got {first}"
    assert last >= 0.9 * CHASE_OPTIMUM, f"{r['name']} ({r['schedule']}): reward per step {last} after {r['frames']} frames, optimum {CHASE_OPTIMUM:.4f}"
    peak = max(c[1] for c in r["curve"])
    assert last >= peak - 0.02, f"{r['name']}: the curve should not collapse once learned: {r['curve']}"
    # more than one reward's worth of value: the estimate bootstraps across respawns through the target network
    assert r["qmax"] > 1.2, f"{r['name']}: max-Q {r['qmax']} — the target network does not seem to propagate value"


def check_block(r):
    optimum = 1.0 - 0.01                            # the default min_eps; a random action earns 0 on average
    first, last = r["curve"][0][1], r["final_reward_per_step"]
    assert first < 0.35, f"{r['name']}: the first window should still be near chance while epsilon is high, got {first}"
    assert last >= 0.9 * optimum, f"{r['name']} ({r['schedule']}): reward per step {last} after {r['frames']} frames, optimum {optimum:.3f}"
    ups = [b[1] - a[1] for a, b in zip(r["curve"], r["curve"][1:])]
    assert min(ups) > -0.05, f"{r['name']}: the curve should not collapse once learned: {r['curve']}"
    assert r["qmax"] > 3.0, f"{r['name']}: max-Q {r['qmax']} — the target network does not seem to propagate value"


@pytest.mark.parametrize("launch", [False, True], ids=["main", "launch"])
@pytest.mark.parametrize("name,algo,extra,frames,env_id", FAMILIES, ids=[f[0] for f in FAMILIES])
def test_every_family_learns_the_chase_task(name, algo, extra, frames, env_id, launch):
    if launch and algo in ("iqn", "fqf"):
        extra = {**extra, **SMALL_Q}
    r = LR.run(algo, extra, frames, launch, env_id=env_id, task="chase")
    r["name"] = name
    assert r["host_loop"] == "python classes"
    check_chase(r)


@pytest.mark.parametrize("extra,frames", [({"replay.policy": "prioritize"}, 4_700_000), ({"replay.policy": "prioritize", "replay.sumtree": "false"}, 5_700_000)],
                         ids=["sum-tree", "flat-priority-vector"])
def test_prioritized_replay_learns_the_chase_task(extra, frames):
    """(the reference-faithful flat priority vector — uniform sampling, importance weights from priorities that quirk Q1 leaves misaligned — reaches the optimum later
    and less steadily: 0.246 at 3.6 M frames, 0.216 at 4.6 M, 0.247 from 5.1 M on)"""
    r = LR.run("dqn", extra, frames, task="chase")
    r["name"] = f"dqn {extra}"
    check_chase(r)


@pytest.mark.parametrize("sabotage", ["discount_zero", "nstep_shift", "stale_next_state", "no_target_sync", "eps_one", "lr_zero"])
def test_runs_with_a_cut_credit_path_do_not_pass(sabotage):
    """The same dqn run with one link cut must fail the criterion.  The first four are breakages of the TEMPORAL credit path, which a bandit task cannot see (on the block
    task a never-refreshed target still yields the optimal policy): no bootstrap term (gamma = 0), every transition labelled with the NEXT step's action, the bootstrap
    taken from st instead of st_next, a target network that never changes.  Measured: 0.000 / 0.004 / 0.018 / 0.013 reward per step against the honest run's 0.247."""
    r = LR.run("dqn", {}, 4_200_000, sabotage=sabotage, task="chase")
    r["name"] = f"dqn_{sabotage}"
    with pytest.raises(AssertionError):
        check_chase(r)
    assert r["final_reward_per_step"] < 0.3 * CHASE_OPTIMUM


@pytest.mark.parametrize("name,algo,extra,frames,launch", [("dqn", "dqn", {}, 4_700_000, False), ("c51_rainbow_lite", "c51", LR.RAINBOW, 2_600_000, False), ("dqn", "dqn", {}, 4_700_000, True)],
                         ids=["dqn-main", "rainbow-lite-main", "dqn-launch"])
def test_the_native_loop_learns_the_chase_task(name, algo, extra, frames, launch, monkeypatch):
    """The same criterion with the loop issued by the library's own handles (agent0_amd/deepq/native_loop.py: the production default for these configurations) — under
    the chase task the merged tail + env-step kernels hold their frame waves at a workgroup barrier until wave 0 has chosen the action."""
    monkeypatch.setenv("A0_NATIVE_LOOP", "1")
    r = LR.run(algo, extra, frames, launch, task="chase")
    r["name"] = name
    assert r["host_loop"] == "native handles"
    check_chase(r)


def test_the_native_loop_learns_the_block_task(monkeypatch):
    """The bandit task of round 4 (+1 for naming the block's quadrant) stays as a second, independent learnable task."""
    monkeypatch.setenv("A0_NATIVE_LOOP", "1")
    r = LR.run("dqn", {}, 2_600_000)
    r["name"] = "dqn"
    assert r["host_loop"] == "native handles"
    check_block(r)
