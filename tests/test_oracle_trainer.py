"""CPU: oracle.trainer.OracleTrainer (the composed reference loop, trainer.py:74-119,171-184 and launch.py:30-63) — structural properties
of the composition itself; tests/test_gpu_trace.py walks the product's Trainer through it link by link on the GPU."""
import numpy as np
import pytest
import torch

import recipe
from oracle.trainer import OracleTrainer


def _make(policy="uniform", sumtree=False, launch=False, n_step=3, algo="dqn", **kw):
    spec = recipe.SPECS[algo]
    args = dict(num_envs=4, sample_steps=5, batch_size=8, replay_size=50, learner_steps=2, training_start_steps=25, policy=policy, sumtree=sumtree, n_step=n_step,
                target_update_freq=3, total_steps=400, exploration_steps=40, launch=launch)
    args.update(kw)
    return OracleTrainer(spec, recipe.make_state_dict(spec, 11), **args)


@pytest.mark.parametrize("policy,sumtree", [("uniform", False), ("prioritize", False), ("prioritize", True)])
def test_loop_bookkeeping(policy, sumtree):
    tr = _make(policy, sumtree)
    n_upd = 0
    for it in range(6):
        res = tr.iteration()
        assert res["frames"] == (it + 1) * 20 and len(tr.replay) == min((it + 1) * 20, 50)
        # training starts once len(replay) > training_start_steps (trainer.py:81): 20 -> no, 40 -> yes
        assert len(res["records"]) == (2 if it >= 1 else 0)
        for rec in res["records"]:
            n_upd += 1
            assert rec.idx.shape == (8,) and rec.slot.min() >= 0 and rec.slot.max() < 50 and rec.q_loss.shape == (8,)
            assert all(tr.replay.slots[s] is not None for s in rec.slot)
            if policy == "prioritize":
                assert rec.weights.max() <= 1.0 and rec.weights.min() > 0
            else:
                assert np.array_equal(rec.weights, np.ones(8, np.float32))
    assert tr.learner.update_steps == n_upd == 10
    if policy == "prioritize":
        # beta advanced once per extend by the number of transitions (replay.py:53, utils.py:25-28: value BEFORE the increment)
        assert tr.replay.beta == pytest.approx(0.4 + 0.6 * (5 * 20) / 400)
        if sumtree:
            t = tr.replay.tree
            p = np.arange(1, t.cap2)
            assert np.array_equal(t.tree[p], t.tree[2 * p] + t.tree[2 * p + 1]) and (t.leaves() > 0).all()


def test_uniform_fetcher_never_returns_its_last_batch():
    tr = _make("uniform", learner_steps=7, training_start_steps=0, replay_size=1000, num_envs=5, sample_steps=10, batch_size=8)   # 50 transitions: 7 batches, 6 returned
    res = tr.iteration()
    seen = np.concatenate([r.idx for r in res["records"][:6]])
    assert len(np.unique(seen)) == 48, "six distinct batches of one permutation of range(50)"
    assert tr.fetcher["pos"] == 1 and tr.fetcher["top"] == 50, "the seventh update opened a new fetcher (StopIteration, trainer.py:84-87)"


def test_launch_schedule_acts_with_stale_weights():
    """launch.py:47-62: rollout k+1 is issued before update block k, i.e. with the weights after block k-1 and epsilon(frame_count before
    step k).  The main schedule rolls out after the block.  Same seeds: the two schedules agree until the first update and then differ."""
    a, b = _make(launch=False, n_step=1), _make(launch=True, n_step=1)
    qa, qb = [], []
    for it in range(4):
        ra, rb = a.iteration(), b.iteration()
        qa.append(list(a.Qs)); qb.append(list(b.Qs))
    # iterations 0, 1: no update has happened before their rollouts in either schedule
    assert qa[1] == qb[1]
    # iteration 2's rollout: main has the weights after block 1, launch still the initial ones -> its max-Q equals a frozen-weights actor's
    assert qa[2] != qb[2]
    frozen = _make(launch=False, n_step=1, training_start_steps=10 ** 9)
    for it in range(3):
        frozen.iteration()
    assert frozen.Qs[:15] == b.Qs[:15], "launch-mode rollouts 0..2 act with the initial weights"
    assert frozen.Qs[15:] != b.Qs[15:] if len(b.Qs) > 15 else True
