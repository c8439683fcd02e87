"""Learning runs on the synthetic env's learnable tasks (``env_task=chase`` — temporal credit — and ``env_task=block`` — a bandit; oracle/synth_env.c) — TEST INFRASTRUCTURE.

The reference's only acceptance evidence is learning curves and a score table (README.md:62-112, imgs/*.png): its loop — epsilon schedule
(trainer.py:46-50), replay, target sync every 500 updates (agent.py:160-161), priorities (trainer.py:103-104) — is validated by the fact
that returns rise.  ``run`` drives the PRODUCT's ``Trainer`` (``main`` schedule, or ``launch`` with ``use_lp=True``) on the block task and
turns the episode returns it reports into REWARD PER STEP: episodes end independently of the actions (terminal = a Philox draw of (env,
step)), so the oracle's terminal map gives every reported episode its length, in the order the actor reports them (step-major, env-major:
agent.py:85-88).  Chance level is 0, the optimum +1 per step (times 1 - eps * (1 - 1/A) under epsilon-greedy).

``python tests/learning_runs.py [out.json]`` runs the set the GPU tests assert on and writes the curves (profiles/r06_learning.json).
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def make_cfg(algo: str, extra=None, E: int = 256, env_id: str = "Breakout", task: str = "block"):
    from agent0_amd.deepq.config import parse_overrides
    kv = {"learner.algo": algo, "actor.num_envs": E, "env_id": env_id, "env_task": task, "wandb": "false", "tb": "false",
          "logdir": "gpurun_out/learning_logs", "replay.size": 200000}
    kv.update(extra or {})
    return parse_overrides([f"{k}={v}" for k, v in kv.items()])


def episode_lengths(seed: int, rank: int, E: int, steps: int) -> np.ndarray:
    """Lengths of the episodes that finish within ``steps`` env steps, in the actor's reporting order (step-major, env-major)."""
    from oracle import core
    term = core.env_terminals(seed, rank, E, steps)
    last = np.zeros(E, dtype=np.int64)
    out = []
    for g, e in zip(*np.nonzero(term)):          # row-major: by step, then env
        out.append((g + 1) - last[e])
        last[e] = g + 1
    return np.asarray(out, dtype=np.int64)


def run(algo: str, extra=None, frames: int = 4_000_000, launch: bool = False, E: int = 256, env_id: str = "Breakout", window: int = 25, sabotage=None, task: str = "block"):
    """Trains for ``frames`` agent steps; returns {"curve": [(frames, reward per step over the episodes that finished in the last ``window`` iterations)], ...}.
    ``task``: "block" (a contextual bandit: optimum +1 per step) or "chase" (temporal credit: the action moves the block, +1 on arrival three to six moves later;
    optimum 0.25 per step).  ``sabotage``: "no_target_sync" (target network never refreshed), "eps_one" (the actor never exploits), "lr_zero" (Adam does not move the
    weights); and, for the chase task, the breakages only a task with temporal credit can see — "discount_zero" (no bootstrap term: gamma = 0), "nstep_shift" (the
    n-step window mis-indexed by one: every transition of a rollout carries the action of the NEXT step, written over the ring right after ``replay.extend``) and
    "stale_next_state" (st_next := st in every stored row: the target bootstraps from the state the action was taken in) — runs that must NOT pass the criterion."""
    from agent0_amd.deepq.trainer import Trainer
    extra = dict(extra or {})
    if sabotage == "discount_zero":
        extra["learner.discount"] = 1e-9              # (the config requires a positive discount)
    if sabotage == "no_target_sync":
        extra["learner.target_update_freq"] = 10 ** 9
    if sabotage == "eps_one":
        extra["actor.min_eps"] = 1.0
    if sabotage == "lr_zero":
        extra["learner.learning_rate"] = 0.0
    cfg = make_cfg(algo, extra, E, env_id, task)
    tr = Trainer(cfg, use_lp=launch)
    T = int(cfg.actor.sample_steps)
    if sabotage in ("nstep_shift", "stale_next_state"):
        rp = tr.replay
        extend = rp.extend

        def tampered(block):
            start = rp.write_cursor()                 # the rollout's rows [start, start + T E) (mod the ring), step-major / env-major
            extend(block)
            idx = (start + np.arange(T * E)) % rp.size
            rows = __import__("torch").as_tensor(idx, device=rp.act.device)
            if sabotage == "nstep_shift":
                rp.act[rows[: (T - 1) * E]] = rp.act[rows[E:]]
            else:
                fr = rp.frames.view(rp.size, 2, -1)
                fr[rows, 1] = fr[rows, 0]
        rp.extend = tampered
    iters = frames // (T * E)
    lengths = episode_lengths(cfg.seed, 0, E, (iters + 2) * T)
    curve, marks = [], [0]
    tic = time.time()
    for it in range(iters):
        res = tr.run_iteration()
        marks.append(len(tr.Rs))
        if (it + 1) % window == 0 or it + 1 == iters:
            lo, hi = marks[max(len(marks) - 1 - window, 0)], marks[-1]
            if hi > lo:
                rps = float(np.sum(tr.Rs[lo:hi]) / np.sum(lengths[lo:hi]))
                curve.append((int(res["frames"]), round(rps, 4)))
    wall = time.time() - tic
    assert len(tr.Rs) <= len(lengths)
    host_loop = "native handles" if getattr(tr, "_nl", None) else "python classes"
    if getattr(tr, "_nl", None):
        tr._nl.close()
    for actor in tr.actors:
        if actor is not None:
            actor.close()
    tail = curve[-1][1] if curve else float("nan")
    return {"host_loop": host_loop, "algo": algo, "extra": {k: str(v) for k, v in extra.items()}, "schedule": "launch" if launch else "main", "sabotage": sabotage, "frames": iters * T * E,
            "A": int(cfg.action_dim), "min_eps": float(cfg.actor.min_eps), "final_reward_per_step": tail, "loss": None if res["loss"] is None else float(res["loss"]),
            "qmax": None if res["qmax"] is None else float(res["qmax"]), "wall_s": round(wall, 1), "curve": curve}


RAINBOW = {"learner.double_q": "true", "learner.dueling_head": "true", "learner.noisy_net": "true", "learner.n_step_q": 3, "replay.policy": "prioritize"}
# (name, algo, extra, frames, env_id): the four algorithm families of BASELINE configs[1..4]
FAMILIES = [("dqn", "dqn", {}, 4_000_000, "Breakout"),
            ("c51_rainbow_lite", "c51", RAINBOW, 4_000_000, "Breakout"),
            ("iqn", "iqn", {}, 3_000_000, "Asterix"),
            ("fqf", "fqf", {}, 3_000_000, "Asterix")]


def main(out_path=None, only=None):
    """The runs tests/test_gpu_learning.py asserts on, with their curves: the four families on the chase task (both schedules), prioritized replay, the six sabotaged
    runs, and the bandit task under the library-handle loop."""
    runs = []

    def keep(r, name):
        r["name"] = name
        print(json.dumps({k: v for k, v in r.items() if k != "curve"}), flush=True)
        print("   curve:", r["curve"], flush=True)
        runs.append(r)

    small_q = {"learner.iqn.K": 16, "learner.iqn.N": 16, "learner.iqn.N_dash": 16, "learner.iqn.F": 16}
    for name, algo, extra, frames, env_id in FAMILIES:
        if only and name not in only:
            continue
        for launch in (False, True):
            x = {**extra, **small_q} if (launch and algo in ("iqn", "fqf")) else extra
            keep(run(algo, x, 2_600_000 if algo == "c51" else 4_700_000 if algo == "dqn" else 4_200_000, launch, env_id=env_id, task="chase"), name)
    if not only or "dqn_prio" in only:
        keep(run("dqn", {"replay.policy": "prioritize"}, 4_700_000, task="chase"), "dqn_prioritized_sumtree")
        keep(run("dqn", {"replay.policy": "prioritize", "replay.sumtree": "false"}, 5_700_000, task="chase"), "dqn_prioritized_flat_vector")
    if not only or "sabotage" in only:
        for sab in ("discount_zero", "nstep_shift", "stale_next_state", "no_target_sync", "eps_one", "lr_zero"):
            keep(run("dqn", {}, 4_200_000, sabotage=sab, task="chase"), f"dqn_{sab}")
    if not only or "native" in only:
        os.environ["A0_NATIVE_LOOP"] = "1"
        keep(run("dqn", {}, 4_700_000, task="chase"), "dqn_native_loop")
        keep(run("c51", RAINBOW, 2_600_000, task="chase"), "c51_rainbow_lite_native_loop")
        keep(run("dqn", {}, 4_700_000, True, task="chase"), "dqn_native_loop_launch")
        keep(run("dqn", {}, 2_600_000), "block_task_dqn_native_loop")
    if out_path:
        with open(out_path, "w") as f:
            json.dump({"task": "env_task=chase (oracle/synth_env.c): the action moves the block on a 4 x 4 lattice, +1 on arrival at the target cell, respawn three to six moves "
                               "away; chance ~0.02, optimum 0.25 per step.  block_task_* runs: env_task=block (a contextual bandit, optimum 1 per step) under the library-handle loop",
                       "metric": "reward per env step over the episodes that finished in the last 25 iterations", "runs": runs}, f, indent=1)
    return runs


if __name__ == "__main__":
    args = sys.argv[1:]
    out = args[0] if args and args[0].endswith(".json") else None
    only = [a for a in args if not a.endswith(".json")] or None
    main(out, only)
