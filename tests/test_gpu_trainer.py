"""GPU (MI355X): the product's Python classes end to end — Actor rollout parity against the oracle actor on the same
synthetic env and the same random draws, replay contents, Trainer iterations for every algorithm family."""
import os

import numpy as np
import pytest
import torch

import recipe
from oracle import actor as oactor
from oracle import core, learner as olearner, nets
from util import assert_close, record_stats

pytestmark = pytest.mark.gpu


def make_cfg(algo="dqn", E=4, **kw):
    from agent0_amd.deepq.config import parse_overrides
    cfg = parse_overrides([f"learner.algo={algo}", f"actor.num_envs={E}", "wandb=false", "tb=false", "logdir=gpurun_out/test_logs"] + [f"{k}={v}" for k, v in kw.items()])
    cfg.obs_shape = (4, 84, 84)
    cfg.action_dim = 4
    return cfg


ROLLOUT_SPECS = dict(recipe.SPECS, fqf4=recipe.NetSpec("fqf", 4))


@pytest.mark.parametrize("n_step,spec_name,task", [(1, "dqn", "stream"), (3, "dqn", "stream"), (1, "dqn_duel", "stream"), (3, "c51", "stream"), (1, "qr", "stream"),
                                                  (1, "iqn_duel", "stream"), (3, "iqn_duel", "stream"), (3, "fqf4", "stream"),
                                                  (3, "dqn", "block"), (1, "dqn_duel", "block"), (3, "c51", "block"), (1, "iqn_duel", "block"), (3, "fqf4", "block"),
                                                  (3, "dqn", "block-unmerged"),
                                                  (1, "dqn", "chase"), (3, "dqn_duel", "chase"), (3, "c51", "chase"), (1, "iqn_duel", "chase"), (3, "fqf4", "chase"),
                                                  (3, "dqn", "chase-unmerged"),
                                                  # 96 envs: terminal steps occur (1 / 500 per env and step) — the merged kernels' terminal paths: all four channels of the
                                                  # new observation are the new frame; on the chase task that frame waits for the action (the whole of conv1 behind it)
                                                  (3, "dqn_duel", "chase@96"), (1, "dqn", "stream@96"), (3, "c51", "chase@96")])
def test_actor_rollout_matches_oracle(n_step, spec_name, task, monkeypatch):
    """dqn / dqn_duel take the fused actor tail (a0_actor_qhead), c51 / qr the distributional tail, iqn / fqf the quantile tail (head GEMM slabs ->
    bias, dueling per quantile, mean / fraction-weighted sum, argmax, epsilon-greedy), each in one launch with the env step
    (a0_actor_qhead_env_step / a0_actor_dist_tail_env_step / a0_actor_quantile_tail_env_step).  IQN draws K = 32 fresh taus per env and step (model.py:238, agent.py:25-28) from the actor's Philox tau stream — the
    oracle gets the same draws through ``taus_fn`` — and under hipGraph replay their offsets come from the device control block.
    ``task=block``: the learnable reward task (the reward depends on the action the tail has just chosen: the three merged kernels and, ``block-unmerged``,
    a0_env_synth_step_commit behind a separate tail)."""
    from agent0_amd.deepq.agent import Actor
    from agent0_amd.deepq.model import DeepQNet
    from agent0_amd.deepq.replay import ReplayDataset
    from agent0_amd.common.utils import DeviceRng

    E, T = 4, 12
    if "@" in task:
        task, big = task.split("@")
        E = int(big)
    spec = ROLLOUT_SPECS[spec_name]
    if task.endswith("-unmerged"):
        monkeypatch.setenv("A0_TAIL_ENV", "0")
        task = task[: -len("-unmerged")]
    cfg = make_cfg(spec.algo, E, **{"env_task": task, "learner.n_step_q": n_step, "actor.sample_steps": 6, "replay.size": max(256, 36 * E), "learner.batch_size": 8,
                                     "learner.dueling_head": str(bool(spec.dueling)).lower(), **({"learner.qr.num_atoms": spec.num_atoms} if spec.algo == "qr" else {})})
    model = DeepQNet(cfg)
    sd = recipe.make_state_dict(spec, 11)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    replay = ReplayDataset(cfg, ops=model.ops)
    actor = Actor(cfg, model, replay=replay, rank=0)
    assert actor.fused_tail == (spec.algo == "dqn") and actor.quant_tail == (spec.algo in ("iqn", "fqf")) and actor.tail_env == (os.environ.get("A0_TAIL_ENV") != "0" and spec.algo in ("dqn", "c51", "qr"))
    # oracle twin: same env definition, same Philox draws (stream ids / offsets as DeviceRng assigns them)
    seed64 = (cfg.seed & 0xFFFFFFFF)
    step_no = [0]

    def draw(E_):
        off = step_no[0] * ((E_ + 3) // 4) * 4
        step_no[0] += 1
        a = (core.rng_u32(seed64, DeviceRng.STREAM_EGREEDY_A, off, E_) % 4).astype(np.int64)
        u = core.rng_uniform(seed64, DeviceRng.STREAM_EGREEDY_U, off, E_)
        return a, u

    tau_call = [0]
    K = int(cfg.learner.iqn.K)

    def taus_fn(E_):
        n = E_ * K
        off = tau_call[0] * ((n + 3) // 4) * 4
        tau_call[0] += 1
        return torch.from_numpy(core.rng_uniform(seed64, DeviceRng.STREAM_TAUS, off, n).reshape(E_, K, 1))

    env = core.SynthVecEnv(E, seed=cfg.seed, rank=0, action_dim=4, task=task)
    ora = oactor.OracleActor(env, olearner.to_params(sd), spec, n_step=n_step, sample_steps=6, draw=draw, taus_fn=taus_fn if spec.algo == "iqn" else None)
    eps = np.float32(0.35)
    for call in range(6):           # calls 0-1 run eagerly, call 2 captures the rollout into a hipGraph, calls 3-5 replay it
        data, rs, qs = actor.sample(float(eps))
        replay.extend(data)
        odata, ors, oqs = ora.sample(eps)
        # fqf: the fraction net's taus differ from torch's by an ulp and q(tau) runs through cos(pi*64*tau), which amplifies that ~200x
        assert_close(qs, oqs, *((5e-4, 5e-5) if spec.algo == "fqf" else (5e-5, 5e-6)), "mean max-Q per step")
        assert rs == [float(x) for x in ors]
        n = len(odata)
        base = call * n
        rows = replay.frames.view(replay.size, -1)[base:base + n].cpu().numpy()
        for i, (fr, at, rt, dt) in enumerate(odata):
            assert np.array_equal(rows[i], fr.reshape(-1)), f"transition {base + i}: packed st||st_next bytes"
            assert int(replay.act[base + i]) == int(at) and float(replay.rew[base + i]) == np.float32(rt) and bool(replay.done[base + i] != 0) == bool(dt)
    assert actor._graph is not None, "the rollout should have been captured"
    assert len(replay) == min(6 * 6 * E, replay.size) and replay.top == min(36 * E, replay.size)
    if E > 4:
        assert core.env_terminals(cfg.seed, 0, E, 36).any(), "the larger case exists for the terminal steps"


ALGOS = [("dqn", {}), ("dqn", {"learner.double_q": "true", "learner.dueling_head": "true", "learner.n_step_q": 3}),
         ("c51", {"learner.double_q": "true", "learner.dueling_head": "true", "learner.noisy_net": "true", "learner.n_step_q": 3, "replay.policy": "prioritize"}),
         ("c51", {"replay.policy": "prioritize", "replay.sumtree": "false"}), ("qr", {}), ("iqr", {"learner.double_q": "true"}), ("fqf", {}),
         ("mdqn", {"learner.n_step_q": 3})]


@pytest.mark.parametrize("algo,extra", ALGOS)
def test_trainer_iterations(algo, extra):
    from agent0_amd.deepq.trainer import Trainer
    cfg = make_cfg(algo, 8, **{"actor.sample_steps": 10, "replay.size": 400, "learner.batch_size": 32, "learner.learner_steps": 3,
                                "trainer.training_start_steps": 100, "learner.target_update_freq": 4, **extra})
    tr = Trainer(cfg)
    flat0 = tr.learner.engine.online.flat.clone()
    res = None
    for it in range(7):
        res = tr.run_iteration()
    assert res["frames"] == 7 * 80 and len(tr.replay) == 400 and tr.replay.written == 560
    assert res["loss"] is not None and np.isfinite(res["loss"]) and res["fps"] > 0 and np.isfinite(res["qmax"])
    tr.logging(res)                        # trainer.py:158-169: the per-iteration line reaches msg.log (level INFO without Hydra's logging config)
    for h in tr.logger.handlers:
        h.flush()
    assert "frames:" in open(os.path.join(cfg.logdir, "msg.log")).read()
    if algo == "fqf":
        assert np.isfinite(res["fraction_loss"])
    eng = tr.learner.engine
    n_updates = 5 * 3            # training starts once len(replay) > 100, i.e. from the 2nd iteration... (80 -> 160 > 100)
    assert tr.learner.update_steps == 6 * 3 and int(eng.state[2]) == 0
    assert not torch.equal(flat0, eng.online.flat) and torch.isfinite(eng.online.flat).all()
    if cfg.replay.policy.name == "prioritize":
        if cfg.replay.sumtree:
            t = tr.replay.tree.cpu().numpy(); c2 = tr.replay.cap2
            p = np.arange(1, c2)
            assert np.array_equal(t[p], t[2 * p] + t[2 * p + 1]) and t[1] > 0
        else:
            assert float(tr.replay.priority.min()) > 0 and tr.replay.max_p >= 1.0
        assert 0.4 <= tr.replay.beta <= 1.0
    sd = tr.learner.model.state_dict()
    assert all(torch.isfinite(v).all() for v in sd.values())
    # reference-signature train(): the 6-tuple of float tensors (trainer.py:88-97)
    B = 32
    fr = torch.from_numpy(recipe.make_frames(B, 3)).reshape(B, -1).float()
    a, r, d, w = recipe.make_transitions(B, 4, 4)
    out = tr.learner.train((fr, torch.from_numpy(a).float(), torch.from_numpy(r), torch.from_numpy(d).float(), torch.from_numpy(w), torch.arange(B).float()))
    assert out["q_loss"].shape == (B,) and out["indices"].dtype == torch.int64
    assert out["q_loss"].device.type == "cpu" and out["indices"].device.type == "cpu", "CPU copies like the reference's .detach().cpu() (agent.py:163-169)"


def test_model_api_matches_reference_shapes():
    from agent0_amd.deepq.model import DeepQNet
    x = torch.from_numpy(recipe.make_frames(3, 1)[:, :4].copy()).float().div(255.0)
    for name in ("dqn", "c51", "qr", "iqn", "fqf"):
        spec = recipe.SPECS[name]
        cfg = make_cfg(name if name != "iqn" else "iqn", 4)
        cfg.action_dim = spec.action_dim
        m = DeepQNet(cfg)
        sd = recipe.make_state_dict(spec, 11)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        got = m.state_dict()
        for k, v in sd.items():
            if not recipe.is_buffer(k):
                assert torch.equal(got[k].cpu(), torch.from_numpy(v)), k       # state_dict round trip through the packed device layout
        p = olearner.to_params(sd)
        with torch.no_grad():
            if name in ("dqn", "c51", "qr"):
                assert_close(m(x), nets.forward(p, spec, x), 5e-5, 5e-6, f"{name} forward")
                assert_close(m.qval(x), nets.qval(p, spec, x), 5e-5, 5e-6, f"{name} qval")
            elif name == "iqn":
                q, taus = m(x, n=8)
                assert q.shape == (3, 8, spec.action_dim) and taus.shape == (3, 8, 1)
                assert_close(q, nets.head_iqn(p, spec, nets.encoder(p, x), taus.cpu()), 5e-5, 5e-6, "iqn forward at the drawn taus")
            else:
                assert_close(m.qval(x), nets.qval(p, spec, x), 1e-3, 1e-4, "fqf qval")
        assert len(list(m.params())) == len([k for k in sd if not recipe.is_buffer(k) and "fraction" not in k])


def test_rccl_collectives_on_the_flat_buffers():
    """The exact collectives the data-parallel path issues (SUM on the fp32 gradient buffer, MAX on the int32 NaN flag, broadcast of
    the parameters, barrier) through RCCL — single rank here (one GPU per box); world_size 2 is covered on CPU/gloo."""
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        from agent0_amd.deepq.trainer import Trainer
        cfg = make_cfg("dqn", 8, **{"actor.sample_steps": 10, "replay.size": 400, "learner.batch_size": 32, "learner.learner_steps": 2,
                                    "trainer.training_start_steps": 50})
        tr = Trainer(cfg)
        eng = tr.learner.engine
        calls = []

        def hook(grads, state):
            g0 = grads[: eng.L.n_adam].clone()
            dist.all_reduce(grads[: eng.L.n_adam], op=dist.ReduceOp.SUM)
            dist.all_reduce(state[0:1], op=dist.ReduceOp.MAX)
            calls.append(bool(torch.equal(g0, grads[: eng.L.n_adam])))

        eng.grad_hook = hook
        dist.broadcast(eng.online.flat, src=0)
        eng.online.refresh_wt(); eng.sync_target(force=True)
        for _ in range(3):
            res = tr.run_iteration()
        dist.barrier()
        torch.cuda.synchronize()
        assert len(calls) == 6 and all(calls) and np.isfinite(res["loss"]) and tr.learner.update_steps == 6
    finally:
        dist.destroy_process_group()


def test_checkpoint_roundtrip_and_modes(tmp_path):
    """N3: save -> load restores weights (reference state_dict keys), optimizer state and counters; mode=play evaluates only."""
    from agent0_amd.deepq.trainer import Trainer
    base = {"actor.sample_steps": 10, "replay.size": 400, "learner.batch_size": 32, "learner.learner_steps": 2, "trainer.training_start_steps": 50,
            "logdir": str(tmp_path / "run")}
    tr = Trainer(make_cfg("c51", 8, **{**base, "learner.dueling_head": "true"}))
    for _ in range(3):
        tr.run_iteration()
    path = tr.save_checkpoint(str(tmp_path / "ck.pth"))
    blob = torch.load(path, weights_only=False)
    assert set(blob["model"]) == set(recipe.state_dict_shapes(recipe.NetSpec("c51", 4, dueling=True))), "reference key names"
    tr2 = Trainer(make_cfg("c51", 8, **{**base, "learner.dueling_head": "true", "seed": 7}))
    assert not torch.equal(tr2.learner.engine.online.flat, tr.learner.engine.online.flat)
    tr2.load_checkpoint(path)
    e1, e2 = tr.learner.engine, tr2.learner.engine
    assert torch.equal(e1.online.flat, e2.online.flat) and torch.equal(e1.target.flat, e2.target.flat)
    assert torch.equal(e1.adam_m, e2.adam_m) and torch.equal(e1.adam_v, e2.adam_v) and torch.equal(e1.state, e2.state) and tr2.frame_count == tr.frame_count
    x = torch.from_numpy(recipe.make_frames(3, 1)[:, :4].copy())
    assert torch.equal(tr.learner.model.qval(x), tr2.learner.model.qval(x))
    # play mode: loads the weights, runs evaluation episodes, trains nothing
    cfg = make_cfg("c51", 8, **{**base, "learner.dueling_head": "true", "mode": "play", "checkpoint": path, "trainer.test_episodes": 1})
    tr3 = Trainer(cfg)
    tr3.run()
    assert tr3.learner.update_steps == 0 and torch.equal(tr3.learner.engine.online.flat, e1.online.flat)
    # the evaluation's video (trainer.py:131-135): newest frame of the first four envs per step, grey channel repeated three times
    v = tr3.last_test_video
    assert v is not None and v.dtype == np.uint8 and v.shape[0] == 4 and v.shape[2:] == (3, 84, 84) and v.shape[1] >= cfg.actor.sample_steps
    assert np.array_equal(v[:, :, 0], v[:, :, 1]) and np.array_equal(v[:, :, 0], v[:, :, 2])


@pytest.mark.parametrize("workers", [0, 2])
def test_host_env_pool_device_frame_stack_is_exact(workers):
    """N1, device frame stack: only the newest frame of an env crosses PCIe when its stack advanced by one frame; stacks that changed in any
    other way (two new frames, a whole new stack) are uploaded whole.  The device observation must equal the host one byte for byte at
    every step, and the number of whole-stack uploads must be exactly the number of such events of the scripted env."""
    import host_slices
    from agent0_amd.common.env_pool import HostEnvPool
    from agent0_amd.ops import HipOps
    E, T = 6, 20
    pool = HostEnvPool(host_slices.scripted_slice(5), E, obs_shape=(4, 84, 84), action_dim=4, num_workers=workers, ops=HipOps(), newest_frame=True)
    ref = host_slices.ScriptedStack(5, 0, E)
    try:
        obs, _ = pool.reset()
        want, _ = ref.reset()
        assert np.array_equal(obs.cpu().numpy().reshape(E, 4, 84, 84), want)
        act = torch.zeros(E, dtype=torch.int32, device="cuda")
        events = 0
        for t in range(1, T + 1):
            obs = pool.step(act)[0]
            want = ref.step(None)[0]
            assert np.array_equal(obs.cpu().numpy().reshape(E, 4, 84, 84), want), f"step {t}"
            events += sum(host_slices.ScriptedStack.mode(e, t) in (0, 3) for e in range(E))
        assert 0 < events < T * E and pool.full_uploads == events
        assert pool.pcie_bytes_per_step < E * 28224 // 3
    finally:
        pool.close()


@pytest.mark.parametrize("workers,newest,algo,knobs", [(0, True, "dqn", {}), (3, True, "dqn", {}), (3, False, "dqn", {}), (3, True, "c51", {}), (2, True, "iqn", {}), (3, True, "c51-noisy", {}),
                                                       (3, True, "dqn", {"A0_ENV_POOL_CALLS": "0"}), (3, True, "dqn", {"A0_HOST_ROLLOUT": "0"})])
def test_host_env_pool_matches_device_env(workers, newest, algo, knobs, monkeypatch):
    """N1: HOST environments behind env_pool.HostEnvPool feed the same device pipeline — worker processes (3 workers over 8 envs: slices
    of 3 / 2 / 3) or in-process stepping (0) write into the page-locked double-buffered ring, observations arrive over the copy stream,
    actions reach the workers by DMA.  With the oracle's CPU twin of the synthetic env inside the workers, the replay contents, episode
    returns and per-step max-Q must equal those produced with the device-resident env, byte for byte, for n-step 3 over 5 rollouts.
    Round 4: the step path's two PCIe legs are one library call each (a0_env_pool_upload; a0_env_pool_send, whose kernels store the actions and the
    step word straight into the page-locked block) and the actor enqueues a step's bookkeeping behind the NEXT step's actions (Actor._rollout_host);
    the knobs select the per-copy torch calls / the step-by-step order, which must give the same bytes."""
    import host_slices
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    from agent0_amd.common.env_pool import HostEnvPool
    from agent0_amd.deepq.agent import Actor
    from agent0_amd.deepq.model import DeepQNet
    from agent0_amd.deepq.replay import ReplayDataset
    E = 8
    outs = []
    for host in (False, True):
        noisy = algo.endswith("-noisy")      # NoisyNet: the noise of every layer is redrawn every reset_noise_freq = 4 steps, in the middle of the 6-step rollouts
        name = algo.split("-")[0]
        cfg = make_cfg(name, E, **{"learner.n_step_q": 3, "actor.sample_steps": 6, "replay.size": 300, "learner.batch_size": 8, "learner.noisy_net": str(noisy).lower()})
        model = DeepQNet(cfg)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(recipe.NetSpec(name, 4, noisy=noisy), 11).items()})
        replay = ReplayDataset(cfg, ops=model.ops)
        envs = HostEnvPool(host_slices.synth_slice(cfg.seed, 0), E, obs_shape=(4, 84, 84), action_dim=4, num_workers=workers, ops=model.ops,
                           newest_frame=newest) if host else None
        actor = Actor(cfg, model, replay=replay, rank=0, envs=envs)
        if host:
            assert envs.library_calls == (newest and knobs.get("A0_ENV_POOL_CALLS") != "0")
        rs_all, qs_all = [], []
        for _ in range(5):
            data, rs, qs = actor.sample(0.3)
            replay.extend(data)
            rs_all += rs
            qs_all += qs
        n = 5 * 6 * E
        outs.append((replay.frames[: n * replay.row_bytes].clone(), replay.act[:n].clone(), replay.rew[:n].clone(), replay.done[:n].clone(), rs_all, qs_all))
        actor.close()
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(a, b)
    assert outs[0][4] == outs[1][4] and outs[0][5] == outs[1][5]
    assert float(outs[0][3].sum()) > 0, "the run contains done flags (life losses), so the flag path is exercised"


@pytest.mark.parametrize("workers", [0, 2])
def test_g12_host_env_pool_rows_equal_the_reference_wrappers(workers):
    """N1 pinned to the reference: the product's wrapper chain over the scripted emulators of fixture group G12, behind HostEnvPool (in-process and
    two worker processes), through Actor.sample into the replay ring.  Every row must hold what the REFERENCE's wrappers returned on the same emulator
    (tests/golden/g12_wrappers_*.npz, generated by importing atari_wrappers.py:11-56): st = the observation before the step, st_next = the one after it (the next
    episode's first observation on an autoreset), the sign-clipped reward, done = (terminated | life_loss) & ~truncated (agent.py:57-62), and the finished episodes'
    unclipped returns in env order.  The emulator's outcomes do not depend on the action, so the actor's own epsilon-greedy choices are free."""
    import host_slices
    from agent0_amd.common.env_pool import HostEnvPool
    from agent0_amd.deepq.agent import Actor
    from agent0_amd.deepq.model import DeepQNet
    from agent0_amd.deepq.replay import ReplayDataset
    from util import golden
    E, T, R = 4, 8, 18                                    # 144 agent steps per env (the shortest case drives 160)
    cfg = make_cfg("dqn", E, **{"learner.n_step_q": 1, "actor.sample_steps": T, "replay.size": E * T * R, "learner.batch_size": 8})
    model = DeepQNet(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(recipe.NetSpec("dqn", 4), 11).items()})
    replay = ReplayDataset(cfg, ops=model.ops)
    envs = HostEnvPool(host_slices.g12_slice(), E, obs_shape=(4, 84, 84), action_dim=4, num_workers=workers, ops=model.ops)
    actor = Actor(cfg, model, replay=replay, rank=0, envs=envs)
    rs_all = []
    try:
        for _ in range(R):
            data, rs, qs = actor.sample(0.5)
            replay.extend(data)
            rs_all += [float(x) for x in rs]
    finally:
        actor.close()
    n = E * T * R
    rows = replay.frames[: n * replay.row_bytes].view(n, 2, 4, 84, 84).cpu().numpy()
    rew, done = replay.rew[:n].cpu().numpy(), replay.done[:n].cpu().numpy()
    counter = lambda stack: int(np.frombuffer(stack[0, 0, :4].tobytes(), np.uint32)[0])
    G = [golden(f"g12_wrappers_{name}") for name in host_slices.G12_NAMES]
    want_rs = []
    prev = [int(g["reset_obs_c"][0]) for g in G]
    nreset = [1] * E
    ret = [0.0] * E
    for s in range(T * R):
        for e in range(E):
            g, row = G[e], s * E + e                     # env-major within a step, step-major overall (agent.py:78-81)
            over = bool(g["terminated"][s] or g["truncated"][s])
            nxt = int(g["reset_obs_c"][nreset[e]]) if over else int(g["obs_c"][s])
            assert counter(rows[row, 0]) == prev[e] and counter(rows[row, 1]) == nxt, (s, e)
            assert (rows[row, 1, 1:] == nxt % 251).all() and (rows[row, 0, 1:] == prev[e] % 251).all()
            assert float(rew[row]) == float(g["reward"][s]), (s, e)
            assert bool(done[row]) == bool((g["terminated"][s] or g["life_loss"][s]) and not g["truncated"][s]), (s, e)
            ret[e] += float(g["raw_reward"][s])
            if over:
                want_rs.append(float(np.float32(ret[e])))
                ret[e] = 0.0
                nreset[e] += 1
            prev[e] = nxt
    assert rs_all == want_rs, "episode returns over the unclipped rewards, in step and env order (agent.py:85-88)"
    assert float(done.sum()) > 20 and len(want_rs) > 20


@pytest.mark.parametrize("algo,n_step,workers,groups", [("dqn", 3, 4, 2), ("dqn", 1, 0, 2), ("c51", 1, 6, 3), ("dqn", 3, 2, 2)])
def test_grouped_host_env_rollouts_match_the_device_env(algo, n_step, workers, groups):
    """Round 4 (N1): env_pool.HostEnvGroups splits the host vector env into groups with their own worker processes; the actor steps one group on the CPU while the
    GPU infers the other (Actor._rollout_groups — what the reference gets from num_actors actor processes, launch.py:30-61).  Every env must see exactly what it
    sees in a one-group rollout: replay rows (bytes, actions, n-step rewards, dones) in the same slots, episode returns in the same order and per-step max-Q equal
    to those of the DEVICE-resident env, for scalar and distributional heads, n-step 1 and 3, uneven groups (8 envs in 3 groups) and in-process stepping."""
    import host_slices
    from agent0_amd.common.env_pool import HostEnvGroups
    from agent0_amd.deepq.agent import Actor
    from agent0_amd.deepq.model import DeepQNet
    from agent0_amd.deepq.replay import ReplayDataset
    E = 8
    outs = []
    spec = recipe.NetSpec(algo, 4)
    for host in (False, True):
        cfg = make_cfg(algo, E, **{"learner.n_step_q": n_step, "actor.sample_steps": 6, "replay.size": 300, "learner.batch_size": 8})
        model = DeepQNet(cfg)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(spec, 11).items()})
        replay = ReplayDataset(cfg, ops=model.ops)
        envs = HostEnvGroups(host_slices.synth_slice(cfg.seed, 0), E, groups=groups, obs_shape=(4, 84, 84), action_dim=4, num_workers=workers, ops=model.ops) if host else None
        actor = Actor(cfg, model, replay=replay, rank=0, envs=envs)
        assert (actor.groups is not None) == host
        rs_all, qs_all = [], []
        for _ in range(5):
            data, rs, qs = actor.sample(0.3)
            replay.extend(data)
            rs_all += rs
            qs_all += qs
        n = 5 * 6 * E
        outs.append((replay.frames[: n * replay.row_bytes].clone(), replay.act[:n].clone(), replay.rew[:n].clone(), replay.done[:n].clone(), rs_all, qs_all))
        actor.close()
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(a, b)
    assert outs[0][4] == outs[1][4] and outs[0][5] == outs[1][5]
    assert float(outs[0][3].sum()) > 0


@pytest.mark.parametrize("algo,extra", [("dqn", []), ("c51", ["learner.noisy_net=true", "learner.n_step_q=3", "replay.policy=prioritize"])])
def test_launch_mode_overlap_is_race_free(algo, extra):
    """Trainer(use_lp=True) — the launch.py schedule: rollout k+1 with a weight snapshot runs on a second stream while the update block
    consumes rollout k.  Issuing the same work on ONE stream must give bit-identical parameters, replay contents and statistics."""
    from agent0_amd.deepq.config import parse_overrides
    from agent0_amd.deepq.trainer import Trainer
    res = []
    for overlap in (True, False):
        cfg = parse_overrides([f"learner.algo={algo}", "actor.num_envs=16", "actor.sample_steps=12", "learner.batch_size=32", "learner.learner_steps=3", "replay.size=500",
                               "trainer.training_start_steps=100", "learner.target_update_freq=4", "wandb=false", "tb=false", "logdir=/tmp/a0_lp"] + extra)
        tr = Trainer(cfg, use_lp=True)
        tr.overlap = overlap
        out = [tr.run_iteration() for _ in range(7)]            # wraps the 500-slot ring, captures + replays the graphs
        torch.cuda.synchronize()
        res.append((tr.learner.engine.online.flat.clone(), tr.replay.frames.clone(), tr.replay.act.clone(), tr.replay.rew.clone(), [o["loss"] for o in out],
                    [o["qmax"] for o in out], tr.frame_count, tr.actors[1].model._dev.flat.clone()))
    a, b = res
    assert a[6] == b[6] == 7 * 16 * 12
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[7], b[7])
    assert a[4] == b[4] and a[5] == b[5]
    assert a[4][-1] is not None and not torch.equal(a[0], a[7])      # training happened; the actor's copy is one update block behind


@pytest.mark.parametrize("algo,extra,workers", [("dqn", [], 3), ("c51", ["learner.noisy_net=true", "learner.n_step_q=3", "replay.policy=prioritize"], 2)])
def test_launch_mode_with_host_envs_trains_beside_the_rollout(algo, extra, workers):
    """Round 4 (N1 x N2): with a HOST env the rollout occupies the Python thread (it waits for the worker processes), so ``run_iteration_lp`` enqueues the update block
    first and runs the rollout on the actor stream beside it — launch.py's actor processes stepping emulators while the learner trains (launch.py:44-63).  Parameters,
    replay contents and statistics must equal, bit for bit, the one-stream order (rollout, then block) on the host env AND the same schedule on the device-resident env."""
    import host_slices
    from agent0_amd.common.env_pool import HostEnvPool
    from agent0_amd.deepq import agent as agents
    from agent0_amd.deepq.config import parse_overrides
    from agent0_amd.deepq.trainer import Trainer
    res = []
    for host, overlap in ((False, False), (True, True), (True, False)):
        cfg = parse_overrides([f"learner.algo={algo}", "actor.num_envs=16", "actor.sample_steps=12", "learner.batch_size=32", "learner.learner_steps=3", "replay.size=500",
                               "trainer.training_start_steps=100", "learner.target_update_freq=4", "wandb=false", "tb=false", "logdir=/tmp/a0_lp"] + extra)
        tr = Trainer(cfg, use_lp=True)
        if host:
            tr.actors[1].close()
            pool = HostEnvPool(host_slices.synth_slice(cfg.seed, 0), 16, obs_shape=(4, 84, 84), action_dim=tr.act_dim, num_workers=workers, ops=tr.ops)
            tr.actors[1] = agents.Actor(cfg, None, replay=tr.stage, ops=tr.ops, rank=0, envs=pool)
        tr.overlap = overlap
        out = [tr.run_iteration() for _ in range(7)]
        torch.cuda.synchronize()
        res.append((tr.learner.engine.online.flat.clone(), tr.replay.frames.clone(), tr.replay.act.clone(), tr.replay.rew.clone(), [o["loss"] for o in out],
                    [o["qmax"] for o in out], tr.frame_count, tr.actors[1].model._dev.flat.clone(), [o["return_train"] for o in out]))
        tr.actors[1].close()
    for b in res[1:]:
        a = res[0]
        assert a[6] == b[6] == 7 * 16 * 12
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[7], b[7])
        assert a[4] == b[4] and a[5] == b[5] and a[8] == b[8]
    assert res[0][4][-1] is not None


def test_launch_mode_uses_stale_weights_like_the_reference():
    """launch.py:58-63 issues the next rollout BEFORE the update block: rollout k+1 acts with the weights after block k-1.  Check that the
    actor's snapshot equals the learner's parameters as they were when the rollout was issued."""
    from agent0_amd.deepq.config import parse_overrides
    from agent0_amd.deepq.trainer import Trainer
    cfg = parse_overrides(["actor.num_envs=16", "actor.sample_steps=12", "learner.batch_size=32", "learner.learner_steps=2", "replay.size=2000",
                           "trainer.training_start_steps=100", "wandb=false", "tb=false", "logdir=/tmp/a0_lp"])
    tr = Trainer(cfg, use_lp=True)
    for _ in range(3):
        tr.run_iteration()
    before = tr.learner.engine.online.flat.clone()
    tr.run_iteration()
    torch.cuda.synchronize()
    assert torch.equal(tr.actors[1].model._dev.flat, before)
    assert not torch.equal(tr.learner.engine.online.flat, before)


def test_graphed_update_with_a_gradient_hook_matches_eager():
    """Data parallelism installs a grad_hook (RCCL all-reduce) between backward and Adam.  The learner then replays two hipGraphs around
    the eager hook; the result must equal the fully eager update bit for bit (the hook here scales the gradients, so skipping or
    misplacing it would show)."""
    from agent0_amd.deepq.config import parse_overrides
    from agent0_amd.deepq.trainer import Trainer
    res = []
    for use_graph in (True, False):
        cfg = parse_overrides(["actor.num_envs=16", "actor.sample_steps=12", "learner.batch_size=32", "learner.learner_steps=3", "replay.size=1000",
                               "trainer.training_start_steps=100", "wandb=false", "tb=false", "logdir=/tmp/a0_hook"])
        tr = Trainer(cfg)
        tr.learner.use_graph = use_graph
        calls = []
        def hook(grads, state, calls=calls):
            grads.mul_(0.5)
            calls.append(1)
        tr.learner.engine.grad_hook = hook
        for _ in range(6):
            tr.run_iteration()
        torch.cuda.synchronize()
        res.append((tr.learner.engine.online.flat.clone(), len(calls), tr.learner.update_steps))
    assert res[0][1] == res[1][1] == res[0][2] and torch.equal(res[0][0], res[1][0])


def test_data_parallel_path_with_one_rank_rccl_group_matches_plain_run(tmp_path):
    """Rehearsal of the multi-GPU bench on a one-GPU box (A0_DP_FORCE=1): a one-rank RCCL process group is alive, the gradient exchange
    (a0_dp_allreduce over RCCL: the dense bucket on a side stream beside the encoder backward, the small convolution bucket behind it) is
    CAPTURED in the update's single hipGraph (asserted below: "captured in the update's hipGraph"), the parameter broadcast / barriers /
    max-over-ranks timing all run and the line carries the per-rank times.  A one-rank all-reduce is the identity and Adam's eps is
    1e-2/(1*B) either way, so the losses must equal those of the plain run bit for bit."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "4", "--replay-size", "40960", "--no-cpu-baseline",
           "--no-ratio320", "--no-other-entry"]
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    outs = []
    for force in ("0", "1"):
        env = dict(os.environ, A0_DP_FORCE=force, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   A0_PROBE="none", HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    plain, dp = outs
    assert "rehearsal" in dp["config"]["workload"] and "inactive" in plain["config"]["workload"]
    assert dp["value"] > 0 and plain["value"] > 0
    assert dp["last_loss"] == plain["last_loss"], (dp["last_loss"], plain["last_loss"])
    # the exchange goes through the C-ABI (a0_dp_allreduce over RCCL) and is part of the update's single hipGraph
    assert dp["config"]["gradient_exchange"].startswith("RcclGradAllReduce") and "captured" in dp["config"]["gradient_exchange"], dp["config"]["gradient_exchange"]
    # one JSON line is enough to diagnose a scaling run: the exchange in use and every rank's own step time beside the max-over-ranks one
    assert dp["gradient_exchange"] == dp["config"]["gradient_exchange"] and len(dp["per_rank_ms_per_step"]["ranks"]) == 1
    assert 0 < dp["per_rank_ms_per_step"]["min"] <= dp["per_rank_ms_per_step"]["max"] and plain["per_rank_ms_per_step"] is None
    # ... and RCCL's own statement of the rank count (a0_dp_info: ncclCommCount / ncclCommUserRank) with an all-reduce of ones through the gradients' communicator
    assert plain["rccl"] is None and dp["rccl"]["nranks"] == 1 and dp["rccl"]["rank"] == 0 and dp["rccl"]["allreduce_of_ones"] == 1.0 and dp["rccl"]["matches_world_size"] is True
    assert dp["rccl"]["backend"].startswith("rccl") and dp["rccl"]["in_graph"] is True
    # ``python bench.py --gpus N`` with no launcher environment: bench.py starts torch.distributed.run itself, as a child (here N = 1)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(A0_DP_FORCE="1", A0_PROBE="none", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd + ["--gpus", "1", "--self-launch"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    launched = json.loads(lines[0])
    assert launched["n_gpus"] == 1 and launched["last_loss"] == plain["last_loss"] and "rehearsal" in launched["config"]["workload"]
    # (what the exchange and the launcher cost in time is a measurement, not a correctness property: tools/ab_native_dp.sh, profiles/r04_native_dp_one_rank_ab.txt;
    # the three lines' rates are left in gpurun_out/test_stats for the record)
    record_stats("dp_rehearsal_rates", {"plain": plain["value"], "dp_one_rank": dp["value"], "self_launched": launched["value"]})


@pytest.mark.parametrize("extra", [[], ["--algo", "c51", "learner.noisy_net=true", "learner.dueling_head=true", "learner.double_q=true", "learner.n_step_q=3", "replay.policy=prioritize"]],
                         ids=["dqn", "c51-rainbow-lite"])
def test_learner_handle_exchanges_gradients_itself_one_rank_group(extra):
    """Round 4, (e) through the boundary: ``a0_learner_set_exchange`` puts the RCCL gradient exchange (dense bucket + NaN flag on a side stream beside the encoder backward,
    convolution bucket behind it, join before Adam — dist.RcclGradAllReduce's buckets and order) into ``a0_learner_update``, so the native host loop (the default since
    round 6; A0_NATIVE_LOOP_DP=0 opts out) and a plain C host can run data-parallel.  Rehearsed with a one-rank RCCL group (the identity): the losses must equal, bit for bit, the plain
    native run, the Python classes' plain run and the Python classes' data-parallel run; the line must say who issued the exchange."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "3", "--replay-size", "40960", "--no-cpu-baseline", "--no-ratio320", "--no-other-entry"] + extra
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    base = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, A0_PROBE="none", HSA_ENABLE_IPC_MODE_LEGACY="0")
    lines = {}
    runs = (("python plain", dict(A0_NATIVE_LOOP="0", A0_DP_FORCE="0")), ("native plain", dict(A0_NATIVE_LOOP="1", A0_DP_FORCE="0")),
            ("python dp", dict(A0_NATIVE_LOOP="1", A0_DP_FORCE="1", A0_NATIVE_LOOP_DP="0")), ("native dp", dict(A0_NATIVE_LOOP="1", A0_DP_FORCE="1")),
            # the form a group of MORE than one rank takes (dense bucket on the side stream beside the encoder backward, three cross-stream hand-offs), forced at one rank
            ("native dp side stream", dict(A0_NATIVE_LOOP="1", A0_DP_FORCE="1", A0_DP_ONE_STREAM="0")))
    if extra:      # the second configuration: the handle's exchange against the plain handle run only (the Python classes' two runs are the first configuration's)
        runs = (runs[1], runs[3], runs[4])
    for name, env in runs:
        r = subprocess.run(cmd, env=dict(base, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        lines[name] = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    handles = "library handles"
    assert handles in lines["native plain"]["config"]["host_loop"] and handles in lines["native dp"]["config"]["host_loop"]
    if not extra:
        assert handles not in lines["python plain"]["config"]["host_loop"] and handles not in lines["python dp"]["config"]["host_loop"]      # A0_NATIVE_LOOP_DP=0 keeps a data-parallel run on the Python classes
        assert "captured" in lines["python dp"]["gradient_exchange"]
    assert "a0_learner_set_exchange" in lines["native dp"]["gradient_exchange"] and "a0_learner_set_exchange" in lines["native dp side stream"]["gradient_exchange"]
    losses = {k: v["last_loss"] for k, v in lines.items()}
    assert len(set(losses.values())) == 1 and losses["native dp"] is not None, losses
    assert lines["native dp"]["rccl"]["nranks"] == 1 and lines["native dp"]["rccl"]["allreduce_of_ones"] == 1.0
    # (the exchange's cost in time — two all-reduce launches and three cross-stream hand-offs per update, ~8 % when issued eagerly — is a measurement: profiles/r04_experiments.md)
    record_stats("native_dp_rates_" + ("c51" if extra else "dqn"), {k: v["value"] for k, v in lines.items()})


@pytest.mark.parametrize("native", ["0", "1"], ids=["python-classes", "native-loop"])
def test_main_entry_point_at_baseline_config0_sizes(tmp_path, native):
    """BASELINE configs[0] — "Breakout dqn, agent0.deepq.main, 16 envs, 100k replay" — as a child process through the reference's entry point
    with the reference's default sizes (config.py:108-120: 16 envs x 80 steps, batch 512, 20 updates per iteration, replay 100 000) on the
    GPU (``device=cpu`` raises by design: there is no CPU product path); only ``training_start_steps`` and ``total_steps`` are lowered so that
    the run trains within seconds.  Checked: exit code 0, the per-iteration records, training really happened, the checkpoint."""
    import csv
    import glob
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logdir = str(tmp_path / "runs")
    cmd = [sys.executable, "-m", "agent0.deepq.main", "env_id=Breakout", "learner.algo=dqn", "actor.num_envs=16", "replay.size=100000",
           "trainer.training_start_steps=5000", "trainer.total_steps=40000", "trainer.test_episodes=4", "device=cuda", "wandb=false", "tb=false", f"logdir={logdir}"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["A0_NATIVE_LOOP"] = native       # "1": the loop issued by the library's handles over the Trainer's buffers (the default outside this suite)
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    runs = glob.glob(os.path.join(logdir, "*-Breakout-dqn-42-*"))
    assert len(runs) == 1, "one run directory named <name>-<env>-<algo>-<seed>-<sha>-<time>-<uuid> (main.py:18-24)"
    rows = list(csv.DictReader(open(os.path.join(runs[0], "progress.csv"))))
    assert len(rows) == 40000 // (16 * 80) + 1 == 32
    assert [int(float(x["frames"])) for x in rows] == [1280 * (i + 1) for i in range(32)], "frame_count advances by sample_steps * num_envs (trainer.py:78)"
    first_loss = next(i for i, x in enumerate(rows) if x["loss"] != "")
    assert first_loss == 3, "training starts once len(replay) > training_start_steps: 5120 > 5000 after the fourth rollout (trainer.py:82)"
    assert all(np.isfinite(float(x["loss"])) for x in rows[first_loss:]) and all(float(x["fps"]) > 0 for x in rows)
    assert any(x["return_train"] != "" for x in rows) and all(x["qmax"] != "" for x in rows)
    log = open(os.path.join(runs[0], "msg.log")).read()
    assert "TEST --->" in log and "nan" not in log.lower()
    assert ("host loop: library handles" in log) == (native == "1")
    ck = torch.load(os.path.join(runs[0], "final.pth"), map_location="cpu", weights_only=True)
    assert int(ck["state"][1]) == (32 - first_loss) * 20 and int(ck["frame_count"]) == 32 * 1280, "20 updates per iteration (learner_steps, config.py:112)"
    assert ck["model"]["encoder.convs.0.weight"].shape == (32, 4, 8, 8) and ck["model"]["head.q_head.weight"].shape == (4, 512)


@pytest.mark.parametrize("launch", [False, True], ids=["main", "launch"])
@pytest.mark.parametrize("algo,extra", [("dqn", {}), ("dqn", {"learner.double_q": "true", "learner.dueling_head": "true", "learner.n_step_q": 3}),
                                        ("c51", {"learner.double_q": "true", "learner.dueling_head": "true"}), ("c51", {})])
def test_pipelined_target_pass_changes_no_number(algo, extra, launch, monkeypatch):
    """Round 4: under uniform replay the Trainer draws batch k + 1 and runs the TARGET network on it on a second stream while update k is in flight
    (Trainer._update_block_pipelined; blocks that contain a target sync run strictly in order).  Same kernels on the same inputs: losses, online and target
    parameters and Adam moments must be BIT-identical to the strictly serial block, across blocks with and without a sync, on both schedules."""
    from agent0_amd.deepq.trainer import Trainer

    def run(pipe):
        monkeypatch.setenv("A0_PIPELINE_TARGET", "1" if pipe else "0")
        cfg = make_cfg(algo, 8, **{"actor.sample_steps": 10, "replay.size": 400, "learner.batch_size": 32, "learner.learner_steps": 5, "trainer.training_start_steps": 100,
                                    "learner.target_update_freq": 7, **extra})
        tr = Trainer(cfg, use_lp=launch)
        for _ in range(10):
            tr.run_iteration()
        if launch:
            tr.actors[1].sample_finish(tr._pending)
        torch.cuda.synchronize()
        eng = tr.learner.engine
        return tr, list(tr.Ls), eng.online.flat.clone(), eng.target.flat.clone(), eng.adam_m.clone(), eng.adam_v.clone(), tr.replay.frames.clone()

    tr0, l0, p0, t0, m0, v0, f0 = run(False)
    tr1, l1, p1, t1, m1, v1, f1 = run(True)
    assert getattr(tr0, "pipelined_blocks", 0) == 0
    assert 3 <= tr1.pipelined_blocks < len(l1) // 5, "some blocks pipelined, the ones with a target sync not"
    assert l0 == l1 and len(l0) == 45
    assert torch.equal(p0, p1) and torch.equal(t0, t1) and torch.equal(m0, m1) and torch.equal(v0, v1) and torch.equal(f0, f1)


@pytest.mark.parametrize("algo,extra", [("dqn", {}), ("dqn", {"replay.policy": "prioritize", "learner.n_step_q": 3}), ("c51", {"learner.noisy_net": "true"}),
                                        ("fqf", {"env_id": "Asterix"})], ids=["dqn", "dqn-per-n3", "c51-noisy", "fqf"])
def test_prefetched_rollouts_change_no_number(algo, extra, monkeypatch):
    """Round 4: ``run_iteration(prefetch=True)`` (Trainer.run, bench.py's `main` schedule) enqueues iteration i + 1's rollout before the host waits for iteration
    i's statistics, which come back through page-locked buffers behind an event.  The stream sees the same launches in the same order with the same arguments,
    so losses, max-Q and return statistics, parameters, Adam moments and the replay ring must be BIT-identical to the loop that stops at every iteration —
    including an epsilon schedule that moves (the next epsilon is a function of the frame count only) and a rollout issued ahead and booked by ``final()``."""
    from agent0_amd.deepq.trainer import Trainer

    def run(ahead):
        cfg = make_cfg(algo, 8, **{"actor.sample_steps": 10, "replay.size": 400, "learner.batch_size": 32, "learner.learner_steps": 5, "trainer.training_start_steps": 100,
                                    "learner.target_update_freq": 7, "trainer.exploration_steps": 600, **extra})
        tr = Trainer(cfg)
        res = []
        for i in range(10):
            res.append(tr.run_iteration(prefetch=ahead and i != 4))       # iteration 5 consumes a prefetched rollout, iteration 6 issues its own
        if not ahead:
            tr.replay.extend(tr.actors[1].sample(tr.epsilon_fn(tr.frame_count))[0])      # what final() books for the prefetching loop: one more rollout
            tr.frame_count += tr.num_transitions
        else:
            assert tr._prefetched is not None
            pending, tr._prefetched = tr._prefetched, None
            tr.replay.extend(tr.actors[1].sample_finish(pending)[0])
            tr.frame_count += tr.num_transitions
        torch.cuda.synchronize()
        eng = tr.learner.engine
        keep = [{k: v for k, v in r.items() if k != "fps"} for r in res]
        return (keep, list(tr.Ls), list(tr.FLs), list(tr.Qs), list(tr.Rs), tr.frame_count, len(tr.replay), eng.online.flat.clone(), eng.target.flat.clone(),
                eng.adam_m.clone(), eng.adam_v.clone(), tr.replay.frames.clone())

    a = run(False)
    b = run(True)
    assert len(a[1]) == 45 and (algo != "fqf" or len(a[2]) == 45)
    for x, y in zip(a[:7], b[:7]):
        assert x == y
    for x, y in zip(a[7:], b[7:]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("algo,extra", [("dqn", {}), ("c51", {"learner.noisy_net": "true", "learner.dueling_head": "true", "replay.policy": "prioritize"}),
                                        ("iqr", {"env_id": "Asterix"}), ("fqf", {"env_id": "Asterix", "learner.double_q": "true"})], ids=["dqn", "c51-noisy-duel-per", "iqr", "fqf"])
def test_dense_reductions_in_the_encoder_launch_change_no_number(algo, extra, monkeypatch):
    """Round 4: without a gradient hook the slab reductions of the dense weight gradients (head, fc1, cosine embedding) are not launched by a0_dense_wgrad_multi but
    handed (a0_pending_reduce) to the encoder weight gradients' reduction launch, and NoisyNet's sigma gradients follow it: one launch less per update, the same sums
    in the same order.  Losses, parameters and Adam moments must be BIT-identical to the two-launch sequence."""
    from agent0_amd.deepq.trainer import Trainer

    def run(defer):
        monkeypatch.setenv("A0_DEFER_DENSE_REDUCE", "1" if defer else "0")
        # batch 256: large enough for the head's and fc1's weight gradients to be split into slabs
        cfg = make_cfg(algo, 8, **{"actor.sample_steps": 40, "replay.size": 1600, "learner.batch_size": 256, "learner.learner_steps": 5, "trainer.training_start_steps": 300,
                                    "learner.target_update_freq": 7, **extra})
        tr = Trainer(cfg)
        eng = tr.learner.engine
        assert eng._defer_dense == defer and (not defer or eng._enc_slab_off > 0)
        for _ in range(6):
            tr.run_iteration()
        torch.cuda.synchronize()
        return list(tr.Ls), list(tr.FLs), eng.online.flat.clone(), eng.target.flat.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.grads.clone()

    a = run(False)
    b = run(True)
    assert len(a[0]) == 30 and a[0] == b[0] and a[1] == b[1]
    for x, y in zip(a[2:], b[2:]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("algo", ["dqn", "c51", "iqr"])
def test_loss_statistic_from_the_adam_launch_equals_mean_rows(algo, monkeypatch):
    """Round 4: the Trainer's per-update `loss` statistic (trainer.py:99,111-113) is taken by workgroup 0 of the Adam launch into a ring (a0_adam_step_sync_wt,
    DeviceLearner.loss_ring) instead of by an a0_mean_rows launch per update — the same reduction statement for statement, so the recorded means must be
    bit-identical, ring wrap-around (1024 slots) aside, and everything else (parameters) untouched."""
    from agent0_amd.deepq.trainer import Trainer

    def run(ring):
        monkeypatch.setenv("A0_LOSS_RING", "1" if ring else "0")
        cfg = make_cfg(algo, 8, **{"actor.sample_steps": 10, "replay.size": 400, "learner.batch_size": 32, "learner.learner_steps": 7, "trainer.training_start_steps": 100,
                                    "learner.target_update_freq": 5, **({"env_id": "Asterix"} if algo == "iqr" else {})})
        tr = Trainer(cfg)
        for _ in range(8):
            tr.run_iteration()
        torch.cuda.synchronize()
        return list(tr.Ls), tr.learner.engine.online.flat.clone(), int(tr.learner.engine.state[6]), tr.learner.updates_issued

    l0, p0, c0, u0 = run(False)
    l1, p1, c1, u1 = run(True)
    assert len(l0) == 7 * 7 and l0 == l1 and torch.equal(p0, p1)
    assert c0 == u0 == c1 == u1 == 49, "the device's ring counter and the host's count of issued updates stay in step"


RAINBOW = {"learner.double_q": "true", "learner.dueling_head": "true", "learner.noisy_net": "true", "learner.n_step_q": 3, "replay.policy": "prioritize"}


@pytest.mark.parametrize("algo,extra", [("dqn", {}), ("dqn", {"learner.double_q": "true", "learner.dueling_head": "true", "learner.n_step_q": 3}),
                                        ("dqn", {"replay.policy": "prioritize", "learner.n_step_q": 3, "env_task": "block"}), ("c51", {}), ("c51", RAINBOW),
                                        ("c51", {**RAINBOW, "env_task": "block", "actor.sample_steps": 10, "learner.reset_noise_freq": 3}),
                                        ("iqn", {"env_id": "Asterix"}), ("iqn", {"env_id": "Asterix", "learner.double_q": "true", "learner.dueling_head": "true", "learner.n_step_q": 3,
                                                                               "replay.policy": "prioritize", "env_task": "block"}),
                                        ("fqf", {"env_id": "Asterix"}), ("fqf", {"env_id": "Asterix", "learner.double_q": "true", "learner.dueling_head": "true", "learner.n_step_q": 3,
                                                                               "replay.policy": "prioritize", "env_task": "block"}),
                                        ("qr", RAINBOW), ("mdqn", {"learner.noisy_net": "true", "learner.dueling_head": "true", "env_task": "block"})],
                         ids=["dqn-uniform", "duel-double-n3", "prioritized-n3-block", "c51", "rainbow-lite", "rainbow-lite-block-noise3", "iqn", "iqn-duel-double-n3-per-block", "fqf",
                              "fqf-duel-double-n3-per-block", "qr-rainbow", "mdqn-noisy-duel-block"])
def test_native_handles_run_the_loop_like_the_python_trainer(algo, extra):
    """Round 4 (opaque handles, library-owned HBM): ``a0_actor`` / ``a0_rbuf`` / ``a0_learner`` (csrc/runtime.hip, learner.hip) restate the host-side
    bookkeeping of the Python classes — cursors, shuffled epochs, Philox offsets, beta, epsilon — in C++, so that a host needs a handful of C calls per iteration.
    Driven here through ctypes with the loop of trainer.py:74-119,171-184 written out, from the Python Trainer's initial weights: after eight iterations (ring
    wrap, target syncs, uniform and sum-tree replay, n-step 1 and 3, both reward tasks) the replay ring, the sum-tree, max_p, the parameters, the target, the update
    counter, every episode return and every per-step max-Q must be BIT-identical to the Python Trainer's."""
    import ctypes as C
    from agent0_amd import _abi
    from agent0_amd.deepq.trainer import Trainer, epsilon_schedule

    class RbufDesc(C.Structure):
        _fields_ = [("size", C.c_longlong), ("obs_bytes", C.c_int), ("B", C.c_int), ("prioritize", C.c_int), ("alpha", C.c_double), ("eps", C.c_double), ("beta0", C.c_double),
                    ("total_steps", C.c_longlong), ("seed", C.c_ulonglong)]

    class ActorDesc(C.Structure):
        _fields_ = [("E", C.c_int), ("T", C.c_int), ("A", C.c_int), ("dueling", C.c_int), ("n_step", C.c_int), ("discount", C.c_double), ("seed", C.c_ulonglong), ("rank", C.c_uint),
                    ("env_task", C.c_int), ("reset_noise_freq", C.c_int)]

    class Batch(C.Structure):
        _fields_ = [(n, C.c_void_p) for n in ("idx", "slot", "act", "rew", "done", "prio", "weights")]

    E, T, B, SIZE, LS, START, TF = 8, 10, 32, 200, 5, 100, 7
    cfg = make_cfg(algo, E, **{"actor.sample_steps": T, "replay.size": SIZE, "learner.batch_size": B, "learner.learner_steps": LS, "trainer.training_start_steps": START,
                                "learner.target_update_freq": TF, "trainer.exploration_steps": 300, **extra})
    tr = Trainer(cfg)
    eng = tr.learner.engine
    lib, ok = _abi.load(), _abi.check
    st = torch.cuda.current_stream().cuda_stream
    prio, duel, dq, n = cfg.replay.policy.name == "prioritize", bool(cfg.learner.dueling_head), bool(cfg.learner.double_q), int(cfg.learner.n_step_q)
    noisy = bool(cfg.learner.noisy_net)
    A = int(cfg.action_dim)
    nat = tr.ops.native_learner(A=A, dueling=duel, double_q=dq, B=B, n_step=n, discount=cfg.learner.discount, lr=cfg.learner.learning_rate, target_update_freq=TF, algo=algo,
                                num_atoms=cfg.learner.qr.num_atoms if algo == "qr" else cfg.learner.c51.num_atoms, vmin=cfg.learner.c51.vmin, vmax=cfg.learner.c51.vmax, noisy=noisy,
                                seed=cfg.seed + 15485863, K=cfg.learner.iqn.K, N=cfg.learner.iqn.N, N_dash=cfg.learner.iqn.N_dash, F=cfg.learner.iqn.F,
                                mdqn_tau=cfg.learner.mdqn.tau, mdqn_lo=cfg.learner.mdqn.lo)
    nat.set_params(eng.online.flat, eng.target.flat)
    rd = RbufDesc(SIZE, 4 * 84 * 84, B, int(prio), cfg.replay.alpha, cfg.replay.eps, cfg.replay.beta0, cfg.trainer.total_steps, cfg.seed + 104729)
    rb = C.c_void_p()
    ok(lib.a0_rbuf_create(C.addressof(rd), C.addressof(rb)), "a0_rbuf_create")
    ad = ActorDesc(E, T, A, int(duel), n, cfg.learner.discount, cfg.seed, 0, {"stream": 0, "block": 1}[cfg.env_task], int(cfg.learner.reset_noise_freq))
    ac = C.c_void_p()
    ok(lib.a0_actor_create(C.addressof(ad), C.addressof(ac)), "a0_actor_create")
    eps_fn = epsilon_schedule(cfg)
    loss_dev, state_dev = tr.ops.empty(B), tr.ops.zeros(8, dtype=torch.int32)
    frame_count, n_rs, n_qs, n_Ls = 0, [], [], []
    qs_h, rs_h, nret = (C.c_float * T)(), (C.c_float * (T * E))(), C.c_int()
    for it in range(8):
        res = tr.run_iteration()
        # ---- the same iteration through the handles (trainer.py:176-182 -> 74-119)
        ok(lib.a0_actor_rollout(ac, nat.h, rb, C.c_float(eps_fn(frame_count)), st), "a0_actor_rollout")
        ok(lib.a0_rbuf_commit(rb, T * E, st), "a0_rbuf_commit")
        frame_count += T * E
        if lib.a0_rbuf_len(rb) > START:
            for _ in range(LS):
                b = Batch()
                ok(lib.a0_rbuf_sample(rb, C.addressof(b), st), "a0_rbuf_sample")
                ok(lib.a0_learner_update(nat.h, _rbuf_frames(lib, rb), b.slot, 2 * 4 * 84 * 84, b.act, b.rew, b.done, b.weights, loss_dev.data_ptr(), st), "a0_learner_update")
                if prio:
                    ok(lib.a0_learner_get(nat.h, None, None, None, None, state_dev.data_ptr(), st), "a0_learner_get")
                    ok(lib.a0_rbuf_update_priority(rb, loss_dev.data_ptr(), state_dev.data_ptr(), st), "a0_rbuf_update_priority")
                n_Ls.append(float(loss_dev.mean()))
        ok(lib.a0_actor_collect(ac, qs_h, rs_h, T * E, C.addressof(nret), st), "a0_actor_collect")
        n_qs += list(qs_h)
        n_rs += list(rs_h)[: nret.value]
        assert frame_count == res["frames"]
    torch.cuda.synchronize()
    rp = tr.replay
    frames, act, rew, done = torch.empty_like(rp.frames), torch.empty_like(rp.act), torch.empty_like(rp.rew), torch.empty_like(rp.done)
    tree = torch.empty_like(rp.tree) if prio else None
    max_p = tr.ops.zeros(1)
    ok(lib.a0_rbuf_read(rb, SIZE, frames.data_ptr(), act.data_ptr(), rew.data_ptr(), done.data_ptr(), None if tree is None else tree.data_ptr(), max_p.data_ptr(), st), "a0_rbuf_read")
    on, tg, m, v, state = nat.get()
    torch.cuda.synchronize()
    assert torch.equal(frames, rp.frames) and torch.equal(act, rp.act) and torch.equal(rew, rp.rew) and torch.equal(done, rp.done), "replay ring"
    if prio:
        assert torch.equal(tree, rp.tree) and float(max_p[0]) == rp.max_p, "sum-tree / max_p"
        beta = C.c_double()
        ok(lib.a0_rbuf_info(rb, None, None, C.addressof(beta)), "a0_rbuf_info")
        assert beta.value == rp.beta
    assert torch.equal(on, eng.online.flat) and torch.equal(tg, eng.target.flat) and torch.equal(m, eng.adam_m) and torch.equal(v, eng.adam_v), "parameters / target / Adam moments"
    assert int(state[1]) == tr.learner.update_steps == 7 * LS and int(state[1]) // TF >= 4
    assert n_rs == [float(x) for x in tr.Rs], "episode returns, in the reference's order"
    assert np.array_equal(np.array(n_qs, dtype=np.float32), np.array(tr.Qs, dtype=np.float32)), "per-step mean max-Q"
    assert np.allclose(n_Ls, tr.Ls, rtol=1e-6, atol=0) and len(n_Ls) == 7 * LS
    lib.a0_actor_destroy(ac); lib.a0_rbuf_destroy(rb); nat.close()


@pytest.mark.parametrize("algo,extra", [("dqn", {}), ("dqn", {"learner.double_q": "true", "learner.dueling_head": "true", "learner.n_step_q": 3, "replay.policy": "prioritize"}),
                                        ("c51", {"env_task": "block"}), ("c51", RAINBOW), ("c51", {**RAINBOW, "learner.reset_noise_freq": 3, "env_task": "block"}),
                                        ("iqn", {"env_id": "Asterix", "env_task": "block"}),
                                        ("iqn", {"env_id": "Asterix", "learner.double_q": "true", "learner.dueling_head": "true", "learner.n_step_q": 3, "replay.policy": "prioritize"}),
                                        ("fqf", {"env_id": "Asterix", "env_task": "block"}),
                                        ("fqf", {"env_id": "Asterix", "learner.double_q": "true", "learner.dueling_head": "true", "learner.n_step_q": 3, "replay.policy": "prioritize"}),
                                        ("qr", {"learner.double_q": "true", "env_task": "block"}), ("mdqn", {"learner.n_step_q": 3, "replay.policy": "prioritize"}),
                                        # round 5: NoisyNet on scalar and quantile heads; the fqf fraction net's gradient clipping (agent.py:143-147)
                                        ("dqn", {"learner.noisy_net": "true", "learner.dueling_head": "true", "env_task": "block"}),
                                        ("iqn", {"env_id": "Asterix", "learner.noisy_net": "true", "learner.double_q": "true", "learner.reset_noise_freq": 5}),
                                        ("fqf", {"env_id": "Asterix", "learner.noisy_net": "true", "learner.dueling_head": "true", "learner.max_grad_norm": 0.05, "replay.policy": "prioritize"}),
                                        # round 5: the launch schedule through the handles (the actor's own network snapshot, the stage ring, the second stream)
                                        ("dqn", {"_launch": True}), ("c51", {**RAINBOW, "_launch": True, "env_task": "block"}),
                                        ("c51", {**RAINBOW, "_launch": True, "learner.reset_noise_freq": 5}),
                                        ("iqn", {"env_id": "Asterix", "learner.n_step_q": 3, "_launch": True}), ("qr", {"learner.double_q": "true", "replay.policy": "prioritize", "_launch": True}),
                                        # round 6: the reference-faithful flat priority vector (replay.sumtree=false: a0_rbuf_desc.prioritize == 2) ...
                                        ("dqn", {"replay.policy": "prioritize", "replay.sumtree": "false", "learner.n_step_q": 3}), ("c51", {**RAINBOW, "replay.sumtree": "false", "env_task": "chase"}),
                                        ("qr", {"replay.policy": "prioritize", "replay.sumtree": "false", "_launch": True}),
                                        # ... and data parallelism through the learner handle (a one-rank RCCL group: the exchange is the identity, every launch of it is issued)
                                        ("dqn", {"_dp": True}), ("c51", {**RAINBOW, "_dp": True}), ("fqf", {"env_id": "Asterix", "replay.policy": "prioritize", "_dp": True})],
                         ids=["dqn", "dqn-duel-double-n3-per", "c51-block", "rainbow-lite", "rainbow-lite-noise3-block", "iqn-block", "iqn-duel-double-n3-per", "fqf-block",
                              "fqf-duel-double-n3-per", "qr-double-block", "mdqn-n3-per", "dqn-noisy-duel-block", "iqn-noisy-double", "fqf-noisy-duel-clip-per",
                              "launch-dqn", "launch-rainbow-lite-block", "launch-rainbow-lite-noise5", "launch-iqn-n3", "launch-qr-double-per",
                              "dqn-n3-flat-per", "rainbow-lite-flat-per-chase", "launch-qr-flat-per", "dp-dqn", "dp-rainbow-lite", "dp-fqf-per"])
def test_native_loop_equals_the_python_classes(algo, extra, monkeypatch):
    """agent0_amd/deepq/native_loop.py: for the configurations the handles cover, ``Trainer.run_iteration`` hands the loop to a0_actor / a0_rbuf / a0_learner created
    OVER the Python classes' own buffers (a0_learner_create_on / a0_rbuf_create_on) — one C call per rollout, batch and update, eager launches from native code.
    The same launches in the same order: after ten iterations (with and without the next rollout issued ahead, ring wrap, target syncs, a rollout booked by
    final()) every statistic, the parameters, the target, the Adam moments, the status words, the replay ring, the sum-tree, max_p and beta must be BIT-identical
    to the run of the Python classes — and the Python views (``state_dict()``, ``len(replay)``, ``replay.tree``) must show the live data without a copy."""
    from agent0_amd.deepq.native_loop import NativeLoop
    from agent0_amd.deepq.trainer import Trainer

    extra = dict(extra)
    launch = bool(extra.pop("_launch", False))
    dp = bool(extra.pop("_dp", False))
    if dp:                                         # A0_DP_FORCE=1: a one-rank RCCL process group in this process (127.0.0.1), closed again at the end of the test
        import socket
        import torch.distributed as dist
        from agent0_amd.deepq.dist import init_process_group, make_grad_hook
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
        for k, v in (("A0_DP_FORCE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", port)):
            monkeypatch.setenv(k, v)
        init_process_group()

    def run(native):
        monkeypatch.setenv("A0_NATIVE_LOOP", "1" if native else "0")
        cfg = make_cfg(algo, 8, **{"actor.sample_steps": 12, "replay.size": 400, "learner.batch_size": 32, "learner.learner_steps": 5, "trainer.training_start_steps": 100,
                                    "learner.target_update_freq": 7, "trainer.exploration_steps": 600, **extra})
        tr = Trainer(cfg, use_lp=launch)
        hook = None
        if dp:                                     # what bench.py / launch.py do for N > 1
            eng0 = tr.learner.engine
            hook = eng0.grad_hook = make_grad_hook(tr.ops, eng0.L.n_adam)
            assert type(hook).__name__ == "RcclGradAllReduce" and hook.active
        res = []
        for i in range(10):
            res.append({k: v for k, v in tr.run_iteration(prefetch=(i % 3 != 1)).items() if k != "fps"})
        assert isinstance(tr._nl, NativeLoop) if native else tr._nl is False, getattr(tr, "native_loop_reason", None)
        sd = {k: v.clone() for k, v in tr.learner.model.state_dict().items()}
        eng, rp = tr.learner.engine, tr.replay
        n_len = len(rp)
        tree = (rp.tree if rp.use_sumtree else rp.priority).clone() if rp.prioritize else torch.zeros(1)      # (flat mode: the priority vector)
        out = (res, list(tr.Ls) + list(tr.FLs), list(tr.Qs), list(tr.Rs), tr.frame_count, n_len, rp.written, float(rp.beta) if rp.prioritize else 0.0, rp.max_p if rp.prioritize else 1.0,
               eng.online.flat.clone(), eng.target.flat.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.state.clone(), rp.frames.clone(), rp.act.clone(), rp.rew.clone(),
               rp.done.clone(), tree, sd)
        if dp:
            calls = hook.report()                  # collective; the exchange really ran: two all-reduce launches per update through this communicator
            assert calls["nranks"] == 1 and calls["allreduce_of_ones"] == 1.0
            if native:
                tr._nl.detach_exchange()
        tr.test = lambda: None
        tr.final(save=False)                       # books the rollout issued ahead, closes the handles
        if hook is not None:
            hook.close()
        return out + (tr.frame_count, list(tr.Qs), len(rp))

    try:
        a = run(False)
        b = run(True)
    finally:
        if dp:
            dist.destroy_process_group()
    assert len(a[1]) == 9 * 5 * (2 if algo == "fqf" else 1) and a[4] == 10 * 96
    for i, (x, y) in enumerate(zip(a, b)):
        if isinstance(x, torch.Tensor):
            assert torch.equal(x, y), f"item {i}"
        elif isinstance(x, dict):
            assert x.keys() == y.keys() and all(torch.equal(x[k], y[k]) for k in x), f"item {i}: state_dict()"
        else:
            assert x == y, f"item {i}"


@pytest.mark.parametrize("algo,extra", [("dqn", {}), ("c51", RAINBOW)], ids=["dqn-configs1", "rainbow-lite-configs2"])
def test_native_loop_equals_the_python_classes_at_the_benchmarked_size(algo, extra, monkeypatch):
    """the comparison above at the size bench.py times — 256 envs x 80 steps per rollout, batch 512, 20 updates per block — on a ring of
    81 920 rows = 4.6 GB, so that slot * 56 448 passes 2^32 for most of the ring; five iterations of 20 updates (three target syncs); the rollout issued ahead of the last one — the sixth, booked by final() — wraps the ring.
    The whole ring, the sum-tree, parameters, target, Adam moments, status words and every statistic must be BIT-identical between the library-handle loop and the
    Python classes (which tests/test_gpu_trace.py holds to the oracle link by link)."""
    from agent0_amd.deepq.native_loop import NativeLoop
    from agent0_amd.deepq.trainer import Trainer

    def run(native):
        monkeypatch.setenv("A0_NATIVE_LOOP", "1" if native else "0")
        cfg = make_cfg(algo, 256, **{"actor.sample_steps": 80, "replay.size": 81920, "learner.batch_size": 512, "learner.learner_steps": 20, "trainer.training_start_steps": 20000,
                                      "learner.target_update_freq": 30, **extra})
        tr = Trainer(cfg)
        res = [{k: v for k, v in tr.run_iteration(prefetch=(i % 2 == 0)).items() if k != "fps"} for i in range(5)]
        assert isinstance(tr._nl, NativeLoop) if native else tr._nl is False, getattr(tr, "native_loop_reason", None)
        tr.test = lambda: None
        tr.final(save=False)                       # books the rollout issued ahead (the one that wraps the ring), closes the handles
        torch.cuda.synchronize()
        eng, rp = tr.learner.engine, tr.replay
        assert rp.written == 6 * 20480 and len(rp) == 81920
        tree = rp.tree if rp.prioritize else torch.zeros(1)
        return (res, list(tr.Ls), list(tr.Qs), list(tr.Rs), tr.frame_count, float(rp.beta) if rp.prioritize else 0.0, rp.max_p if rp.prioritize else 1.0, eng.online.flat, eng.target.flat,
                eng.adam_m, eng.adam_v, eng.state, rp.frames, rp.act, rp.rew, rp.done, tree), tr

    a, tr_a = run(False)
    b, tr_b = run(True)
    assert len(a[1]) == 5 * 20 and int(a[11][1]) == 100 and a[4] == 6 * 20480
    assert a[12].numel() == 81920 * 56448 > (1 << 32)
    for i, (x, y) in enumerate(zip(a, b)):
        if isinstance(x, torch.Tensor):
            assert torch.equal(x, y), f"item {i}"
        else:
            assert x == y, f"item {i}"


def test_native_loop_resumes_from_a_checkpoint(tmp_path, monkeypatch):
    """A run under the native loop saves the reference-keyed checkpoint from the buffers the handles work on (no copy back), and a fresh Trainer that loaded it —
    frame counter, weights, Adam moments, update counter restored — hands ITS loop to the handles too and continues: epsilon from the restored frame count, the
    update counter counting on, finite losses."""
    from agent0_amd.deepq.native_loop import NativeLoop
    from agent0_amd.deepq.trainer import Trainer
    monkeypatch.setenv("A0_NATIVE_LOOP", "1")
    base = {"actor.sample_steps": 10, "replay.size": 400, "learner.batch_size": 32, "learner.learner_steps": 3, "trainer.training_start_steps": 50, "logdir": str(tmp_path / "run"),
            **RAINBOW}
    tr = Trainer(make_cfg("c51", 8, **base))
    for _ in range(4):
        tr.run_iteration(prefetch=True)
    assert isinstance(tr._nl, NativeLoop) and tr.learner.update_steps == 12
    tr.test = lambda: None
    tr.final(save=False)
    path = tr.save_checkpoint(str(tmp_path / "ck.pth"))
    tr2 = Trainer(make_cfg("c51", 8, **{**base, "seed": 7}))
    tr2.load_checkpoint(path)
    e1, e2 = tr.learner.engine, tr2.learner.engine
    assert torch.equal(e1.online.flat, e2.online.flat) and torch.equal(e1.adam_v, e2.adam_v) and tr2.frame_count == tr.frame_count == 5 * 80
    res = [tr2.run_iteration(prefetch=(i != 2)) for i in range(3)]
    assert isinstance(tr2._nl, NativeLoop), getattr(tr2, "native_loop_reason", None)
    assert tr2.learner.update_steps == 12 + 3 * 3, "the update counter counts on from the checkpoint's"
    assert res[-1]["frames"] == 8 * 80 and np.isfinite(res[-1]["loss"]) and not torch.equal(e1.online.flat, e2.online.flat)
    tr2.test = lambda: None
    tr2.final(save=False)


def _rbuf_frames(lib, rb):
    import ctypes as C
    p = C.c_void_p()
    _ = lib.a0_rbuf_buffers(rb, C.addressof(p), None, None, None, None, None)
    return p


def _python_run_fingerprint(config, iters, size):
    """tests/c_host_loop.c's run, by the Python Trainer (library-handle host loop) from the same LCG weights and seeds; its fingerprint as the C host prints it."""
    from agent0_amd.deepq.native_loop import NativeLoop
    from agent0_amd.deepq.trainer import Trainer
    keep = os.environ.get("A0_NATIVE_LOOP")
    os.environ["A0_NATIVE_LOOP"] = "1"
    try:
        cfg = make_cfg("c51" if config == 2 else "dqn", 256, **{"actor.sample_steps": 80, "replay.size": size, "learner.batch_size": 512, "learner.learner_steps": 20,
                                                                  "trainer.training_start_steps": size // 2, "env_task": "block", **(RAINBOW if config == 2 else {})})
        tr = Trainer(cfg)
        eng = tr.learner.engine
        n = eng.L.n_params_padded
        # c_host_loop.c: x_{i+1} = x_i * 1664525 + 1013904223 (mod 2^32) from x_0 = 7; weight i = ((float)((x_{i+1} >> 8) % 2001) - 1000) * 2e-5f.  Closed form in wrapping
        # uint32 arithmetic: x_i = a^i x_0 + c (1 + a + ... + a^(i-1))
        with np.errstate(over="ignore"):
            apow = np.multiply.accumulate(np.full(n, 1664525, dtype=np.uint32), dtype=np.uint32)                     # a^1 .. a^n
            geo = np.concatenate([np.ones(1, np.uint32), apow[:-1]]).cumsum(dtype=np.uint32)          # 1 + a + ... + a^(i-1), i = 1 .. n
            x = apow * np.uint32(7) + np.uint32(1013904223) * geo
        hp = (((x >> np.uint32(8)) % np.uint32(2001)).astype(np.float32) - np.float32(1000.0)) * np.float32(2e-5)
        eng.online.flat.copy_(torch.from_numpy(hp))
        eng.online.refresh_wt()
        eng.sync_target(force=True)
        for _ in range(iters):
            tr.run_iteration()
        assert isinstance(tr._nl, NativeLoop), getattr(tr, "native_loop_reason", None)
        torch.cuda.synchronize()
        rp = tr.replay

        def fold(acc, t):
            w = t.contiguous().view(-1).view(torch.uint8).cpu().numpy().view(np.uint32).astype(np.uint64)
            pos = np.arange(acc[2] + 1, acc[2] + 1 + w.size, dtype=np.uint64)
            return [int((np.uint64(acc[0]) + w.sum(dtype=np.uint64)) & np.uint64(0xFFFFFFFFFFFFFFFF)), int((np.uint64(acc[1]) + (w * pos).sum(dtype=np.uint64)) & np.uint64(0xFFFFFFFFFFFFFFFF)),
                    acc[2] + w.size]
        with np.errstate(over="ignore"):
            a = [0, 0, 0]
            for t in (eng.online.flat, eng.target.flat, eng.adam_m, eng.adam_v):
                a = fold(a, t)
            b = [0, 0, 0]
            for t in (rp.act, rp.rew, rp.done):
                b = fold(b, t)
            fr = rp.frames.view(rp.size, -1)
            for row in range(0, size, 997):
                b = fold(b, fr[row])
        tr.test = lambda: None
        tr.final(save=False)
        return {"learner": a[:2], "ring": b[:2]}
    finally:
        if keep is None:
            os.environ.pop("A0_NATIVE_LOOP", None)
        else:
            os.environ["A0_NATIVE_LOOP"] = keep


@pytest.mark.parametrize("config", [1, 2, 3, 4])
def test_plain_c_host_runs_baseline_config1(tmp_path, config):
    """tests/c_host_loop.c: BASELINE configs[1]'s workload (256 envs x 80 steps + 20 updates of batch 512 per iteration; a 40 000-slot ring here) — and configs[2]'s
    (c51 rainbow-lite on prioritized replay) configs[3]'s (Asterix-shaped iqn) and configs[4]'s per-GPU share (fqf) — driven from plain C through the a0_actor / a0_rbuf / a0_learner handles — no Python, no torch in the process.
    Compiled with gcc against the in-tree library, run as a child."""
    import json, os, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no C compiler / ROCm headers on this box")
    exe, lib = str(tmp_path / "c_host_loop"), os.path.join(root, "agent0_amd", "lib")
    r = subprocess.run([gcc, "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "tests", "c_host_loop.c"), "-I/opt/rocm/include", "-I", os.path.join(root, "include"), "-L", lib,
                        "-lagent0_hip", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "12" if config < 3 else "4", "40000", "1", str(config)], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(x) for x in r.stdout.strip().splitlines() if x.startswith("{")]
    out = next(x for x in lines if "host" in x)
    print(out)
    it = 12 if config < 3 else 4
    assert out["iterations_timed"] == it and out["updates"] == it * 20 and out["finite"] == 1 and out["episodes"] > 100
    record_stats(f"c_host_loop_config{config}", out)
    if config in (1, 2):
        # the C host's run against the Python run of the same seed — the LCG weights c_host_loop.c starts from loaded into a Trainer of the same
        # configuration (library-handle loop), the same twelve iterations: parameters, target, Adam moments, the ring's actions / n-step rewards / dones and every 997th
        # row's bytes must have the same position-weighted word sums
        fp = next(x for x in lines if "fingerprint" in x)["fingerprint"]
        assert fp == _python_run_fingerprint(config, it, 40000), "plain C host vs Python run of the same seed"
    if config in (1, 2):
        # (e) without Python: the same host with a ONE-rank RCCL communicator in the learner handle (a0_dp_unique_id / a0_dp_init / a0_learner_set_exchange) — every update
        # all-reduces its two gradient buckets; a one-rank sum is the identity, so the run must end on the same numbers
        r = subprocess.run([exe, "12", "40000", "1", str(config), "1"], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0, r.stdout + r.stderr
        dp = next(json.loads(x) for x in r.stdout.strip().splitlines() if x.startswith("{") and "host" in x)
        print(dp)
        assert "a0_learner_set_exchange" in dp["gradient_exchange"] and out["gradient_exchange"] == "none"
        for k in ("updates", "episodes", "mean_return", "last_mean_loss", "last_qmax", "finite", "frames"):
            assert dp[k] == out[k], (k, dp[k], out[k])
