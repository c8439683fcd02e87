"""N4 and the boundary's config surface: the CSV aggregation in the column shape of the reference's summary.py:20-44,
Hydra's ConfigStore path when Hydra is importable (main.py:16,38-41), and — on the GPU — the exact tensorboard / wandb key set of
trainer.py:111-118,139-150,158-169 recorded by stub SummaryWriter / wandb modules."""
import csv
import os
import sys
import types

import numpy as np
import pytest
import torch


# ----------------------------------------------------------------------------- summary aggregation (CPU)
def _fake_run(root, exp, run, game, algo, returns, frames):
    d = os.path.join(root, exp, run)
    os.makedirs(d)
    torch.save({"ITRs": [float(x) for x in returns], "frame_count": frames, "game": game, "algo": algo, "sha": "abcdef123456", "name": exp}, os.path.join(d, "final.pth"))
    return d


def test_summary_tables_have_the_reference_shape(tmp_path):
    from agent0_amd import summary

    root = str(tmp_path)
    _fake_run(root, "expA", "r1", "Breakout", "dqn", [10, 20, 30], 1000)
    _fake_run(root, "expA", "r2", "Breakout", "c51", [50, 70], 2000)
    _fake_run(root, "expA", "r3", "Enduro", "dqn", [5, 5], 1000)
    _fake_run(root, "expA", "r4", "Enduro", "c51", [1, 3], 1000)
    _fake_run(root, "expA", "r5", "Pong", "dqn", [21], 10)                      # excluded from the ranking (summary.py:66)
    assert summary.main([root]) == 0
    rows = list(csv.reader(open(os.path.join(root, "summary.csv"))))
    assert rows[0] == ["", "exp_name", "commit", "algo", "game", "mean", "std", "max", "min", "size", "frames"]       # summary.py:20-33 (+ pandas' index column)
    by = {(r[3], r[4]): r for r in rows[1:]}
    assert len(rows) == 6 and by[("dqn", "Breakout")][5:] == ["20.0", str(np.std([10, 20, 30])), "30.0", "10.0", "3", "1000"] and by[("dqn", "Breakout")][2] == "abcdef"
    rank = list(csv.DictReader(open(os.path.join(root, "rank.csv"))))
    assert [r["game"] for r in rank] == ["Breakout", "Enduro", "avg", "final"]
    assert rank[0]["expA_c51"] == "0" and rank[0]["expA_dqn"] == "1" and rank[1]["expA_dqn"] == "0" and rank[1]["expA_c51"] == "1"
    assert float(rank[2]["expA_c51"]) == 0.5 and {rank[3]["expA_c51"], rank[3]["expA_dqn"]} == {"0", "1"}
    score = list(csv.DictReader(open(os.path.join(root, "score.csv"))))
    assert float(score[0]["expA_c51"]) == 60.0 and float(score[1]["expA_dqn"]) == 5.0
    import agent0.summary as alias
    assert alias.SUMMARY_COLUMNS == summary.SUMMARY_COLUMNS


# ----------------------------------------------------------------------------- Hydra when importable (CPU, stub modules)
def test_main_uses_hydra_config_store_when_importable(monkeypatch):
    """With ``hydra`` importable ``main()`` registers ExpConfig in the ConfigStore under "config" and runs through ``@hydra.main(
    version_base=None, config_name="config")`` (main.py:16,38-41); the composed config reaches the Trainer with the overrides applied,
    enums by name and the ``iqr`` alias included.  The stand-in modules below implement just that contract."""
    from agent0_amd.deepq import config as C, main as M

    store = {}
    hydra = types.ModuleType("hydra")
    core = types.ModuleType("hydra.core")
    cs_mod = types.ModuleType("hydra.core.config_store")

    class ConfigStore:
        _inst = None

        @classmethod
        def instance(cls):
            cls._inst = cls._inst or cls()
            return cls._inst

        def store(self, name, node):
            store[name] = node

    def hydra_main(version_base=None, config_name=None):
        assert version_base is None and config_name == "config"

        def deco(fn):
            def run():
                node = store[config_name]
                cfg = C.parse_overrides(sys.argv[1:], node())              # what Hydra's override grammar does for dotted key=value
                return fn(C.to_dict(cfg))                                   # a plain container, like OmegaConf.to_container(DictConfig)
            return run
        return deco

    cs_mod.ConfigStore = ConfigStore
    hydra.main = hydra_main
    hydra.core = core
    core.config_store = cs_mod
    oc = types.ModuleType("omegaconf")
    oc.OmegaConf = type("OmegaConf", (), {"to_container": staticmethod(lambda x, resolve=True: dict(x))})
    for name, mod in (("hydra", hydra), ("hydra.core", core), ("hydra.core.config_store", cs_mod), ("omegaconf", oc)):
        monkeypatch.setitem(sys.modules, name, mod)
    assert M.hydra_available()
    seen = {}
    monkeypatch.setattr(M, "_run", lambda cfg: seen.setdefault("cfg", cfg))
    monkeypatch.setattr(M, "make_atari", lambda env_id, num_envs=1, **k: types.SimpleNamespace(
        close=lambda: None, observation_space=types.SimpleNamespace(shape=(1, 4, 84, 84)), action_space=[types.SimpleNamespace(n=9)]))
    monkeypatch.setattr(sys, "argv", ["main.py", "env_id=Asterix", "learner.algo=iqr", "learner.double_q=true", "actor.num_envs=64", "replay.policy=prioritize"])
    M.main()
    cfg = seen["cfg"]
    assert store["config"] is C.ExpConfig
    assert cfg.env_id == "Asterix" and cfg.learner.algo == C.AlgoEnum.iqn and cfg.learner.double_q is True and cfg.actor.num_envs == 64
    assert cfg.replay.policy == C.ReplayEnum.prioritize and cfg.action_dim == 9 and tuple(cfg.obs_shape) == (4, 84, 84)
    assert "Asterix-iqn" in cfg.logdir
    # without Hydra the built-in parser produces the same tree
    for name in ("hydra", "hydra.core", "hydra.core.config_store", "omegaconf"):
        monkeypatch.setitem(sys.modules, name, None)
    assert not M.hydra_available()
    seen.clear()
    M.main()
    assert C.to_dict(seen["cfg"]) | {"logdir": ""} == C.to_dict(cfg) | {"logdir": ""}


# ----------------------------------------------------------------------------- tensorboard / wandb key set (GPU: needs a Trainer)
@pytest.mark.gpu
def test_tb_and_wandb_receive_the_reference_key_set(monkeypatch, tmp_path):
    from agent0_amd.deepq.config import parse_overrides

    scalars, videos, wlogs, winit = [], [], [], []

    class SummaryWriter:
        def __init__(self, logdir):
            self.logdir = logdir

        def add_scalar(self, k, v, step):
            scalars.append((k, float(v), step))

        def add_video(self, k, v, step, fps=None):
            videos.append((k, tuple(v.shape), step, fps))

    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = SummaryWriter
    monkeypatch.setitem(sys.modules, "torch.utils.tensorboard", tb)
    wandb = types.ModuleType("wandb")
    wandb.init = lambda project=None, config=None: winit.append((project, config))
    wandb.log = lambda d: wlogs.append(d)
    wandb.Video = lambda v, fps=None, format=None: ("video", tuple(v.shape), fps, format)
    monkeypatch.setitem(sys.modules, "wandb", wandb)
    from agent0_amd.deepq.trainer import PROGRESS_COLUMNS, Trainer

    cfg = parse_overrides(["learner.algo=fqf", "actor.num_envs=8", "actor.sample_steps=10", "replay.size=400", "learner.batch_size=32", "learner.learner_steps=2",
                           "trainer.training_start_steps=50", "trainer.test_episodes=1", "wandb=true", "tb=true", f"logdir={tmp_path}/run"])
    tr = Trainer(cfg)
    assert winit and winit[0][0] == cfg.name and winit[0][1]["learner"]["algo"] == "fqf"       # trainer.py:52-53: project = cfg.name, config = the tree
    for _ in range(3):
        res = tr.run_iteration()
        tr.logging(res)
    fc = tr.frame_count
    keys = {"frames", "fraction_loss", "loss", "return_train", "return_train_max", "qmax", "fps"}      # trainer.py:111-118 + fps (180-181); None values are skipped (161-162)
    logged = {k for k, _, _ in scalars}
    assert logged <= keys and {"frames", "loss", "fraction_loss", "qmax", "fps"} <= logged
    assert all(step == s for (_, _, step), s in zip(scalars[-len(logged):], [fc] * len(logged)))
    assert all(set(d) - {"frame"} <= keys and len(d) == 2 and "frame" in d for d in wlogs)              # wandb.log({k: v, "frame": frame_count}) per key (165-166)
    n_before = len(scalars)
    tr.final(save=True)
    test_keys = [k for k, _, _ in scalars[n_before:]]
    assert test_keys == ["return_test", "return_test_max"]                                               # trainer.py:139-143
    assert videos and videos[0][0] == "test_video" and videos[0][3] == 60 and videos[0][1][0] == 4 and videos[0][1][2:] == (3, 84, 84)
    tail = wlogs[-3:]
    assert [sorted(d) for d in tail] == [["frame", "return_test"], ["frame", "return_test_max"], ["frame", "test_video"]] and tail[2]["test_video"][2:] == (60, "mp4")
    # progress.csv: one row per logged iteration, columns = the result keys; final.pth carries what the summary tables need
    rows = list(csv.DictReader(open(os.path.join(cfg.logdir, "progress.csv"))))
    assert len(rows) == 3 and tuple(rows[0].keys()) == PROGRESS_COLUMNS and int(rows[-1]["frames"]) == 240
    blob = torch.load(os.path.join(cfg.logdir, "final.pth"), weights_only=True)
    assert blob["game"] == cfg.env_id and blob["algo"] == "fqf" and len(blob["ITRs"]) >= 1 and blob["frame_count"] == 240
    from agent0_amd import summary
    assert summary.progress_tail(cfg.logdir)["frames"] == "240"
    assert len(summary.read_runs(str(tmp_path))) == 1
