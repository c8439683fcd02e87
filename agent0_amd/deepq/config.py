"""Experiment configuration: the reference's ``ExpConfig`` tree, field for field, plus a Hydra-style override parser.

Mirrors /root/reference agent0/deepq/config.py:6-145 (enums 6-39, algorithm sub-configs 42-69, LearnerConfig 72-95,
TrainerConfig 98-105, ActorConfig 108-115, ReplayConfig 118-124, ExpConfig 127-145).  The reference builds the tree
with Hydra's ConfigStore + dacite (main.py:16-41); neither is installed here, so ``parse_overrides`` accepts the same
dotted ``key=value`` command-line syntax (README.md:44-53) and coerces enums / bools / numbers itself.

Additions (all default to the reference's behaviour being available):
  replay.sumtree   bool, default True — with ``replay.policy=prioritize`` draw batches proportionally from the HBM
                   sum-tree (what the north star builds).  False reproduces the reference, which despite its name
                   samples uniformly and only re-weights (quirks Q1/Q2/Q7 of SURVEY.md).
  learner.algo     accepts ``iqr`` (README spelling) as an alias of ``iqn`` (quirk Q10).
  env_task         reward task of the device-resident synthetic env (include/agent0_hip.h A0_ENV_TASK_*): ``stream`` (default; an action-independent reward
                   stream — the throughput workload) or ``block`` (learnable: +1 for naming the quadrant of the bright block in the newest frame,
                   -1 for the next class; chance 0, optimum +1 per step) or ``chase`` (temporal credit: the action moves the block on a 4 x 4 lattice, +1 only on
                   arrival at the target cell three to six moves away; optimum 0.25 per step) — what the learning tests train on.  Ignored by real Atari envs.
  device           ``cuda`` is the only supported device: this build has no CPU path (it raises instead).
  checkpoint       path of a checkpoint written by ``Trainer.save_checkpoint``; read when ``mode`` is ``finetune`` (resume training)
                   or ``play`` (evaluate only) — the reference declares those modes (config.py:26-29) but never implements them.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field, fields, is_dataclass
from enum import Enum
from typing import Any, List, Sequence, get_type_hints


class AlgoEnum(Enum):
    dqn = 0
    c51 = 1
    qr = 2
    iqn = 3
    fqf = 4
    mdqn = 5


class ActorEnum(Enum):
    greedy = 0
    random = 1
    epsilon = 2


class ReplayEnum(Enum):
    uniform = 0
    prioritize = 1


class ModeEnum(Enum):
    train = 0
    finetune = 1
    play = 2


class EnvEnum(Enum):
    atari = 0
    mujoco = 1


class DeviceEnum(Enum):
    cuda = "cuda"
    cpu = "cpu"


ALGO_ALIASES = {"iqr": "iqn"}


@dataclass
class C51Config:
    num_atoms: int = 51
    vmax: float = 10
    vmin: float = -10


@dataclass
class QRConfig:
    num_atoms: int = 200
    vmax: Any = None
    vmin: Any = None


@dataclass
class IQNConfig:
    K: int = 32
    N: int = 64
    N_dash: int = 64
    num_cosines: int = 64
    F: int = 32


@dataclass
class MDQNConfig:
    tau: float = 0.03
    alpha: float = 0.9
    lo: float = -1


@dataclass
class LearnerConfig:
    algo: AlgoEnum = AlgoEnum.dqn
    discount: float = 0.99
    batch_size: int = 512
    learning_rate: float = 5e-4
    fraction_lr: float = 2.5e-8
    max_grad_norm: float = -1.0
    target_update_freq: int = 500
    learner_steps: int = 20
    double_q: bool = False
    dueling_head: bool = False
    n_step_q: int = 1
    noisy_net: bool = False
    reset_noise_freq: int = 4
    c51: C51Config = field(default_factory=C51Config)
    qr: QRConfig = field(default_factory=QRConfig)      # the reference passes the CLASS as default (quirk Q10); instances here
    iqn: IQNConfig = field(default_factory=IQNConfig)
    mdqn: MDQNConfig = field(default_factory=MDQNConfig)


@dataclass
class TrainerConfig:
    total_steps: int = int(1e7)
    training_start_steps: int = int(1e5)
    exploration_steps: int = int(1e6)
    log_freq: int = 10
    test_freq: int = 500
    test_episodes: int = 20


@dataclass
class ActorConfig:
    policy: ActorEnum = ActorEnum.random
    num_envs: int = 16
    sample_steps: int = 80
    test_steps: int = 800
    min_eps: float = 0.01
    test_eps: float = 0.001


@dataclass
class ReplayConfig:
    size: int = int(1e6)
    policy: ReplayEnum = ReplayEnum.uniform
    beta0: float = 0.4
    alpha: float = 0.5
    eps: float = 0.01
    sumtree: bool = True


@dataclass
class ExpConfig:
    env_id: str = "Breakout"
    env_type: EnvEnum = EnvEnum.atari
    obs_shape: Any = (0,)
    action_dim: int = 0
    num_actors: int = 3
    seed: int = 42
    device: DeviceEnum = DeviceEnum.cuda
    name: str = "agent0"
    mode: ModeEnum = ModeEnum.train
    logdir: str = "logs"
    wandb: bool = True
    tb: bool = True
    checkpoint: str = ""
    env_task: str = "stream"
    learner: LearnerConfig = field(default_factory=LearnerConfig)
    trainer: TrainerConfig = field(default_factory=TrainerConfig)
    actor: ActorConfig = field(default_factory=ActorConfig)
    replay: ReplayConfig = field(default_factory=ReplayConfig)


# ----------------------------------------------------------------------------- overrides
def _coerce(raw: str, current: Any, annotation: Any):
    if isinstance(current, Enum):
        enum_t = type(current)
        name = ALGO_ALIASES.get(raw.lower(), raw) if enum_t is AlgoEnum else raw
        for member in enum_t:
            if member.name.lower() == name.lower() or str(member.value).lower() == name.lower():
                return member
        raise ValueError(f"{raw!r} is not one of {[m.name for m in enum_t]}")
    if isinstance(current, bool):
        if raw.lower() in ("true", "1", "yes", "on"):
            return True
        if raw.lower() in ("false", "0", "no", "off"):
            return False
        raise ValueError(f"{raw!r} is not a boolean")
    if isinstance(current, int) and not isinstance(current, bool):
        try:
            return int(raw)
        except ValueError:
            f = float(raw)            # e.g. total_steps=1e7
            if f != int(f):
                raise
            return int(f)
    if isinstance(current, float):
        return float(raw)
    if isinstance(current, (tuple, list)):
        inner = raw.strip("()[] ")
        vals = [int(v) for v in inner.split(",") if v.strip()]
        return tuple(vals)
    if current is None:
        if raw.lower() in ("null", "none"):
            return None
        try:
            return float(raw)
        except ValueError:
            return raw
    return raw


def apply_override(cfg: Any, dotted: str, raw: str) -> None:
    obj = cfg
    parts = dotted.split(".")
    for p in parts[:-1]:
        if not hasattr(obj, p):
            raise KeyError(f"unknown config group {p!r} in {dotted!r}")
        obj = getattr(obj, p)
    leaf = parts[-1]
    if not is_dataclass(obj) or leaf not in {f.name for f in fields(obj)}:
        raise KeyError(f"unknown config key {dotted!r}")
    setattr(obj, leaf, _coerce(raw, getattr(obj, leaf), None))


def parse_overrides(argv: Sequence[str], cfg: ExpConfig | None = None) -> ExpConfig:
    """``["env_id=Enduro", "learner.algo=c51", "actor.num_envs=256"]`` -> ExpConfig (Hydra's dotted override syntax)."""
    cfg = cfg or ExpConfig()
    for item in argv:
        if "=" not in item:
            raise ValueError(f"override {item!r} is not of the form key=value")
        key, raw = item.split("=", 1)
        apply_override(cfg, key.lstrip("+"), raw)
    return cfg


def to_dict(cfg: Any) -> dict:
    out = {}
    for f in fields(cfg):
        v = getattr(cfg, f.name)
        out[f.name] = to_dict(v) if is_dataclass(v) else (v.name if isinstance(v, Enum) else v)
    return out


def from_dict(d: dict, cfg: ExpConfig | None = None) -> ExpConfig:
    """dacite.from_dict equivalent for this tree (main.py:28)."""
    cfg = cfg or ExpConfig()

    def fill(obj, dd):
        for k, v in dd.items():
            cur = getattr(obj, k)
            if is_dataclass(cur) and isinstance(v, dict):
                fill(cur, v)
            elif isinstance(cur, Enum) and not isinstance(v, Enum):
                setattr(obj, k, _coerce(str(v), cur, None))
            else:
                setattr(obj, k, v)

    fill(cfg, d)
    return cfg
