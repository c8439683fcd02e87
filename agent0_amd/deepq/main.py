"""``python -m agent0.deepq.main [key=value ...]`` — single-process entry point.

Mirrors /root/reference agent0/deepq/main.py:16-41: builds the run directory name
``<name>-<env>-<algo>-<seed>-<sha>-<time>-<uuid>`` (18-24,30), probes the env for ``obs_shape`` / ``action_dim``
(25-32), seeds, and runs ``Trainer(cfg).run()``.  Hydra / dacite / gitpython / shortuuid are not installed in this
image; the same dotted overrides are parsed by ``config.parse_overrides`` and the git sha / uuid fall back to stdlib.
"""
from __future__ import annotations

import os
import subprocess
import sys
import uuid as _uuid
from time import localtime, strftime

from agent0_amd.common.atari_wrappers import make_atari
from agent0_amd.common.utils import set_random_seed
from .config import ExpConfig, parse_overrides


def _git_sha() -> str:
    try:
        return subprocess.check_output(["git", "rev-parse", "HEAD"], stderr=subprocess.DEVNULL, cwd=os.path.dirname(__file__)).decode()[:8]
    except Exception:
        return "nogit000"      # the reference requires a git checkout (quirk Q18); we do not


def run_subdir(cfg: ExpConfig) -> str:
    """main.py:18-24,30.  ``A0_RUN_SUBDIR`` (set by launch.py's parent process) makes every rank of one job use the same directory."""
    return os.environ.get("A0_RUN_SUBDIR") or _fresh_subdir(cfg)


def _fresh_subdir(cfg: ExpConfig) -> str:
    return f"{cfg.name}-{cfg.env_id}-{cfg.learner.algo.name}-{cfg.seed}-{_git_sha()}-{strftime('%Y%m%d-%H%M%S', localtime())}-{_uuid.uuid4().hex[:4]}"


def build_config(argv, subdir: str | None = None) -> ExpConfig:
    cfg = parse_overrides(argv)
    subdir = subdir or run_subdir(cfg)
    dummy_env = make_atari(cfg.env_id, num_envs=1)
    dummy_env.close()
    cfg.logdir = os.path.join(cfg.logdir, subdir)
    cfg.obs_shape = tuple(dummy_env.observation_space.shape[1:])
    cfg.action_dim = int(dummy_env.action_space[0].n)
    return cfg


def main(argv=None):
    from .trainer import Trainer

    cfg = build_config(sys.argv[1:] if argv is None else argv)
    set_random_seed(cfg.seed)
    Trainer(cfg).run()


if __name__ == "__main__":
    main()
