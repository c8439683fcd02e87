"""``python -m agent0.deepq.main [key=value ...]`` — single-process entry point.

Mirrors /root/reference agent0/deepq/main.py:16-41: builds the run directory name
``<name>-<env>-<algo>-<seed>-<sha>-<time>-<uuid>`` (18-24,30), probes the env for ``obs_shape`` / ``action_dim``
(25-32), seeds, and runs ``Trainer(cfg).run()``.  When Hydra is importable the reference's own mechanism is used —
``ConfigStore.store(name="config", node=ExpConfig)`` + ``@hydra.main(version_base=None, config_name="config")`` (main.py:16,38-41), the
DictConfig converted by ``config.from_dict`` where the reference calls dacite (main.py:28); otherwise (this image has neither Hydra
nor dacite, gitpython or shortuuid) the same dotted overrides are parsed by ``config.parse_overrides`` and the git sha / uuid fall
back to stdlib.
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


def build_config(argv, subdir: str | None = None, cfg: ExpConfig | None = None) -> ExpConfig:
    cfg = parse_overrides(argv) if cfg is None else cfg
    subdir = subdir or run_subdir(cfg)
    dummy_env = make_atari(cfg.env_id, num_envs=1)
    dummy_env.close()
    cfg.logdir = os.path.join(cfg.logdir, subdir)
    cfg.obs_shape = tuple(dummy_env.observation_space.shape[1:])
    cfg.action_dim = int(dummy_env.action_space[0].n)
    return cfg


def _run(cfg: ExpConfig):
    from .trainer import Trainer

    set_random_seed(cfg.seed)
    Trainer(cfg).run()


def hydra_available() -> bool:
    try:
        import hydra  # noqa: F401
        from hydra.core.config_store import ConfigStore  # noqa: F401
        from omegaconf import OmegaConf  # noqa: F401
    except ImportError:
        return False
    return True


def main(argv=None):
    if argv is None and hydra_available():
        # the reference's path (main.py:16,38-41): Hydra composes the structured config from the command line
        import hydra
        from hydra.core.config_store import ConfigStore
        from omegaconf import OmegaConf

        from .config import from_dict

        ConfigStore.instance().store(name="config", node=ExpConfig)

        @hydra.main(version_base=None, config_name="config")
        def hydra_main(hcfg):
            plain = hcfg if isinstance(hcfg, dict) else OmegaConf.to_container(hcfg, resolve=True)
            _run(build_config([], cfg=from_dict(plain)))

        return hydra_main()
    _run(build_config(sys.argv[1:] if argv is None else argv))


if __name__ == "__main__":
    main()
