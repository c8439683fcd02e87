"""Alias of agent0_amd.deepq.trainer (same public names as the reference's agent0/deepq/trainer.py)."""
from agent0_amd.deepq.trainer import *  # noqa: F401,F403
from agent0_amd.deepq import trainer as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

if __name__ == "__main__" and hasattr(_impl, "main"):
    _impl.main()
