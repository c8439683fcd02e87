"""Alias of agent0_amd.summary (same module path as the reference's agent0/summary.py)."""
from agent0_amd.summary import *  # noqa: F401,F403
from agent0_amd import summary as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

if __name__ == "__main__":
    raise SystemExit(_impl.main())
