"""Alias of agent0_amd.common.atari_wrappers (same public names as the reference's agent0/common/atari_wrappers.py)."""
from agent0_amd.common import atari_wrappers as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
