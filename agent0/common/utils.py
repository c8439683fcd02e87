"""Alias of agent0_amd.common.utils (same public names as the reference's agent0/common/utils.py)."""
from agent0_amd.common import utils as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
