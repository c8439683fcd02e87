"""Picklable ``make_slice`` callables for the env-pool tests (worker processes import this module by name)."""
import functools

from oracle import core


def _synth(seed, rank, e0, k):
    return core.SynthVecEnv(k, seed=seed, rank=rank, env_offset=e0)


def synth_slice(seed=42, rank=0):
    """envs [e0, e0 + k) of the oracle's CPU twin of the synthetic vector env."""
    return functools.partial(_synth, seed, rank)
