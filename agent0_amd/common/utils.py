"""Schedules, seeding and the device RNG cursor.

LinearSchedule / set_random_seed follow /root/reference agent0/common/utils.py:12-28,77-82.  The reference's
DataLoaderX / DataPrefetcher (utils.py:31-61) have no counterpart: batches never leave HBM, so there is nothing to
prefetch.
"""
from __future__ import annotations

import random

import numpy as np
import torch


class LinearSchedule:
    """Value moves linearly from ``start`` to ``end`` over ``steps``; a call returns the value BEFORE advancing."""

    def __init__(self, start, end=None, steps=None):
        if end is None:
            end, steps = start, 1
        self.inc = (end - start) / float(steps)
        self.current = start
        self.end = end
        self._clip = min if end > start else max

    def __call__(self, steps=1):
        value = self.current
        self.current = self._clip(self.current + self.inc * steps, self.end)
        return value


def set_random_seed(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(np.random.randint(int(1e6)))


class DeviceRng:
    """Philox streams on the device (agent0_amd/csrc/rng.hip).  Every consumer owns a stream id and a running offset,
    so a run is reproducible from (seed, rank) alone and ranks never share draws."""

    STREAM_EGREEDY_U, STREAM_EGREEDY_A, STREAM_TAUS, STREAM_NOISE, STREAM_SUMTREE, STREAM_PERM = 1, 2, 3, 4, 5, 6

    CTRL_INDEX = {3: 5, 4: 6}      # STREAM_TAUS -> A0_CTRL_RNG_TAUS, STREAM_NOISE -> A0_CTRL_RNG_NOISE (include/agent0_hip.h)

    def __init__(self, ops, seed: int, rank: int = 0):
        self.ops = ops
        self.seed = (int(seed) & 0xFFFFFFFF) | ((int(rank) & 0xFFFF) << 32)
        self.offsets = {}
        self.ctrl = None      # device int64[8]: when set, taus / noise fills add ctrl[idx] to their offset (hipGraph replay, see Actor)

    def _advance(self, stream: int, n: int) -> int:
        off = self.offsets.get(stream, 0)
        self.offsets[stream] = off + ((n + 3) // 4) * 4
        return off

    def uniform(self, stream: int, out: torch.Tensor, n: int):
        if self.ctrl is not None and stream in self.CTRL_INDEX:
            self.ops.rng_uniform_ctrl(self.seed, stream, self._advance(stream, n), out, n, self.ctrl, self.CTRL_INDEX[stream])
        else:
            self.ops.rng_uniform(self.seed, stream, self._advance(stream, n), out, n)

    def uniform_cos(self, stream: int, out: torch.Tensor, cos_out: torch.Tensor, n: int, D: int):
        """``uniform`` plus the cosine features cos(pi (d + 1) out[r]), d < D, of the draws in the same launch (a0_tau_cos_features: the actor's IQN step)."""
        ctrl = self.ctrl if (self.ctrl is not None and stream in self.CTRL_INDEX) else None
        self.ops.tau_cos_features(self.seed, stream, self._advance(stream, n), out, cos_out, n, D, ctrl=ctrl, ctrl_idx=self.CTRL_INDEX.get(stream, 0))

    def randint(self, stream: int, hi: int, out: torch.Tensor, n: int):
        self.ops.rng_randint(self.seed, stream, self._advance(stream, n), hi, out, n)

    def normal(self, stream: int, std: float, out: torch.Tensor, n: int):
        if self.ctrl is not None and stream in self.CTRL_INDEX:
            self.ops.rng_normal_ctrl(self.seed, stream, self._advance(stream, n), std, out, n, self.ctrl, self.CTRL_INDEX[stream])
        else:
            self.ops.rng_normal(self.seed, stream, self._advance(stream, n), std, out, n)

    def reserve(self, stream: int, n: int) -> int:
        """Claims ``n`` draws of ``stream`` for a kernel that generates them itself; returns their offset."""
        return self._advance(stream, n)

    def next_seed32(self, stream: int) -> int:
        off = self._advance(stream, 4)
        return (self.seed * 0x9E3779B1 + off * 0x85EBCA77 + stream) & 0xFFFFFFFF
