"""Helpers shared by the parity tests."""
from __future__ import annotations

import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)


def assert_close(got, want, rtol, atol, what=""):
    def _np(x):
        if hasattr(x, "detach"):
            x = x.detach().cpu().numpy()
        return np.asarray(x, dtype=np.float64)

    got, want = _np(got), _np(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {bad.sum()}/{bad.size} outside tol (rtol={rtol}, atol={atol}); worst at {i}: got {got[i]!r} want {want[i]!r}")
