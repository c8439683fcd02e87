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


def assert_mostly_close(got, want, atol, max_bad_frac, max_rel_l2, what=""):
    """Robust comparison for chaotic paths (FQF: q(tau) runs through cos(pi*64*tau), so ulp-level differences in tau can
    flip an isolated ReLU and change a handful of elements by O(1)): bounds the FRACTION of elements outside ``atol``
    and the relative L2 error instead of the worst element."""
    def _np(x):
        if hasattr(x, "detach"):
            x = x.detach().cpu().numpy()
        return np.asarray(x, dtype=np.float64)

    got, want = _np(got), _np(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    bad = np.abs(got - want) > atol
    frac = bad.mean() if bad.size else 0.0
    rel = np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30)
    assert frac <= max_bad_frac and rel <= max_rel_l2, f"{what}: {bad.sum()}/{bad.size} elements outside {atol} (allowed {max_bad_frac:.3%}), rel-L2 {rel:.3e} (allowed {max_rel_l2})"


def record_stats(name, payload):
    """Measured statistics of a GPU test, for profiles/ (gpurun merges gpurun_out/ back): one JSON file per statistic and case."""
    import json
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "test_stats")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as f:
            json.dump(payload, f, indent=1, sort_keys=True)
    except OSError:
        pass


def pinned_oracle(case: str):
    """The oracle's FQF steps of G6 case ``case`` computed in a child process in the CPU-independent mode (tests/golden/pinned.py, tests/pinned_oracle.py)."""
    import os, sys, tempfile
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import pinned
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "out.npz")
        r = pinned.run([os.path.join(here, "pinned_oracle.py"), case, path], capture_output=True, text=True)
        assert r.returncode == 0, f"pinned oracle child failed:\n{r.stdout}\n{r.stderr}"
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
