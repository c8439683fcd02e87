"""Diagnostics (GPU box): per-tensor gradient error of the HIP update and of the torch-fp32 oracle against a float64 evaluation of the
same step, at B = 512.  Usage: python tools/diag_grad_accuracy.py [dqn|iqn|fqf ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch

import recipe
import test_engine_emul as E
from agent0_amd.ops import HipOps
from oracle.losses import Hyper

ops = HipOps()
for algo in (sys.argv[1:] or ["dqn", "iqn", "fqf"]):
    A = 4 if algo == "dqn" else 9
    spec = recipe.NetSpec(algo, A)
    hp = Hyper(double_q=(algo == "iqn"), n_step=1, K=32, N=64, N_dash=64)
    spec, L, ora, dev, results = E.run_both(ops, algo, 512, hp.double_q, 1, steps=1, target_freq=1, spec=spec, hp=hp, arbiter=True)
    res_o, out, g_o, g_d, got, tgt, want_p, want_t, g64 = results[0]
    g_ref = L.unpack(g_d)
    print(f"== {algo}: key, scale, |hip-fp64|/scale, |torch32-fp64|/scale, |hip-torch32|/scale")
    for k, g in g_o.items():
        if g is None:
            continue
        s = float(g64[k].abs().max()) + 1e-30
        print(f"{k:32s} {s:10.3e} {float((g_ref[k].double().cpu() - g64[k]).abs().max()) / s:10.3e} {float((g.double() - g64[k]).abs().max()) / s:10.3e} "
              f"{float((g_ref[k].cpu() - g).abs().max()) / s:10.3e}")
