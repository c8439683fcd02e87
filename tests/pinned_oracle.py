"""The oracle's FQF train steps in the CPU-independent mode of tests/golden/pinned.py — TEST INFRASTRUCTURE, run as a CHILD process (the mode's environment
variables must be set before torch is imported):  python tests/golden/pinned.py tests/pinned_oracle.py <case> <out.npz>

Writes, per step s: the per-sample losses, the proposed fractions of the step's two prop_taus calls (what the HIP path is fed, tests/test_gpu_engine.py), the
fingerprints recipe.checksum of every gradient / parameter / target tensor, and the FULL post-step parameters, target parameters and Adam moments (the state the
device continues from, so that every step is compared from a common state)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")]
import numpy as np
import torch

import pinned

pinned.configure()
import recipe
from oracle import learner as olearner, nets
from oracle.losses import Hyper


def main(case: str, out_path: str, steps: int = 3):
    name, b, dq, n = case.rsplit("_", 3)
    B, dq, n = int(b[1:]), bool(int(dq[2:])), int(n[1:])
    spec = recipe.SPECS[name]
    ora = olearner.OracleLearner(spec, recipe.make_state_dict(spec, 11), recipe.make_state_dict(spec, 12), Hyper(double_q=dq, n_step=n), batch_size=B, target_update_freq=2)
    out = {}
    for s in range(steps):
        frames = recipe.make_frames(B, 61 + s, spec.obs_shape)
        a, r, d, w = recipe.make_transitions(B, spec.action_dim, 62 + s)
        nets.TAU_LOG = []
        res = ora.train(frames.reshape(B, -1), a, r, d.astype(np.float32), w, np.arange(B))
        taus, nets.TAU_LOG = nets.TAU_LOG, None
        out[f"s{s}::q_loss"], out[f"s{s}::fraction_loss"] = res["q_loss"].numpy(), res["fraction_loss"].numpy()
        for i, (t, th) in enumerate(taus):
            out[f"s{s}::taus_{i}"], out[f"s{s}::tau_hats_{i}"] = t.numpy(), th.numpy()
        for k, v in ora.last_grads.items():
            if v is not None:
                out[f"s{s}::grad::{k}"] = recipe.checksum(v.numpy())
        for k, v in ora.po.items():
            out[f"s{s}::param::{k}"] = recipe.checksum(v.detach().numpy())
            out[f"s{s}::po::{k}"] = v.detach().numpy().copy()
        for k, v in ora.pt.items():
            out[f"s{s}::target::{k}"] = recipe.checksum(v.detach().numpy())
            out[f"s{s}::pt::{k}"] = v.detach().numpy().copy()
        for k, v in ora.adam.m.items():
            out[f"s{s}::adam_m::{k}"] = v.numpy().copy()
        for k, v in ora.adam.v.items():
            out[f"s{s}::adam_v::{k}"] = v.numpy().copy()
        out[f"s{s}::update_steps"] = np.array(ora.update_steps)
    out["cpu_capability"] = np.array(torch.backends.cpu.get_cpu_capability())
    np.savez(out_path, **out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
