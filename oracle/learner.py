"""Oracle: one full learner update (torch CPU fp32 + hand-written Adam/RMSprop).

Restates /root/reference agent0/deepq/agent.py:97-169 (BaseLearner.__init__/train)
and agent.py:331-338 (FQF's RMSprop on the fraction net).  The optimizers are
spelled out (rather than calling torch.optim) because the HIP ``a0_adam_step``
/ ``a0_rmsprop_step`` kernels are checked against exactly this arithmetic;
tests/test_oracle_golden.py pins them against the reference's torch.optim
results (fixture group G6).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from . import losses, nets
from .losses import Hyper


class Adam:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps) single-tensor arithmetic."""

    def __init__(self, lr: float, eps: float, b1: float = 0.9, b2: float = 0.999):
        self.lr, self.eps, self.b1, self.b2 = lr, eps, b1, b2
        self.t = 0
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}

    @torch.no_grad()
    def step(self, params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor]):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        step_size = self.lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        for k, g in grads.items():
            if g is None:
                continue
            if k not in self.m:
                self.m[k] = torch.zeros_like(g)
                self.v[k] = torch.zeros_like(g)
            m, v = self.m[k], self.v[k]
            m.add_((g - m) * (1.0 - self.b1))  # lerp form, as torch does
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            denom = (v.sqrt() / bc2_sqrt).add_(self.eps)
            params[k].addcdiv_(m, denom, value=-step_size)


class RMSprop:
    """torch.optim.RMSprop(lr, alpha, eps), no momentum, not centered."""

    def __init__(self, lr: float, alpha: float = 0.95, eps: float = 1e-5):
        self.lr, self.alpha, self.eps = lr, alpha, eps
        self.sq: Dict[str, torch.Tensor] = {}

    @torch.no_grad()
    def step(self, params, grads):
        for k, g in grads.items():
            if g is None:
                continue
            if k not in self.sq:
                self.sq[k] = torch.zeros_like(g)
            sq = self.sq[k]
            sq.mul_(self.alpha).addcmul_(g, g, value=1.0 - self.alpha)
            params[k].addcdiv_(g, sq.sqrt().add_(self.eps), value=-self.lr)


def to_params(sd_np: "OrderedDict[str, np.ndarray]") -> "OrderedDict[str, torch.Tensor]":
    p = OrderedDict()
    for k, v in sd_np.items():
        p[k] = torch.from_numpy(np.array(v, dtype=np.float32, copy=True))
    return p


def exact_gradients(spec, hp: Hyper, po, pt, frames_u8: np.ndarray, actions, rewards, terminals, weights, rand=None, taus=None):
    """The gradients of one ``train`` call evaluated in float64 on the SAME fp32 inputs (parameters, fl(x/255) observations, draws) —
    the arbiter for long reductions: at B = 512 a conv1 weight gradient sums 204 800 products per element, and two correct fp32
    evaluations (torch's blocked sums, the MFMA's k-ordered chain) differ from each other by more than they each differ from this.
    ``taus`` (FQF): the (taus, taus_hat) pairs the fp32 run used, in call order.  -> {key: float64 tensor}, per-sample loss (float64)."""
    assert not spec.noisy
    keys = nets.trainable_keys(po)
    p64 = OrderedDict((k, v.detach().double().requires_grad_(k in keys)) for k, v in po.items())
    t64 = OrderedDict((k, v.detach().double()) for k, v in pt.items())
    C = spec.obs_shape[0]
    fr = torch.from_numpy(np.ascontiguousarray(frames_u8)).float().reshape(-1, 2 * C, *spec.obs_shape[1:]).div(255.0).double()
    obs, next_obs = torch.split(fr, C, 1)
    a, r, d, w = torch.as_tensor(actions).long(), torch.as_tensor(rewards).double(), torch.as_tensor(terminals).double(), torch.as_tensor(weights).double()
    rnd = None if rand is None else [torch.as_tensor(x).double() for x in rand]
    nets.TAU_OVERRIDE = None if taus is None else [(t.double(), th.double()) for t, th in taus]
    try:
        q_loss, f_loss = losses.train_step(p64, t64, spec, hp, obs, a, r, d, next_obs, rnd)
    finally:
        nets.TAU_OVERRIDE = None
    q_keys = [k for k in keys if "fraction" not in k]
    out = dict(zip(q_keys, torch.autograd.grad((q_loss * w).sum(), [p64[k] for k in q_keys], allow_unused=True, retain_graph=f_loss is not None)))
    if f_loss is not None:
        f_keys = [k for k in keys if "fraction" in k]
        out.update(zip(f_keys, torch.autograd.grad((f_loss * w).sum(), [p64[k] for k in f_keys])))
    return out, q_loss.detach()


class OracleLearner:
    def __init__(self, spec, online_sd, target_sd, hp: Hyper, batch_size: int, lr: float = 5e-4,
                 target_update_freq: int = 500, max_grad_norm: float = -1.0):
        self.spec, self.hp, self.B = spec, hp, batch_size
        self.po = to_params(online_sd)
        self.pt = to_params(target_sd)
        self.train_keys = nets.trainable_keys(self.po)
        self.q_keys = [k for k in self.train_keys if "fraction" not in k]  # model.params(), model.py:332-333
        self.f_keys = [k for k in self.train_keys if "fraction" in k]
        for k in self.train_keys:
            self.po[k].requires_grad_(True)
        self.adam = Adam(lr, eps=1e-2 / batch_size)
        self.rms = RMSprop(lr / 2e4, alpha=0.95, eps=1e-5) if spec.algo == "fqf" else None
        self.update_steps = 0
        self.target_update_freq = target_update_freq
        self.max_grad_norm = max_grad_norm
        self.last_grads: Dict[str, torch.Tensor] = {}

    def set_noise(self, which: str, draws: List[np.ndarray]):
        """draws: per NoisyLinear in module order (first_dense, q_head, value_head):
        noise_in, noise_out_weight, noise_out_bias  (model.py:73-76)."""
        p = self.po if which == "online" else self.pt
        it = iter(draws)
        for prefix in nets.dense_prefixes(self.spec):
            with torch.no_grad():
                for leaf in ("noise_in", "noise_out_weight", "noise_out_bias"):
                    p[f"{prefix}.{leaf}"] = torch.from_numpy(np.array(next(it), dtype=np.float32))
                nets.compose_noise(p, prefix)

    def train(self, frames_u8: np.ndarray, actions, rewards, terminals, weights, indices,
              rand: Optional[list] = None, noise_online=None, noise_target=None):
        spec = self.spec
        if spec.noisy:
            self.set_noise("online", noise_online)
            self.set_noise("target", noise_target)
        C = spec.obs_shape[0]
        fr = torch.from_numpy(np.ascontiguousarray(frames_u8)).float().reshape(-1, 2 * C, *spec.obs_shape[1:]).div(255.0)
        obs, next_obs = torch.split(fr, C, 1)
        a = torch.as_tensor(actions).long()
        r = torch.as_tensor(rewards).float()
        d = torch.as_tensor(terminals).float()
        w = torch.as_tensor(weights).float()
        rnd = None if rand is None else [torch.as_tensor(x).float() for x in rand]
        for k in self.train_keys:
            self.po[k].grad = None
        q_loss, f_loss = losses.train_step(self.po, self.pt, spec, self.hp, obs, a, r, d, next_obs, rnd)

        if spec.algo == "fqf":
            fgrads = torch.autograd.grad((f_loss * w).sum(), [self.po[k] for k in self.f_keys], retain_graph=True)
            fg = dict(zip(self.f_keys, fgrads))
            if self.max_grad_norm > 0:
                tot = torch.sqrt(sum((g * g).sum() for g in fgrads))
                coef = min(1.0, self.max_grad_norm / (float(tot) + 1e-6))
                fg = {k: g * coef for k, g in fg.items()}
            self.rms.step({k: self.po[k].data for k in self.f_keys}, fg)
            self.last_grads.update(fg)

        skipped = bool(torch.isnan(q_loss).any())
        if not skipped:
            grads = torch.autograd.grad((q_loss * w).sum(), [self.po[k] for k in self.q_keys], allow_unused=True)
            g = dict(zip(self.q_keys, grads))
            self.last_grads.update(g)
            self.adam.step({k: self.po[k].data for k in self.q_keys}, g)
            self.update_steps += 1
        if self.update_steps % self.target_update_freq == 0:
            self.pt = OrderedDict((k, v.detach().clone()) for k, v in self.po.items())
        return {
            "q_loss": None if skipped else q_loss.detach().clone(),
            "fraction_loss": None if f_loss is None else f_loss.detach().clone(),
            "indices": torch.as_tensor(indices).long(),
        }
