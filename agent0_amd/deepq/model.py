"""``DeepQNet``: the reference's network classes as an ``nn.Module`` SHELL over packed device parameters.

Mirrors the public surface of /root/reference agent0/deepq/model.py — ``NoisyLinear`` (28-87), ``ConvEncoder``
(90-105), the five heads (108-284) and ``DeepQNet`` (287-338): same attribute names (``.encoder``, ``.head``),
same ``state_dict()`` keys/shapes/initialisation, ``.forward``, ``.qval``, ``.params()``, ``.reset_noise()``.
The arithmetic does not run in these modules: parameters live in one flat HBM buffer in kernel layout
(agent0_amd/deepq/layout.py) and every forward pass is a sequence of HIP kernels (agent0_amd/deepq/engine.py).  The
shell's tensors are refreshed from / written to that buffer on ``state_dict()`` / ``load_state_dict()``, so weights
interchange with the reference.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from itertools import chain
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .config import AlgoEnum, ExpConfig
from .engine import DeviceNet, Workspace
from .layout import NetLayout


def _orthogonal(m: nn.Module, gain: float = 1.0):
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        nn.init.orthogonal_(m.weight.data, gain)
        nn.init.zeros_(m.bias.data)


def _xavier(m: nn.Module, gain: float = 1.0):
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        nn.init.xavier_uniform_(m.weight, gain=gain)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


class NoisyLinear(nn.Module):
    """Parameter/buffer container with the reference's names and initial values (model.py:28-52,64-71)."""

    def __init__(self, in_features: int, out_features: int, std_init: float = 0.4, noisy_layer_std: float = 0.1):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.std_init, self.noisy_layer_std = std_init, noisy_layer_std
        self.weight_mu = nn.Parameter(torch.zeros(out_features, in_features))
        self.weight_sigma = nn.Parameter(torch.zeros(out_features, in_features))
        self.register_buffer("weight_epsilon", torch.zeros(out_features, in_features))
        self.bias_mu = nn.Parameter(torch.zeros(out_features))
        self.bias_sigma = nn.Parameter(torch.zeros(out_features))
        self.register_buffer("bias_epsilon", torch.zeros(out_features))
        self.register_buffer("noise_in", torch.zeros(in_features))
        self.register_buffer("noise_out_weight", torch.zeros(out_features))
        self.register_buffer("noise_out_bias", torch.zeros(out_features))
        bound = 1 / np.sqrt(in_features)
        self.weight_mu.data.uniform_(-bound, bound)
        self.weight_sigma.data.fill_(std_init / np.sqrt(in_features))
        self.bias_mu.data.uniform_(-bound, bound)
        self.bias_sigma.data.fill_(std_init / np.sqrt(out_features))


class ConvEncoder(nn.Module):
    def __init__(self, chan_dim: int):
        super().__init__()
        self.convs = nn.Sequential(nn.Conv2d(chan_dim, 32, 8, stride=4), nn.ReLU(), nn.Conv2d(32, 64, 4, stride=2), nn.ReLU(),
                                   nn.Conv2d(64, 64, 3, stride=1), nn.ReLU(), nn.Flatten())
        self.convs.apply(lambda m: _orthogonal(m, nn.init.calculate_gain("relu")))


class _Head(nn.Module):
    def __init__(self, L: NetLayout, cfg: ExpConfig):
        super().__init__()
        Dense = NoisyLinear if L.noisy else nn.Linear
        self.first_dense = Dense(L.feat, 512)
        self.first_dense.apply(lambda m: _orthogonal(m, nn.init.calculate_gain("relu")))
        self.q_head = Dense(512, L.Nq)
        self.q_head.apply(lambda m: _orthogonal(m, 0.01))
        if L.dueling:
            self.value_head = Dense(512, L.V)
            self.value_head.apply(lambda m: _orthogonal(m, 1.0))
        else:
            self.value_head = None
        self.action_dim = L.A
        if L.algo == "c51":
            c = cfg.learner.c51
            self.register_buffer("atoms", torch.linspace(c.vmin, c.vmax, c.num_atoms).view(1, 1, -1))
            self.delta = (c.vmax - c.vmin) / (c.num_atoms - 1)
        if L.algo == "qr":
            n = cfg.learner.qr.num_atoms
            self.register_buffer("cumulative_density", (2 * torch.arange(n) + 1) / (2.0 * n))
        if L.quantile:
            self.cfg = cfg.learner.iqn
            self.cosine_emb = nn.Sequential(nn.Linear(L.num_cosines, L.feat), nn.ReLU())
            self.cosine_emb.apply(lambda m: _orthogonal(m, nn.init.calculate_gain("relu")))
        if L.algo == "fqf":
            self.fraction_net = nn.Linear(L.feat, L.F)
            self.fraction_net.apply(lambda m: _xavier(m, 0.01))


def layout_from_cfg(cfg: ExpConfig) -> NetLayout:
    lc = cfg.learner
    algo = lc.algo.name
    atoms = lc.c51.num_atoms if algo == "c51" else (lc.qr.num_atoms if algo == "qr" else 1)
    if not cfg.obs_shape or len(tuple(cfg.obs_shape)) != 3 or cfg.action_dim < 1:
        raise ValueError("cfg.obs_shape / cfg.action_dim must be set before building a network (main.py:31-32)")
    return NetLayout(algo, int(cfg.action_dim), lc.dueling_head, lc.noisy_net, atoms, tuple(int(v) for v in cfg.obs_shape), lc.iqn.num_cosines, lc.iqn.F)


class DeepQNet(nn.Module):
    def __init__(self, cfg: ExpConfig, ops=None, dev_net: Optional[DeviceNet] = None, rng=None):
        super().__init__()
        if cfg.device.value != "cuda":
            raise RuntimeError("agent0_amd runs on MI355X only: set device=cuda (there is deliberately no CPU fallback; the CPU "
                               "restatement used for testing lives in oracle/).")
        self.cfg = cfg
        self.L = layout_from_cfg(cfg)
        if dev_net is None:
            if ops is None:
                from agent0_amd.ops import HipOps
                ops = HipOps()
            dev_net = DeviceNet(ops, self.L, ops.net(self.L.C, self.L.H, self.L.W))
        self.ops = dev_net.ops
        object.__setattr__(self, "_dev", dev_net)       # not a submodule
        self.encoder = ConvEncoder(self.L.C)
        self.head = _Head(self.L, cfg)
        self.to(self.ops.device)
        self._ws = {}
        self._rng = rng
        self._noise_calls = 0
        self.push()
        if self.L.noisy:
            self.reset_noise()

    # ------------------------------------------------------------------ shell <-> device
    def push(self):
        """shell tensors -> packed device parameters"""
        self._dev.load_state_dict(OrderedDict((k, v.detach()) for k, v in nn.Module.state_dict(self).items()))

    def pull(self):
        """packed device parameters -> shell tensors"""
        fresh = self._dev.state_dict()
        own = dict(self.named_parameters())
        own.update(dict(self.named_buffers()))
        with torch.no_grad():
            for k, v in fresh.items():
                own[k].copy_(v.reshape(own[k].shape))

    def state_dict(self, *args, **kwargs):
        self.pull()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.push()
        return out

    def params(self):
        return chain(v for k, v in self.named_parameters() if "fraction" not in k)

    # ------------------------------------------------------------------ noise
    def reset_noise(self, rng=None, compose: bool = True):
        """NoisyLinear.reset_noise for every noisy layer (model.py:73-83,335-338): N(0, 0.1^2) draws on the device.  The noise vectors are
        adjacent in one buffer in draw order (DeviceNet), so ONE fill produces the same Philox draws as a fill per vector.
        ``compose=False``: the caller composes the effective weights itself before the next forward (the learner's update does)."""
        if not self.L.noisy:
            return
        if rng is None:
            if self._rng is None:
                from agent0_amd.common.utils import DeviceRng
                self._rng = DeviceRng(self.ops, self.cfg.seed + 7919)
            rng = self._rng
        rng.normal(rng.STREAM_NOISE, 0.1, self._dev.noise_buf, self._dev.noise_len)
        if compose:
            self._dev.compose_noise()

    # ------------------------------------------------------------------ forward
    def _workspace(self, B: int, n_tau: int) -> Workspace:
        key = (B, n_tau)
        if key not in self._ws:
            self._ws[key] = Workspace(self.ops, self.L, B, n_tau)
        return self._ws[key]

    @staticmethod
    def _to_u8(x: torch.Tensor) -> torch.Tensor:
        if x.dtype == torch.uint8:
            return x.contiguous()
        return (x * 255.0).round().clamp_(0, 255).to(torch.uint8).contiguous()    # exact for inputs that are uint8/255 (agent.py:27,132)

    def _taus(self, B: int, n: int, taus: Optional[torch.Tensor]):
        if taus is not None:
            return taus.to(self.ops.device, torch.float32).reshape(-1).contiguous()
        if self._rng is None:
            from agent0_amd.common.utils import DeviceRng
            self._rng = DeviceRng(self.ops, self.cfg.seed + 7919)
        t = self.ops.empty(B * n)
        self._rng.uniform(self._rng.STREAM_TAUS, t, B * n)
        return t

    def _run(self, x: torch.Tensor, n: Optional[int] = None, taus: Optional[torch.Tensor] = None):
        L, dev = self.L, self._dev
        u8 = self._to_u8(x.to(self.ops.device))
        B = u8.shape[0]
        obs_bytes = L.C * L.H * L.W
        if L.algo == "fqf":
            ws = self._workspace(B, L.F)
            dev.encode(ws, u8.reshape(-1), None, obs_bytes, 0, B, keep=False)
            dev.fqf_taus(ws, B)
            dev.head(ws, B, ws.tau_hat, L.F)
            return ws, B, L.F, ws.tau_hat
        if L.algo == "iqn":
            n = taus.shape[1] if taus is not None else (n or self.cfg.learner.iqn.K)
            ws = self._workspace(B, n)
            t = self._taus(B, n, taus)
            dev.encode(ws, u8.reshape(-1), None, obs_bytes, 0, B, keep=False)
            dev.head(ws, B, t, n)
            return ws, B, n, t
        ws = self._workspace(B, 1)
        dev.encode(ws, u8.reshape(-1), None, obs_bytes, 0, B, keep=False)
        dev.head(ws, B)
        return ws, B, 1, None

    def forward(self, x, n: Optional[int] = None, taus: Optional[torch.Tensor] = None):
        """Same outputs as the reference: [B,A] (dqn/mdqn), [B,A,atoms] (c51/qr), ([B,n,A], taus [B,n,1]) (iqn/fqf)."""
        L = self.L
        ws, B, nt, t = self._run(x, n, taus)
        if L.quantile:
            return ws.q[: B * nt * L.A].view(B, nt, L.A).clone(), t[: B * nt].view(B, nt, 1).clone()
        q = ws.q[: B * L.A * L.T].clone()
        return q.view(B, L.A) if L.T == 1 else q.view(B, L.A, L.T)

    def qval(self, x, n: Optional[int] = None):
        L = self.L
        ws, B, nt, _ = self._run(x, n)
        qsel = self.ops.empty(B * L.A)
        a_star = self.ops.zeros(B, dtype=torch.int32)
        atoms = self.head.atoms.reshape(-1).contiguous() if L.algo == "c51" else None
        self._dev.select(ws, B, nt, a_star, qsel=qsel, atoms=atoms)
        return qsel.view(B, L.A)
