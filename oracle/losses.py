"""Oracle: per-sample TD losses of the six learners (torch CPU fp32).

Restates /root/reference agent0/deepq/agent.py:
  BaseLearner.huber_qr_loss       agent.py:110-114
  BaseLearner.log_softmax_stable  agent.py:116-119
  DQNLearner.train_step           agent.py:173-190
  MDQNLearner.train_step          agent.py:194-215
  C51Learner.train_step           agent.py:219-269
  QRLearner.train_step            agent.py:273-293
  IQNLearner.train_step           agent.py:297-327
  FQFLearner.train_step           agent.py:340-388

All random draws (IQN taus) arrive through ``rand`` — a list consumed in the
order the reference calls ``torch.rand`` (action-selection taus first, then
target N', then online N).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn.functional as F

from . import nets


@dataclass
class Hyper:
    discount: float = 0.99
    n_step: int = 1
    double_q: bool = False
    vmin: float = -10.0
    vmax: float = 10.0
    K: int = 32
    N: int = 64
    N_dash: int = 64
    mdqn_tau: float = 0.03
    mdqn_lo: float = -1.0

    @property
    def gamma_n(self) -> float:
        return self.discount ** self.n_step


def huber_quantile(q: torch.Tensor, target: torch.Tensor, taus: torch.Tensor) -> torch.Tensor:
    """q [B,N] online quantiles, target [B,N'] , taus [B|1, N] -> loss [B].

    Pairwise element (j over targets, i over online): smooth_l1(q_i - T_j) * |tau_i - 1{T_j < q_i}|,
    summed over i, averaged over j.
    """
    qi = q.unsqueeze(1)  # [B,1,N]
    tj = target.unsqueeze(2)  # [B,N',1]
    d = qi - tj
    ad = d.abs()
    hub = torch.where(ad < 1.0, 0.5 * d * d, ad - 0.5)
    ind = (tj < qi).detach().to(q.dtype)
    w = (taus.unsqueeze(1) - ind).abs()
    return (hub * w).sum(dim=-1).mean(dim=-1)


def log_softmax_tau(logits: torch.Tensor, tau: float) -> torch.Tensor:
    z = logits - logits.max(dim=-1, keepdim=True)[0]
    return z - tau * torch.logsumexp(z / tau, dim=-1, keepdim=True)


def _rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return x[torch.arange(x.shape[0]), idx]


def dqn_loss(po, pt, spec, hp: Hyper, obs, actions, rewards, terminals, next_obs):
    with torch.no_grad():
        q_next = nets.forward(pt, spec, next_obs)
        a_next = nets.qval(po, spec, next_obs).argmax(dim=-1) if hp.double_q else q_next.argmax(dim=-1)
        y = rewards + hp.gamma_n * (1 - terminals) * _rows(q_next, a_next)
    q = _rows(nets.forward(po, spec, obs), actions)
    return F.smooth_l1_loss(q, y, reduction="none")


def mdqn_loss(po, pt, spec, hp: Hyper, obs, actions, rewards, terminals, next_obs):
    with torch.no_grad():
        ql = nets.forward(pt, spec, next_obs)
        v_next = (ql.softmax(dim=-1) * (ql - log_softmax_tau(ql, hp.mdqn_tau))).sum(dim=-1)
        add_on = _rows(log_softmax_tau(nets.forward(pt, spec, obs), hp.mdqn_tau), actions).clamp(hp.mdqn_lo, 0)
        y = rewards + hp.mdqn_tau * add_on + hp.gamma_n * (1 - terminals) * v_next
    q = _rows(nets.forward(po, spec, obs), actions)
    return F.smooth_l1_loss(q, y, reduction="none")


def c51_project(prob_next: torch.Tensor, rewards, terminals, atoms: torch.Tensor, hp: Hyper, delta: float) -> torch.Tensor:
    """Categorical projection of the Bellman-shifted support (agent.py:230-264).  prob_next [B,atoms]."""
    B, n = prob_next.shape
    tz = rewards.view(-1, 1) + hp.gamma_n * (1 - terminals.view(-1, 1)) * atoms.view(1, -1)
    tz = tz.clamp(hp.vmin, hp.vmax)
    b = (tz - hp.vmin) / delta
    lo, up = b.floor().long(), b.ceil().long()
    lo = torch.where((up > 0) & (lo == up), lo - 1, lo)
    up = torch.where((lo < (n - 1)) & (lo == up), up + 1, up)
    m = torch.zeros_like(prob_next)
    m.scatter_add_(1, lo, prob_next * (up.float() - b))
    m.scatter_add_(1, up, prob_next * (b - lo.float()))
    return m


def c51_loss(po, pt, spec, hp: Hyper, obs, actions, rewards, terminals, next_obs, return_target=False):
    atoms = nets.c51_atoms(spec, hp.vmin, hp.vmax)
    delta = (hp.vmax - hp.vmin) / (spec.num_atoms - 1)
    with torch.no_grad():
        prob_next = nets.forward(pt, spec, next_obs).softmax(dim=-1)
        if hp.double_q:
            a_next = nets.qval(po, spec, next_obs).argmax(dim=-1)
        else:
            a_next = (prob_next * atoms.view(1, 1, -1)).sum(dim=-1).argmax(dim=-1)
        m = c51_project(_rows(prob_next, a_next), rewards, terminals, atoms, hp, delta)
    logp = _rows(nets.forward(po, spec, obs).log_softmax(dim=-1), actions)
    loss = -(m * logp).sum(dim=-1)
    return (loss, m) if return_target else loss


def qr_loss(po, pt, spec, hp: Hyper, obs, actions, rewards, terminals, next_obs):
    with torch.no_grad():
        q_next = nets.forward(pt, spec, next_obs)
        a_next = nets.qval(po, spec, next_obs).argmax(dim=-1) if hp.double_q else q_next.mean(dim=-1).argmax(dim=-1)
        y = rewards.view(-1, 1) + hp.gamma_n * (1 - terminals.view(-1, 1)) * _rows(q_next, a_next)
    q = _rows(nets.forward(po, spec, obs), actions)
    n = spec.num_atoms
    taus = ((2 * torch.arange(n) + 1) / (2.0 * n)).view(1, -1)
    return huber_quantile(q, y, taus)


def iqn_loss(po, pt, spec, hp: Hyper, obs, actions, rewards, terminals, next_obs, rand: List[torch.Tensor]):
    """rand = [taus_K [B,K,1], taus_Ndash [B,N',1], taus_N [B,N,1]] in the reference's draw order."""
    t_sel, t_tgt, t_on = rand
    B = obs.shape[0]
    ar = torch.arange(B)
    with torch.no_grad():
        feat_next_t = nets.encoder(pt, next_obs)
        if hp.double_q:
            a_next = nets.head_iqn(po, spec, nets.encoder(po, next_obs), t_sel).mean(dim=1).argmax(dim=-1)
        else:
            a_next = nets.head_iqn(pt, spec, feat_next_t, t_sel).mean(dim=1).argmax(dim=-1)
        q_next = nets.head_iqn(pt, spec, feat_next_t, t_tgt)[ar, :, a_next]
        y = rewards.view(-1, 1) + hp.gamma_n * (1 - terminals.view(-1, 1)) * q_next
    q = nets.head_iqn(po, spec, nets.encoder(po, obs), t_on)[ar, :, actions]
    return huber_quantile(q, y, t_on[:, :, 0])


def fqf_loss(po, pt, spec, hp: Hyper, obs, actions, rewards, terminals, next_obs):
    """-> (quantile loss [B], fraction loss [B]).  Note Q16: the target net is
    evaluated at the ONLINE net's tau-hats."""
    B = obs.shape[0]
    ar = torch.arange(B)
    feat = nets.encoder(po, obs)
    taus, taus_hat, _ = nets.fqf_prop_taus(po, spec, feat.detach())
    q_hat = nets.head_iqn(po, spec, feat, taus_hat)[ar, :, actions]
    with torch.no_grad():
        feat_next_t = nets.encoder(pt, next_obs)
        if hp.double_q:
            a_next = nets.qval_from_feat(po, spec, nets.encoder(po, next_obs)).argmax(dim=-1)
        else:
            a_next = nets.qval_from_feat(pt, spec, feat_next_t).argmax(dim=-1)
        q_next = nets.head_iqn(pt, spec, feat_next_t, taus_hat)[ar, :, a_next]
        y = rewards.view(-1, 1) + hp.gamma_n * (1 - terminals.view(-1, 1)) * q_next
    loss = huber_quantile(q_hat, y, taus_hat[:, :, 0])
    with torch.no_grad():
        q = nets.head_iqn(po, spec, feat, taus[:, 1:-1])[ar, :, actions]
        v1 = q - q_hat[:, :-1]
        s1 = q > torch.cat((q_hat[:, :1], q[:, :-1]), dim=1)
        v2 = q - q_hat[:, 1:]
        s2 = q < torch.cat((q[:, 1:], q_hat[:, -1:]), dim=1)
    g = torch.where(s1, v1, -v1) + torch.where(s2, v2, -v2)
    fraction_loss = (g * taus[:, 1:-1, 0]).sum(dim=1)
    return loss, fraction_loss


def train_step(po, pt, spec, hp: Hyper, obs, actions, rewards, terminals, next_obs, rand: Optional[list] = None):
    algo = spec.algo
    if algo == "dqn":
        return dqn_loss(po, pt, spec, hp, obs, actions, rewards, terminals, next_obs), None
    if algo == "mdqn":
        return mdqn_loss(po, pt, spec, hp, obs, actions, rewards, terminals, next_obs), None
    if algo == "c51":
        return c51_loss(po, pt, spec, hp, obs, actions, rewards, terminals, next_obs), None
    if algo == "qr":
        return qr_loss(po, pt, spec, hp, obs, actions, rewards, terminals, next_obs), None
    if algo == "iqn":
        return iqn_loss(po, pt, spec, hp, obs, actions, rewards, terminals, next_obs, rand), None
    if algo == "fqf":
        return fqf_loss(po, pt, spec, hp, obs, actions, rewards, terminals, next_obs)
    raise ValueError(algo)
