"""Oracle: Nature-CNN Q-network forward passes (torch CPU fp32, functional).

Restates /root/reference agent0/deepq/model.py:
  NoisyLinear.forward/reset_noise/transform_noise  model.py:54-62,73-87
  ConvEncoder                                      model.py:90-105
  DQNHead.forward                                  model.py:123-131
  C51Head.forward/qval                             model.py:163-177
  QRHead.qval                                      model.py:190-192
  IQNHead.feature_emb/forward/qval                 model.py:219-257
  FQFHead.prop_taus/qval                           model.py:268-284
  DeepQNet.forward/qval                            model.py:323-330

Parameters are a plain ``dict[str, Tensor]`` keyed by the reference's
``state_dict`` names (tests/golden/recipe.py:state_dict_shapes).  Random draws
(IQN taus) are always passed in by the caller — never drawn here.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# ReLU decisions of the differentiated pass, injected (parity tests at full size).  A ReLU's VALUE is continuous in its input but its
# gradient is not: where a pre-activation sits within fp32 rounding of zero, two correct evaluations (torch's blocked sums, the MFMA's
# k-ordered chain) can disagree on the 0/1 decision, and each such disagreement moves a weight gradient by one whole term of its sum.  When
# RELU_MASKS = {"conv1" | "conv2" | "conv3" | "fc1" | "cos": 0/1 tensor in this module's layout} is set, every ReLU evaluated with autograd
# enabled keeps its forward value but back-propagates through the given mask, and RELU_STATS[name] records (number of decisions that differ
# from this module's own, number of decisions, largest |pre-activation| among the differing ones relative to the layer's largest).
RELU_MASKS: Optional[dict] = None
RELU_STATS: dict = {}


def _relu(x: torch.Tensor, name: str) -> torch.Tensor:
    if RELU_MASKS is None or not torch.is_grad_enabled() or name not in RELU_MASKS:
        return F.relu(x)
    m = RELU_MASKS[name].reshape(x.shape).to(x.dtype)
    differ = (x > 0) != (m > 0)
    worst = float(x.detach().abs()[differ].max()) / (float(x.detach().abs().max()) + 1e-30) if bool(differ.any()) else 0.0
    RELU_STATS[name] = (int(differ.sum()), differ.numel(), worst)
    return F.relu(x).detach() + m * (x - x.detach())


def noise_f(x: torch.Tensor) -> torch.Tensor:
    """f(x) = sign(x) * sqrt(|x|)   (model.py:85-87)"""
    return x.sign() * x.abs().sqrt()


def compose_noise(p: Params, prefix: str) -> None:
    """Recompute weight_epsilon / bias_epsilon from the noise vectors (model.py:73-83)."""
    p[f"{prefix}.weight_epsilon"] = torch.outer(noise_f(p[f"{prefix}.noise_out_weight"]), noise_f(p[f"{prefix}.noise_in"]))
    p[f"{prefix}.bias_epsilon"] = noise_f(p[f"{prefix}.noise_out_bias"])


def dense_prefixes(spec) -> list:
    out = ["head.first_dense", "head.q_head"]
    if spec.dueling:
        out.append("head.value_head")
    return out


def dense(p: Params, prefix: str, x: torch.Tensor, noisy: bool) -> torch.Tensor:
    """nn.Linear or NoisyLinear in training mode (the reference never calls .eval(), Q11)."""
    if noisy:
        w = p[f"{prefix}.weight_mu"] + p[f"{prefix}.weight_sigma"] * p[f"{prefix}.weight_epsilon"]
        b = p[f"{prefix}.bias_mu"] + p[f"{prefix}.bias_sigma"] * p[f"{prefix}.bias_epsilon"]
    else:
        w, b = p[f"{prefix}.weight"], p[f"{prefix}.bias"]
    return F.linear(x, w, b)


def encoder(p: Params, x: torch.Tensor, return_all: bool = False):
    """x: [B,C,H,W] fp32 in [0,1] -> [B, 64*h*w] flattened in (C,H,W) order."""
    a1 = _relu(F.conv2d(x, p["encoder.convs.0.weight"], p["encoder.convs.0.bias"], stride=4), "conv1")
    a2 = _relu(F.conv2d(a1, p["encoder.convs.2.weight"], p["encoder.convs.2.bias"], stride=2), "conv2")
    a3 = _relu(F.conv2d(a2, p["encoder.convs.4.weight"], p["encoder.convs.4.bias"], stride=1), "conv3")
    feat = a3.flatten(1)
    if return_all:
        return feat, (a1, a2, a3)
    return feat


def normalize(frames_u8: torch.Tensor) -> torch.Tensor:
    """uint8 -> fp32 / 255 (agent.py:27, agent.py:129-134): a true fp32 division."""
    return frames_u8.float().div(255.0)


# ----------------------------------------------------------------------------- heads
def head_dqn(p: Params, spec, feat: torch.Tensor) -> torch.Tensor:
    h = _relu(dense(p, "head.first_dense", feat, spec.noisy), "fc1")
    q = dense(p, "head.q_head", h, spec.noisy)
    if spec.dueling:
        v = dense(p, "head.value_head", h, spec.noisy)
        q = v + (q - q.mean(dim=-1, keepdim=True))
    return q


def head_dist(p: Params, spec, feat: torch.Tensor) -> torch.Tensor:
    """C51 / QR: [B, A, atoms]; dueling value is [B,1,atoms], mean over the ACTION dim."""
    h = _relu(dense(p, "head.first_dense", feat, spec.noisy), "fc1")
    q = dense(p, "head.q_head", h, spec.noisy).view(feat.shape[0], spec.action_dim, spec.num_atoms)
    if spec.dueling:
        v = dense(p, "head.value_head", h, spec.noisy).view(feat.shape[0], 1, spec.num_atoms)
        q = v + (q - q.mean(dim=1, keepdim=True))
    return q


def c51_atoms(spec, vmin: float = -10.0, vmax: float = 10.0) -> torch.Tensor:
    return torch.linspace(vmin, vmax, spec.num_atoms)


def cos_features(p: Params, spec, feat: torch.Tensor, taus: torch.Tensor) -> torch.Tensor:
    """taus: [B,n,1] -> Hadamard(state, relu(Linear(cos(pi*i*tau)))) as [(B n), D]  (model.py:235-251)."""
    B, n, _ = taus.shape
    ipi = math.pi * torch.arange(1, spec.num_cosines + 1, dtype=feat.dtype).view(1, 1, -1)
    cosine = (ipi * taus).cos().reshape(B * n, spec.num_cosines)
    emb = _relu(F.linear(cosine, p["head.cosine_emb.0.weight"], p["head.cosine_emb.0.bias"]), "cos").view(B, n, -1)
    return (emb * feat.unsqueeze(1)).reshape(B * n, -1)


def head_iqn(p: Params, spec, feat: torch.Tensor, taus: torch.Tensor) -> torch.Tensor:
    """[B,n,A] quantile values at the given taus [B,n,1]."""
    B, n, _ = taus.shape
    x = cos_features(p, spec, feat, taus)
    h = _relu(dense(p, "head.first_dense", x, spec.noisy), "fc1")
    q = dense(p, "head.q_head", h, spec.noisy)
    if spec.dueling:
        v = dense(p, "head.value_head", h, spec.noisy)
        q = v + (q - q.mean(dim=-1, keepdim=True))
    return q.view(B, n, spec.action_dim)


# When a list, every fqf_prop_taus call appends its (taus, taus_hat): the parity tests hand these to the device path so that both sides
# evaluate q(tau) at bit-identical fractions (cos(pi*64*tau) amplifies ulp-level differences of two softmax/cumsum evaluations ~200x).
TAU_LOG: Optional[list] = None
# When a list of (taus, taus_hat) pairs, fqf_prop_taus takes its VALUES from there, call by call (the autograd path to the fraction net is
# kept): lets the fp64 evaluation of a step (learner.exact_gradients) run at the same fractions as the fp32 one it arbitrates.
TAU_OVERRIDE: Optional[list] = None


def fqf_prop_taus(p: Params, spec, feat_detached: torch.Tensor):
    """(taus [B,F+1,1], taus_hat [B,F,1], entropies [B,1])  model.py:268-278."""
    logp = F.linear(feat_detached, p["head.fraction_net.weight"], p["head.fraction_net.bias"]).log_softmax(dim=-1)
    probs = logp.exp()
    taus = torch.cat((torch.zeros(feat_detached.shape[0], 1, dtype=probs.dtype), torch.cumsum(probs, dim=-1)), dim=-1)
    taus_hat = (taus[:, :-1] + taus[:, 1:]).detach() / 2.0
    if TAU_OVERRIDE is not None:
        t_in, th_in = TAU_OVERRIDE.pop(0)
        taus = taus + (t_in.to(taus.dtype) - taus).detach()
        taus_hat = th_in.to(taus.dtype)
    ent = -(probs * logp).sum(dim=-1, keepdim=True)
    if TAU_LOG is not None:
        TAU_LOG.append((taus.detach().clone(), taus_hat.detach().clone()))
    return taus.unsqueeze(-1), taus_hat.unsqueeze(-1), ent


# ----------------------------------------------------------------------------- whole net
def forward(p: Params, spec, x: torch.Tensor, taus: Optional[torch.Tensor] = None) -> torch.Tensor:
    feat = encoder(p, x)
    if spec.algo in ("dqn", "mdqn"):
        return head_dqn(p, spec, feat)
    if spec.algo in ("c51", "qr"):
        return head_dist(p, spec, feat)
    assert taus is not None, "iqn/fqf forward needs injected taus"
    return head_iqn(p, spec, feat, taus)


def qval_from_feat(p: Params, spec, feat: torch.Tensor, taus: Optional[torch.Tensor] = None) -> torch.Tensor:
    if spec.algo in ("dqn", "mdqn"):
        return head_dqn(p, spec, feat)
    if spec.algo == "c51":
        return (head_dist(p, spec, feat).softmax(dim=-1) * c51_atoms(spec).view(1, 1, -1)).sum(dim=-1)
    if spec.algo == "qr":
        return head_dist(p, spec, feat).mean(dim=-1)
    if spec.algo == "iqn":
        assert taus is not None
        return head_iqn(p, spec, feat, taus).mean(dim=1)
    if spec.algo == "fqf":
        t, t_hat, _ = fqf_prop_taus(p, spec, feat.detach())
        q_hat = head_iqn(p, spec, feat, t_hat)
        return ((t[:, 1:, :] - t[:, :-1, :]) * q_hat).sum(dim=1)
    raise ValueError(spec.algo)


def qval(p: Params, spec, x: torch.Tensor, taus: Optional[torch.Tensor] = None) -> torch.Tensor:
    return qval_from_feat(p, spec, encoder(p, x), taus)


def trainable_keys(p: Params) -> list:
    """Keys of nn.Parameters in registration order (buffers excluded)."""
    skip = ("weight_epsilon", "bias_epsilon", "noise_in", "noise_out_weight", "noise_out_bias", "atoms", "cumulative_density")
    return [k for k in p if not k.endswith(skip)]
