"""Shared input recipe for golden fixtures (test infrastructure, own code).

Both ``gen_golden.py`` (runs in the build container, imports the reference from
/root/reference) and the parity tests (run anywhere, never read the reference)
regenerate *inputs* from this recipe: network weights, u8 frames, actions,
rewards, terminals.  Only the reference's *outputs* (and small injected random
draws such as IQN taus / NoisyNet noise) are stored in ``tests/golden/*.npz``.

The tensor inventory below restates the reference's ``state_dict`` layout
(/root/reference agent0/deepq/model.py:28-52 NoisyLinear, :90-105 ConvEncoder,
:108-121 DQNHead, :137-161 C51Head, :180-188 QRHead, :195-217 IQNHead,
:260-266 FQFHead); ``gen_golden.py`` asserts it equals the reference model's
actual ``state_dict()`` keys and shapes, so the inventory itself is pinned.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Tuple

import numpy as np

OBS_SHAPE = (4, 84, 84)
FRAME_BYTES = 8 * 84 * 84  # st || st_next, agent.py:78-81


@dataclass(frozen=True)
class NetSpec:
    """Everything that determines the parameter inventory of a DeepQNet."""

    algo: str = "dqn"  # dqn | mdqn | c51 | qr | iqn | fqf
    action_dim: int = 4
    dueling: bool = False
    noisy: bool = False
    num_atoms: int = 51  # c51: 51, qr: 200
    num_cosines: int = 64
    F: int = 32
    obs_shape: Tuple[int, int, int] = OBS_SHAPE
    tag: str = field(default="", compare=False)

    @property
    def conv_out_hw(self) -> Tuple[int, int, int]:
        c, h, w = self.obs_shape
        h1, w1 = (h - 8) // 4 + 1, (w - 8) // 4 + 1
        h2, w2 = (h1 - 4) // 2 + 1, (w1 - 4) // 2 + 1
        h3, w3 = h2 - 2, w2 - 2
        return 64, h3, w3

    @property
    def feat_dim(self) -> int:
        c, h, w = self.conv_out_hw
        return c * h * w

    @property
    def head_out(self) -> int:
        if self.algo in ("dqn", "mdqn", "iqn", "fqf"):
            return self.action_dim
        return self.action_dim * self.num_atoms

    @property
    def value_out(self) -> int:
        if self.algo in ("c51", "qr"):
            return self.num_atoms
        return 1


def _dense(prefix: str, out_f: int, in_f: int, noisy: bool) -> "OrderedDict[str, tuple]":
    d: "OrderedDict[str, tuple]" = OrderedDict()
    if noisy:
        d[f"{prefix}.weight_mu"] = (out_f, in_f)
        d[f"{prefix}.weight_sigma"] = (out_f, in_f)
        d[f"{prefix}.bias_mu"] = (out_f,)
        d[f"{prefix}.bias_sigma"] = (out_f,)
        d[f"{prefix}.weight_epsilon"] = (out_f, in_f)
        d[f"{prefix}.bias_epsilon"] = (out_f,)
        d[f"{prefix}.noise_in"] = (in_f,)
        d[f"{prefix}.noise_out_weight"] = (out_f,)
        d[f"{prefix}.noise_out_bias"] = (out_f,)
    else:
        d[f"{prefix}.weight"] = (out_f, in_f)
        d[f"{prefix}.bias"] = (out_f,)
    return d


def state_dict_shapes(spec: NetSpec) -> "OrderedDict[str, tuple]":
    """Key -> shape in the reference's ``state_dict()`` order."""
    c = spec.obs_shape[0]
    d: "OrderedDict[str, tuple]" = OrderedDict()
    d["encoder.convs.0.weight"] = (32, c, 8, 8)
    d["encoder.convs.0.bias"] = (32,)
    d["encoder.convs.2.weight"] = (64, 32, 4, 4)
    d["encoder.convs.2.bias"] = (64,)
    d["encoder.convs.4.weight"] = (64, 64, 3, 3)
    d["encoder.convs.4.bias"] = (64,)
    # buffers registered directly on the head come first in state_dict order
    head_buffers: "OrderedDict[str, tuple]" = OrderedDict()
    if spec.algo == "c51":
        head_buffers["head.atoms"] = (1, 1, spec.num_atoms)
    if spec.algo == "qr":
        head_buffers["head.cumulative_density"] = (spec.num_atoms,)
    d.update(head_buffers)
    d.update(_dense("head.first_dense", 512, spec.feat_dim, spec.noisy))
    d.update(_dense("head.q_head", spec.head_out, 512, spec.noisy))
    if spec.dueling:
        d.update(_dense("head.value_head", spec.value_out, 512, spec.noisy))
    if spec.algo in ("iqn", "fqf"):
        d["head.cosine_emb.0.weight"] = (spec.feat_dim, spec.num_cosines)
        d["head.cosine_emb.0.bias"] = (spec.feat_dim,)
    if spec.algo == "fqf":
        d["head.fraction_net.weight"] = (spec.F, spec.feat_dim)
        d["head.fraction_net.bias"] = (spec.F,)
    return d


BUFFER_SUFFIXES = (
    "weight_epsilon",
    "bias_epsilon",
    "noise_in",
    "noise_out_weight",
    "noise_out_bias",
    "atoms",
    "cumulative_density",
)


def is_buffer(key: str) -> bool:
    return key.endswith(BUFFER_SUFFIXES)


def gen(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def make_state_dict(spec: NetSpec, seed: int) -> "OrderedDict[str, np.ndarray]":
    """Deterministic fp32 weights.  Magnitudes follow fan-in scaling so that
    activations stay O(1); biases are non-zero on purpose (the reference
    initialises them to zero, which would hide a missing bias add)."""
    g = gen(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in state_dict_shapes(spec).items():
        if key == "head.atoms":
            # torch.linspace(-10, 10, 51) restated bit-for-bit is not needed:
            # the generator overwrites this entry with the reference's buffer.
            out[key] = np.linspace(-10.0, 10.0, spec.num_atoms, dtype=np.float32).reshape(shape)
            continue
        if key == "head.cumulative_density":
            n = spec.num_atoms
            out[key] = ((2 * np.arange(n) + 1) / (2.0 * n)).astype(np.float32)
            continue
        if is_buffer(key):
            out[key] = np.zeros(shape, dtype=np.float32)
            continue
        leaf = key.rsplit(".", 1)[1]
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else None
        if leaf in ("weight", "weight_mu"):
            scale = np.sqrt(2.0 / fan_in)
            if "q_head" in key or "value_head" in key:
                scale = 0.5 / np.sqrt(fan_in)
            if "fraction_net" in key:
                scale = 2.0 / np.sqrt(fan_in)
            out[key] = (g.standard_normal(shape) * scale).astype(np.float32)
        elif leaf == "weight_sigma":
            out[key] = g.uniform(0.2, 0.6, shape).astype(np.float32) / np.float32(np.sqrt(shape[1]))
        elif leaf in ("bias", "bias_mu"):
            out[key] = (g.standard_normal(shape) * 0.05).astype(np.float32)
        elif leaf == "bias_sigma":
            out[key] = g.uniform(0.01, 0.05, shape).astype(np.float32)
        else:  # pragma: no cover
            raise KeyError(key)
    return out


def make_frames(batch: int, seed: int, obs_shape=OBS_SHAPE) -> np.ndarray:
    """u8 [B, 2*C, H, W]: st || st_next as stored by the actor (agent.py:78-81).
    Smooth-ish structure + noise so conv activations are not degenerate."""
    g = gen(seed)
    c, h, w = obs_shape
    base = g.integers(0, 256, size=(batch, 2 * c, h, w), dtype=np.uint8)
    # make ~half of the pixels dark like an Atari frame
    mask = g.integers(0, 2, size=(batch, 2 * c, h, w), dtype=np.uint8)
    return (base * mask).astype(np.uint8)


def make_transitions(batch: int, action_dim: int, seed: int):
    """actions int64 [B], rewards f32 [B] in {-1,0,1} n-step style sums, terminals bool [B]."""
    g = gen(seed)
    actions = g.integers(0, action_dim, size=batch, dtype=np.int64)
    rewards = g.choice(np.array([-1.0, 0.0, 0.0, 1.0, 1.99, -0.99], dtype=np.float32), size=batch)
    terminals = g.random(batch) < 0.25
    weights = g.uniform(0.2, 1.0, size=batch).astype(np.float32)
    return actions, rewards.astype(np.float32), terminals, weights


def checksum(x: np.ndarray) -> np.ndarray:
    """(sum, l2, first-8) fingerprint of a tensor, float64 accumulations."""
    flat = np.asarray(x, dtype=np.float64).ravel()
    head = np.zeros(8, dtype=np.float64)
    head[: min(8, flat.size)] = flat[:8]
    return np.concatenate(([flat.sum(), np.sqrt((flat * flat).sum())], head))


# Canonical network variants pinned by the fixtures.
SPECS: Dict[str, NetSpec] = {
    "dqn": NetSpec("dqn", 4),
    "dqn_duel": NetSpec("dqn", 4, dueling=True),
    "mdqn": NetSpec("mdqn", 4),
    "c51": NetSpec("c51", 4, num_atoms=51),
    "c51_duel_noisy": NetSpec("c51", 4, dueling=True, noisy=True, num_atoms=51),
    "qr": NetSpec("qr", 4, num_atoms=200),
    "qr_duel": NetSpec("qr", 6, dueling=True, num_atoms=200),
    "iqn": NetSpec("iqn", 9),
    "iqn_duel": NetSpec("iqn", 4, dueling=True),
    "fqf": NetSpec("fqf", 9),
    # the reference's own 8-game suite (README.md:62-112, atari8_double_duel_prior): fqf + double-Q + dueling + prioritized; and the largest
    # action set of that suite (Seaquest: the full 18 ALE actions)
    "fqf_duel": NetSpec("fqf", 9, dueling=True),
    "dqn_duel_a18": NetSpec("dqn", 18, dueling=True),
    "fqf_duel_a18": NetSpec("fqf", 18, dueling=True),
    # tiny geometry: conv stack output 1x1x64 (36x36 input), fast kernel tests
    "dqn_tiny": NetSpec("dqn", 4, obs_shape=(4, 36, 36)),
    "c51_tiny": NetSpec("c51", 6, dueling=True, noisy=True, obs_shape=(4, 36, 36)),
}
