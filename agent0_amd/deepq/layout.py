"""Packed parameter layout of the device network and its mapping to the reference ``state_dict``.

The HIP kernels keep every trainable tensor in ONE flat fp32 buffer (shared by Adam and, under data parallelism, by
the gradient all-reduce).  Layers are stored as ``[W (N x K) | b (N)]`` blocks in kernel order:

  conv1  W [32][C*8*8]      (c,kh,kw)  == reference layout (agent0/deepq/model.py:94)
  conv2  W [64][4*4*32]     (kh,kw,c)  <- reference (64,32,4,4) permuted        (model.py:96)
  conv3  W [64][3*3*64]     (kh,kw,c)  <- reference (64,64,3,3) permuted        (model.py:98)
  fc1    W [512][feat]      columns (h,w,c) <- reference (c,h,w) flatten order  (model.py:100,112)
  head   W [Npad][512]      rows = q_head (A*T) then value_head (V), zero-padded to a multiple of 32 (model.py:114-119)
  cos    W [feat][64]       rows (h,w,c)   (IQN/FQF cosine embedding, model.py:214-216)
  frac   W [32][feat]       columns (h,w,c), rows F zero-padded to 32 (FQF fraction_net, model.py:265; RMSprop, not Adam)

NoisyLinear layers (model.py:28-52) contribute a ``mu`` and a ``sigma`` block each (both trained by Adam); their
composed weights live in a separate scratch buffer.  ``pack`` / ``unpack`` convert between this layout and the
reference's ``state_dict`` keys and shapes, so checkpoints interchange with the reference.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch


def ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass(frozen=True)
class Block:
    name: str          # conv1 | conv2 | conv3 | fc1 | head | cos | frac  (+ ".mu" / ".sigma" for noisy dense layers)
    offset: int        # float offset of W in the flat buffer; bias follows at offset + N*K
    N: int             # padded rows
    K: int
    n_real: int        # rows that map to reference parameters

    @property
    def size(self) -> int:
        return self.N * self.K + self.N

    @property
    def w(self) -> slice:
        return slice(self.offset, self.offset + self.N * self.K)

    @property
    def b(self) -> slice:
        return slice(self.offset + self.N * self.K, self.offset + self.size)

    @property
    def all(self) -> slice:
        return slice(self.offset, self.offset + self.size)


class NetLayout:
    """Layout for one network variant.  ``spec`` needs: algo, action_dim, dueling, noisy, num_atoms, num_cosines, F, obs_shape."""

    def __init__(self, algo: str, action_dim: int, dueling: bool, noisy: bool, num_atoms: int, obs_shape, num_cosines: int = 64, F: int = 32):
        self.algo, self.A, self.dueling, self.noisy = algo, action_dim, dueling, noisy
        self.C, self.H, self.W = obs_shape
        self.H1, self.W1 = (self.H - 8) // 4 + 1, (self.W - 8) // 4 + 1
        self.H2, self.W2 = (self.H1 - 4) // 2 + 1, (self.W1 - 4) // 2 + 1
        self.H3, self.W3 = self.H2 - 2, self.W2 - 2
        self.feat = self.H3 * self.W3 * 64
        self.T = num_atoms if algo in ("c51", "qr") else 1          # atoms per action in the head output
        self.num_cosines, self.F = num_cosines, F
        self.Nq = self.A * self.T
        self.V = self.T if dueling else 0
        self.Npad = ceil_to(self.Nq + self.V, 32)
        self.quantile = algo in ("iqn", "fqf")
        self.Fpad = 32 if algo == "fqf" else 0
        assert F <= 32, "fraction net wider than 32 is not supported by the packed layout"

        blocks: List[Block] = []
        off = 0

        def add(name, N, K, n_real):
            nonlocal off
            b = Block(name, off, N, K, n_real)
            blocks.append(b)
            off += b.size
            return b

        add("conv1", 32, self.C * 64, 32)
        add("conv2", 64, 512, 64)
        add("conv3", 64, 576, 64)
        self.conv_end = off                     # [0, conv_end): convolution blocks; [conv_end, n_adam): dense blocks (the two gradient buckets of dist.GradAllReduce)
        if noisy:
            add("fc1.mu", 512, self.feat, 512)
            add("fc1.sigma", 512, self.feat, 512)
            add("head.mu", self.Npad, 512, self.Nq + self.V)
            add("head.sigma", self.Npad, 512, self.Nq + self.V)
        else:
            add("fc1", 512, self.feat, 512)
            add("head", self.Npad, 512, self.Nq + self.V)
        if self.quantile:
            add("cos", self.feat, num_cosines, self.feat)
        self.n_adam = off                       # Adam covers [0, n_adam)
        if algo == "fqf":
            add("frac", self.Fpad, self.feat, F)
        self.n_params = off
        self.n_params_padded = ceil_to(off, 4)
        self.blocks: "OrderedDict[str, Block]" = OrderedDict((b.name, b) for b in blocks)
        # scratch with the composed (effective) weights of noisy dense layers: same [W | b] blocks
        self.eff: "OrderedDict[str, Block]" = OrderedDict()
        if noisy:
            e = 0
            for name, N, K, nr in (("fc1", 512, self.feat, 512), ("head", self.Npad, 512, self.Nq + self.V)):
                self.eff[name] = Block(name, e, N, K, nr)
                e += N * K + N
            self.n_eff = e
        else:
            self.n_eff = 0
        # noise vectors per NoisyLinear module, in the reference's module order (first_dense, q_head, value_head)
        self.noise_modules: List[Tuple[str, str, int, int, int]] = []   # (ref prefix, block, r0, r1, in_features)
        if noisy:
            self.noise_modules.append(("head.first_dense", "fc1", 0, 512, self.feat))
            self.noise_modules.append(("head.q_head", "head", 0, self.Nq, 512))
            if dueling:
                self.noise_modules.append(("head.value_head", "head", self.Nq, self.Nq + self.V, 512))

    @classmethod
    def from_spec(cls, spec) -> "NetLayout":
        return cls(spec.algo, spec.action_dim, spec.dueling, spec.noisy, getattr(spec, "num_atoms", 51), tuple(spec.obs_shape),
                   getattr(spec, "num_cosines", 64), getattr(spec, "F", 32))

    # ------------------------------------------------------------------ permutations
    def _feat_cols_to_hwc(self, w: torch.Tensor) -> torch.Tensor:
        """[..., 64*H3*W3] with (c,h,w) columns -> (h,w,c) columns."""
        lead = w.shape[:-1]
        return w.reshape(*lead, 64, self.H3, self.W3).permute(*range(len(lead)), len(lead) + 1, len(lead) + 2, len(lead)).reshape(*lead, self.feat)

    def _feat_cols_to_chw(self, w: torch.Tensor) -> torch.Tensor:
        lead = w.shape[:-1]
        return w.reshape(*lead, self.H3, self.W3, 64).permute(*range(len(lead)), len(lead) + 2, len(lead), len(lead) + 1).reshape(*lead, self.feat)

    # ------------------------------------------------------------------ reference state_dict -> flat
    def pack(self, sd: Dict[str, torch.Tensor], flat: torch.Tensor) -> None:
        """Writes every trainable tensor of ``sd`` (reference keys/shapes) into ``flat`` (len >= n_params)."""
        dev, dt = flat.device, flat.dtype
        flat[: self.n_params_padded].zero_()

        def put(block: Block, w: torch.Tensor, b: torch.Tensor, r0: int = 0):
            N, K = block.N, block.K
            w = w.to(device=dev, dtype=dt).reshape(-1, K)
            rows = w.shape[0]
            flat[block.offset + r0 * K: block.offset + (r0 + rows) * K] = w.reshape(-1)
            flat[block.offset + N * K + r0: block.offset + N * K + r0 + rows] = b.to(device=dev, dtype=dt).reshape(-1)

        B = self.blocks
        put(B["conv1"], sd["encoder.convs.0.weight"].reshape(32, -1), sd["encoder.convs.0.bias"])
        put(B["conv2"], sd["encoder.convs.2.weight"].permute(0, 2, 3, 1).reshape(64, -1), sd["encoder.convs.2.bias"])
        put(B["conv3"], sd["encoder.convs.4.weight"].permute(0, 2, 3, 1).reshape(64, -1), sd["encoder.convs.4.bias"])
        for kind, wk, bk in ((".mu", "weight_mu", "bias_mu"), (".sigma", "weight_sigma", "bias_sigma")) if self.noisy else (("", "weight", "bias"),):
            put(B["fc1" + kind], self._feat_cols_to_hwc(sd[f"head.first_dense.{wk}"]), sd[f"head.first_dense.{bk}"])
            put(B["head" + kind], sd[f"head.q_head.{wk}"], sd[f"head.q_head.{bk}"])
            if self.dueling:
                put(B["head" + kind], sd[f"head.value_head.{wk}"], sd[f"head.value_head.{bk}"], r0=self.Nq)
        if self.quantile:
            w = sd["head.cosine_emb.0.weight"]          # [feat (c,h,w)][64]
            w = w.reshape(64, self.H3, self.W3, self.num_cosines).permute(1, 2, 0, 3).reshape(self.feat, self.num_cosines)
            b = sd["head.cosine_emb.0.bias"].reshape(64, self.H3, self.W3).permute(1, 2, 0).reshape(self.feat)
            put(B["cos"], w, b)
        if self.algo == "fqf":
            put(B["frac"], self._feat_cols_to_hwc(sd["head.fraction_net.weight"]), sd["head.fraction_net.bias"])

    # ------------------------------------------------------------------ flat -> reference state_dict entries
    def unpack(self, flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        B = self.blocks

        def get(block: Block, r0: int, rows: int):
            N, K = block.N, block.K
            w = flat[block.offset + r0 * K: block.offset + (r0 + rows) * K].reshape(rows, K).clone()
            b = flat[block.offset + N * K + r0: block.offset + N * K + r0 + rows].clone()
            return w, b

        w, b = get(B["conv1"], 0, 32)
        out["encoder.convs.0.weight"], out["encoder.convs.0.bias"] = w.reshape(32, self.C, 8, 8), b
        w, b = get(B["conv2"], 0, 64)
        out["encoder.convs.2.weight"], out["encoder.convs.2.bias"] = w.reshape(64, 4, 4, 32).permute(0, 3, 1, 2).contiguous(), b
        w, b = get(B["conv3"], 0, 64)
        out["encoder.convs.4.weight"], out["encoder.convs.4.bias"] = w.reshape(64, 3, 3, 64).permute(0, 3, 1, 2).contiguous(), b
        for kind, wk, bk in ((".mu", "weight_mu", "bias_mu"), (".sigma", "weight_sigma", "bias_sigma")) if self.noisy else (("", "weight", "bias"),):
            w, b = get(B["fc1" + kind], 0, 512)
            out[f"head.first_dense.{wk}"], out[f"head.first_dense.{bk}"] = self._feat_cols_to_chw(w).contiguous(), b
            w, b = get(B["head" + kind], 0, self.Nq)
            out[f"head.q_head.{wk}"], out[f"head.q_head.{bk}"] = w, b
            if self.dueling:
                w, b = get(B["head" + kind], self.Nq, self.V)
                out[f"head.value_head.{wk}"], out[f"head.value_head.{bk}"] = w, b
        if self.quantile:
            w, b = get(B["cos"], 0, self.feat)
            out["head.cosine_emb.0.weight"] = w.reshape(self.H3, self.W3, 64, self.num_cosines).permute(2, 0, 1, 3).reshape(self.feat, self.num_cosines).contiguous()
            out["head.cosine_emb.0.bias"] = b.reshape(self.H3, self.W3, 64).permute(2, 0, 1).reshape(self.feat).contiguous()
        if self.algo == "fqf":
            w, b = get(B["frac"], 0, self.F)
            out["head.fraction_net.weight"], out["head.fraction_net.bias"] = self._feat_cols_to_chw(w).contiguous(), b
        return out

    def noise_in_to_kernel(self, prefix: str, v: torch.Tensor) -> torch.Tensor:
        """noise_in of first_dense indexes features in (c,h,w) order; the kernels want (h,w,c)."""
        return self._feat_cols_to_hwc(v) if prefix == "head.first_dense" else v

    def noise_in_from_kernel(self, prefix: str, v: torch.Tensor) -> torch.Tensor:
        return self._feat_cols_to_chw(v) if prefix == "head.first_dense" else v
