"""Device-side network and learner: composes the HIP kernels (through ``ops``) into the forward passes, losses,
backward passes and optimizer steps of the six reference learners.

``ops`` is the only door to the GPU (agent0_amd.ops.HipOps).  The composition is written against that interface so
the CPU test-suite can drive the very same code with an emulation backend (tests/cpu_ops.py) that runs the shared
C++ layer orchestration on the host — the product itself never does that.

Reference mapping (paths relative to the reference repo):
  DeviceNet.forward / qvalues   agent0/deepq/model.py:323-330 + heads 123-131, 163-177, 190-192, 219-257, 268-284
  DeviceLearner.update          agent0/deepq/agent.py:124-169 (BaseLearner.train) + train_step of each learner
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch

from .layout import Block, NetLayout

MODE_IDENT, MODE_MEAN, MODE_C51, MODE_FQF = 0, 1, 2, 3


class Workspace:
    """Activations of one forward pass over ``B`` observations (``n_tau`` quantile samples each for IQN/FQF)."""

    def __init__(self, ops, L: NetLayout, B: int, n_tau: int = 1, grads: bool = False):
        self.B, self.n_tau, self.grads = B, n_tau, grads
        R = B * n_tau if L.quantile else B
        self.R = R
        self.act1 = ops.empty(B * L.H1 * L.W1 * 32)
        self.act2 = ops.empty(B * L.H2 * L.W2 * 64)
        self.act3 = ops.empty(B * L.feat)
        self.h = ops.empty(R * 512)
        self.raw = ops.empty(R * L.Npad)
        self.q = ops.empty(R * L.A * L.T)
        if L.quantile:
            self.cosx = ops.empty(R * L.num_cosines)
            self.emb = ops.empty(R * L.feat)
            self.x = ops.empty(R * L.feat)
            self.taus = ops.empty(R)
        if L.algo == "fqf":
            self.frac_logits = ops.empty(B * L.Fpad)
            self.tau_all = ops.empty(B * (L.F + 1))
            self.tau_hat = ops.empty(B * L.F)
        if grads:
            self.dq = ops.zeros(R * L.A * L.T)
            self.draw = ops.empty(R * L.Npad)
            self.dh = ops.empty(R * 512)
            self.d3 = ops.empty(B * L.feat)
            self.d2 = ops.empty(B * L.H2 * L.W2 * 64)
            self.d1 = ops.empty(B * L.H1 * L.W1 * 32)
            if L.quantile:
                self.dx = ops.empty(R * L.feat)
                self.demb = ops.empty(R * L.feat)


class DeviceNet:
    """One set of network parameters on the device in the packed layout, plus forward-pass composition."""

    def __init__(self, ops, L: NetLayout, net_handle):
        self.ops, self.L, self.net = ops, L, net_handle
        self.flat = ops.zeros(L.n_params_padded)
        self.eff = ops.zeros(max(L.n_eff, 4)) if L.noisy else None
        # noise vectors in KERNEL order (noise_in of first_dense permuted to (h,w,c))
        # All of them live in ONE buffer, in the order reset_noise draws them and each padded to a multiple of four floats — the stride of
        # the Philox offsets (DeviceRng._advance) — so a single normal fill over the buffer produces exactly the draws of nine fills.
        self.noise: Dict[str, Dict[str, torch.Tensor]] = {}
        sizes = [(prefix, leaf, n) for prefix, block, r0, r1, in_f in L.noise_modules
                 for leaf, n in (("noise_in", in_f), ("noise_out_weight", r1 - r0), ("noise_out_bias", r1 - r0))]
        self.noise_buf = ops.zeros(max(sum((n + 3) // 4 * 4 for _, _, n in sizes), 4))
        off = 0
        for prefix, leaf, n in sizes:
            self.noise.setdefault(prefix, {})[leaf] = self.noise_buf[off:off + n]
            off += (n + 3) // 4 * 4
        self.noise_len = off
        self._noise_sizes = sizes
        self._scratch: Optional[torch.Tensor] = None
        # fused per-observation encoder (encoder_fused.hip): needs k-major copies of the conv weights, refreshed when they change
        self.fused = bool(ops.fused_supported(L.C, L.H, L.W))
        self.fused_dgrad = self.fused and bool(ops.dgrad_fused_supported(L.C, L.H, L.W))
        self.wt = ops.zeros(ops.conv_wt_floats(L.C)) if self.fused else None

    def adopt_noise_buf(self, buf: torch.Tensor):
        """Move the noise vectors into ``buf`` (noise_len floats of a buffer the caller owns): the learner puts its two networks' vectors back to back, in
        draw order, so that ONE normal fill serves both resets of a train call (agent.py:125-127)."""
        assert buf.numel() >= self.noise_len
        buf[: self.noise_len].copy_(self.noise_buf[: self.noise_len])
        self.noise_buf = buf[: self.noise_len]
        off = 0
        for prefix, leaf, n in self._noise_sizes:
            self.noise[prefix][leaf] = buf[off:off + n]
            off += (n + 3) // 4 * 4

    def refresh_wt(self):
        if self.fused:
            self.ops.conv_wt_refresh(self.encoder_weights(), self.L.C, self.wt)

    def copy_from(self, other: "DeviceNet"):
        """Device-to-device snapshot of another DeviceNet of the same layout: parameters, k-major conv copies, NoisyNet noise
        vectors and composed weights.  This is what replaces shipping ``state_dict()`` to an actor (launch.py:34-36,58-62)."""
        self.flat.copy_(other.flat)
        if self.wt is not None:
            self.wt.copy_(other.wt)
        if self.eff is not None:
            self.eff.copy_(other.eff)
        if self.noise:
            self.noise_buf.copy_(other.noise_buf)

    # ------------------------------------------------------------------ weights
    def block(self, name: str) -> Block:
        return self.L.blocks[name]

    def wb(self, name: str):
        """(W, b) views of the EFFECTIVE weights of a dense layer (composed ones for NoisyNet)."""
        if self.L.noisy and name in ("fc1", "head"):
            blk, buf = self.L.eff[name], self.eff
        else:
            blk, buf = self.L.blocks[name], self.flat
        return buf[blk.w], buf[blk.b]

    def encoder_weights(self):
        f, B = self.flat, self.L.blocks
        return {"w1": f[B["conv1"].w], "b1": f[B["conv1"].b], "w2": f[B["conv2"].w], "b2": f[B["conv2"].b], "w3": f[B["conv3"].w], "b3": f[B["conv3"].b]}

    def compose_mods(self):
        L = self.L
        mods = []
        for prefix, block, r0, r1, in_f in L.noise_modules:
            mu, sg, ef = L.blocks[block + ".mu"], L.blocks[block + ".sigma"], L.eff[block]
            nz = self.noise[prefix]
            mods.append((self.flat[mu.all], self.flat[sg.all], self.eff[ef.all], mu.N, mu.K, r0, r1, nz["noise_in"], nz["noise_out_weight"], nz["noise_out_bias"]))
        return mods

    def compose_noise(self):
        """NoisyLinear.reset_noise's weight_epsilon/bias_epsilon + forward composition (model.py:54-62,78-83)."""
        mods = self.compose_mods()
        if mods:
            self.ops.noisy_multi(False, mods)              # the two or three modules in one launch

    def set_noise(self, prefix: str, noise_in, noise_out_weight, noise_out_bias):
        """Install the three noise vectors of one NoisyLinear (reference order/layout; model.py:73-76)."""
        nz = self.noise[prefix]
        dev = self.flat.device
        nz["noise_in"].copy_(self.L.noise_in_to_kernel(prefix, torch.as_tensor(noise_in, dtype=torch.float32).to(dev)))
        nz["noise_out_weight"].copy_(torch.as_tensor(noise_out_weight, dtype=torch.float32).to(dev))
        nz["noise_out_bias"].copy_(torch.as_tensor(noise_out_bias, dtype=torch.float32).to(dev))

    def load_state_dict(self, sd):
        """Reference-format state_dict (keys/shapes of agent0/deepq/model.py) -> packed device parameters."""
        dev = self.flat.device
        sd = {k: torch.as_tensor(v).to(dev) for k, v in sd.items()}
        self.L.pack(sd, self.flat)
        self.refresh_wt()
        for prefix, *_ in self.L.noise_modules:
            if f"{prefix}.noise_in" in sd:
                self.set_noise(prefix, sd[f"{prefix}.noise_in"], sd[f"{prefix}.noise_out_weight"], sd[f"{prefix}.noise_out_bias"])
        if self.L.noisy:
            self.compose_noise()

    def state_dict(self):
        """Packed device parameters -> reference-format entries (trainables + NoisyNet buffers)."""
        out = self.L.unpack(self.flat)
        for prefix, *_ in self.L.noise_modules:
            nz = self.noise[prefix]
            nin = self.L.noise_in_from_kernel(prefix, nz["noise_in"]).clone()
            f = lambda x: x.sign() * x.abs().sqrt()
            out[f"{prefix}.weight_epsilon"] = torch.outer(f(nz["noise_out_weight"]), f(nin))
            out[f"{prefix}.bias_epsilon"] = f(nz["noise_out_bias"])
            out[f"{prefix}.noise_in"] = nin
            out[f"{prefix}.noise_out_weight"] = nz["noise_out_weight"].clone()
            out[f"{prefix}.noise_out_bias"] = nz["noise_out_bias"].clone()
        return out

    def scratch(self, n: int) -> Optional[torch.Tensor]:
        if n <= 0:
            return None
        if self._scratch is None or self._scratch.numel() < n:
            self._scratch = self.ops.empty(n)
        return self._scratch

    # ------------------------------------------------------------------ forward
    def encode(self, ws: Workspace, frames, slot, sample_stride, chan_off, B, keep: bool = True):
        """frames -> act3 (and act1/act2 when ``keep``: only a pass that is differentiated needs them in HBM)."""
        if self.fused:
            self.ops.encoder_fwd_fused(self.net, self.wt, self.encoder_weights(), frames, slot, sample_stride, chan_off, B,
                                       ws.act1 if keep else None, ws.act2 if keep else None, ws.act3)
        else:
            self.ops.encoder_fwd(self.net, self.encoder_weights(), frames, slot, sample_stride, chan_off, B, ws.act1, ws.act2, ws.act3)

    def _dense(self, X, ldx, name, Y, R, relu):
        W, b = self.wb(name)
        blk = self.L.eff[name] if (self.L.noisy and name in ("fc1", "head")) else self.L.blocks[name]
        N, K = blk.N, blk.K
        self.ops.dense_fwd(X, ldx, W, b, Y, R, N, K, relu, self.scratch(self.ops.dense_fwd_scratch(R, N, K)))

    def head(self, ws: Workspace, B, taus: Optional[torch.Tensor] = None, n_tau: int = 1, feat: Optional[torch.Tensor] = None):
        """features -> q.  Dense algos: q [B][A][T].  Quantile algos: taus [B*n_tau] -> q [B][n_tau][A].
        ``feat`` overrides ws.act3 (evaluating a head on another pass's features)."""
        L, ops = self.L, self.ops
        feat = ws.act3 if feat is None else feat
        if not L.quantile:
            self._dense(feat, L.feat, "fc1", ws.h, B, True)
            self._dense(ws.h, 512, "head", ws.raw, B, False)
            ops.dueling_fwd(ws.raw, L.Npad, ws.q, B, L.A, L.T, L.dueling)
            return ws.q
        R = B * n_tau
        ops.cos_features(taus, ws.cosx, R, L.num_cosines)
        if not ws.grads and ops.dense_fwd_scratch(R, L.feat, L.num_cosines) == 0:
            # a pass that is not differentiated: embedding x features in the GEMM's epilogue, the embedding never reaches HBM
            Wc, bc = self.wb("cos")
            ops.dense_fwd_mul(ws.cosx, L.num_cosines, Wc, bc, feat, n_tau, ws.x, R, L.feat, L.num_cosines, True)
        elif ws.grads and ops.dense_fwd_mul_keep_ok(R, L.feat, L.num_cosines, L.num_cosines):
            # the differentiated pass: the embedding is kept for the backward pass, embedding x features written beside it in the same launch
            Wc, bc = self.wb("cos")
            ops.dense_fwd_mul_keep(ws.cosx, L.num_cosines, Wc, bc, feat, n_tau, ws.emb, ws.x, R, L.feat, L.num_cosines, True)
        else:
            self._dense(ws.cosx, L.num_cosines, "cos", ws.emb, R, True)
            ops.hadamard_fwd(ws.emb, feat, ws.x, B, n_tau, L.feat)
        self._dense(ws.x, L.feat, "fc1", ws.h, R, True)
        self._dense(ws.h, 512, "head", ws.raw, R, False)
        ops.dueling_fwd(ws.raw, L.Npad, ws.q, R, L.A, 1, L.dueling)
        return ws.q

    def refresh_fc1_planes(self):
        """fc1's EFFECTIVE weights as bf16 term planes (a0_split_planes) for ``head_slabs(.., w_planes=True)``: the actor refreshes them when a rollout starts and after
        every NoisyNet compose, and reuses them for the rollout's GEMMs of E * K rows (same exact terms as the GEMM would form per tile: same bits)."""
        W, _ = self.wb("fc1")
        if getattr(self, "fc1_planes", None) is None:
            self.fc1_planes = torch.empty(self.ops.weight_planes_words(512, self.L.feat), dtype=torch.int32, device=self.flat.device)
        self.ops.split_planes(W, self.fc1_planes, 512, self.L.feat)

    def head_slabs(self, ws: Workspace, B, taus: torch.Tensor, n_tau: int, slabs: torch.Tensor, cos_ready: bool = False, w_planes: bool = False) -> int:
        """Quantile heads of a pass that is not differentiated, up to the head GEMM's split-K slabs [ns][B * n_tau][Npad] (the consumer kernel finishes
        the layer: a0_actor_quantile_tail_env_step).  Returns the slab count.  ``cos_ready``: ``ws.cosx`` already holds the fractions' cosine features (the launch that
        produced the fractions wrote them: a0_tau_cos_features / a0_fqf_taus_cos)."""
        L, ops = self.L, self.ops
        R = B * n_tau
        if not cos_ready:
            ops.cos_features(taus, ws.cosx, R, L.num_cosines)
        Wc, bc = self.wb("cos")
        ops.dense_fwd_mul(ws.cosx, L.num_cosines, Wc, bc, ws.act3, n_tau, ws.x, R, L.feat, L.num_cosines, True)
        if w_planes:      # ``refresh_fc1_planes`` has run since fc1's effective weights last changed
            ops.dense_fwd_wplanes(ws.x, L.feat, self.fc1_planes, self.wb("fc1")[1], ws.h, R, 512, L.feat, True)
        else:
            self._dense(ws.x, L.feat, "fc1", ws.h, R, True)
        Wh, _ = self.wb("head")
        return ops.dense_fwd_partial(ws.h, 512, Wh, R, L.Npad, 512, slabs)

    def fc1(self, ws: Workspace, B):
        """features -> relu(fc1) only (the fused DQN head kernel takes it from there)."""
        self._dense(ws.act3, self.L.feat, "fc1", ws.h, B, True)

    def fqf_taus(self, ws: Workspace, B, with_cos: bool = False):
        """FQFHead.prop_taus (model.py:268-278): fraction net on (detached) features -> taus, tau_hats (``with_cos``: and the tau_hats' cosine features into ``ws.cosx``)."""
        L = self.L
        self._dense(ws.act3, L.feat, "frac", ws.frac_logits, B, False)
        if with_cos:
            self.ops.fqf_taus_cos(ws.frac_logits, L.Fpad, ws.tau_all, ws.tau_hat, ws.cosx, L.num_cosines, B, L.F)
        else:
            self.ops.fqf_taus(ws.frac_logits, L.Fpad, ws.tau_all, ws.tau_hat, B, L.F)

    def select(self, ws: Workspace, B, n_tau, a_star, qsel=None, qmax=None, atoms=None):
        """argmax_a head.qval (greedy action) from the activations in ``ws``."""
        L, ops = self.L, self.ops
        if L.algo in ("dqn", "mdqn"):
            ops.select_action(ws.q, L.A, 1, 1, B, L.A, 1, MODE_IDENT, None, a_star, qsel, qmax)
        elif L.algo == "qr":
            ops.select_action(ws.q, L.A * L.T, L.T, 1, B, L.A, L.T, MODE_MEAN, None, a_star, qsel, qmax)
        elif L.algo == "c51":
            ops.select_action(ws.q, L.A * L.T, L.T, 1, B, L.A, L.T, MODE_C51, atoms, a_star, qsel, qmax)
        elif L.algo == "iqn":
            ops.select_action(ws.q, n_tau * L.A, 1, L.A, B, L.A, n_tau, MODE_MEAN, None, a_star, qsel, qmax)
        else:  # fqf: sum_i (tau_{i+1} - tau_i) q(tau_hat_i)
            ops.select_action(ws.q, n_tau * L.A, 1, L.A, B, L.A, n_tau, MODE_FQF, ws.tau_all, a_star, qsel, qmax)


class DeviceLearner:
    """Online + target network, gradients, Adam / RMSprop state and the per-algorithm update."""

    def __init__(self, ops, L: NetLayout, batch_size: int, *, discount=0.99, n_step=1, double_q=False, lr=5e-4,
                 target_update_freq=500, vmin=-10.0, vmax=10.0, K=32, N=64, N_dash=64, max_grad_norm=-1.0, adam_eps=None,
                 mdqn_tau=0.03, mdqn_lo=-1.0):
        self.ops, self.L, self.B = ops, L, batch_size
        self.net = ops.net(L.C, L.H, L.W)
        self.online = DeviceNet(ops, L, self.net)
        self.target = DeviceNet(ops, L, self.net)
        self.noise_joint = None
        if L.noisy and self.online.noise_len % 4 == 0:
            # both networks' noise vectors back to back in the order a train call draws them (online, then target: agent.py:125-127)
            n = self.online.noise_len
            self.noise_joint = ops.zeros(2 * n)
            self.online.adopt_noise_buf(self.noise_joint[:n])
            self.target.adopt_noise_buf(self.noise_joint[n:])
        self.grads = ops.zeros(L.n_params_padded + 4)     # tail slot [n_params_padded]: the NaN flag as a float, reduced with the dense bucket
        self.adam_m = ops.zeros(L.n_params_padded)
        self.adam_v = ops.zeros(L.n_params_padded)
        self.state = ops.zeros(8, dtype=torch.int32)
        self.scalars = ops.zeros(4)
        # batch means of the last 1024 updates' per-sample losses, written by the Adam launch itself (a0_adam_step_sync_wt: ring slot = state[6] % 1024)
        self.loss_ring = ops.zeros(1024)
        self.discount, self.n_step, self.double_q = discount, n_step, double_q
        self.gamma_n = float(discount ** n_step)
        self.lr, self.target_update_freq = lr, target_update_freq
        self.adam_eps = (1e-2 / batch_size) if adam_eps is None else adam_eps
        self.vmin, self.vmax = float(vmin), float(vmax)
        self.K, self.N, self.N_dash = K, N, N_dash
        self.max_grad_norm = max_grad_norm
        self.mdqn_tau, self.mdqn_lo = float(mdqn_tau), float(mdqn_lo)
        B = batch_size
        if L.quantile:
            n_on = N if L.algo == "iqn" else L.F
            n_tg = N_dash if L.algo == "iqn" else L.F
            n_sel = K if L.algo == "iqn" else L.F
            self.ws_o = Workspace(ops, L, B, n_on, grads=True)
            self.ws_t = Workspace(ops, L, B, max(n_tg, n_sel))
            self.ws_s = Workspace(ops, L, B, n_sel) if double_q else None
            if L.algo == "fqf":
                self.ws_f = Workspace(ops, L, B, L.F)        # q at taus[1:-1] for the fraction loss (F-1 used)
                self.rms_sq = ops.zeros(L.blocks["frac"].size)
                self.frac_loss = ops.empty(B)
                self.dfrac_logits = ops.zeros(B * L.Fpad)
                self.clip = ops.zeros(1)
            self.y = ops.empty(B * n_tg)
        else:
            self.ws_o = Workspace(ops, L, B, grads=True)
            self.ws_t = Workspace(ops, L, B)
            self.ws_s = Workspace(ops, L, B) if double_q else None
            if L.algo == "qr":
                self.y = ops.empty(B * L.T)
                self.qr_taus = ((2 * torch.arange(L.T, dtype=torch.float32) + 1) / (2.0 * L.T)).to(ops.device)
            if L.algo == "mdqn":
                self.ws_m = Workspace(ops, L, B)
        if L.algo == "c51":
            self.atoms = torch.linspace(self.vmin, self.vmax, L.T).to(ops.device)
            self.m_proj = ops.empty(B * L.T)
        self.a_star = ops.zeros(B, dtype=torch.int32)
        self.loss = ops.empty(B)
        n_slab = max(ops.encoder_bwd_scratch(self.net, B),
                     ops.dense_wgrad_scratch(self.ws_o.R, L.Npad, 512), ops.dense_wgrad_scratch(self.ws_o.R, 512, L.feat),
                     ops.dense_wgrad_scratch(self.ws_o.R, L.feat, L.num_cosines) if L.quantile else 0,
                     ops.dense_wgrad_scratch(B, L.Fpad, L.feat) if L.algo == "fqf" else 0, 4)
        # head + fc1 (+ cosine embedding) reduce in disjoint regions of one launch
        shapes = [(self.ws_o.R, L.Npad, 512), (self.ws_o.R, 512, L.feat)] + ([(self.ws_o.R, L.feat, L.num_cosines)] if L.quantile else [])
        n_slab = max(n_slab, ops.dense_wgrad_multi_scratch(shapes))
        # Round 4: without a gradient hook the dense layers' slab reductions ride in the encoder weight gradients' reduction launch (a0_pending_reduce: one launch less
        # per update, same sums) — their slabs then have to survive until that launch, so the encoder's get a region of their own behind them
        self._enc_slab_off, self._defer_dense = 0, False
        if hasattr(ops, "pending_reduce") and self.online.fused and self.online.fused_dgrad and os.environ.get("A0_DEFER_DENSE_REDUCE", "1") != "0":
            self._enc_slab_off, self._defer_dense = ops.dense_wgrad_multi_scratch(shapes), True
            n_slab = max(n_slab, self._enc_slab_off + ops.encoder_bwd_scratch(self.net, B))
        self._pend = None
        self.slabs = ops.empty(n_slab)
        self.obs_bytes = L.C * L.H * L.W
        self.grad_hook = None       # data parallelism: callable(grads, state) run between backward and the optimizer (dist.GradAllReduce)

    # ------------------------------------------------------------------ helpers
    def sync_target(self, force=True):
        self.ops.target_sync(self.target.flat, self.online.flat, self.L.n_params_padded, self.state, force)
        self.target.refresh_wt()

    def _grad(self, name: str) -> torch.Tensor:
        L = self.L
        key = name + ".mu" if (L.noisy and name in ("fc1", "head")) else name
        return self.grads[L.blocks[key].all]

    def _backward_dense(self, ws: Workspace, B, have_draw: bool = False, have_dh: bool = False):
        """dq (w.r.t. the combined head output) -> the gradients of every dense block (head, fc1, cosine embedding, NoisyNet sigmas)
        and d3, the gradient w.r.t. the encoder output.  After this call the flat gradient range [L.conv_end, L.n_adam) is final:
        the data-parallel exchange of that range (95 % of the parameters) can run while the encoder backward is still computing."""
        L, ops, on = self.L, self.ops, self.online
        R, T = ws.R, (1 if L.quantile else L.T)
        if not have_draw:
            ops.dueling_bwd(ws.dq, ws.draw, L.Npad, R, L.A, T, L.dueling)
        Wh, _ = on.wb("head")
        Wf, _ = on.wb("fc1")
        # the data gradients first, then every dense weight gradient with ONE slab reduction
        wg = [(ws.draw, ws.h, 512, self._grad("head"), R, L.Npad, 512)]
        if not have_dh:
            if hasattr(ops, "dense_dgrad_wgrad_ok2") and ops.dense_dgrad_wgrad_ok2(R, L.Npad, 512):
                # round 5: a distributional head's data gradient (64 tiles) and weight gradient (32 - 104 tiles) side by side in one launch
                ops.dense_dgrad_wgrad(ws.draw, Wh, ws.h, 512, ws.dh, self._grad("head"), R, L.Npad, 512)
                wg = []
            else:
                ops.dense_dgrad(ws.draw, Wh, ws.h, ws.dh, R, L.Npad, 512)
        if not L.quantile:
            if wg and hasattr(ops, "dense_dgrad_wgrad2") and ops.dense_dgrad_wgrad2_ok(R, 512, L.feat, L.Npad, 512):
                # round 5: ... and the head's weight gradient, which waits for the loss kernel only, in the same launch (its few tiles fill slots the pair leaves empty)
                ops.dense_dgrad_wgrad2(ws.dh, Wf, ws.act3, L.feat, ws.d3, self._grad("fc1"), R, 512, L.feat, ws.draw, ws.h, 512, self._grad("head"), L.Npad, 512)
                wg = []
            elif hasattr(ops, "dense_dgrad_wgrad") and ops.dense_dgrad_wgrad_ok(R, 512, L.feat):
                # fc1's data gradient and weight gradient — 392 tiles each at R = 512 — as ONE launch that keeps the chip's workgroup slots filled (bit-identical)
                ops.dense_dgrad_wgrad(ws.dh, Wf, ws.act3, L.feat, ws.d3, self._grad("fc1"), R, 512, L.feat)
            else:
                wg.append((ws.dh, ws.act3, L.feat, self._grad("fc1"), R, 512, L.feat))
                ops.dense_dgrad(ws.dh, Wf, ws.act3, ws.d3, R, 512, L.feat)
        else:
            n = ws.n_tau
            wg.append((ws.dh, ws.x, L.feat, self._grad("fc1"), R, 512, L.feat))
            if hasattr(ops, "dense_dgrad_hadamard") and ops.dense_dgrad_hadamard_ok(R, 512, L.feat, n):
                # round 6: dx = dh W never reaches HBM — the embedding product's backward runs in the data-gradient GEMM's epilogue
                ops.dense_dgrad_hadamard(ws.dh, Wf, ws.emb, ws.act3, ws.demb, ws.d3, R, 512, L.feat, n)
            else:
                ops.dense_dgrad(ws.dh, Wf, None, ws.dx, R, 512, L.feat)
                ops.hadamard_bwd(ws.dx, ws.emb, ws.act3, ws.demb, ws.d3, B, n, L.feat)
            wg.append((ws.demb, ws.cosx, L.num_cosines, self._grad("cos"), R, L.feat, L.num_cosines))
        # data parallelism exchanges the dense range right after this call: its reductions cannot wait for the encoder's launch then
        defer = self._defer_dense and self.grad_hook is None
        self._pend = ops.pending_reduce() if defer else None
        if wg:
            ops.dense_wgrad_multi(wg, self.slabs, **({"pend": self._pend} if defer else {}))
        if L.noisy and not defer:
            self._noisy_sigma_grads()

    def _noisy_sigma_grads(self):
        """d sigma = d eff * eps for every NoisyLinear module, from the (reduced) gradients in the mu blocks."""
        L, on = self.L, self.online
        mods = []
        for prefix, block, r0, r1, in_f in L.noise_modules:
            mu, sg = L.blocks[block + ".mu"], L.blocks[block + ".sigma"]
            nz = on.noise[prefix]
            mods.append((self.grads[mu.all], None, self.grads[sg.all], mu.N, mu.K, r0, r1, nz["noise_in"], nz["noise_out_weight"], nz["noise_out_bias"]))
        self.ops.noisy_multi(True, mods)

    def backward_encoder(self):
        """d3 -> the three convolution blocks' gradients (flat range [0, L.conv_end)); the second half of the backward pass, on the
        batch of the last forward_dense call."""
        L, ops, on = self.L, self.ops, self.online
        ws, frames, slot, stride, B = self._bw
        g1, g2, g3 = self.grads[L.blocks["conv1"].all], self.grads[L.blocks["conv2"].all], self.grads[L.blocks["conv3"].all]
        if on.fused and on.fused_dgrad:
            # both data gradients per observation in one kernel (LDS-resident d2), then the three weight-gradient GEMMs
            ops.encoder_dgrad_fused(self.net, on.wt, ws.d3, ws.act1, ws.act2, B, ws.d2, ws.d1)
            pend, self._pend = self._pend, None
            if pend is not None:
                ops.encoder_wgrad(self.net, on.encoder_weights(), frames, slot, stride, 0, B, ws.act1, ws.act2, ws.d3, ws.d2, ws.d1, g1, g2, g3, self.slabs[self._enc_slab_off:],
                                  pend=pend)
                if L.noisy:
                    self._noisy_sigma_grads()
            else:
                ops.encoder_wgrad(self.net, on.encoder_weights(), frames, slot, stride, 0, B, ws.act1, ws.act2, ws.d3, ws.d2, ws.d1, g1, g2, g3, self.slabs)
        else:
            ops.encoder_bwd(self.net, on.encoder_weights(), frames, slot, stride, 0, B, ws.act1, ws.act2, ws.d3, ws.d2, ws.d1, g1, g2, g3, self.slabs)

    # ------------------------------------------------------------------ the update
    def update(self, frames, slot, sample_stride, act, rew, done, wgt, rand: Optional[List[torch.Tensor]] = None, tstage: Optional[int] = None):
        """One BaseLearner.train step (agent.py:124-169) on a batch that stays on the device: forward + backward, the data-parallel
        gradient exchange when a ``grad_hook`` is installed, optimizer step.  ``tstage``: the target network's pass on this batch has already
        run into the buffers of that parity (``target_stage``)."""
        out = self.forward_dense(frames, slot, sample_stride, act, rew, done, wgt, rand, tstage=tstage)
        self.exchange_begin()
        self.backward_encoder()
        self.exchange_end()
        self.apply()
        return out

    def _bucketed_hook(self) -> bool:
        return self.grad_hook is not None and getattr(self.grad_hook, "bucketed", False)

    def exchange_begin(self):
        """Data parallelism, first bucket: the dense blocks' gradients are final once forward_dense returns; a hook with a
        ``start_dense`` method (dist.GradAllReduce) reduces them asynchronously while backward_encoder runs."""
        h = self.grad_hook
        if self._bucketed_hook():
            h.start_dense(self.grads, self.L.conv_end, self.L.n_params_padded + 1)

    def exchange_end(self):
        """Second bucket (convolution blocks + the NaN flag) and the join with the first; a plain callable hook gets one call with
        the whole buffer instead."""
        h = self.grad_hook
        if h is None:
            return
        if self._bucketed_hook():
            h.finish(self.grads, None, self.L.conv_end)
        else:
            h(self.grads, self.state)

    def forward_backward(self, frames, slot, sample_stride, act, rew, done, wgt, rand: Optional[List[torch.Tensor]] = None):
        """Losses and every parameter gradient (into ``self.grads``); no parameter is modified (FQF's fraction net aside, which the
        reference also steps separately, agent.py:140-147)."""
        out = self.forward_dense(frames, slot, sample_stride, act, rew, done, wgt, rand)
        self.backward_encoder()
        return out

    def apply(self):
        """Adam on the flat buffer (NaN-skip and step counter on the device), refresh of the fused kernels' weight copies, target sync."""
        L, ops, on, tg = self.L, self.ops, self.online, self.target
        if L.algo == "fqf":           # unconditional, like the reference's fqf_optimizer.step() in front of the NaN guard (agent.py:139-148)
            blk = L.blocks["frac"]
            ops.rmsprop_step(on.flat[blk.all], self.grads[blk.all], self.rms_sq, blk.size, self.lr / 2e4, 0.95, 1e-5, self.max_grad_norm, self.clip)
        tail = self.grads[L.n_params_padded: L.n_params_padded + 1] if self._bucketed_hook() else None
        if on.fused:
            # two launches: Adam with its bookkeeping and the target copy folded in; the online conv copies, mirrored to the target's on a sync
            ops.adam_step_sync_wt(on.flat, self.grads, self.adam_m, self.adam_v, L.n_adam, self.state, self.scalars, self.lr, 0.9, 0.999, self.adam_eps,
                                  self.target_update_freq, tg.flat, L.n_params_padded, tail, on.encoder_weights(), L.C, on.wt, tg.wt, self.loss, self.B, self.loss_ring)
        else:
            ops.adam_step_sync(on.flat, self.grads, self.adam_m, self.adam_v, L.n_adam, self.state, self.scalars, self.lr, 0.9, 0.999, self.adam_eps,
                               self.target_update_freq, tg.flat, L.n_params_padded, tail)

    def _encode_passes(self, frames, slot, sample_stride, passes):
        """passes: [(net, ws, chan_off, keep)] — the encoder forward passes of one update over the same batch.  They are independent of one another (the reference
        runs them one after the other, agent.py:176-181 / 222-231): on the fused split-operand kernel they go out as ONE launch (a0_net_encoder_fwd_fused_multi)."""
        L, ops, B = self.L, self.ops, self.B
        multi = (len(passes) > 1 and self.online.fused and (L.C, L.H, L.W) == (4, 84, 84) and hasattr(ops, "encoder_fwd_fused_multi")
                 and os.environ.get("A0_ENC_MULTI", "1") != "0" and os.environ.get("A0_NO_X9") is None)
        if not multi:
            for net, ws, chan_off, keep in passes:
                net.encode(ws, frames, slot, sample_stride, chan_off, B, keep=keep)
            return
        ops.encoder_fwd_fused_multi(self.net, [(net.wt, net.encoder_weights(), frames, slot, sample_stride, chan_off, B, ws.act1 if keep else None, ws.act2 if keep else None, ws.act3)
                                               for net, ws, chan_off, keep in passes])

    # ------------------------------------------------------------------ the target network's pass as a stage of its own
    @property
    def target_stage_supported(self) -> bool:
        """The target network's forward pass on the next observations (encoder + fc1 GEMM: agent.py:176 / 222) depends on neither the online weights nor the
        previous update's gradients, so the Trainer may run it for batch k + 1 on a second stream while update k is in flight — for the paths whose
        tail consumes the target's fc1 slabs (dqn's and c51's fused heads) and without NoisyNet (whose per-update noise draws are ordered on the host)."""
        L = self.L
        if L.noisy or not self.online.fused:
            return False
        return (L.algo == "dqn" and L.A + (1 if L.dueling else 0) <= 24) or (L.algo == "c51" and hasattr(self.ops, "c51_head_loss_slabs") and os.environ.get("A0_C51_SEPARATE", "0") != "1")

    def _tstage_buf(self, p: int):
        if getattr(self, "_tst", None) is None:
            self._tst = {}
        if p not in self._tst:
            ns = self.ops.dense_fwd_partial_slabs(self.B, 512, self.L.feat)
            self._tst[p] = (self.ops.empty(self.B * self.L.feat), self.ops.empty(ns * self.B * 512))
        return self._tst[p]

    def target_stage(self, frames, slot, sample_stride, p: int):
        """target(next_obs) up to the fc1 GEMM's split-K slabs, into the buffers of parity ``p`` (what ``forward_dense(..., tstage=p)`` then consumes)."""
        L, ops, B, tg = self.L, self.ops, self.B, self.target
        act3, slabs = self._tstage_buf(p)
        Wf_t, _ = tg.wb("fc1")
        ops.encoder_fwd_fused(tg.net, tg.wt, tg.encoder_weights(), frames, slot, sample_stride, self.obs_bytes, B, None, None, act3)
        ops.dense_fwd_partial(act3, L.feat, Wf_t, B, 512, L.feat, slabs)

    def _qr_fused_ok(self) -> bool:
        """QR's head path from the GEMM slabs to the loss in one launch (a0_qr_head_loss_slabs): the staged head outputs of a sample must fit in LDS."""
        L, ops = self.L, self.ops
        if not hasattr(ops, "qr_head_loss_slabs") or os.environ.get("A0_QR_SEPARATE", "0") == "1":       # 1: tuning aid (same numbers, the eleven separate launches)
            return False
        R_on = 2 * self.B if self.double_q else self.B
        return (L.A <= 32 and (3 * L.Npad + (L.T + 3) // 4 * 4) * 4 <= 150 * 1024 and L.Npad % 4 == 0
                and max(ops.dense_fwd_partial_slabs(R_on, L.Npad, 512), ops.dense_fwd_partial_slabs(self.B, L.Npad, 512)) <= 8)

    def _dist_heads_to_slabs(self, frames, slot, sample_stride, tstage):
        """The distributional learners' passes (online on s, online on s' under double-Q, target on s': agent.py:219-231 / 272-280) up to the head GEMMs' split-K slabs:
        the encoders as one launch, the fc1 GEMMs as one grouped launch (the online passes' rows interleaved into one [splits][R_on][512] buffer), ONE reduction launch
        (bias + ReLU), the head GEMMs as one grouped launch.  Returns (buffers, online head slab count, target head slab count, R_on); buffers["hs_on"] holds rows
        [0, B) = s and, under double-Q, rows [B, 2B) = s'."""
        L, ops, B = self.L, self.ops, self.B
        on, tg = self.online, self.target
        wo, wt, wsel = self.ws_o, self.ws_t, self.ws_s
        nxt = self.obs_bytes
        dq_ = self.double_q
        ns = ops.dense_fwd_partial_slabs(B, 512, L.feat)
        R_on = 2 * B if dq_ else B
        ns_on = ops.dense_fwd_partial_slabs(R_on, 512, L.feat)
        if getattr(self, "_c51_buf", None) is None:
            nh_on, nh_tg = ops.dense_fwd_partial_slabs(R_on, L.Npad, 512), ops.dense_fwd_partial_slabs(B, L.Npad, 512)
            self._c51_buf = dict(fc1_on=ops.empty(ns_on * R_on * 512), fc1_tg=ops.empty(ns * B * 512), h_on=ops.empty(R_on * 512), act3_on=ops.empty(R_on * L.feat),
                                 hs_on=ops.empty(nh_on * R_on * L.Npad), hs_tg=ops.empty(nh_tg * B * L.Npad), R_on=R_on)
            wo.h = self._c51_buf["h_on"][: B * 512]                  # h(s) of the online network: what the backward pass reads
            # the online network's features of s and (double-Q) of s' back to back: fc1 and the head run over both as ONE GEMM each (same weights)
            wo.act3 = self._c51_buf["act3_on"][: B * L.feat]
            if dq_:
                wsel.act3 = self._c51_buf["act3_on"][B * L.feat:]
        buf = self._c51_buf
        (Wf_o, bf_o), (Wf_t, bf_t) = on.wb("fc1"), tg.wb("fc1")
        (Wh_o, _), (Wh_t, _) = on.wb("head"), tg.wb("head")
        s_tg = buf["fc1_tg"]
        self._encode_passes(frames, slot, sample_stride, ([(tg, wt, nxt, False)] if tstage is None else []) + ([(on, wsel, nxt, False)] if dq_ else []) + [(on, wo, 0, True)])
        npass = 3 if dq_ else 2
        grouped = (tstage is None and hasattr(ops, "dense_fwd_partial_multi") and ops.dense_fwd_partial_multi_ok(npass, B, 512, L.feat)
                   and ops.dense_fwd_partial_multi_ok(npass, B, L.Npad, 512))
        if grouped:
            # the passes (online on s, online on s' under double-Q, target on s') are GEMMs of one shape each for fc1 and for the head: ONE grouped launch per layer,
            # the online passes' rows interleaved into the [splits][R_on][N] buffers the reduction and the loss kernel read
            a3, f1 = buf["act3_on"], buf["fc1_on"]
            Xs = [a3[: B * L.feat]] + ([a3[B * L.feat:]] if dq_ else []) + [wt.act3]
            sl = [f1] + ([f1[B * 512:]] if dq_ else []) + [s_tg]
            st = [R_on * 512] * (npass - 1) + [B * 512]
            ns_on = ns = ops.dense_fwd_partial_multi(Xs, L.feat, [Wf_o] * (npass - 1) + [Wf_t], B, 512, L.feat, sl, st)
        else:
            if tstage is None:
                ops.dense_fwd_partial(wt.act3, L.feat, Wf_t, B, 512, L.feat, s_tg)
            else:
                s_tg = self._tstage_buf(tstage)[1]
            ops.dense_fwd_partial(buf["act3_on"], L.feat, Wf_o, R_on, 512, L.feat, buf["fc1_on"])
        layers = [(buf["fc1_on"], ns_on, bf_o, buf["h_on"], R_on), (s_tg, ns, bf_t, wt.h, B)]
        ops.reduce_bias_act_multi(layers, 512, True)
        if grouped:
            h, hs = buf["h_on"], buf["hs_on"]
            Xs = [h[: B * 512]] + ([h[B * 512:]] if dq_ else []) + [wt.h]
            sl = [hs] + ([hs[B * L.Npad:]] if dq_ else []) + [buf["hs_tg"]]
            st = [R_on * L.Npad] * (npass - 1) + [B * L.Npad]
            nh_on = nh_tg = ops.dense_fwd_partial_multi(Xs, 512, [Wh_o] * (npass - 1) + [Wh_t], B, L.Npad, 512, sl, st)
        else:
            nh_on = ops.dense_fwd_partial(buf["h_on"], 512, Wh_o, R_on, L.Npad, 512, buf["hs_on"])
            nh_tg = ops.dense_fwd_partial(wt.h, 512, Wh_t, B, L.Npad, 512, buf["hs_tg"])
        return buf, nh_on, nh_tg, R_on

    def forward_dense(self, frames, slot, sample_stride, act, rew, done, wgt, rand: Optional[List[torch.Tensor]] = None, tstage: Optional[int] = None):
        """Forward passes, losses and the dense half of the backward pass (every gradient but the convolution blocks', which
        backward_encoder adds); no parameter is modified (FQF's fraction net aside, which the reference also steps separately,
        agent.py:140-147).

        frames: u8 replay rows (st || st_next); slot: optional int32 row indices; act int32, rew/done/wgt fp32 [B].
        rand (IQN): [taus_K [B*K], taus_N' [B*N'], taus_N [B*N]] in the reference's draw order.
        rand (FQF, parity tests only): [taus [B*(F+1)], tau_hats [B*F]] of the online net on the observations, then the same pair for the
        action-selection pass on the next observations — fractions evaluated elsewhere (the oracle's), written over the fraction net's own
        so that both sides evaluate q(tau) at bit-identical fractions; the fraction net still runs (its logits feed the fraction loss).
        Returns the per-sample loss tensor (device) — and for FQF also the fraction loss.
        """
        L, ops, B = self.L, self.ops, self.B
        on, tg = self.online, self.target
        nxt = self.obs_bytes
        if tstage is not None and not self.target_stage_supported:
            raise ValueError("tstage: this learner's target pass cannot run as a separate stage")
        if L.noisy:
            mods = on.compose_mods() + tg.compose_mods()
            if len(mods) <= 6:
                ops.noisy_multi(False, mods)            # both networks' effective weights in one launch
            else:
                on.compose_noise()
                tg.compose_noise()
        algo = L.algo
        wo, wt, wsel = self.ws_o, self.ws_t, self.ws_s
        frac = None
        have_draw = have_dh = False
        if algo == "mdqn" and L.A + (1 if L.dueling else 0) <= 24 and hasattr(ops, "mdqn_head_loss_slabs") and os.environ.get("A0_MDQN_SEPARATE", "0") != "1":
            # round 5: dqn's path with the Munchausen target (a0_mdqn_head_loss_slabs) — the three passes' encoders in one launch, their fc1 GEMMs in one grouped launch,
            # and one kernel from the fc1 slabs to loss, head gradient and dh.  The third pass is the TARGET network on the current observation (agent.py:202-204).
            wm = self.ws_m
            (Wo, bo), (Wt, bt) = on.wb("head"), tg.wb("head")
            ns = ops.dense_fwd_partial_slabs(B, 512, L.feat)
            if getattr(self, "_fc1_slabs", None) is None or self._fc1_slabs[0].numel() < ns * B * 512:
                self._fc1_slabs = [ops.empty(ns * B * 512) for _ in range(3)]
                self._q_cur = ops.empty(B * L.A)
            (Wf_o, bf_o), (Wf_t, bf_t) = on.wb("fc1"), tg.wb("fc1")
            self._encode_passes(frames, slot, sample_stride, [(tg, wt, nxt, False), (tg, wm, 0, False), (on, wo, 0, True)])
            if hasattr(ops, "dense_fwd_partial_multi") and ops.dense_fwd_partial_multi_ok(3, B, 512, L.feat):
                ns = ops.dense_fwd_partial_multi([wo.act3, wt.act3, wm.act3], L.feat, [Wf_o, Wf_t, Wf_t], B, 512, L.feat, self._fc1_slabs[:3])
            else:
                ops.dense_fwd_partial(wt.act3, L.feat, Wf_t, B, 512, L.feat, self._fc1_slabs[1])
                ops.dense_fwd_partial(wm.act3, L.feat, Wf_t, B, 512, L.feat, self._fc1_slabs[2])
                ops.dense_fwd_partial(wo.act3, L.feat, Wf_o, B, 512, L.feat, self._fc1_slabs[0])
            ops.mdqn_head_loss_slabs(self._fc1_slabs[0], self._fc1_slabs[1], self._fc1_slabs[2], ns, bf_o, bf_t, wo.h, Wo, bo, Wt, bt, L.A, L.dueling, L.Npad, act, rew, done, wgt,
                                     self.gamma_n, self.mdqn_tau, self.mdqn_lo, B, self.loss, wo.q, wt.q, self._q_cur, wo.draw, self.state, wo.dh)
            have_dh = True
            have_draw = True
        elif algo == "mdqn":
            wm = self.ws_m
            # target net on the next AND on the current observation (agent.py:202-204), online net on the current one: one launch
            self._encode_passes(frames, slot, sample_stride, [(tg, wt, nxt, False), (tg, wm, 0, False), (on, wo, 0, True)])
            tg.head(wt, B)
            tg.head(wm, B)
            on.head(wo, B)
            ops.loss_mdqn(wo.q, wt.q, wm.q, L.A, act, rew, done, wgt, self.gamma_n, self.mdqn_tau, self.mdqn_lo, B, self.loss, wo.dq, self.state)
        elif algo == "dqn" and L.A + (1 if L.dueling else 0) <= 24:
            # heads, loss and head gradient in one kernel that also finishes fc1 from the GEMMs' split-K slabs (a0_dqn_head_loss_slabs):
            # only fc1 runs as a GEMM, and there are no reduction launches
            (Wo, bo), (Wt, bt) = on.wb("head"), tg.wb("head")
            ns = ops.dense_fwd_partial_slabs(B, 512, L.feat)
            if getattr(self, "_fc1_slabs", None) is None or self._fc1_slabs[0].numel() < ns * B * 512:
                self._fc1_slabs = [ops.empty(ns * B * 512) for _ in range(3 if self.double_q else 2)]
            (Wf_o, bf_o), (Wf_t, bf_t) = on.wb("fc1"), tg.wb("fc1")
            s_tg = self._fc1_slabs[1]
            self._encode_passes(frames, slot, sample_stride, ([(tg, wt, nxt, False)] if tstage is None else []) + ([(on, wsel, nxt, False)] if self.double_q else []) + [(on, wo, 0, True)])
            n_fc1 = 3 if self.double_q else 2
            if tstage is None and hasattr(ops, "dense_fwd_partial_multi") and ops.dense_fwd_partial_multi_ok(n_fc1, B, 512, L.feat):
                # the passes' fc1 GEMMs as ONE launch: together they fill the chip with a half / a third of the splits each would need alone
                ns = ops.dense_fwd_partial_multi([wo.act3, wt.act3] + ([wsel.act3] if self.double_q else []), L.feat, [Wf_o, Wf_t] + ([Wf_o] if self.double_q else []),
                                                 B, 512, L.feat, self._fc1_slabs[:n_fc1])
            else:
                if tstage is None:
                    ops.dense_fwd_partial(wt.act3, L.feat, Wf_t, B, 512, L.feat, s_tg)
                else:
                    s_tg = self._tstage_buf(tstage)[1]
                if self.double_q:
                    ops.dense_fwd_partial(wsel.act3, L.feat, Wf_o, B, 512, L.feat, self._fc1_slabs[2])
                ops.dense_fwd_partial(wo.act3, L.feat, Wf_o, B, 512, L.feat, self._fc1_slabs[0])
            ops.dqn_head_loss_slabs(self._fc1_slabs[0], s_tg, self._fc1_slabs[2] if self.double_q else None, ns, bf_o, bf_t, wo.h, Wo, bo, Wt, bt,
                                    L.A, L.dueling, L.Npad, act, rew, done, wgt, self.gamma_n, B, self.loss, wo.q, wt.q, wo.draw, self.state, wo.dh)
            have_dh = True           # ... and the head's backward-data pass: dh is written by the same kernel
            have_draw = True
        elif algo == "c51" and hasattr(ops, "c51_head_loss_slabs") and os.environ.get("A0_C51_SEPARATE", "0") != "1":      # 1: tuning aid (same numbers, the nine separate launches)
            # three fc1 GEMMs (their split-K slabs finished by ONE reduction launch), the online head as ONE GEMM over [s ; s'] rows, the target head, and
            # one launch for everything behind them (a0_c51_head_loss_slabs: slab sums, dueling, greedy next action, projection + cross entropy, head gradient)
            buf, nh_on, nh_tg, R_on = self._dist_heads_to_slabs(frames, slot, sample_stride, tstage)
            _, bh_o = on.wb("head")
            _, bh_t = tg.wb("head")
            ops.c51_head_loss_slabs(buf["hs_on"], nh_on, R_on, buf["hs_tg"], nh_tg, B if self.double_q else -1, bh_o, bh_t, L.Npad, L.A, L.T, L.dueling, act, rew, done, wgt,
                                    self.atoms, self.gamma_n, self.vmin, self.vmax, B, self.loss, wo.draw, self.state, q_on=wo.q, q_tg=wt.q, m_out=self.m_proj,
                                    a_star=self.a_star)
            have_draw = True
        elif algo == "qr" and self._qr_fused_ok():
            # round 5: the same layer structure as c51 (grouped fc1 GEMMs, one reduction launch, grouped head GEMMs) and ONE launch from the head slabs to the
            # quantile Huber loss and the head gradient (a0_qr_head_loss_slabs; agent.py:272-293)
            buf, nh_on, nh_tg, R_on = self._dist_heads_to_slabs(frames, slot, sample_stride, tstage)
            _, bh_o = on.wb("head")
            _, bh_t = tg.wb("head")
            ops.qr_head_loss_slabs(buf["hs_on"], nh_on, R_on, buf["hs_tg"], nh_tg, B if self.double_q else -1, bh_o, bh_t, L.Npad, L.A, L.T, L.dueling, act, rew, done, wgt,
                                   self.qr_taus, self.gamma_n, B, self.loss, wo.draw, self.state, q_on=wo.q, q_tg=wt.q, a_star=self.a_star)
            have_draw = True
        elif algo in ("dqn", "c51", "qr"):
            self._encode_passes(frames, slot, sample_stride, [(tg, wt, nxt, False)] + ([(on, wsel, nxt, False)] if self.double_q else []) + [(on, wo, 0, True)])
            tg.head(wt, B)
            if self.double_q:
                on.head(wsel, B)
                on.select(wsel, B, 1, self.a_star, atoms=getattr(self, "atoms", None))
            else:
                tg.select(wt, B, 1, self.a_star, atoms=getattr(self, "atoms", None))
            on.head(wo, B)
            if algo == "dqn":
                ops.loss_dqn(wo.q, wt.q, L.A, act, self.a_star, rew, done, wgt, self.gamma_n, B, self.loss, wo.dq, self.state)
            elif algo == "c51":
                ops.loss_c51(wo.q, wt.q, L.A, L.T, act, self.a_star, rew, done, wgt, self.atoms, self.gamma_n, self.vmin, self.vmax, B,
                             self.loss, wo.dq, self.m_proj, self.state)
            else:
                ops.quantile_target(wt.q, L.A * L.T, 1, L.T, self.a_star, rew, done, self.gamma_n, B, L.T, self.y)
                wo.dq.zero_()
                ops.loss_quantile_huber(wo.q, L.A * L.T, 1, L.T, self.y, self.qr_taus, 0, act, wgt, B, L.T, L.T, self.loss, wo.dq, self.state)
        elif algo == "iqn":
            t_sel, t_tgt, t_on = rand
            K, Nd, N = self.K, self.N_dash, self.N
            self._encode_passes(frames, slot, sample_stride, [(tg, wt, nxt, False)] + ([(on, wsel, nxt, False)] if self.double_q else []) + [(on, wo, 0, True)])
            if self.double_q:
                on.head(wsel, B, t_sel, K)
                on.select(wsel, B, K, self.a_star)
            else:
                tg.head(wt, B, t_sel, K)
                tg.select(wt, B, K, self.a_star)
            tg.head(wt, B, t_tgt, Nd)
            ops.quantile_target(wt.q, Nd * L.A, L.A, 1, self.a_star, rew, done, self.gamma_n, B, Nd, self.y)
            on.head(wo, B, t_on, N)
            wo.dq.zero_()
            ops.loss_quantile_huber(wo.q, N * L.A, L.A, 1, self.y, t_on, N, act, wgt, B, N, Nd, self.loss, wo.dq, self.state)
        elif algo == "fqf":
            F = L.F
            def taus(net, ws, k):
                net.fqf_taus(ws, B)
                if rand is not None:
                    ws.tau_all[: B * (F + 1)].copy_(rand[2 * k].reshape(-1)); ws.tau_hat[: B * F].copy_(rand[2 * k + 1].reshape(-1))
            self._encode_passes(frames, slot, sample_stride, [(on, wo, 0, True), (tg, wt, nxt, False)] + ([(on, wsel, nxt, False)] if self.double_q else []))
            taus(on, wo, 0)
            on.head(wo, B, wo.tau_hat, F)
            if self.double_q:
                taus(on, wsel, 1)
                on.head(wsel, B, wsel.tau_hat, F)
                on.select(wsel, B, F, self.a_star)
            else:
                taus(tg, wt, 1)
                tg.head(wt, B, wt.tau_hat, F)
                tg.select(wt, B, F, self.a_star)
            tg.head(wt, B, wo.tau_hat, F)           # quirk Q16: target evaluated at the ONLINE tau-hats
            ops.quantile_target(wt.q, F * L.A, L.A, 1, self.a_star, rew, done, self.gamma_n, B, F, self.y)
            wo.dq.zero_()
            ops.loss_quantile_huber(wo.q, F * L.A, L.A, 1, self.y, wo.tau_hat, F, act, wgt, B, F, F, self.loss, wo.dq, self.state)
            # fraction loss: q at the interior taus (no grad), its gradient w.r.t. the fraction logits, RMSprop
            wf = self.ws_f
            ops.fqf_inner_taus(wo.tau_all, wf.taus, B, F)                      # taus[:, 1:-1] -> [B][F-1]
            on.head(wf, B, wf.taus, F - 1, feat=wo.act3)
            ops.fqf_fraction_loss(wf.q, wo.q, wo.tau_all, act, wgt, B, F, L.A, L.Fpad, self.frac_loss, self.dfrac_logits, wo.frac_logits)
            gfr = self.grads[L.blocks["frac"].all]
            ops.dense_wgrad(self.dfrac_logits, wo.act3, L.feat, gfr, B, L.Fpad, L.feat, self.slabs)
            # the fraction net's RMSprop step (agent.py:140-147) runs in apply(): nothing in this update reads the fraction net again, and
            # under data parallelism its gradient has then been reduced with the dense bucket like every other block
            frac = self.frac_loss
        else:
            raise NotImplementedError(f"algo {algo} has no device learner yet")
        self._backward_dense(wo, B, have_draw, have_dh)
        self._bw = (wo, frames, slot, sample_stride, B)
        if self._bucketed_hook():      # the NaN flag rides at the tail of the dense gradient bucket
            ops.nan_flag_export(self.state, self.grads[L.n_params_padded: L.n_params_padded + 1])
        return (self.loss, frac) if frac is not None else self.loss
