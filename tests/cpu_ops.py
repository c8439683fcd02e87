"""CPU emulation backend with the same interface as agent0_amd.ops.HipOps — TEST INFRASTRUCTURE.

Lets the CPU test-suite drive agent0_amd.deepq.engine (parameter packing, forward/backward wiring, NoisyNet,
dueling, optimizer sequencing) without a GPU:
  * the GEMM-shaped layers run tests/host_emul.cpp, i.e. the SAME C++ orchestration + operand policies the HIP
    library uses (agent0_amd/csrc/net_impl.h, operands.h), evaluated in plain loops;
  * the small elementwise kernels are restated here with torch CPU ops.
The product never imports this module.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import torch

from agent0_amd._abi import EncoderWeights, FramesArg

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SAN = os.environ.get("A0_SANITIZE") == "1"          # tools/asan.sh: ASan + UBSan build of the host emulation
_SO = os.path.join(_HERE, "_build", *(("asan",) if _SAN else ()), "libhost_emul.so")


def build_emul() -> str:
    src = os.path.join(_HERE, "host_emul.cpp")
    deps = [src] + [os.path.join(_ROOT, "agent0_amd", "csrc", f) for f in ("net_impl.h", "operands.h", "net_tables.h", "a0_defs.h")]
    deps.append(os.path.join(_ROOT, "include", "agent0_hip.h"))
    if not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        opt = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if _SAN else ["-O2"]
        subprocess.check_call(["g++", *opt, "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", _SO, src])
    return _SO


class _Net:
    def __init__(self, lib, C_, H, W):
        self.lib = lib
        self.h = lib.emul_net_create(C_, H, W)
        assert self.h, "bad geometry"
        geo = (C.c_int * 8)()
        lib.emul_net_geometry(self.h, geo)
        self.C, self.H, self.W = C_, H, W
        self.H1, self.W1, self.H2, self.W2, self.H3, self.W3, self.feat, self.K1 = list(geo)
        self.K2, self.K3 = 512, 576


def _p(t):
    return None if t is None else t.data_ptr()


class CpuOps:
    name = "cpu-emulation"

    def __init__(self):
        lib = C.CDLL(build_emul())
        lib.emul_net_create.restype = C.c_void_p
        lib.emul_net_create.argtypes = [C.c_int] * 3
        lib.emul_net_geometry.argtypes = [C.c_void_p, C.c_void_p]
        lib.emul_encoder_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 3
        lib.emul_encoder_bwd_scratch.restype = C.c_longlong
        lib.emul_encoder_bwd_scratch.argtypes = [C.c_void_p, C.c_int]
        lib.emul_encoder_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 9
        lib.emul_dense_fwd_scratch.restype = C.c_longlong
        lib.emul_dense_fwd_scratch.argtypes = [C.c_int] * 3
        lib.emul_dense_fwd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.emul_dense_dgrad.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3
        lib.emul_dense_wgrad_scratch.restype = C.c_longlong
        lib.emul_dense_wgrad_scratch.argtypes = [C.c_int] * 3
        lib.emul_dense_wgrad.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        self.lib = lib
        self.device = torch.device("cpu")

    def empty(self, *shape, dtype=torch.float32):
        return torch.full(shape, float("nan") if dtype.is_floating_point else 0, dtype=dtype)  # poison: catches reads of unwritten memory

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype)

    def net(self, C_, H, W):
        return _Net(self.lib, C_, H, W)

    # ------------------------------------------------------------------ GEMM-shaped layers: shared C++ orchestration
    @staticmethod
    def _fa(frames, slot, stride, chan_off):
        return FramesArg(frames.data_ptr(), _p(slot), stride, chan_off)

    @staticmethod
    def _ew(w):
        return EncoderWeights(*[w[k].data_ptr() for k in ("w1", "b1", "w2", "b2", "w3", "b3")])

    def encoder_fwd(self, net, w, frames, slot, sample_stride, chan_off, B, act1, act2, act3):
        fa, ew = self._fa(frames, slot, sample_stride, chan_off), self._ew(w)
        self.lib.emul_encoder_fwd(net.h, C.addressof(ew), C.addressof(fa), B, _p(act1), _p(act2), _p(act3))

    def dgrad_fused_supported(self, C_, H, W):
        return False

    def fused_supported(self, C_, H, W):
        return False           # the fused encoder exists only as a HIP kernel; the emulation runs the unfused orchestration

    def encoder_bwd_scratch(self, net, B):
        return int(self.lib.emul_encoder_bwd_scratch(net.h, B))

    def encoder_bwd(self, net, w, frames, slot, sample_stride, chan_off, B, act1, act2, d3, d2, d1, g1, g2, g3, slabs):
        fa, ew = self._fa(frames, slot, sample_stride, chan_off), self._ew(w)
        assert slabs is None or slabs.numel() >= self.encoder_bwd_scratch(net, B)
        self.lib.emul_encoder_bwd(net.h, C.addressof(ew), C.addressof(fa), B, _p(act1), _p(act2), _p(d3), _p(d2), _p(d1), _p(g1), _p(g2), _p(g3), _p(slabs))

    def dense_fwd_scratch(self, R, N, K):
        return int(self.lib.emul_dense_fwd_scratch(R, N, K))

    def dense_wgrad_scratch(self, R, N, K):
        return int(self.lib.emul_dense_wgrad_scratch(R, N, K))

    def dense_fwd(self, X, ldx, W, b, Y, R, N, K, relu, scratch):
        assert N % 4 == 0 and K % 4 == 0 and ldx % 4 == 0 and W.numel() >= N * K and b.numel() >= N and Y.numel() >= R * N
        assert scratch is None or scratch.numel() >= self.dense_fwd_scratch(R, N, K)
        self.lib.emul_dense_fwd(_p(X), ldx, _p(W), _p(b), _p(Y), R, N, K, int(relu), _p(scratch))

    def dense_dgrad(self, dY, W, mask, dX, R, N, K):
        assert N % 4 == 0 and K % 4 == 0 and dX.numel() >= R * K
        self.lib.emul_dense_dgrad(_p(dY), _p(W), _p(mask), _p(dX), R, N, K)

    def dense_wgrad(self, dY, X, ldx, grad, R, N, K, slabs):
        assert N % 4 == 0 and K % 4 == 0 and grad.numel() >= N * K + N
        assert slabs is None or slabs.numel() >= self.dense_wgrad_scratch(R, N, K)
        self.lib.emul_dense_wgrad(_p(dY), _p(X), ldx, _p(grad), R, N, K, _p(slabs))

    # ------------------------------------------------------------------ heads / losses (torch restatements)
    def dueling_fwd(self, raw, ld, q, R, A, T, dueling):
        x = raw[: R * ld].view(R, ld)
        adv = x[:, : A * T].reshape(R, A, T)
        out = adv if not dueling else x[:, A * T: A * T + T].reshape(R, 1, T) + (adv - adv.sum(1, keepdim=True) / A)
        q[: R * A * T] = out.reshape(-1)

    def dueling_bwd(self, dq, draw, ld, R, A, T, dueling):
        g = dq[: R * A * T].view(R, A, T)
        out = torch.zeros(R, ld)
        if dueling:
            s = g.sum(1)
            out[:, : A * T] = (g - s.unsqueeze(1) / A).reshape(R, A * T)
            out[:, A * T: A * T + T] = s
        else:
            out[:, : A * T] = g.reshape(R, A * T)
        draw[: R * ld] = out.reshape(-1)

    def select_action(self, x, sb, sa, st, B, A, T, mode, aux, a_star, qsel, qmax):
        v = x.as_strided((B, A, T), (sb, sa, st))
        if mode == 0:
            val = v[:, :, 0]
        elif mode == 1:
            val = v.sum(-1) / T
        elif mode == 2:
            val = (v.softmax(-1) * aux[:T].view(1, 1, T)).sum(-1)
        else:
            tau = aux[: B * (T + 1)].view(B, T + 1)
            val = ((tau[:, 1:] - tau[:, :-1]).unsqueeze(1) * v).sum(-1)
        if a_star is not None:
            a_star[:B] = val.argmax(-1).to(torch.int32)
        if qsel is not None:
            qsel[: B * A] = val.reshape(-1)
        if qmax is not None:
            qmax[:B] = val.max(-1)[0]

    @staticmethod
    def _flag_nan(loss, state):
        if torch.isnan(loss).any():
            state[0] |= 1

    def loss_dqn(self, q, q_next, A, act, a_star, rew, done, wgt, gamma_n, B, loss, dq, state):
        ar = torch.arange(B)
        y = rew[:B] + (gamma_n * (1 - done[:B])) * q_next[: B * A].view(B, A)[ar, a_star[:B].long()]
        d = q[: B * A].view(B, A)[ar, act[:B].long()] - y
        l = torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5)
        loss[:B] = l
        g = torch.zeros(B, A)
        g[ar, act[:B].long()] = wgt[:B] * d.clamp(-1, 1)
        dq[: B * A] = g.reshape(-1)
        self._flag_nan(l, state)

    def loss_mdqn(self, q, q_next, q_cur_tgt, A, act, rew, done, wgt, gamma_n, tau, lo, B, loss, dq, state):
        ar = torch.arange(B)
        def lp(x):
            z = x - x.max(-1, keepdim=True)[0]
            return z - tau * torch.logsumexp(z / tau, -1, keepdim=True)
        qn, qc = q_next[: B * A].view(B, A), q_cur_tgt[: B * A].view(B, A)
        v_next = (qn.softmax(-1) * (qn - lp(qn))).sum(-1)
        add_on = lp(qc)[ar, act[:B].long()].clamp(lo, 0)
        y = rew[:B] + tau * add_on + (gamma_n * (1 - done[:B])) * v_next
        d = q[: B * A].view(B, A)[ar, act[:B].long()] - y
        l = torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5)
        loss[:B] = l
        g = torch.zeros(B, A)
        g[ar, act[:B].long()] = wgt[:B] * d.clamp(-1, 1)
        dq[: B * A] = g.reshape(-1)
        self._flag_nan(l, state)

    def loss_c51(self, logits, tgt_logits, A, T, act, a_star, rew, done, wgt, atoms, gamma_n, vmin, vmax, B, loss, dlogits, m_out, state):
        ar = torch.arange(B)
        p = tgt_logits[: B * A * T].view(B, A, T)[ar, a_star[:B].long()].softmax(-1)
        delta = (vmax - vmin) / (T - 1)
        tz = (rew[:B].view(-1, 1) + (gamma_n * (1 - done[:B].view(-1, 1))) * atoms[:T].view(1, -1)).clamp(vmin, vmax)
        b = (tz - vmin) / delta
        lo, up = b.floor().long(), b.ceil().long()
        lo = torch.where((up > 0) & (lo == up), lo - 1, lo)
        up = torch.where((lo < T - 1) & (lo == up), up + 1, up)
        m = torch.zeros(B, T)
        m.scatter_add_(1, lo, p * (up.float() - b))
        m.scatter_add_(1, up, p * (b - lo.float()))
        lg = logits[: B * A * T].view(B, A, T)[ar, act[:B].long()]
        logp = lg.log_softmax(-1)
        l = -(m * logp).sum(-1)
        loss[:B] = l
        g = torch.zeros(B, A, T)
        g[ar, act[:B].long()] = wgt[:B].view(-1, 1) * (logp.exp() * m.sum(-1, keepdim=True) - m)
        dlogits[: B * A * T] = g.reshape(-1)
        if m_out is not None:
            m_out[: B * T] = m.reshape(-1)
        self._flag_nan(l, state)

    def quantile_target(self, q_next, sb, sj, sa, a_star, rew, done, gamma_n, B, Nd, y):
        A = int(a_star[:B].max()) + 1
        v = q_next.as_strided((B, Nd, A), (sb, sj, sa))[torch.arange(B), :, a_star[:B].long()]
        y[: B * Nd] = (rew[:B].view(-1, 1) + (gamma_n * (1 - done[:B].view(-1, 1))) * v).reshape(-1)

    def loss_quantile_huber(self, q, sb, si, sa, y, taus, tb, act, wgt, B, N, Nd, loss, dq, state):
        A = int(act[:B].max()) + 1
        ar = torch.arange(B)
        qi = q.as_strided((B, N, A), (sb, si, sa))[ar, :, act[:B].long()]          # [B,N]
        tj = y[: B * Nd].view(B, Nd)
        tau = taus.as_strided((B, N), (tb, 1))
        d = qi.unsqueeze(1) - tj.unsqueeze(2)                                       # [B,Nd,N]
        h = torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5)
        w = (tau.unsqueeze(1) - (tj.unsqueeze(2) < qi.unsqueeze(1)).float()).abs()
        l = (h * w).sum(-1).mean(-1)
        loss[:B] = l
        g = wgt[:B].view(-1, 1) * (d.clamp(-1, 1) * w).sum(1) / Nd
        dq.as_strided((B, N, A), (sb, si, sa))[ar, :, act[:B].long()] = g
        self._flag_nan(l, state)

    # ------------------------------------------------------------------ IQN / FQF
    def cos_features(self, taus, out, R, D):
        ipi = (torch.tensor(math.pi, dtype=torch.float32) * torch.arange(1, D + 1, dtype=torch.float32)).view(1, D)
        out[: R * D] = (ipi * taus[:R].view(R, 1)).cos().reshape(-1)

    def hadamard_fwd(self, emb, feat, x, B, n, D):
        x[: B * n * D] = (emb[: B * n * D].view(B, n, D) * feat[: B * D].view(B, 1, D)).reshape(-1)

    def hadamard_bwd(self, dx, emb, feat, demb, d3, B, n, D):
        g, e, f = dx[: B * n * D].view(B, n, D), emb[: B * n * D].view(B, n, D), feat[: B * D].view(B, 1, D)
        demb[: B * n * D] = torch.where(e > 0, g * f, torch.zeros(())).reshape(-1)
        d3[: B * D] = torch.where(f[:, 0] > 0, (g * e).sum(1), torch.zeros(())).reshape(-1)

    def fqf_taus(self, logits, ld, taus, tau_hat, B, F):
        p = logits[: B * ld].view(B, ld)[:, :F].log_softmax(-1).exp()
        t = torch.cat((torch.zeros(B, 1), torch.cumsum(p, -1)), -1)
        taus[: B * (F + 1)] = t.reshape(-1)
        tau_hat[: B * F] = ((t[:, :-1] + t[:, 1:]) / 2.0).reshape(-1)

    def fqf_inner_taus(self, taus, out, B, F):
        out[: B * (F - 1)] = taus[: B * (F + 1)].view(B, F + 1)[:, 1:-1].reshape(-1)

    def fqf_fraction_loss(self, q, qh, taus, act, wgt, B, F, A, ldl, loss, dlogits, logits):
        ar = torch.arange(B)
        qi = q[: B * (F - 1) * A].view(B, F - 1, A)[ar, :, act[:B].long()]
        qhat = qh[: B * F * A].view(B, F, A)[ar, :, act[:B].long()]
        t = taus[: B * (F + 1)].view(B, F + 1)
        v1, v2 = qi - qhat[:, :-1], qi - qhat[:, 1:]
        s1 = qi > torch.cat((qhat[:, :1], qi[:, :-1]), 1)
        s2 = qi < torch.cat((qi[:, 1:], qhat[:, -1:]), 1)
        g = torch.where(s1, v1, -v1) + torch.where(s2, v2, -v2)
        loss[:B] = (g * t[:, 1:-1]).sum(1)
        p = logits[: B * ldl].view(B, ldl)[:, :F].softmax(-1)
        dp = torch.zeros(B, F)
        dp[:, : F - 1] = torch.flip(torch.cumsum(torch.flip(g, [1]), 1), [1])
        dp = dp * wgt[:B].view(-1, 1)
        out = torch.zeros(B, ldl)
        out[:, :F] = p * (dp - (p * dp).sum(-1, keepdim=True))
        dlogits[: B * ldl] = out.reshape(-1)

    # ------------------------------------------------------------------ optimizer
    def adam_step(self, params, grads, m, v, n, state, scalars, lr, b1, b2, eps, target_freq):
        skip = int(state[0]) != 0
        steps = int(state[1]) + (0 if skip else 1)
        if skip:
            state[2] += 1
        t = max(steps, 1)
        step_size = torch.tensor(lr / (1.0 - b1 ** t), dtype=torch.float32)
        bc2_sqrt = torch.tensor(math.sqrt(1.0 - b2 ** t), dtype=torch.float32)
        state[1], state[3], state[0] = steps, int(skip), 0
        state[4] = 1 if (target_freq > 0 and steps % target_freq == 0) else 0
        if skip:
            return
        g = grads[:n]
        m[:n] += (g - m[:n]) * torch.tensor(1.0 - b1, dtype=torch.float32)
        v[:n] = v[:n] * torch.tensor(b2, dtype=torch.float32) + (torch.tensor(1.0 - b2, dtype=torch.float32) * g) * g
        params[:n] -= step_size * (m[:n] / (v[:n].sqrt() / bc2_sqrt + torch.tensor(eps, dtype=torch.float32)))

    def rmsprop_step(self, params, grads, sq, n, lr, alpha, eps, max_grad_norm, clip_scratch):
        g = grads[:n]
        if max_grad_norm > 0:
            g = g * min(1.0, max_grad_norm / (float(g.norm()) + 1e-6))
        sq[:n] = sq[:n] * torch.tensor(alpha, dtype=torch.float32) + (torch.tensor(1.0 - alpha, dtype=torch.float32) * g) * g
        params[:n] -= torch.tensor(lr, dtype=torch.float32) * (g / (sq[:n].sqrt() + torch.tensor(eps, dtype=torch.float32)))

    def target_sync(self, target, online, n, state, force):
        if force or int(state[4]):
            target[:n] = online[:n]

    @staticmethod
    def _f(x):
        return x.sign() * x.abs().sqrt()

    def noisy_compose(self, mu, sigma, eff, N, K, r0, r1, noise_in, noise_out_w, noise_out_b):
        rows = slice(r0 * K, r1 * K)
        eps_w = torch.outer(self._f(noise_out_w[: r1 - r0]), self._f(noise_in[:K])).reshape(-1)
        eff[rows] = mu[rows] + sigma[rows] * eps_w
        bs = slice(N * K + r0, N * K + r1)
        eff[bs] = mu[bs] + sigma[bs] * self._f(noise_out_b[: r1 - r0])

    def noisy_grad_sigma(self, gmu, gsigma, N, K, r0, r1, noise_in, noise_out_w, noise_out_b):
        rows = slice(r0 * K, r1 * K)
        gsigma[rows] = gmu[rows] * torch.outer(self._f(noise_out_w[: r1 - r0]), self._f(noise_in[:K])).reshape(-1)
        bs = slice(N * K + r0, N * K + r1)
        gsigma[bs] = gmu[bs] * self._f(noise_out_b[: r1 - r0])

    # ------------------------------------------------------------------ the product's fused entry points, composed from the pieces above
    # (the HIP library has one kernel for each; agent0_amd.deepq.engine calls only these, so the emulation provides the same arithmetic
    # as compositions — test scaffolding lives here, not in the product's control flow)
    def noisy_multi(self, grad: bool, mods):
        for m in mods:
            if grad:
                self.noisy_grad_sigma(m[0], m[2], *m[3:])
            else:
                self.noisy_compose(*m)

    def dense_fwd_mul(self, X, ldx, W, b, M, group, Y, R, N, K, relu):
        tmp = torch.empty(R * N)
        self.dense_fwd(X, ldx, W, b, tmp, R, N, K, relu, self.empty(max(self.dense_fwd_scratch(R, N, K), 1)))
        Y[: R * N] = (tmp.view(R // group, group, N) * M[: (R // group) * N].view(R // group, 1, N)).reshape(-1)

    def dense_fwd_mul_keep_ok(self, R, N, K, ldx) -> bool:
        return bool(self.lib.emul_short_k_shape(R, N, K, ldx))

    def dense_fwd_mul_keep(self, X, ldx, W, b, M, group, E, Y, R, N, K, relu):
        assert self.dense_fwd_mul_keep_ok(R, N, K, ldx)
        self.dense_fwd(X, ldx, W, b, E, R, N, K, relu, None)
        Y[: R * N] = (E[: R * N].view(R // group, group, N) * M[: (R // group) * N].view(R // group, 1, N)).reshape(-1)

    def dense_wgrad_multi_scratch(self, shapes) -> int:
        return max([self.dense_wgrad_scratch(R, N, K) for (R, N, K) in shapes] + [0])

    def dense_wgrad_multi(self, layers, slabs):
        for (dY, X, ldx, grad, R, N, K) in layers:
            self.dense_wgrad(dY, X, ldx, grad, R, N, K, slabs)

    def nan_flag_export(self, state, out):
        out[0] = 1.0 if int(state[0]) != 0 else 0.0

    def adam_step_sync(self, params, grads, m, v, n, state, scalars, lr, b1, b2, eps, target_freq, target, n_total, extra_nan_flag=None):
        if extra_nan_flag is not None and float(extra_nan_flag[0]) != 0.0:
            state[0] = 1
        self.adam_step(params, grads, m, v, n, state, scalars, lr, b1, b2, eps, target_freq)
        self.target_sync(target, params, n_total, state, False)

    def adam_step_sync_wt(self, params, grads, m, v, n, state, scalars, lr, b1, b2, eps, target_freq, target, n_total, extra_nan_flag, w, C_, wt, wt_target, loss=None, loss_n=0,
                          loss_ring=None):
        if loss is not None:
            c = int(state[6])
            loss_ring[c % loss_ring.numel()] = loss[:loss_n].mean()
            state[6] = c + 1
        self.adam_step_sync(params, grads, m, v, n, state, scalars, lr, b1, b2, eps, target_freq, target, n_total, extra_nan_flag)
        self.conv_wt_refresh_sync(w, C_, wt, wt_target, state)

    def dense_fwd_partial_slabs(self, R, N, K) -> int:
        return 1

    def dense_fwd_partial(self, X, ldx, W, R, N, K, slabs):
        self.dense_fwd(X, ldx, W, torch.zeros(N), slabs, R, N, K, False, self.empty(max(self.dense_fwd_scratch(R, N, K), 1)))
        return 1

    def reduce_bias_act_multi(self, layers, N, relu=True):
        for slabs, nslab, bias, out, rows in layers:
            v = slabs[: nslab * rows * N].view(nslab, rows, N).sum(0) + bias[:N]
            out[: rows * N] = (torch.relu(v) if relu else v).reshape(-1)

    def c51_head_loss_slabs(self, s_on, nslab_on, rows_on, s_tg, nslab_tg, sel_off, bias_on, bias_tg, ld, A, T, dueling, act, rew, done, wgt, atoms, gamma_n, vmin, vmax, B,
                            loss, draw, state, q_on=None, q_tg=None, m_out=None, a_star=None):
        """The composition the HIP kernel replaces: slab sums + bias, dueling, greedy next action, C51 loss, dueling backward."""
        raw_on = s_on[: nslab_on * rows_on * ld].view(nslab_on, rows_on, ld).sum(0) + bias_on[:ld]
        raw_tg = s_tg[: nslab_tg * B * ld].view(nslab_tg, B, ld).sum(0) + bias_tg[:ld]
        q_o = q_on if q_on is not None else torch.empty(B * A * T)
        q_t = q_tg if q_tg is not None else torch.empty(B * A * T)
        self.dueling_fwd(raw_on[:B].reshape(-1).contiguous(), ld, q_o, B, A, T, dueling)
        self.dueling_fwd(raw_tg.reshape(-1).contiguous(), ld, q_t, B, A, T, dueling)
        a_s = a_star if a_star is not None else torch.zeros(B, dtype=torch.int32)
        if sel_off >= 0:
            q_s = torch.empty(B * A * T)
            self.dueling_fwd(raw_on[sel_off: sel_off + B].reshape(-1).contiguous(), ld, q_s, B, A, T, dueling)
            self.select_action(q_s, A * T, T, 1, B, A, T, 2, atoms, a_s, None, None)
        else:
            self.select_action(q_t, A * T, T, 1, B, A, T, 2, atoms, a_s, None, None)
        dq = torch.zeros(B * A * T)
        self.loss_c51(q_o, q_t, A, T, act, a_s, rew, done, wgt, atoms, gamma_n, vmin, vmax, B, loss, dq, m_out, state)
        self.dueling_bwd(dq, draw, ld, B, A, T, dueling)

    def qr_head_loss_slabs(self, s_on, nslab_on, rows_on, s_tg, nslab_tg, sel_off, bias_on, bias_tg, ld, A, T, dueling, act, rew, done, wgt, taus, gamma_n, B, loss, draw, state,
                           q_on=None, q_tg=None, a_star=None):
        """The composition the HIP kernel replaces: slab sums + bias, dueling, greedy next action (mean over quantiles), target quantiles, quantile Huber, dueling backward."""
        raw_on = s_on[: nslab_on * rows_on * ld].view(nslab_on, rows_on, ld).sum(0) + bias_on[:ld]
        raw_tg = s_tg[: nslab_tg * B * ld].view(nslab_tg, B, ld).sum(0) + bias_tg[:ld]
        q_o = q_on if q_on is not None else torch.empty(B * A * T)
        q_t = q_tg if q_tg is not None else torch.empty(B * A * T)
        self.dueling_fwd(raw_on[:B].reshape(-1).contiguous(), ld, q_o, B, A, T, dueling)
        self.dueling_fwd(raw_tg.reshape(-1).contiguous(), ld, q_t, B, A, T, dueling)
        a_s = a_star if a_star is not None else torch.zeros(B, dtype=torch.int32)
        if sel_off >= 0:
            q_s = torch.empty(B * A * T)
            self.dueling_fwd(raw_on[sel_off: sel_off + B].reshape(-1).contiguous(), ld, q_s, B, A, T, dueling)
            self.select_action(q_s, A * T, T, 1, B, A, T, 1, None, a_s, None, None)
        else:
            self.select_action(q_t, A * T, T, 1, B, A, T, 1, None, a_s, None, None)
        y = torch.empty(B * T)
        self.quantile_target(q_t, A * T, 1, T, a_s, rew, done, gamma_n, B, T, y)
        dq = torch.zeros(B * A * T)
        self.loss_quantile_huber(q_o, A * T, 1, T, y, taus, 0, act, wgt, B, T, T, loss, dq, state)
        self.dueling_bwd(dq, draw, ld, B, A, T, dueling)

    def mdqn_head_loss_slabs(self, s_on, s_tg, s_cur, nslab, b1_on, b1_tg, h_on, W_on, b_on, W_tg, b_tg, A, dueling, ld, act, rew, done, wgt, gamma_n, tau, lo, B, loss,
                             q_on, q_tg, q_cur, draw, state, dh=None):
        def fc1(slabs, bias):
            return torch.relu(slabs[: nslab * B * 512].view(nslab, B, 512).sum(0) + bias[:512]).reshape(-1).contiguous()

        def head(h, W, b, q):
            raw = torch.empty(B * ld)
            self.dense_fwd(h, 512, W, b, raw, B, ld, 512, False, self.empty(max(self.dense_fwd_scratch(B, ld, 512), 1)))
            self.dueling_fwd(raw, ld, q, B, A, 1, dueling)

        h_on[: B * 512] = fc1(s_on, b1_on)
        q_t = q_tg if q_tg is not None else torch.empty(B * A)
        q_c = q_cur if q_cur is not None else torch.empty(B * A)
        head(h_on, W_on, b_on, q_on)
        head(fc1(s_tg, b1_tg), W_tg, b_tg, q_t)
        head(fc1(s_cur, b1_tg), W_tg, b_tg, q_c)
        dq = torch.zeros(B * A)
        self.loss_mdqn(q_on, q_t, q_c, A, act, rew, done, wgt, gamma_n, tau, lo, B, loss, dq, state)
        self.dueling_bwd(dq, draw, ld, B, A, 1, dueling)
        if dh is not None:
            self.dense_dgrad(draw, W_on, h_on, dh, B, ld, 512)

    def dqn_head_loss_slabs(self, s_on, s_tg, s_sel, nslab, b1_on, b1_tg, h_on, W_on, b_on, W_tg, b_tg, A, dueling, ld, act, rew, done, wgt, gamma_n, B, loss, q_on, q_tg,
                            draw, state, dh=None):
        def fc1(slabs, bias):
            return torch.relu(slabs[: nslab * B * 512].view(nslab, B, 512).sum(0) + bias[:512]).reshape(-1).contiguous()

        def head(h, W, b, q):
            raw = torch.empty(B * ld)
            self.dense_fwd(h, 512, W, b, raw, B, ld, 512, False, self.empty(max(self.dense_fwd_scratch(B, ld, 512), 1)))
            self.dueling_fwd(raw, ld, q, B, A, 1, dueling)

        h_on[: B * 512] = fc1(s_on, b1_on)
        q_t = q_tg if q_tg is not None else torch.empty(B * A)
        head(h_on, W_on, b_on, q_on)
        head(fc1(s_tg, b1_tg), W_tg, b_tg, q_t)
        a_star = torch.zeros(B, dtype=torch.int32)
        if s_sel is not None:
            q_s = torch.empty(B * A)
            head(fc1(s_sel, b1_on), W_on, b_on, q_s)
            self.select_action(q_s, A, 1, 1, B, A, 1, 0, None, a_star, None, None)
        else:
            self.select_action(q_t, A, 1, 1, B, A, 1, 0, None, a_star, None, None)
        dq = torch.zeros(B * A)
        self.loss_dqn(q_on, q_t, A, act, a_star, rew, done, wgt, gamma_n, B, loss, dq, state)
        self.dueling_bwd(dq, draw, ld, B, A, 1, dueling)
        if dh is not None:
            self.dense_dgrad(draw, W_on, h_on, dh, B, ld, 512)
