"""Tensor-level wrappers over the C-ABI (one method per exported kernel group).

Every method validates dtype / device / contiguity / size on the host before a
pointer reaches a kernel, and launches on torch's current HIP stream.  The
product code (agent0_amd.deepq.*) talks to the GPU only through this class.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _abi
from ._abi import A0Error, EncoderPass, EncoderWeights, FramesArg, LearnerDesc, NetDesc, check


def _stream() -> int:
    """Raw handle of torch's current stream on the current device.  ``torch.cuda.current_stream().cuda_stream`` returns the same handle but spends ~8 us per call
    in torch's device-index bookkeeping on this image (a device-count query every time: cProfile of the host-env step path, profiles/r04_experiments.md) — per
    LAUNCH of every eager path."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _req(t: Optional[torch.Tensor], dtype, min_numel: int, name: str, optional: bool = False):
    if t is None:
        if optional:
            return None
        raise A0Error(f"{name}: tensor required")
    if not t.is_cuda:
        raise A0Error(f"{name}: expected a HIP device tensor (got {t.device}); agent0_amd has no CPU path")
    if t.dtype != dtype:
        raise A0Error(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise A0Error(f"{name}: must be contiguous")
    if t.numel() < min_numel:
        raise A0Error(f"{name}: needs >= {min_numel} elements, has {t.numel()}")
    return t.data_ptr()


class Net:
    """a0_net handle: geometry + gather tables for one observation shape."""

    def __init__(self, lib, C_, H, W):
        self.lib = lib
        self.h = C.c_void_p()
        desc = NetDesc(C_, H, W)
        check(lib.a0_net_create(C.addressof(desc), C.addressof(self.h)), "a0_net_create")
        geo = (C.c_int * 8)()
        check(lib.a0_net_geometry(self.h, C.addressof(geo)), "a0_net_geometry")
        self.C, self.H, self.W = C_, H, W
        self.H1, self.W1, self.H2, self.W2, self.H3, self.W3, self.feat, self.K1 = list(geo)
        self.K2, self.K3 = 512, 576

    def __del__(self):
        try:
            if self.h:
                self.lib.a0_net_destroy(self.h)
                self.h = None
        except Exception:
            pass


class NativeLearner:
    """``a0_learner``: a whole learner (dqn; c51, also with NoisyLinear layers; dueling; double-Q; n-step) behind one handle whose HBM the library owns; ``update`` is BaseLearner.train as ONE
    C call (include/agent0_hip.h).  What a non-Python host binds; here it exists for the parity test against the per-kernel composition of deepq/engine.py."""

    ALGOS = {"dqn": 0, "c51": 1, "iqn": 2, "fqf": 3, "qr": 4, "mdqn": 5}

    def __init__(self, lib, A, dueling, double_q, B, n_step=1, discount=0.99, lr=5e-4, adam_eps=0.0, target_update_freq=500, algo="dqn", num_atoms=51, vmin=-10.0,
                 vmax=10.0, noisy=False, seed=0, K=32, N=64, N_dash=64, F=32, mdqn_tau=0.03, mdqn_lo=-1.0, max_grad_norm=-1.0):
        self.lib, self.B, self.T = lib, B, int(num_atoms)
        desc = LearnerDesc(int(A), int(bool(dueling)), int(bool(double_q)), int(B), int(n_step), float(discount), float(lr), float(adam_eps), int(target_update_freq),
                           self.ALGOS[algo], int(num_atoms), float(vmin), float(vmax), int(bool(noisy)), int(seed) & 0xFFFFFFFFFFFFFFFF, int(K), int(N), int(N_dash), int(F), float(mdqn_tau), float(mdqn_lo),
                           float(max_grad_norm))
        h = C.c_void_p()
        check(lib.a0_learner_create(C.addressof(desc), C.addressof(h)), "a0_learner_create")
        self.h = h
        self.n = int(lib.a0_learner_param_floats(h))

    def frac_loss(self, out):
        """fqf: the per-sample fraction losses of the last update, copied into ``out`` [B]."""
        check(self.lib.a0_learner_get_frac_loss(self.h, _req(out, torch.float32, self.B, "out"), _stream()), "a0_learner_get_frac_loss")
        return out

    def set_support(self, atoms):
        a = (C.c_float * self.T)(*[float(x) for x in atoms])
        check(self.lib.a0_learner_set_support(self.h, a), "a0_learner_set_support")

    def set_params(self, online, target=None):
        check(self.lib.a0_learner_set_params(self.h, _req(online, torch.float32, self.n, "online"), _req(target, torch.float32, self.n, "target", optional=True), _stream()),
              "a0_learner_set_params")

    def get(self):
        dev = torch.device("cuda", torch.cuda.current_device())
        on, tg, m, v = (torch.empty(self.n, device=dev) for _ in range(4))
        st = torch.zeros(8, dtype=torch.int32, device=dev)
        check(self.lib.a0_learner_get(self.h, on.data_ptr(), tg.data_ptr(), m.data_ptr(), v.data_ptr(), st.data_ptr(), _stream()), "a0_learner_get")
        return on, tg, m, v, st

    def update(self, frames, slot, row_bytes, act, rew, done, wgt, loss_out=None):
        B = self.B
        check(self.lib.a0_learner_update(self.h, _req(frames, torch.uint8, row_bytes, "frames"), _req(slot, torch.int32, B, "slot", optional=True), int(row_bytes),
                                         _req(act, torch.int32, B, "act"), _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"), _req(wgt, torch.float32, B, "wgt"),
                                         _req(loss_out, torch.float32, B, "loss_out", optional=True), _stream()), "a0_learner_update")

    def close(self):
        if self.h is not None:
            self.lib.a0_learner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 — interpreter shutdown
            pass


class _TraceRange:
    def __init__(self, lib, name: bytes):
        self.lib, self.name = lib, name

    def __enter__(self):
        self.lib.a0_trace_push(self.name)

    def __exit__(self, *exc):
        self.lib.a0_trace_pop()
        return False


class _NoRange:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_RANGE = _NoRange()


class HipOps:
    name = "hip"

    def __init__(self):
        if not torch.cuda.is_available():
            raise A0Error("agent0_amd needs an AMD GPU visible to PyTorch-ROCm (torch.cuda.is_available() is False)")
        self.lib = _abi.load()
        self.device = torch.device("cuda", torch.cuda.current_device())

    # ------------------------------------------------------------------ allocation helpers
    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def net(self, C_, H, W) -> Net:
        return Net(self.lib, C_, H, W)

    def native_learner(self, **kw) -> NativeLearner:
        return NativeLearner(self.lib, **kw)

    # ------------------------------------------------------------------ encoder
    def _frames(self, net, frames, slot, sample_stride, chan_off, B):
        hw = net.H * net.W
        need_rows = B if slot is None else 1
        fp = _req(frames, torch.uint8, need_rows * sample_stride if slot is None else chan_off + net.C * hw, "frames")
        if chan_off + net.C * hw > sample_stride:
            raise A0Error("frames: chan_off + C*H*W exceeds sample_stride")
        sp = _req(slot, torch.int32, B, "slot", optional=True)
        return FramesArg(fp, sp, sample_stride, chan_off)

    @staticmethod
    def _enc_w(w):
        return EncoderWeights(*[w[k].data_ptr() for k in ("w1", "b1", "w2", "b2", "w3", "b3")])

    def encoder_fwd(self, net, w, frames, slot, sample_stride, chan_off, B, act1, act2, act3):
        fa = self._frames(net, frames, slot, sample_stride, chan_off, B)
        ew = self._enc_w(w)
        a1 = _req(act1, torch.float32, B * net.H1 * net.W1 * 32, "act1")
        a2 = _req(act2, torch.float32, B * net.H2 * net.W2 * 64, "act2")
        a3 = _req(act3, torch.float32, B * net.feat, "act3")
        check(self.lib.a0_net_encoder_fwd(net.h, C.addressof(ew), C.addressof(fa), B, a1, a2, a3, _stream()), "a0_net_encoder_fwd")

    def fused_supported(self, C_, H, W) -> bool:
        return bool(self.lib.a0_net_encoder_fused_supported(C_, H, W))

    def conv_wt_floats(self, C_) -> int:
        return int(self.lib.a0_net_conv_wt_floats(C_))

    def conv_wt_refresh(self, w, C_, wt):
        ew = self._enc_w(w)
        check(self.lib.a0_net_conv_wt_refresh(C.addressof(ew), C_, _req(wt, torch.float32, self.conv_wt_floats(C_), "wt"), _stream()), "a0_net_conv_wt_refresh")

    def conv_wt_refresh_sync(self, w, C_, wt, wt_target, state):
        ew = self._enc_w(w)
        n = self.conv_wt_floats(C_)
        check(self.lib.a0_net_conv_wt_refresh_sync(C.addressof(ew), C_, _req(wt, torch.float32, n, "wt"), _req(wt_target, torch.float32, n, "wt_target"),
                                                   _req(state, torch.int32, 8, "state"), _stream()), "a0_net_conv_wt_refresh_sync")

    def encoder_fwd_fused(self, net, wt, w, frames, slot, sample_stride, chan_off, B, act1, act2, act3):
        fa = self._frames(net, frames, slot, sample_stride, chan_off, B)
        ew = self._enc_w(w)
        check(self.lib.a0_net_encoder_fwd_fused(net.C, net.H, net.W, _req(wt, torch.float32, self.conv_wt_floats(net.C), "wt"), C.addressof(ew), C.addressof(fa), B,
                                                _req(act1, torch.float32, B * net.H1 * net.W1 * 32, "act1", optional=True),
                                                _req(act2, torch.float32, B * net.H2 * net.W2 * 64, "act2", optional=True),
                                                _req(act3, torch.float32, B * net.feat, "act3"), _stream()), "a0_net_encoder_fwd_fused")

    def encoder_fwd_fused_multi(self, net, passes):
        """passes: [(wt, w, frames, slot, sample_stride, chan_off, B, act1, act2, act3)] (at most three): the forward passes of ``encoder_fwd_fused`` in one launch."""
        keep = []
        arr = (EncoderPass * len(passes))()
        for i, (wt, w, frames, slot, sample_stride, chan_off, B, act1, act2, act3) in enumerate(passes):
            fa, ew = self._frames(net, frames, slot, sample_stride, chan_off, B), self._enc_w(w)
            keep += [fa, ew]
            arr[i] = EncoderPass(_req(wt, torch.float32, self.conv_wt_floats(net.C), "wt"), C.addressof(ew), C.addressof(fa), B,
                                 _req(act1, torch.float32, B * net.H1 * net.W1 * 32, "act1", optional=True), _req(act2, torch.float32, B * net.H2 * net.W2 * 64, "act2", optional=True),
                                 _req(act3, torch.float32, B * net.feat, "act3"))
        check(self.lib.a0_net_encoder_fwd_fused_multi(net.C, net.H, net.W, len(passes), C.addressof(arr), _stream()), "a0_net_encoder_fwd_fused_multi")

    def encoder_bwd_scratch(self, net, B) -> int:
        return int(self.lib.a0_net_encoder_bwd_scratch(net.h, B))

    def encoder_bwd(self, net, w, frames, slot, sample_stride, chan_off, B, act1, act2, d3, d2, d1, g1, g2, g3, slabs):
        fa = self._frames(net, frames, slot, sample_stride, chan_off, B)
        ew = self._enc_w(w)
        need = self.encoder_bwd_scratch(net, B)
        check(self.lib.a0_net_encoder_bwd(
            net.h, C.addressof(ew), C.addressof(fa), B,
            _req(act1, torch.float32, B * net.H1 * net.W1 * 32, "act1"), _req(act2, torch.float32, B * net.H2 * net.W2 * 64, "act2"),
            _req(d3, torch.float32, B * net.feat, "d3"), _req(d2, torch.float32, B * net.H2 * net.W2 * 64, "d2"),
            _req(d1, torch.float32, B * net.H1 * net.W1 * 32, "d1"),
            _req(g1, torch.float32, 32 * net.K1 + 32, "g1"), _req(g2, torch.float32, 64 * net.K2 + 64, "g2"), _req(g3, torch.float32, 64 * net.K3 + 64, "g3"),
            _req(slabs, torch.float32, need, "slabs", optional=(need == 0)), _stream()), "a0_net_encoder_bwd")

    def dgrad_fused_supported(self, C_, H, W) -> bool:
        return bool(self.lib.a0_net_encoder_dgrad_fused_supported(C_, H, W))

    def encoder_dgrad_fused(self, net, wt, d3, act1, act2, B, d2, d1):
        check(self.lib.a0_net_encoder_dgrad_fused(net.C, net.H, net.W, _req(wt, torch.float32, self.conv_wt_floats(net.C), "wt"),
                                                  _req(d3, torch.float32, B * net.feat, "d3"), _req(act1, torch.float32, B * net.H1 * net.W1 * 32, "act1"),
                                                  _req(act2, torch.float32, B * net.H2 * net.W2 * 64, "act2"), B,
                                                  _req(d2, torch.float32, B * net.H2 * net.W2 * 64, "d2"), _req(d1, torch.float32, B * net.H1 * net.W1 * 32, "d1"),
                                                  _stream()), "a0_net_encoder_dgrad_fused")

    def pending_reduce(self):
        """An empty a0_pending_reduce: dense_wgrad_multi(..., pend=p) appends its slab reductions, encoder_wgrad(..., pend=p) launches them with its own."""
        from ._abi import PendingReduce
        p = PendingReduce()
        p.n = 0
        return p

    def encoder_wgrad(self, net, w, frames, slot, sample_stride, chan_off, B, act1, act2, d3, d2, d1, g1, g2, g3, slabs, pend=None):
        fa = self._frames(net, frames, slot, sample_stride, chan_off, B)
        ew = self._enc_w(w)
        need = self.encoder_bwd_scratch(net, B)
        check(self.lib.a0_net_encoder_wgrad(
            net.h, C.addressof(ew), C.addressof(fa), B,
            _req(act1, torch.float32, B * net.H1 * net.W1 * 32, "act1"), _req(act2, torch.float32, B * net.H2 * net.W2 * 64, "act2"),
            _req(d3, torch.float32, B * net.feat, "d3"), _req(d2, torch.float32, B * net.H2 * net.W2 * 64, "d2"),
            _req(d1, torch.float32, B * net.H1 * net.W1 * 32, "d1"),
            _req(g1, torch.float32, 32 * net.K1 + 32, "g1"), _req(g2, torch.float32, 64 * net.K2 + 64, "g2"), _req(g3, torch.float32, 64 * net.K3 + 64, "g3"),
            _req(slabs, torch.float32, need, "slabs", optional=(need == 0)), None if pend is None else C.addressof(pend), _stream()), "a0_net_encoder_wgrad")

    # ------------------------------------------------------------------ dense
    def dense_fwd_scratch(self, R, N, K) -> int:
        return int(self.lib.a0_dense_fwd_scratch(R, N, K))

    def dense_wgrad_scratch(self, R, N, K) -> int:
        return int(self.lib.a0_dense_wgrad_scratch(R, N, K))

    def dense_fwd(self, X, ldx, W, b, Y, R, N, K, relu, scratch):
        need = self.dense_fwd_scratch(R, N, K)
        check(self.lib.a0_dense_fwd(_req(X, torch.float32, (R - 1) * ldx + K, "X"), ldx, _req(W, torch.float32, N * K, "W"), _req(b, torch.float32, N, "b"),
                                    _req(Y, torch.float32, R * N, "Y"), R, N, K, int(relu),
                                    _req(scratch, torch.float32, need, "scratch", optional=(need == 0)), _stream()), "a0_dense_fwd")

    def dense_dgrad_hadamard_ok(self, R, N, K, n) -> bool:
        return bool(self.lib.a0_dense_dgrad_hadamard_ok(R, N, K, n))

    def dense_dgrad_hadamard(self, dY, W, emb, feat, demb, d3, R, N, K, n):
        """dense_dgrad(dY, W, no mask) + hadamard_bwd in one launch: dx = dY W is consumed in the GEMM's epilogue (a0_dense_dgrad_hadamard)."""
        check(self.lib.a0_dense_dgrad_hadamard(_req(dY, torch.float32, R * N, "dY"), _req(W, torch.float32, N * K, "W"), _req(emb, torch.float32, R * K, "emb"),
                                               _req(feat, torch.float32, (R // n) * K, "feat"), _req(demb, torch.float32, R * K, "demb"), _req(d3, torch.float32, (R // n) * K, "d3"),
                                               R, N, K, n, _stream()), "a0_dense_dgrad_hadamard")

    def weight_planes_words(self, N, K) -> int:
        return int(self.lib.a0_weight_planes_words(N, K))

    def split_planes(self, W, planes, N, K):
        """The three bf16 term planes of W [N][K] for ``dense_fwd_wplanes`` (a0_split_planes)."""
        check(self.lib.a0_split_planes(_req(W, torch.float32, N * K, "W"), _req(planes, torch.int32, self.weight_planes_words(N, K), "planes"), N, K, _stream()), "a0_split_planes")

    def dense_fwd_wplanes_ok(self, R, N, K) -> bool:
        return bool(self.lib.a0_dense_fwd_wplanes_ok(R, N, K))

    def dense_fwd_wplanes(self, X, ldx, planes, b, Y, R, N, K, relu):
        check(self.lib.a0_dense_fwd_wplanes(_req(X, torch.float32, (R - 1) * ldx + K, "X"), ldx, _req(planes, torch.int32, self.weight_planes_words(N, K), "planes"),
                                            _req(b, torch.float32, N, "b"), _req(Y, torch.float32, R * N, "Y"), R, N, K, int(relu), _stream()), "a0_dense_fwd_wplanes")

    def dense_fwd_mul(self, X, ldx, W, b, M, group, Y, R, N, K, relu):
        check(self.lib.a0_dense_fwd_mul(_req(X, torch.float32, (R - 1) * ldx + K, "X"), ldx, _req(W, torch.float32, N * K, "W"), _req(b, torch.float32, N, "b"),
                                        _req(M, torch.float32, ((R - 1) // group + 1) * N, "M"), group, _req(Y, torch.float32, R * N, "Y"), R, N, K, int(relu), _stream()),
              "a0_dense_fwd_mul")

    def dense_fwd_mul_keep_ok(self, R, N, K, ldx) -> bool:
        return bool(self.lib.a0_dense_fwd_mul_keep_ok(R, N, K, ldx))

    def dense_fwd_mul_keep(self, X, ldx, W, b, M, group, E, Y, R, N, K, relu):
        check(self.lib.a0_dense_fwd_mul_keep(_req(X, torch.float32, (R - 1) * ldx + K, "X"), ldx, _req(W, torch.float32, N * K, "W"), _req(b, torch.float32, N, "b"),
                                             _req(M, torch.float32, ((R - 1) // group + 1) * N, "M"), group, _req(E, torch.float32, R * N, "E"),
                                             _req(Y, torch.float32, R * N, "Y"), R, N, K, int(relu), _stream()), "a0_dense_fwd_mul_keep")

    def dense_dgrad(self, dY, W, mask, dX, R, N, K):
        check(self.lib.a0_dense_dgrad(_req(dY, torch.float32, R * N, "dY"), _req(W, torch.float32, N * K, "W"),
                                      _req(mask, torch.float32, R * K, "mask", optional=True), _req(dX, torch.float32, R * K, "dX"), R, N, K, _stream()), "a0_dense_dgrad")

    def dense_wgrad(self, dY, X, ldx, grad, R, N, K, slabs):
        need = self.dense_wgrad_scratch(R, N, K)
        check(self.lib.a0_dense_wgrad(_req(dY, torch.float32, R * N, "dY"), _req(X, torch.float32, (R - 1) * ldx + K, "X"), ldx,
                                      _req(grad, torch.float32, N * K + N, "grad"), R, N, K,
                                      _req(slabs, torch.float32, need, "slabs", optional=(need == 0)), _stream()), "a0_dense_wgrad")

    # ------------------------------------------------------------------ heads / losses
    def dueling_fwd(self, raw, ld, q, R, A, T, dueling):
        check(self.lib.a0_dueling_fwd(_req(raw, torch.float32, R * ld, "raw"), ld, _req(q, torch.float32, R * A * T, "q"), R, A, T, int(dueling), _stream()), "a0_dueling_fwd")

    def dueling_bwd(self, dq, draw, ld, R, A, T, dueling):
        check(self.lib.a0_dueling_bwd(_req(dq, torch.float32, R * A * T, "dq"), _req(draw, torch.float32, R * ld, "draw"), ld, R, A, T, int(dueling), _stream()), "a0_dueling_bwd")

    def select_action(self, x, sb, sa, st, B, A, T, mode, aux, a_star, qsel, qmax):
        span = (B - 1) * sb + (A - 1) * sa + (T - 1) * st + 1
        check(self.lib.a0_select_action(_req(x, torch.float32, span, "x"), sb, sa, st, B, A, T, mode,
                                        _req(aux, torch.float32, T if mode == 2 else (B * (T + 1) if mode == 3 else 0), "aux", optional=mode < 2),
                                        _req(a_star, torch.int32, B, "a_star", optional=True), _req(qsel, torch.float32, B * A, "qsel", optional=True),
                                        _req(qmax, torch.float32, B, "qmax", optional=True), _stream()), "a0_select_action")

    def loss_dqn(self, q, q_next, A, act, a_star, rew, done, wgt, gamma_n, B, loss, dq, state):
        check(self.lib.a0_loss_dqn(_req(q, torch.float32, B * A, "q"), _req(q_next, torch.float32, B * A, "q_next"), A, _req(act, torch.int32, B, "act"),
                                   _req(a_star, torch.int32, B, "a_star"), _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"),
                                   _req(wgt, torch.float32, B, "wgt"), gamma_n, B, _req(loss, torch.float32, B, "loss"), _req(dq, torch.float32, B * A, "dq"),
                                   _req(state, torch.int32, 8, "state"), _stream()), "a0_loss_dqn")

    def loss_mdqn(self, q, q_next, q_cur_tgt, A, act, rew, done, wgt, gamma_n, tau, lo, B, loss, dq, state):
        check(self.lib.a0_loss_mdqn(_req(q, torch.float32, B * A, "q"), _req(q_next, torch.float32, B * A, "q_next"), _req(q_cur_tgt, torch.float32, B * A, "q_cur_tgt"), A,
                                    _req(act, torch.int32, B, "act"), _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"),
                                    _req(wgt, torch.float32, B, "wgt"), gamma_n, tau, lo, B, _req(loss, torch.float32, B, "loss"), _req(dq, torch.float32, B * A, "dq"),
                                    _req(state, torch.int32, 8, "state"), _stream()), "a0_loss_mdqn")

    def loss_c51(self, logits, tgt_logits, A, T, act, a_star, rew, done, wgt, atoms, gamma_n, vmin, vmax, B, loss, dlogits, m_out, state):
        check(self.lib.a0_loss_c51(_req(logits, torch.float32, B * A * T, "logits"), _req(tgt_logits, torch.float32, B * A * T, "tgt_logits"), A, T,
                                   _req(act, torch.int32, B, "act"), _req(a_star, torch.int32, B, "a_star"), _req(rew, torch.float32, B, "rew"),
                                   _req(done, torch.float32, B, "done"), _req(wgt, torch.float32, B, "wgt"), _req(atoms, torch.float32, T, "atoms"),
                                   gamma_n, vmin, vmax, B, _req(loss, torch.float32, B, "loss"), _req(dlogits, torch.float32, B * A * T, "dlogits"),
                                   _req(m_out, torch.float32, B * T, "m_out", optional=True), _req(state, torch.int32, 8, "state"), _stream()), "a0_loss_c51")

    def quantile_target(self, q_next, sb, sj, sa, a_star, rew, done, gamma_n, B, Nd, y):
        check(self.lib.a0_quantile_target(_req(q_next, torch.float32, (B - 1) * sb + (Nd - 1) * sj + 1, "q_next"), sb, sj, sa, _req(a_star, torch.int32, B, "a_star"),
                                          _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"), gamma_n, B, Nd,
                                          _req(y, torch.float32, B * Nd, "y"), _stream()), "a0_quantile_target")

    def loss_quantile_huber(self, q, sb, si, sa, y, taus, tb, act, wgt, B, N, Nd, loss, dq, state):
        span = (B - 1) * sb + (N - 1) * si + 1
        check(self.lib.a0_loss_quantile_huber(_req(q, torch.float32, span, "q"), sb, si, sa, _req(y, torch.float32, B * Nd, "y"),
                                              _req(taus, torch.float32, (B - 1) * tb + N, "taus"), tb, _req(act, torch.int32, B, "act"),
                                              _req(wgt, torch.float32, B, "wgt"), B, N, Nd, _req(loss, torch.float32, B, "loss"),
                                              _req(dq, torch.float32, span, "dq"), _req(state, torch.int32, 8, "state"), _stream()), "a0_loss_quantile_huber")

    # ------------------------------------------------------------------ IQN / FQF
    def cos_features(self, taus, out, R, D):
        check(self.lib.a0_cos_features(_req(taus, torch.float32, R, "taus"), _req(out, torch.float32, R * D, "out"), R, D, _stream()), "a0_cos_features")

    def tau_cos_features(self, seed, stream_id, offset, taus, out, R, D, ctrl=None, ctrl_idx=0):
        """taus = the Philox uniform draws [offset, offset + R) of the stream (== rng_uniform / rng_uniform_ctrl) and out = cos_features(taus), one launch (a0_tau_cos_features)."""
        check(self.lib.a0_tau_cos_features(seed, stream_id, offset, _req(ctrl, torch.int64, 8, "ctrl", optional=True), int(ctrl_idx), _req(taus, torch.float32, R, "taus"),
                                           _req(out, torch.float32, R * D, "out"), R, D, _stream()), "a0_tau_cos_features")

    def fqf_taus_cos(self, logits, ld, taus, tau_hat, cos_out, D, B, F):
        check(self.lib.a0_fqf_taus_cos(_req(logits, torch.float32, B * ld, "logits"), ld, _req(taus, torch.float32, B * (F + 1), "taus"), _req(tau_hat, torch.float32, B * F, "tau_hat"),
                                       _req(cos_out, torch.float32, B * F * D, "cos_out"), D, B, F, _stream()), "a0_fqf_taus_cos")

    def hadamard_fwd(self, emb, feat, x, B, n, D):
        check(self.lib.a0_hadamard_fwd(_req(emb, torch.float32, B * n * D, "emb"), _req(feat, torch.float32, B * D, "feat"), _req(x, torch.float32, B * n * D, "x"), B, n, D, _stream()), "a0_hadamard_fwd")

    def hadamard_bwd(self, dx, emb, feat, demb, d3, B, n, D):
        check(self.lib.a0_hadamard_bwd(_req(dx, torch.float32, B * n * D, "dx"), _req(emb, torch.float32, B * n * D, "emb"), _req(feat, torch.float32, B * D, "feat"),
                                       _req(demb, torch.float32, B * n * D, "demb"), _req(d3, torch.float32, B * D, "d3"), B, n, D, _stream()), "a0_hadamard_bwd")

    def fqf_taus(self, logits, ld, taus, tau_hat, B, F):
        check(self.lib.a0_fqf_taus(_req(logits, torch.float32, B * ld, "logits"), ld, _req(taus, torch.float32, B * (F + 1), "taus"),
                                   _req(tau_hat, torch.float32, B * F, "tau_hat"), B, F, _stream()), "a0_fqf_taus")

    def fqf_inner_taus(self, taus, out, B, F):
        check(self.lib.a0_fqf_inner_taus(_req(taus, torch.float32, B * (F + 1), "taus"), _req(out, torch.float32, B * (F - 1), "out"), B, F, _stream()), "a0_fqf_inner_taus")

    def fqf_fraction_loss(self, q, qh, taus, act, wgt, B, F, A, ldl, loss, dlogits, logits):
        check(self.lib.a0_fqf_fraction_loss(_req(q, torch.float32, B * (F - 1) * A, "q"), _req(qh, torch.float32, B * F * A, "qh"), _req(taus, torch.float32, B * (F + 1), "taus"),
                                            _req(act, torch.int32, B, "act"), _req(wgt, torch.float32, B, "wgt"), B, F, A, ldl, _req(loss, torch.float32, B, "loss"),
                                            _req(dlogits, torch.float32, B * ldl, "dlogits"), _req(logits, torch.float32, B * ldl, "logits"), _stream()), "a0_fqf_fraction_loss")

    # ------------------------------------------------------------------ optimizer
    def adam_step(self, params, grads, m, v, n, state, scalars, lr, b1, b2, eps, target_freq):
        check(self.lib.a0_adam_step(_req(params, torch.float32, n, "params"), _req(grads, torch.float32, n, "grads"), _req(m, torch.float32, n, "m"),
                                    _req(v, torch.float32, n, "v"), n, _req(state, torch.int32, 8, "state"), _req(scalars, torch.float32, 2, "scalars"),
                                    lr, b1, b2, eps, target_freq, _stream()), "a0_adam_step")

    def adam_step_sync(self, params, grads, m, v, n, state, scalars, lr, b1, b2, eps, target_freq, target, n_total, extra_nan_flag=None):
        check(self.lib.a0_adam_step_sync(_req(params, torch.float32, n_total, "params"), _req(grads, torch.float32, n, "grads"), _req(m, torch.float32, n, "m"),
                                         _req(v, torch.float32, n, "v"), n, _req(state, torch.int32, 8, "state"), _req(scalars, torch.float32, 2, "scalars"),
                                         lr, b1, b2, eps, target_freq, _req(target, torch.float32, n_total, "target"), n_total,
                                         _req(extra_nan_flag, torch.float32, 1, "extra_nan_flag", optional=True), _stream()), "a0_adam_step_sync")

    def adam_step_sync_wt(self, params, grads, m, v, n, state, scalars, lr, b1, b2, eps, target_freq, target, n_total, extra_nan_flag, w, C_, wt, wt_target, loss=None, loss_n=0,
                          loss_ring=None):
        """``loss`` [loss_n] + ``loss_ring``: the Adam launch also writes the batch mean of the per-sample losses to ring slot state[6] % len(ring)."""
        ew = self._enc_w(w)
        nw = self.conv_wt_floats(C_)
        check(self.lib.a0_adam_step_sync_wt(_req(params, torch.float32, n_total, "params"), _req(grads, torch.float32, n, "grads"), _req(m, torch.float32, n, "m"),
                                            _req(v, torch.float32, n, "v"), n, _req(state, torch.int32, 8, "state"), _req(scalars, torch.float32, 2, "scalars"),
                                            lr, b1, b2, eps, target_freq, _req(target, torch.float32, n_total, "target"), n_total,
                                            _req(extra_nan_flag, torch.float32, 1, "extra_nan_flag", optional=True), C.addressof(ew), C_,
                                            _req(wt, torch.float32, nw, "wt"), _req(wt_target, torch.float32, nw, "wt_target"),
                                            _req(loss, torch.float32, max(loss_n, 1), "loss", optional=True), int(loss_n), _req(loss_ring, torch.float32, 1, "loss_ring", optional=True),
                                            0 if loss_ring is None else int(loss_ring.numel()), _stream()), "a0_adam_step_sync_wt")

    def nan_flag_export(self, state, out):
        check(self.lib.a0_nan_flag_export(_req(state, torch.int32, 8, "state"), _req(out, torch.float32, 1, "out"), _stream()), "a0_nan_flag_export")

    def rmsprop_step(self, params, grads, sq, n, lr, alpha, eps, max_grad_norm, clip_scratch):
        check(self.lib.a0_rmsprop_step(_req(params, torch.float32, n, "params"), _req(grads, torch.float32, n, "grads"), _req(sq, torch.float32, n, "sq"), n,
                                       lr, alpha, eps, max_grad_norm, _req(clip_scratch, torch.float32, 1, "clip_scratch", optional=True), _stream()), "a0_rmsprop_step")

    def target_sync(self, target, online, n, state, force):
        check(self.lib.a0_target_sync(_req(target, torch.float32, n, "target"), _req(online, torch.float32, n, "online"), n,
                                      _req(state, torch.int32, 8, "state", optional=bool(force)), int(force), _stream()), "a0_target_sync")

    def noisy_compose(self, mu, sigma, eff, N, K, r0, r1, noise_in, noise_out_w, noise_out_b):
        check(self.lib.a0_noisy_compose(_req(mu, torch.float32, N * K + N, "mu"), _req(sigma, torch.float32, N * K + N, "sigma"), _req(eff, torch.float32, N * K + N, "eff"),
                                        N, K, r0, r1, _req(noise_in, torch.float32, K, "noise_in"), _req(noise_out_w, torch.float32, r1 - r0, "noise_out_w"),
                                        _req(noise_out_b, torch.float32, r1 - r0, "noise_out_b"), _stream()), "a0_noisy_compose")

    def noisy_grad_sigma(self, gmu, gsigma, N, K, r0, r1, noise_in, noise_out_w, noise_out_b):
        check(self.lib.a0_noisy_grad_sigma(_req(gmu, torch.float32, N * K + N, "gmu"), _req(gsigma, torch.float32, N * K + N, "gsigma"), N, K, r0, r1,
                                           _req(noise_in, torch.float32, K, "noise_in"), _req(noise_out_w, torch.float32, r1 - r0, "noise_out_w"),
                                           _req(noise_out_b, torch.float32, r1 - r0, "noise_out_b"), _stream()), "a0_noisy_grad_sigma")

    # ------------------------------------------------------------------ replay
    def noisy_multi(self, grad: bool, mods):
        """mods: [(mu, sigma, out, N, K, r0, r1, noise_in, noise_out_w, noise_out_b)] (at most six NoisyLinear modules), one launch."""
        n = len(mods)
        PP, II = C.c_void_p * n, C.c_int * n
        blk = lambda t, m, nm: _req(t, torch.float32, m[3] * m[4] + m[3], nm)
        check(self.lib.a0_noisy_multi(int(grad), n, PP(*[blk(m[0], m, "mu") for m in mods]), PP(*[None if grad else blk(m[1], m, "sigma") for m in mods]),
                                      PP(*[blk(m[2], m, "out") for m in mods]), II(*[m[3] for m in mods]), II(*[m[4] for m in mods]), II(*[m[5] for m in mods]),
                                      II(*[m[6] for m in mods]), PP(*[_req(m[7], torch.float32, m[4], "noise_in") for m in mods]),
                                      PP(*[_req(m[8], torch.float32, m[6] - m[5], "noise_out_w") for m in mods]),
                                      PP(*[_req(m[9], torch.float32, m[6] - m[5], "noise_out_b") for m in mods]), _stream()), "a0_noisy_multi")

    def replay_insert(self, frames, cap, obs_bytes, start_slot, n, obs, obs_next, act, rew, done, r_act, r_rew, r_done, ctrl=None):
        check(self.lib.a0_replay_insert(_req(frames, torch.uint8, cap * 2 * obs_bytes, "frames"), cap, obs_bytes, start_slot, n,
                                        _req(obs, torch.uint8, n * obs_bytes, "obs"), _req(obs_next, torch.uint8, n * obs_bytes, "obs_next"),
                                        _req(act, torch.int32, n, "act"), _req(rew, torch.float32, n, "rew"), _req(done, torch.float32, n, "done"),
                                        _req(r_act, torch.int32, cap, "r_act"), _req(r_rew, torch.float32, cap, "r_rew"), _req(r_done, torch.float32, cap, "r_done"),
                                        _req(ctrl, torch.int64, 8, "ctrl", optional=True), _stream()), "a0_replay_insert")

    def replay_lookup(self, idx, B, top, head, cap, slot, r_act, r_rew, r_done, priority, act, rew, done, prio, idx_out):
        check(self.lib.a0_replay_lookup(_req(idx, torch.int64, B, "idx"), B, top, head, cap, _req(slot, torch.int32, B, "slot"),
                                        _req(r_act, torch.int32, cap, "r_act"), _req(r_rew, torch.float32, cap, "r_rew"), _req(r_done, torch.float32, cap, "r_done"),
                                        _req(priority, torch.float32, top, "priority", optional=True), _req(act, torch.int32, B, "act"),
                                        _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"), _req(prio, torch.float32, B, "prio", optional=True),
                                        _req(idx_out, torch.int64, B, "idx_out", optional=True), _stream()), "a0_replay_lookup")

    def replay_gather(self, frames, row_bytes, slot, B, out, rows_available):
        check(self.lib.a0_replay_gather(_req(frames, torch.uint8, rows_available * row_bytes, "frames"), row_bytes, _req(slot, torch.int32, B, "slot"), B,
                                        _req(out, torch.uint8, B * row_bytes, "out"), _stream()), "a0_replay_gather")

    def replay_sample_slots(self, start, n_perm, seed, top, head, cap, r_act, r_rew, r_done, priority, B, idx_out, slot_out, act, rew, done, prio):
        check(self.lib.a0_replay_sample_slots(start, n_perm, seed & 0xFFFFFFFF, top, head, cap, _req(r_act, torch.int32, cap, "r_act"), _req(r_rew, torch.float32, cap, "r_rew"),
                                              _req(r_done, torch.float32, cap, "r_done"), _req(priority, torch.float32, top, "priority", optional=True), B,
                                              _req(idx_out, torch.int64, B, "idx_out"), _req(slot_out, torch.int32, B, "slot_out"), _req(act, torch.int32, B, "act"),
                                              _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"), _req(prio, torch.float32, B, "prio", optional=True),
                                              _stream()), "a0_replay_sample_slots")

    def replay_sample_gather(self, mode, start, n_perm, seed, tree, cap2, xi, top, head, cap, frames, row_bytes, r_act, r_rew, r_done, priority, B, out,
                             idx_out, slot_out, act, rew, done, prio):
        check(self.lib.a0_replay_sample_gather(mode, start, n_perm, seed & 0xFFFFFFFF, _req(tree, torch.float32, 2 * cap2, "tree", optional=(mode == 0)), cap2,
                                               _req(xi, torch.float32, B, "xi", optional=(mode == 0)), top, head, cap, _req(frames, torch.uint8, cap * row_bytes, "frames"),
                                               row_bytes, _req(r_act, torch.int32, cap, "r_act"), _req(r_rew, torch.float32, cap, "r_rew"),
                                               _req(r_done, torch.float32, cap, "r_done"), _req(priority, torch.float32, top, "priority", optional=True), B,
                                               _req(out, torch.uint8, B * row_bytes, "out"), _req(idx_out, torch.int64, B, "idx_out"), _req(slot_out, torch.int32, B, "slot_out"),
                                               _req(act, torch.int32, B, "act"), _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"),
                                               _req(prio, torch.float32, B, "prio", optional=True), _stream()), "a0_replay_sample_gather")

    fused_dqn_head = True

    def dqn_head_loss(self, h_on, h_tg, h_sel, W_on, b_on, W_tg, b_tg, A, dueling, ld, act, rew, done, wgt, gamma_n, B, loss, q_on, q_tg, draw, state):
        nq = A + (1 if dueling else 0)
        check(self.lib.a0_dqn_head_loss(_req(h_on, torch.float32, B * 512, "h_on"), _req(h_tg, torch.float32, B * 512, "h_tg"),
                                        _req(h_sel, torch.float32, B * 512, "h_sel", optional=True), _req(W_on, torch.float32, nq * 512, "W_on"),
                                        _req(b_on, torch.float32, nq, "b_on"), _req(W_tg, torch.float32, nq * 512, "W_tg"), _req(b_tg, torch.float32, nq, "b_tg"),
                                        A, int(dueling), ld, _req(act, torch.int32, B, "act"), _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"),
                                        _req(wgt, torch.float32, B, "wgt"), float(gamma_n), B, _req(loss, torch.float32, B, "loss"),
                                        _req(q_on, torch.float32, B * A, "q_on"), _req(q_tg, torch.float32, B * A, "q_tg", optional=True),
                                        _req(draw, torch.float32, B * ld, "draw"), _req(state, torch.int32, 4, "state"), _stream()), "a0_dqn_head_loss")

    fused_dqn_head_slabs = True

    def dense_fwd_partial_slabs(self, R, N, K) -> int:
        return int(self.lib.a0_dense_fwd_partial_slabs(R, N, K))

    def dense_fwd_partial(self, X, ldx, W, R, N, K, slabs):
        ns = self.dense_fwd_partial_slabs(R, N, K)
        check(self.lib.a0_dense_fwd_partial(_req(X, torch.float32, (R - 1) * ldx + K, "X"), ldx, _req(W, torch.float32, N * K, "W"), R, N, K,
                                            _req(slabs, torch.float32, ns * R * N, "slabs"), _stream()), "a0_dense_fwd_partial")
        return ns

    def dense_fwd_partial_multi_ok(self, n, R, N, K) -> bool:
        return bool(self.lib.a0_dense_fwd_partial_multi_ok(n, R, N, K))

    def dense_fwd_partial_multi(self, Xs, ldx, Ws, R, N, K, slabs, strides=None):
        """n = 2 or 3 passes of one layer shape in one launch (a0_dense_fwd_partial_multi); returns the slab count every pass leaves.  ``strides``: floats between
        a pass's consecutive slabs (default R * N) — two passes may interleave their rows in one [splits][2R][N] buffer."""
        n = len(Xs)
        ns = int(self.lib.a0_dense_fwd_partial_multi_slabs(n, R, N, K))
        PP = C.c_void_p * n
        st = None if strides is None else (C.c_longlong * n)(*[int(x) for x in strides])
        need = [((ns - 1) * (R * N if strides is None else int(strides[i])) + R * N) for i in range(n)]
        check(self.lib.a0_dense_fwd_partial_multi(n, PP(*[_req(x, torch.float32, (R - 1) * ldx + K, "X") for x in Xs]), ldx, PP(*[_req(w, torch.float32, N * K, "W") for w in Ws]),
                                                  R, N, K, PP(*[_req(sl, torch.float32, need[i], "slabs") for i, sl in enumerate(slabs)]), st, _stream()), "a0_dense_fwd_partial_multi")
        return ns

    def dqn_head_loss_slabs(self, s_on, s_tg, s_sel, nslab, b1_on, b1_tg, h_on, W_on, b_on, W_tg, b_tg, A, dueling, ld, act, rew, done, wgt, gamma_n, B, loss, q_on, q_tg,
                            draw, state, dh=None):
        nq = A + (1 if dueling else 0)
        n = nslab * B * 512
        check(self.lib.a0_dqn_head_loss_slabs(_req(s_on, torch.float32, n, "slabs_on"), _req(s_tg, torch.float32, n, "slabs_tg"),
                                              _req(s_sel, torch.float32, n, "slabs_sel", optional=True), B * 512, nslab, _req(b1_on, torch.float32, 512, "b1_on"),
                                              _req(b1_tg, torch.float32, 512, "b1_tg"), _req(h_on, torch.float32, B * 512, "h_on"),
                                              _req(W_on, torch.float32, nq * 512, "W_on"), _req(b_on, torch.float32, nq, "b_on"), _req(W_tg, torch.float32, nq * 512, "W_tg"),
                                              _req(b_tg, torch.float32, nq, "b_tg"), A, int(dueling), ld, _req(act, torch.int32, B, "act"), _req(rew, torch.float32, B, "rew"),
                                              _req(done, torch.float32, B, "done"), _req(wgt, torch.float32, B, "wgt"), float(gamma_n), B, _req(loss, torch.float32, B, "loss"),
                                              _req(q_on, torch.float32, B * A, "q_on"), _req(q_tg, torch.float32, B * A, "q_tg", optional=True),
                                              _req(draw, torch.float32, B * ld, "draw"), _req(state, torch.int32, 4, "state"),
                                              _req(dh, torch.float32, B * 512, "dh", optional=True), _stream()), "a0_dqn_head_loss_slabs")

    def mdqn_head_loss_slabs(self, s_on, s_tg, s_cur, nslab, b1_on, b1_tg, h_on, W_on, b_on, W_tg, b_tg, A, dueling, ld, act, rew, done, wgt, gamma_n, tau, lo, B, loss,
                             q_on, q_tg, q_cur, draw, state, dh=None):
        """MDQNLearner.train_step from the fc1 GEMMs' slabs on (a0_mdqn_head_loss_slabs): s_cur = the TARGET network's fc1 slabs on the current observation."""
        nq = A + (1 if dueling else 0)
        n = nslab * B * 512
        check(self.lib.a0_mdqn_head_loss_slabs(_req(s_on, torch.float32, n, "slabs_on"), _req(s_tg, torch.float32, n, "slabs_tg"), _req(s_cur, torch.float32, n, "slabs_cur"),
                                               B * 512, nslab, _req(b1_on, torch.float32, 512, "b1_on"), _req(b1_tg, torch.float32, 512, "b1_tg"),
                                               _req(h_on, torch.float32, B * 512, "h_on"), _req(W_on, torch.float32, nq * 512, "W_on"), _req(b_on, torch.float32, nq, "b_on"),
                                               _req(W_tg, torch.float32, nq * 512, "W_tg"), _req(b_tg, torch.float32, nq, "b_tg"), A, int(dueling), ld,
                                               _req(act, torch.int32, B, "act"), _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"),
                                               _req(wgt, torch.float32, B, "wgt"), float(gamma_n), float(tau), float(lo), B, _req(loss, torch.float32, B, "loss"),
                                               _req(q_on, torch.float32, B * A, "q_on"), _req(q_tg, torch.float32, B * A, "q_tg", optional=True),
                                               _req(q_cur, torch.float32, B * A, "q_cur", optional=True), _req(draw, torch.float32, B * ld, "draw"),
                                               _req(state, torch.int32, 4, "state"), _req(dh, torch.float32, B * 512, "dh", optional=True), _stream()), "a0_mdqn_head_loss_slabs")

    def qr_head_loss_slabs(self, s_on, nslab_on, rows_on, s_tg, nslab_tg, sel_off, bias_on, bias_tg, ld, A, T, dueling, act, rew, done, wgt, taus, gamma_n, B, loss, draw, state,
                           q_on=None, q_tg=None, a_star=None):
        """QRLearner.train_step from the head GEMMs' slabs on (a0_qr_head_loss_slabs): the slab layout of ``c51_head_loss_slabs``; taus [T] = the quantile midpoints."""
        check(self.lib.a0_qr_head_loss_slabs(_req(s_on, torch.float32, nslab_on * rows_on * ld, "slabs_on"), rows_on * ld, nslab_on, rows_on,
                                             _req(s_tg, torch.float32, nslab_tg * B * ld, "slabs_tg"), B * ld, nslab_tg, int(sel_off),
                                             _req(bias_on, torch.float32, ld, "bias_on"), _req(bias_tg, torch.float32, ld, "bias_tg"), ld, A, T, int(dueling),
                                             _req(act, torch.int32, B, "act"), _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"),
                                             _req(wgt, torch.float32, B, "wgt"), _req(taus, torch.float32, T, "taus"), float(gamma_n), B,
                                             _req(loss, torch.float32, B, "loss"), _req(draw, torch.float32, B * ld, "draw"),
                                             _req(q_on, torch.float32, B * A * T, "q_on", optional=True), _req(q_tg, torch.float32, B * A * T, "q_tg", optional=True),
                                             _req(a_star, torch.int32, B, "a_star", optional=True), _req(state, torch.int32, 4, "state"), _stream()), "a0_qr_head_loss_slabs")

    def reduce_bias_act_multi(self, layers, N, relu=True):
        """layers: [(slabs, nslab, bias, out, rows)] (at most four, all of width N): out = act(sum of the layer's split-K slabs + bias), one launch."""
        n = len(layers)
        PP, II, LL = C.c_void_p * n, C.c_int * n, C.c_longlong * n
        check(self.lib.a0_reduce_bias_act_multi(n, PP(*[_req(l[0], torch.float32, l[1] * l[4] * N, "slabs") for l in layers]), LL(*[l[4] * N for l in layers]),
                                                II(*[l[1] for l in layers]), PP(*[_req(l[2], torch.float32, N, "bias") for l in layers]),
                                                PP(*[_req(l[3], torch.float32, l[4] * N, "out") for l in layers]), II(*[l[4] for l in layers]), N, int(relu), _stream()),
              "a0_reduce_bias_act_multi")

    def c51_head_loss_slabs(self, s_on, nslab_on, rows_on, s_tg, nslab_tg, sel_off, bias_on, bias_tg, ld, A, T, dueling, act, rew, done, wgt, atoms, gamma_n, vmin, vmax, B,
                            loss, draw, state, q_on=None, q_tg=None, m_out=None, a_star=None):
        """C51Learner.train_step from the head GEMMs' slabs on (a0_c51_head_loss_slabs): online slabs [nslab_on][rows_on][ld] (rows [0, B) = s, [sel_off, sel_off + B) = s'
        under double-Q, sel_off < 0 otherwise), target slabs [nslab_tg][B][ld]."""
        nb = ld
        check(self.lib.a0_c51_head_loss_slabs(_req(s_on, torch.float32, nslab_on * rows_on * ld, "slabs_on"), rows_on * ld, nslab_on, rows_on,
                                              _req(s_tg, torch.float32, nslab_tg * B * ld, "slabs_tg"), B * ld, nslab_tg, int(sel_off),
                                              _req(bias_on, torch.float32, nb, "bias_on"), _req(bias_tg, torch.float32, nb, "bias_tg"), ld, A, T, int(dueling),
                                              _req(act, torch.int32, B, "act"), _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"),
                                              _req(wgt, torch.float32, B, "wgt"), _req(atoms, torch.float32, T, "atoms"), float(gamma_n), float(vmin), float(vmax), B,
                                              _req(loss, torch.float32, B, "loss"), _req(draw, torch.float32, B * ld, "draw"),
                                              _req(q_on, torch.float32, B * A * T, "q_on", optional=True), _req(q_tg, torch.float32, B * A * T, "q_tg", optional=True),
                                              _req(m_out, torch.float32, B * T, "m_out", optional=True), _req(a_star, torch.int32, B, "a_star", optional=True),
                                              _req(state, torch.int32, 4, "state"), _stream()), "a0_c51_head_loss_slabs")

    def actor_dist_tail(self, slabs, nslab, bias, ld, A, T, dueling, mode, atoms, E, seed, stream_a, stream_u, off_a, off_u, eps, action, qmax, ctrl=None, eps_ptr=None):
        check(self.lib.a0_actor_dist_tail(_req(slabs, torch.float32, nslab * E * ld, "slabs"), E * ld, nslab, _req(bias, torch.float32, ld, "bias"), ld, A, T, int(dueling),
                                          mode, _req(atoms, torch.float32, T, "atoms", optional=(mode != 2)), E, seed, stream_a, stream_u, off_a, off_u, float(eps),
                                          _req(ctrl, torch.int64, 8, "ctrl", optional=True), _req(eps_ptr, torch.float32, 1, "eps_ptr", optional=True),
                                          _req(action, torch.int32, E, "action"), _req(qmax, torch.float32, E, "qmax"), _stream()), "a0_actor_dist_tail")

    def actor_dist_tail_env_step(self, slabs, nslab, bias, ld, A, T, dueling, mode, atoms, E, seed, stream_a, stream_u, off_a, off_u, eps, action, qmax, ctrl, eps_ptr,
                                 env_seed, rank, g, obs_in, obs_out, ep_ret, final_mask, final_ret, n, steps, gamma, ring_act, ring_rew, ring_done, obs0, frames, cap,
                                 start_slot, r_act, r_rew, r_done, task=0):
        nb = E * 4 * 84 * 84
        check(self.lib.a0_actor_dist_tail_env_step(
            _req(slabs, torch.float32, nslab * E * ld, "slabs"), E * ld, nslab, _req(bias, torch.float32, ld, "bias"), ld, A, T, int(dueling),
            mode, _req(atoms, torch.float32, T, "atoms", optional=(mode != 2)), E, seed, stream_a, stream_u, off_a, off_u, float(eps),
            _req(ctrl, torch.int64, 8, "ctrl", optional=True), _req(eps_ptr, torch.float32, 1, "eps_ptr", optional=True),
            _req(action, torch.int32, E, "action"), _req(qmax, torch.float32, E, "qmax"),
            env_seed, rank, g, _req(obs_in, torch.uint8, nb, "obs_in"), _req(obs_out, torch.uint8, nb, "obs_out"), _req(ep_ret, torch.float32, E, "ep_ret"),
            _req(final_mask, torch.float32, E, "final_mask"), _req(final_ret, torch.float32, E, "final_ret"), n, steps, float(gamma),
            _req(ring_act, torch.int32, n * E, "ring_act"), _req(ring_rew, torch.float32, n * E, "ring_rew"), _req(ring_done, torch.float32, n * E, "ring_done"),
            _req(obs0, torch.uint8, nb, "obs0"), _req(frames, torch.uint8, cap * 8 * 84 * 84, "frames"), cap, start_slot, _req(r_act, torch.int32, cap, "r_act"),
            _req(r_rew, torch.float32, cap, "r_rew"), _req(r_done, torch.float32, cap, "r_done"), int(task), _stream()), "a0_actor_dist_tail_env_step")

    def actor_dist_tail_env_step_enc(self, slabs, nslab, bias, ld, A, T, dueling, mode, atoms, E, seed, stream_a, stream_u, off_a, off_u, eps, action, qmax, ctrl, eps_ptr,
                                     env_seed, rank, g, obs_in, obs_out, ep_ret, final_mask, final_ret, n, steps, gamma, ring_act, ring_rew, ring_done, obs0, frames, cap,
                                     start_slot, r_act, r_rew, r_done, task=0, wt=None, enc_w=None, act3_next=None):
        """``actor_dist_tail_env_step`` whose kernel goes on to encode the env's new observation into ``act3_next`` [E][3136] (a0_actor_dist_tail_env_step_enc)."""
        nb = E * 4 * 84 * 84
        ew = self._enc_w(enc_w)
        check(self.lib.a0_actor_dist_tail_env_step_enc(
            _req(slabs, torch.float32, nslab * E * ld, "slabs"), E * ld, nslab, _req(bias, torch.float32, ld, "bias"), ld, A, T, int(dueling),
            mode, _req(atoms, torch.float32, T, "atoms", optional=(mode != 2)), E, seed, stream_a, stream_u, off_a, off_u, float(eps),
            _req(ctrl, torch.int64, 8, "ctrl", optional=True), _req(eps_ptr, torch.float32, 1, "eps_ptr", optional=True),
            _req(action, torch.int32, E, "action"), _req(qmax, torch.float32, E, "qmax"),
            env_seed, rank, g, _req(obs_in, torch.uint8, nb, "obs_in"), _req(obs_out, torch.uint8, nb, "obs_out"), _req(ep_ret, torch.float32, E, "ep_ret"),
            _req(final_mask, torch.float32, E, "final_mask"), _req(final_ret, torch.float32, E, "final_ret"), n, steps, float(gamma),
            _req(ring_act, torch.int32, n * E, "ring_act"), _req(ring_rew, torch.float32, n * E, "ring_rew"), _req(ring_done, torch.float32, n * E, "ring_done"),
            _req(obs0, torch.uint8, nb, "obs0"), _req(frames, torch.uint8, cap * 8 * 84 * 84, "frames"), cap, start_slot, _req(r_act, torch.int32, cap, "r_act"),
            _req(r_rew, torch.float32, cap, "r_rew"), _req(r_done, torch.float32, cap, "r_done"), int(task),
            _req(wt, torch.float32, self.conv_wt_floats(4), "wt"), C.addressof(ew), _req(act3_next, torch.float32, E * 3136, "act3_next"), _stream()), "a0_actor_dist_tail_env_step_enc")

    def actor_quantile_tail_env_step(self, slabs, nslab, bias, ld, A, T, dueling, mode, taus, E, seed, stream_a, stream_u, off_a, off_u, eps, action, qmax, ctrl, eps_ptr,
                                     env_seed, rank, g, obs_in, obs_out, ep_ret, final_mask, final_ret, n, steps, gamma, ring_act, ring_rew, ring_done, obs0, frames, cap,
                                     start_slot, r_act, r_rew, r_done, task=0):
        nb = E * 4 * 84 * 84
        check(self.lib.a0_actor_quantile_tail_env_step(
            _req(slabs, torch.float32, nslab * E * T * ld, "slabs"), E * T * ld, nslab, _req(bias, torch.float32, A + (1 if dueling else 0), "bias"), ld, A, T, int(dueling),
            mode, _req(taus, torch.float32, E * (T + 1), "taus", optional=(mode != 3)), E, seed, stream_a, stream_u, off_a, off_u, float(eps),
            _req(ctrl, torch.int64, 8, "ctrl", optional=True), _req(eps_ptr, torch.float32, 1, "eps_ptr", optional=True),
            _req(action, torch.int32, E, "action"), _req(qmax, torch.float32, E, "qmax"),
            env_seed, rank, g, _req(obs_in, torch.uint8, nb, "obs_in"), _req(obs_out, torch.uint8, nb, "obs_out"), _req(ep_ret, torch.float32, E, "ep_ret"),
            _req(final_mask, torch.float32, E, "final_mask"), _req(final_ret, torch.float32, E, "final_ret"), n, steps, float(gamma),
            _req(ring_act, torch.int32, n * E, "ring_act"), _req(ring_rew, torch.float32, n * E, "ring_rew"), _req(ring_done, torch.float32, n * E, "ring_done"),
            _req(obs0, torch.uint8, nb, "obs0"), _req(frames, torch.uint8, cap * 8 * 84 * 84, "frames"), cap, start_slot, _req(r_act, torch.int32, cap, "r_act"),
            _req(r_rew, torch.float32, cap, "r_rew"), _req(r_done, torch.float32, cap, "r_done"), int(task), _stream()), "a0_actor_quantile_tail_env_step")

    def actor_quantile_tail_env_step_enc(self, slabs, nslab, bias, ld, A, T, dueling, mode, taus, E, seed, stream_a, stream_u, off_a, off_u, eps, action, qmax, ctrl, eps_ptr,
                                         env_seed, rank, g, obs_in, obs_out, ep_ret, final_mask, final_ret, n, steps, gamma, ring_act, ring_rew, ring_done, obs0, frames, cap,
                                         start_slot, r_act, r_rew, r_done, task=0, wt=None, enc_w=None, act3_next=None):
        """``actor_quantile_tail_env_step`` whose kernel goes on to encode the env's new observation into ``act3_next`` [E][3136] (a0_actor_quantile_tail_env_step_enc)."""
        nb = E * 4 * 84 * 84
        ew = self._enc_w(enc_w)
        check(self.lib.a0_actor_quantile_tail_env_step_enc(
            _req(slabs, torch.float32, nslab * E * T * ld, "slabs"), E * T * ld, nslab, _req(bias, torch.float32, A + (1 if dueling else 0), "bias"), ld, A, T, int(dueling),
            mode, _req(taus, torch.float32, E * (T + 1), "taus", optional=(mode != 3)), E, seed, stream_a, stream_u, off_a, off_u, float(eps),
            _req(ctrl, torch.int64, 8, "ctrl", optional=True), _req(eps_ptr, torch.float32, 1, "eps_ptr", optional=True),
            _req(action, torch.int32, E, "action"), _req(qmax, torch.float32, E, "qmax"),
            env_seed, rank, g, _req(obs_in, torch.uint8, nb, "obs_in"), _req(obs_out, torch.uint8, nb, "obs_out"), _req(ep_ret, torch.float32, E, "ep_ret"),
            _req(final_mask, torch.float32, E, "final_mask"), _req(final_ret, torch.float32, E, "final_ret"), n, steps, float(gamma),
            _req(ring_act, torch.int32, n * E, "ring_act"), _req(ring_rew, torch.float32, n * E, "ring_rew"), _req(ring_done, torch.float32, n * E, "ring_done"),
            _req(obs0, torch.uint8, nb, "obs0"), _req(frames, torch.uint8, cap * 8 * 84 * 84, "frames"), cap, start_slot, _req(r_act, torch.int32, cap, "r_act"),
            _req(r_rew, torch.float32, cap, "r_rew"), _req(r_done, torch.float32, cap, "r_done"), int(task),
            _req(wt, torch.float32, self.conv_wt_floats(4), "wt"), C.addressof(ew), _req(act3_next, torch.float32, E * 3136, "act3_next"), _stream()), "a0_actor_quantile_tail_env_step_enc")

    def actor_qhead_scratch(self, E, K) -> int:
        return int(self.lib.a0_actor_qhead_scratch(E, K))

    def actor_qhead(self, feat, E, K, W1, b1, W2, b2, A, dueling, scratch, seed, stream_a, stream_u, off_a, off_u, eps, action, qmax, ctrl=None, eps_ptr=None):
        nq = A + (1 if dueling else 0)
        check(self.lib.a0_actor_qhead(_req(feat, torch.float32, E * K, "feat"), E, K, _req(W1, torch.float32, 512 * K, "W1"), _req(b1, torch.float32, 512, "b1"),
                                      _req(W2, torch.float32, nq * 512, "W2"), _req(b2, torch.float32, nq, "b2"), A, int(dueling),
                                      _req(scratch, torch.float32, self.actor_qhead_scratch(E, K), "scratch"), seed, stream_a, stream_u, off_a, off_u, float(eps),
                                      _req(ctrl, torch.int64, 8, "ctrl", optional=True), _req(eps_ptr, torch.float32, 1, "eps_ptr", optional=True),
                                      _req(action, torch.int32, E, "action"), _req(qmax, torch.float32, E, "qmax"), _stream()), "a0_actor_qhead")

    def actor_qhead_env_step(self, feat, E, K, W1, b1, W2, b2, A, dueling, scratch, seed, stream_a, stream_u, off_a, off_u, eps, action, qmax, ctrl, eps_ptr,
                             env_seed, rank, g, obs_in, obs_out, ep_ret, final_mask, final_ret, n, steps, gamma, ring_act, ring_rew, ring_done, obs0, frames, cap,
                             start_slot, r_act, r_rew, r_done, task=0):
        nq = A + (1 if dueling else 0)
        nb = E * 4 * 84 * 84
        check(self.lib.a0_actor_qhead_env_step(
            _req(feat, torch.float32, E * K, "feat"), E, K, _req(W1, torch.float32, 512 * K, "W1"), _req(b1, torch.float32, 512, "b1"),
            _req(W2, torch.float32, nq * 512, "W2"), _req(b2, torch.float32, nq, "b2"), A, int(dueling),
            _req(scratch, torch.float32, self.actor_qhead_scratch(E, K), "scratch"), seed, stream_a, stream_u, off_a, off_u, float(eps),
            _req(ctrl, torch.int64, 8, "ctrl", optional=True), _req(eps_ptr, torch.float32, 1, "eps_ptr", optional=True),
            _req(action, torch.int32, E, "action"), _req(qmax, torch.float32, E, "qmax"),
            env_seed, rank, g, _req(obs_in, torch.uint8, nb, "obs_in"), _req(obs_out, torch.uint8, nb, "obs_out"), _req(ep_ret, torch.float32, E, "ep_ret"),
            _req(final_mask, torch.float32, E, "final_mask"), _req(final_ret, torch.float32, E, "final_ret"), n, steps, float(gamma),
            _req(ring_act, torch.int32, n * E, "ring_act"), _req(ring_rew, torch.float32, n * E, "ring_rew"), _req(ring_done, torch.float32, n * E, "ring_done"),
            _req(obs0, torch.uint8, nb, "obs0"), _req(frames, torch.uint8, cap * 8 * 84 * 84, "frames"), cap, start_slot, _req(r_act, torch.int32, cap, "r_act"),
            _req(r_rew, torch.float32, cap, "r_rew"), _req(r_done, torch.float32, cap, "r_done"), int(task), _stream()), "a0_actor_qhead_env_step")

    def actor_qhead_env_step_enc(self, feat, E, K, W1, b1, W2, b2, A, dueling, scratch, seed, stream_a, stream_u, off_a, off_u, eps, action, qmax, ctrl, eps_ptr,
                                 env_seed, rank, g, obs_in, obs_out, ep_ret, final_mask, final_ret, n, steps, gamma, ring_act, ring_rew, ring_done, obs0, frames, cap,
                                 start_slot, r_act, r_rew, r_done, task=0, wt=None, enc_w=None, act3_next=None):
        """``actor_qhead_env_step`` whose tail kernel goes on to encode the env's new observation into ``act3_next`` (a0_actor_qhead_env_step_enc): a step in two launches."""
        nq = A + (1 if dueling else 0)
        nb = E * 4 * 84 * 84
        ew = self._enc_w(enc_w)
        check(self.lib.a0_actor_qhead_env_step_enc(
            _req(feat, torch.float32, E * K, "feat"), E, K, _req(W1, torch.float32, 512 * K, "W1"), _req(b1, torch.float32, 512, "b1"),
            _req(W2, torch.float32, nq * 512, "W2"), _req(b2, torch.float32, nq, "b2"), A, int(dueling),
            _req(scratch, torch.float32, self.actor_qhead_scratch(E, K), "scratch"), seed, stream_a, stream_u, off_a, off_u, float(eps),
            _req(ctrl, torch.int64, 8, "ctrl", optional=True), _req(eps_ptr, torch.float32, 1, "eps_ptr", optional=True),
            _req(action, torch.int32, E, "action"), _req(qmax, torch.float32, E, "qmax"),
            env_seed, rank, g, _req(obs_in, torch.uint8, nb, "obs_in"), _req(obs_out, torch.uint8, nb, "obs_out"), _req(ep_ret, torch.float32, E, "ep_ret"),
            _req(final_mask, torch.float32, E, "final_mask"), _req(final_ret, torch.float32, E, "final_ret"), n, steps, float(gamma),
            _req(ring_act, torch.int32, n * E, "ring_act"), _req(ring_rew, torch.float32, n * E, "ring_rew"), _req(ring_done, torch.float32, n * E, "ring_done"),
            _req(obs0, torch.uint8, nb, "obs0"), _req(frames, torch.uint8, cap * 8 * 84 * 84, "frames"), cap, start_slot, _req(r_act, torch.int32, cap, "r_act"),
            _req(r_rew, torch.float32, cap, "r_rew"), _req(r_done, torch.float32, cap, "r_done"), int(task),
            _req(wt, torch.float32, self.conv_wt_floats(4), "wt"), C.addressof(ew), _req(act3_next, torch.float32, E * K, "act3_next"), _stream()), "a0_actor_qhead_env_step_enc")

    def mean_rows(self, x, T, E, out):
        check(self.lib.a0_mean_rows(_req(x, torch.float32, T * E, "x"), T, E, _req(out, torch.float32, T, "out"), _stream()), "a0_mean_rows")

    def fill_f32(self, p, n, v):
        check(self.lib.a0_fill_f32(_req(p, torch.float32, n, "p"), n, v, _stream()), "a0_fill_f32")

    def priority_update(self, priority, ids, loss, B, eps, alpha, pstate, state):
        check(self.lib.a0_priority_update(_req(priority, torch.float32, 1, "priority"), _req(ids, torch.int64, B, "ids"), _req(loss, torch.float32, B, "loss"), B,
                                          eps, alpha, _req(pstate, torch.float32, 1, "pstate"), _req(state, torch.int32, 8, "state", optional=True), _stream()), "a0_priority_update")

    def priority_tail(self, priority, size, n, pstate, alpha):
        check(self.lib.a0_priority_tail(_req(priority, torch.float32, size, "priority"), size, n, _req(pstate, torch.float32, 1, "pstate"), alpha, _stream()), "a0_priority_tail")

    def sum_f32(self, x, n, scratch256, out):
        check(self.lib.a0_sum_f32(_req(x, torch.float32, n, "x"), n, _req(scratch256, torch.float32, 256, "scratch256"), _req(out, torch.float32, 1, "out"), _stream()), "a0_sum_f32")

    def is_weights(self, prio, B, psum, top, beta, w):
        check(self.lib.a0_is_weights(_req(prio, torch.float32, B, "prio"), B, _req(psum, torch.float32, 1, "psum"), top, beta, _req(w, torch.float32, B, "w"), _stream()), "a0_is_weights")

    def perm_batch(self, start, count, n, seed, out):
        check(self.lib.a0_perm_batch(start, count, n, seed & 0xFFFFFFFF, _req(out, torch.int64, count, "out"), _stream()), "a0_perm_batch")

    def sumtree_set(self, tree, cap2, idx, val, n, state=None):
        check(self.lib.a0_sumtree_set(_req(tree, torch.float32, 2 * cap2, "tree"), cap2, _req(idx, torch.int64, n, "idx"), _req(val, torch.float32, n, "val"), n,
                                      _req(state, torch.int32, 8, "state", optional=True), _stream()), "a0_sumtree_set")

    def sumtree_sample_batch(self, seed, stream, offset, tree, cap2, B, top, cap, beta, r_act, r_rew, r_done, idx_out, slot_out, act, rew, done, prio, w, rebuild_top=False):
        check(self.lib.a0_sumtree_sample_batch(seed, stream, offset, _req(tree, torch.float32, 2 * cap2, "tree"), cap2, B, top, cap, float(beta),
                                               _req(r_act, torch.int32, cap, "r_act"), _req(r_rew, torch.float32, cap, "r_rew"), _req(r_done, torch.float32, cap, "r_done"),
                                               _req(idx_out, torch.int64, B, "idx_out"), _req(slot_out, torch.int32, B, "slot_out"), _req(act, torch.int32, B, "act"),
                                               _req(rew, torch.float32, B, "rew"), _req(done, torch.float32, B, "done"), _req(prio, torch.float32, B, "prio"),
                                               _req(w, torch.float32, B, "w"), 1 if rebuild_top else 0, _stream()), "a0_sumtree_sample_batch")

    def sumtree_set_range(self, tree, cap2, start, n, size, val):
        check(self.lib.a0_sumtree_set_range(_req(tree, torch.float32, 2 * cap2, "tree"), cap2, start, n, size, _req(val, torch.float32, 1, "val"), _stream()),
              "a0_sumtree_set_range")

    def sumtree_rebuild(self, tree, cap2):
        check(self.lib.a0_sumtree_rebuild(_req(tree, torch.float32, 2 * cap2, "tree"), cap2, _stream()), "a0_sumtree_rebuild")

    def sumtree_sample(self, tree, cap2, xi, B, out_idx, out_p):
        check(self.lib.a0_sumtree_sample(_req(tree, torch.float32, 2 * cap2, "tree"), cap2, _req(xi, torch.float32, B, "xi"), B,
                                         _req(out_idx, torch.int64, B, "out_idx"), _req(out_p, torch.float32, B, "out_p"), _stream()), "a0_sumtree_sample")

    def sumtree_set_from_loss_ok(self, cap2) -> bool:
        return bool(self.lib.a0_sumtree_set_from_loss_ok(cap2))

    def sumtree_top_rebuild(self, tree, cap2):
        check(self.lib.a0_sumtree_top_rebuild(_req(tree, torch.float32, 2 * cap2, "tree"), cap2, _stream()), "a0_sumtree_top_rebuild")

    def sumtree_set_from_loss(self, tree, cap2, idx, loss, n, eps, alpha, pstate, state=None, defer_top=False):
        check(self.lib.a0_sumtree_set_from_loss(_req(tree, torch.float32, 2 * cap2, "tree"), cap2, _req(idx, torch.int64, n, "idx"), _req(loss, torch.float32, n, "loss"), n,
                                                float(eps), float(alpha), _req(pstate, torch.float32, 1, "pstate"), _req(state, torch.int32, 8, "state", optional=True),
                                                1 if defer_top else 0, _stream()),
              "a0_sumtree_set_from_loss")

    def priority_from_loss(self, loss, n, eps, alpha, val, pstate, state=None):
        check(self.lib.a0_priority_from_loss(_req(loss, torch.float32, n, "loss"), n, eps, alpha, _req(val, torch.float32, n, "val"), _req(pstate, torch.float32, 1, "pstate"),
                                             _req(state, torch.int32, 8, "state", optional=True), _stream()), "a0_priority_from_loss")

    # ------------------------------------------------------------------ actor / rng / env
    def actor_egreedy(self, greedy, rand_action, u, eps, E, action, qmax, qs_out):
        check(self.lib.a0_actor_egreedy(_req(greedy, torch.int32, E, "greedy"), _req(rand_action, torch.int32, E, "rand_action"), _req(u, torch.float32, E, "u"), eps, E,
                                        _req(action, torch.int32, E, "action"), _req(qmax, torch.float32, E, "qmax", optional=True),
                                        _req(qs_out, torch.float32, 1, "qs_out", optional=True), _stream()), "a0_actor_egreedy")

    def actor_egreedy_rng(self, greedy, seed, stream_a, stream_u, off_a, off_u, A, eps, E, action, qmax, qs_out, ctrl=None, eps_ptr=None):
        check(self.lib.a0_actor_egreedy_rng(_req(greedy, torch.int32, E, "greedy"), seed, stream_a, stream_u, off_a, off_u, A, eps, E,
                                            _req(action, torch.int32, E, "action"), _req(qmax, torch.float32, E, "qmax", optional=True),
                                            _req(qs_out, torch.float32, 1, "qs_out", optional=True), _req(ctrl, torch.int64, 8, "ctrl", optional=True),
                                            _req(eps_ptr, torch.float32, 1, "eps_ptr", optional=True), _stream()), "a0_actor_egreedy_rng")

    def actor_nstep(self, E, n, steps, gamma, action, reward, terminal, truncated, life_loss, ring_act, ring_rew, ring_done, out_act, out_rew, out_done, ctrl=None):
        check(self.lib.a0_actor_nstep(E, n, steps, gamma, _req(action, torch.int32, E, "action"), _req(reward, torch.float32, E, "reward"),
                                      _req(terminal, torch.float32, E, "terminal"), _req(truncated, torch.float32, E, "truncated"),
                                      _req(life_loss, torch.float32, E, "life_loss", optional=True), _req(ring_act, torch.int32, n * E, "ring_act"),
                                      _req(ring_rew, torch.float32, n * E, "ring_rew"), _req(ring_done, torch.float32, n * E, "ring_done"),
                                      _req(out_act, torch.int32, E, "out_act"), _req(out_rew, torch.float32, E, "out_rew"), _req(out_done, torch.float32, E, "out_done"),
                                      _req(ctrl, torch.int64, 8, "ctrl", optional=True), _stream()), "a0_actor_nstep")

    def env_frame_stack(self, prev, newest, advance, out, E, nstack, frame_bytes):
        check(self.lib.a0_env_frame_stack(_req(prev, torch.uint8, E * nstack * frame_bytes, "prev"), _req(newest, torch.uint8, E * frame_bytes, "newest"),
                                          _req(advance, torch.float32, E, "advance"), _req(out, torch.uint8, E * nstack * frame_bytes, "out"), E, nstack,
                                          frame_bytes, _stream()), "a0_env_frame_stack")

    def rng_uniform(self, seed, stream_id, offset, out, n):
        check(self.lib.a0_rng_uniform(seed, stream_id, offset, _req(out, torch.float32, n, "out"), n, _stream()), "a0_rng_uniform")

    def rng_uniform_ctrl(self, seed, stream_id, offset, out, n, ctrl, ctrl_idx):
        check(self.lib.a0_rng_uniform_ctrl(seed, stream_id, offset, _req(out, torch.float32, n, "out"), n, _req(ctrl, torch.int64, 8, "ctrl"), ctrl_idx, _stream()), "a0_rng_uniform_ctrl")

    def rng_normal_ctrl(self, seed, stream_id, offset, std, out, n, ctrl, ctrl_idx):
        check(self.lib.a0_rng_normal_ctrl(seed, stream_id, offset, std, _req(out, torch.float32, n, "out"), n, _req(ctrl, torch.int64, 8, "ctrl"), ctrl_idx, _stream()), "a0_rng_normal_ctrl")

    def rng_u32(self, seed, stream_id, offset, out, n):
        check(self.lib.a0_rng_u32(seed, stream_id, offset, _req(out, torch.int32, n, "out"), n, _stream()), "a0_rng_u32")

    def rng_randint(self, seed, stream_id, offset, hi, out, n):
        check(self.lib.a0_rng_randint(seed, stream_id, offset, hi, _req(out, torch.int32, n, "out"), n, _stream()), "a0_rng_randint")

    def rng_normal(self, seed, stream_id, offset, std, out, n):
        check(self.lib.a0_rng_normal(seed, stream_id, offset, std, _req(out, torch.float32, n, "out"), n, _stream()), "a0_rng_normal")

    def env_reset(self, seed, rank, E, obs, ep_ret, task=0):
        check(self.lib.a0_env_synth_reset_task(seed, rank, E, _req(obs, torch.uint8, E * 4 * 84 * 84, "obs"), _req(ep_ret, torch.float32, E, "ep_ret"), int(task), _stream()),
              "a0_env_synth_reset_task")

    def env_step(self, seed, rank, E, g, obs_in, obs_out, ep_ret, reward, terminal, truncated, life_loss, final_mask, final_ret, ctrl=None, action=None, A=1, task=0):
        """``task`` 0: action-independent reward stream; 1: the learnable block task, 2: the chase task (both need ``action`` [E] int32 and the action count ``A``)."""
        n = E * 4 * 84 * 84
        check(self.lib.a0_env_synth_step(seed, rank, E, g, _req(obs_in, torch.uint8, n, "obs_in"), _req(obs_out, torch.uint8, n, "obs_out"),
                                         _req(ep_ret, torch.float32, E, "ep_ret"), _req(reward, torch.float32, E, "reward"), _req(terminal, torch.float32, E, "terminal"),
                                         _req(truncated, torch.float32, E, "truncated"), _req(life_loss, torch.float32, E, "life_loss"),
                                         _req(final_mask, torch.float32, E, "final_mask"), _req(final_ret, torch.float32, E, "final_ret"),
                                         _req(action, torch.int32, E, "action", optional=(task == 0)), int(A), int(task),
                                         _req(ctrl, torch.int64, 8, "ctrl", optional=True), _stream()), "a0_env_synth_step")

    def env_step_commit(self, seed, rank, E, g, obs_in, obs_out, ep_ret, final_mask, final_ret, n, steps, gamma, action, ring_act, ring_rew, ring_done, obs0,
                        frames, cap, start_slot, r_act, r_rew, r_done, ctrl=None, A=1, task=0):
        nb = E * 4 * 84 * 84
        check(self.lib.a0_env_synth_step_commit(seed, rank, E, g, _req(obs_in, torch.uint8, nb, "obs_in"), _req(obs_out, torch.uint8, nb, "obs_out"),
                                                _req(ep_ret, torch.float32, E, "ep_ret"), _req(final_mask, torch.float32, E, "final_mask"),
                                                _req(final_ret, torch.float32, E, "final_ret"), n, steps, float(gamma), _req(action, torch.int32, E, "action"),
                                                _req(ring_act, torch.int32, n * E, "ring_act"), _req(ring_rew, torch.float32, n * E, "ring_rew"),
                                                _req(ring_done, torch.float32, n * E, "ring_done"), _req(obs0, torch.uint8, nb, "obs0"),
                                                _req(frames, torch.uint8, cap * 8 * 84 * 84, "frames"), cap, start_slot, _req(r_act, torch.int32, cap, "r_act"),
                                                _req(r_rew, torch.float32, cap, "r_rew"), _req(r_done, torch.float32, cap, "r_done"), int(A), int(task),
                                                _req(ctrl, torch.int64, 8, "ctrl", optional=True), _stream()), "a0_env_synth_step_commit")

    # ------------------------------------------------------------------ data-parallel gradient exchange (RCCL, include/agent0_hip.h a0_dp_*)
    def dp_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        check(self.lib.a0_dp_unique_id(buf), "a0_dp_unique_id")
        return buf.raw

    def dp_init(self, unique_id: bytes, rank: int, world: int) -> int:
        if len(unique_id) != 128:
            raise ValueError("the RCCL rendezvous blob is 128 bytes")
        comm = int(self.lib.a0_dp_init(C.create_string_buffer(unique_id, 128), rank, world))
        if comm == 0:
            check(-1, "a0_dp_init")
        return comm

    def dp_allreduce(self, comm: int, buf, n: int, stream=None):
        """In-place fp32 SUM over the communicator's ranks, enqueued on ``stream`` (a torch.cuda.Stream; default: the current one)."""
        check(self.lib.a0_dp_allreduce(comm, _req(buf, torch.float32, n, "buf"), n, _stream() if stream is None else stream.cuda_stream), "a0_dp_allreduce")

    def dp_info(self, comm: int):
        """(ncclCommCount, ncclCommUserRank, ncclCommCuDevice) of a communicator, as RCCL reports them."""
        out = (C.c_int * 3)()
        check(self.lib.a0_dp_info(comm, out), "a0_dp_info")
        return int(out[0]), int(out[1]), int(out[2])

    def dp_destroy(self, comm: int):
        check(self.lib.a0_dp_destroy(comm), "a0_dp_destroy")

    # ------------------------------------------------------------------ measurement
    PROBE_TAGS = {"conv1_fwd": 1, "conv2_fwd": 2, "conv3_fwd": 3, "dense_fwd": 4, "dense_dgrad": 5, "dense_wgrad": 6, "conv3_wgrad": 7,
                  "conv3_dgrad": 8, "conv2_wgrad": 9, "conv2_dgrad": 10, "conv1_wgrad": 11, "encoder_fused": 12, "encoder_dgrad_fused": 13, "actor_step_enc": 14}

    def dense_dgrad_wgrad_ok(self, R, N, K) -> bool:
        return bool(self.lib.a0_dense_dgrad_wgrad_ok(R, N, K))

    def dense_dgrad_wgrad_ok2(self, R, N, K) -> bool:
        return bool(self.lib.a0_dense_dgrad_wgrad_ok2(R, N, K))

    def dense_dgrad_wgrad(self, dY, W, X, ldx, dX, grad, R, N, K):
        """One layer's masked data gradient and unsplit weight gradient in one launch (a0_dense_dgrad_wgrad)."""
        check(self.lib.a0_dense_dgrad_wgrad(_req(dY, torch.float32, R * N, "dY"), _req(W, torch.float32, N * K, "W"), _req(X, torch.float32, (R - 1) * ldx + K, "X"), ldx,
                                            _req(dX, torch.float32, R * K, "dX"), _req(grad, torch.float32, N * K + N, "grad"), R, N, K, _stream()), "a0_dense_dgrad_wgrad")

    def dense_dgrad_wgrad2_ok(self, R, N, K, N2, K2) -> bool:
        return bool(self.lib.a0_dense_dgrad_wgrad2_ok(R, N, K, N2, K2))

    def dense_dgrad_wgrad2(self, dY, W, X, ldx, dX, grad, R, N, K, dY2, X2, ldx2, grad2, N2, K2):
        """``dense_dgrad_wgrad`` plus the next layer's unsplit weight gradient (dY2^T X2 into grad2 [N2 x K2 | N2]) in the same launch (a0_dense_dgrad_wgrad2)."""
        check(self.lib.a0_dense_dgrad_wgrad2(_req(dY, torch.float32, R * N, "dY"), _req(W, torch.float32, N * K, "W"), _req(X, torch.float32, (R - 1) * ldx + K, "X"), ldx,
                                             _req(dX, torch.float32, R * K, "dX"), _req(grad, torch.float32, N * K + N, "grad"), R, N, K,
                                             _req(dY2, torch.float32, R * N2, "dY2"), _req(X2, torch.float32, (R - 1) * ldx2 + K2, "X2"), ldx2,
                                             _req(grad2, torch.float32, N2 * K2 + N2, "grad2"), N2, K2, _stream()), "a0_dense_dgrad_wgrad2")

    def dense_wgrad_multi(self, layers, slabs, pend=None):
        """layers: [(dY, X, ldx, grad, R, N, K)] (at most four); their slab reductions run as one launch — or, with ``pend``, in the next encoder_wgrad(pend=...)'s."""
        n = len(layers)
        offs, total = [], 0
        for (_, _, _, _, R, N, K) in layers:
            offs.append(total)
            total += (self.dense_wgrad_scratch(R, N, K) + 3) // 4 * 4
        PP = C.c_void_p * n
        II = C.c_int * n
        dY = PP(*[_req(l[0], torch.float32, l[4] * l[5], "dY") for l in layers])
        X = PP(*[_req(l[1], torch.float32, (l[4] - 1) * l[2] + l[6], "X") for l in layers])
        G = PP(*[_req(l[3], torch.float32, l[5] * l[6] + l[5], "grad") for l in layers])
        check(self.lib.a0_dense_wgrad_multi(n, dY, X, II(*[l[2] for l in layers]), G, II(*[l[4] for l in layers]), II(*[l[5] for l in layers]), II(*[l[6] for l in layers]),
                                            _req(slabs, torch.float32, total, "slabs", optional=(total == 0)), (C.c_longlong * n)(*offs), None if pend is None else C.addressof(pend), _stream()), "a0_dense_wgrad_multi")

    def dense_wgrad_multi_scratch(self, shapes) -> int:
        return sum((self.dense_wgrad_scratch(R, N, K) + 3) // 4 * 4 for (R, N, K) in shapes)

    def probe_begin(self, name: str, max_launches: int = 8192):
        self._probe_name = name
        check(self.lib.a0_probe_begin(self.PROBE_TAGS[name], max_launches), "a0_probe_begin")

    def probe_end(self):
        out = (C.c_double * 3)()
        check(self.lib.a0_probe_end(C.addressof(out)), "a0_probe_end")
        return {"kernel": self._probe_name, "launches": int(out[0]), "ms": float(out[1]), "flop": float(out[2])}

    def gemm_mode(self, mode: int = -1) -> int:
        """1 = split-operand bf16 MFMA GEMMs (default), 0 = fp32 MFMA fmaf chain; returns the previous mode (mode < 0: query only)."""
        return int(self.lib.a0_gemm_mode(int(mode)))

    # ------------------------------------------------------------------ roctx ranges (A0_ROCTX=1; rocprofv3 --marker-trace)
    def trace_enabled(self) -> bool:
        if getattr(self, "_trace_on", None) is None:
            self._trace_on = bool(self.lib.a0_trace_enabled())
        return self._trace_on

    def trace_rank(self, rank: int):
        self.lib.a0_trace_rank(int(rank))

    def range(self, name: str):
        """Context manager: a named host-side range around the enqueue of whatever runs inside (a shared no-op object unless A0_ROCTX=1)."""
        return _TraceRange(self.lib, name.encode()) if self.trace_enabled() else _NO_RANGE

    def x9_products(self, n: int = -1) -> int:
        """Cross products of the split-operand kernels: 6 (default) or 9 (strict); returns the previous value (other n: query only).  Not during graph capture."""
        return int(self.lib.a0_x9_products(int(n)))

    def device_info(self):
        cu = C.c_int()
        mem = C.c_longlong()
        name = C.create_string_buffer(64)
        check(self.lib.a0_device_info(C.addressof(cu), C.addressof(mem), C.addressof(name)), "a0_device_info")
        return cu.value, mem.value, name.value.decode()

    def build_info(self) -> str:
        return (self.lib.a0_build_info() or b"").decode()
