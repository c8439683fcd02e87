"""The actor -> replay -> learner loop of ``Trainer.run_iteration`` issued by the library's own handles (``a0_actor`` / ``a0_rbuf`` / ``a0_learner``,
csrc/runtime.hip + learner.hip) instead of by Python: one C call per rollout, per batch, per update.

Why: the launches of a rollout (240) and of an update block (20 x ~16) cost a fast native host ~2 us each and are issued back to back — no graph capture, no warm-up
runs, no graph cache, and 0 - 0.8 % faster than replaying them from hipGraphs (alternating same-box runs, profiles/r04_experiments.md); issued one by one from Python
they would be much slower than either.  The handles are created OVER the buffers the Python classes already hold (``a0_learner_create_on`` / ``a0_rbuf_create_on``): parameters,
target, Adam moments, status words, loss ring, weight copies, NoisyNet buffers, replay ring, sum-tree — so ``state_dict()``, checkpoints, the test actor and every
reader of ``trainer.replay`` keep seeing the live data, with no copies in either direction.  What the handles own themselves: the actor's env state and Philox
offsets, the sampler's state (epochs / beta), the workspaces.

Scope = what the handles cover (include/agent0_hip.h): all six learners (scalar heads with A + dueling <= 24), with or without NoisyNet, on 4 x 84 x 84 observations, the
device-resident env's stream / block / chase tasks, uniform, sum-tree or (round 6) the reference-faithful flat-priority replay, one GPU or — round 6, the default — data
parallelism over RCCL (the learner handle issues the two all-reduces itself: ``a0_learner_set_exchange``), the ``main`` AND the ``launch`` schedule — there the actor
handle owns a copy of the network (a0_actor_bind(.., 1) / a0_actor_snapshot) and rolls out into the Trainer's stage ring on the actor stream while the update block runs
(``run_iteration_lp``).  What stays on the Python classes: host environments (the worker pool is Python) and any Trainer whose hot-loop methods a test harness has wrapped.  Same launches, same order, same arguments: a run is BIT-identical either way (tests/test_gpu_trainer.py::test_native_loop_equals_the_python_classes).
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Optional

import numpy as np
import torch

from agent0_amd import _abi
from .config import ReplayEnum


class _RbufDesc(C.Structure):
    _fields_ = [("size", C.c_longlong), ("obs_bytes", C.c_int), ("B", C.c_int), ("prioritize", C.c_int), ("alpha", C.c_double), ("eps", C.c_double), ("beta0", C.c_double),
                ("total_steps", C.c_longlong), ("seed", C.c_ulonglong)]


class _ActorDesc(C.Structure):
    _fields_ = [("E", C.c_int), ("T", C.c_int), ("A", C.c_int), ("dueling", C.c_int), ("n_step", C.c_int), ("discount", C.c_double), ("seed", C.c_ulonglong), ("rank", C.c_uint),
                ("env_task", C.c_int), ("reset_noise_freq", C.c_int)]


class _LearnerBuffers(C.Structure):
    _fields_ = [("online", C.c_void_p), ("target", C.c_void_p), ("grads", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("state", C.c_void_p), ("scalars", C.c_void_p),
                ("loss_ring", C.c_void_p), ("loss_ring_cap", C.c_int), ("wt_online", C.c_void_p), ("wt_target", C.c_void_p), ("eff_online", C.c_void_p), ("eff_target", C.c_void_p),
                ("noise", C.c_void_p), ("rms_sq", C.c_void_p)]


class _Batch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("idx", "slot", "act", "rew", "done", "prio", "weights")]


def _wrapped(obj) -> bool:
    """An instance-level function on one of the hot-loop objects: a test harness intercepting calls (tests/test_gpu_trace.py)."""
    return any(type(v).__name__ in ("function", "method") for k, v in vars(obj).items() if k != "epsilon_fn")


def hook_ok(hook) -> bool:
    """No gradient hook, or the RCCL exchange of dist.RcclGradAllReduce, which the learner handle issues itself (``a0_learner_set_exchange``: same communicator, buckets and
    order as the captured form — the dense bucket beside the encoder backward on a side stream when the group has more than one rank, on the update's own stream in a
    one-rank group, where there is nothing to overlap).  The default since round 6, so that an N > 1 job runs the host loop every one-GPU number was measured with;
    A0_NATIVE_LOOP_DP=0 (the same on every rank) keeps data parallelism on the Python classes + hipGraphs.  Any other hook (the gloo test exchange) stays there too."""
    if hook is None:
        return True
    from .dist import RcclGradAllReduce
    return os.environ.get("A0_NATIVE_LOOP_DP", "1") != "0" and isinstance(hook, RcclGradAllReduce)


def eligible(tr) -> Optional[str]:
    """None when the Trainer's configuration is one the handles cover, else the reason it is not."""
    from agent0_amd.common.atari_wrappers import DeviceSynthVecEnv
    cfg = tr.cfg
    lc = cfg.learner
    if os.environ.get("A0_NATIVE_LOOP", "1") == "0":
        return "A0_NATIVE_LOOP=0"
    algo = lc.algo.name
    if algo == "dqn":
        if cfg.action_dim + (1 if lc.dueling_head else 0) > 24:
            return "dqn handle: A + dueling <= 24"
    elif algo in ("iqn", "fqf"):
        if cfg.action_dim + (1 if lc.dueling_head else 0) > 32 or not getattr(tr.actors[1], "quant_tail", False):
            return "quantile handles: A + dueling <= 32, the merged quantile tail"
    elif algo == "qr":
        if not getattr(tr.actors[1], "dist_tail", False):
            return "qr handle: the distributional tail kernel"
    elif algo == "mdqn":
        if not getattr(tr.actors[1], "fused_tail", False):
            return "mdqn handle: A + dueling <= 24"
    elif algo != "c51":
        return f"no handle for {algo}"
    if tuple(cfg.obs_shape) != (4, 84, 84):
        return "observations other than 4 x 84 x 84"
    actor = tr.actors[1]
    if not isinstance(actor.envs, DeviceSynthVecEnv) or actor.groups is not None:
        return "host environments"
    rp = tr.replay
    if rp.use_sumtree and lc.batch_size > 1024:
        return "prioritized batches above 1024"
    eng = tr.learner.engine
    if not (eng.online.fused and eng.online.fused_dgrad) or tr.ops.gemm_mode() != 1 or not getattr(eng, "_defer_dense", False):
        return "a tuning mode of the Python composition"
    if lc.learner_steps > eng.loss_ring.numel():
        return "block longer than the loss ring"
    if tr.use_lp and not tr.overlap:
        return "the launch schedule on one stream (a test mode of the Python classes)"
    if algo == "c51" and not (2 <= lc.c51.num_atoms <= 64):
        return "support size"
    return None


class NativeLoop:
    def __init__(self, tr):
        self.tr = tr
        self.learner = self.rbuf = self.actor = None        # close() destroys whatever exists, also when a later create call fails
        self.lib, self.ok = _abi.load(), _abi.check
        try:
            self._create(tr)
        except Exception:
            self.close()
            raise

    def _create(self, tr):
        cfg, ops = tr.cfg, tr.ops
        lc, rc = cfg.learner, cfg.replay
        lib, ok = self.lib, self.ok
        eng, rp, actor = tr.learner.engine, tr.replay, tr.actors[1]
        L = eng.L
        self.E, self.T, self.B = int(cfg.actor.num_envs), int(cfg.actor.sample_steps), int(lc.batch_size)
        self.prio = rp.prioritize
        # ---- learner over the engine's buffers
        desc = _abi.LearnerDesc(int(cfg.action_dim), int(bool(lc.dueling_head)), int(bool(lc.double_q)), self.B, int(lc.n_step_q), float(lc.discount), float(lc.learning_rate),
                                float(eng.adam_eps), int(lc.target_update_freq), {"dqn": 0, "c51": 1, "iqn": 2, "fqf": 3, "qr": 4, "mdqn": 5}[lc.algo.name], int(lc.qr.num_atoms if lc.algo.name == "qr" else lc.c51.num_atoms), float(lc.c51.vmin), float(lc.c51.vmax),
                                int(bool(lc.noisy_net)), (int(cfg.seed) + 15485863) & 0xFFFFFFFFFFFFFFFF, int(lc.iqn.K), int(lc.iqn.N), int(lc.iqn.N_dash), int(lc.iqn.F), float(lc.mdqn.tau), float(lc.mdqn.lo),
                                float(lc.max_grad_norm))
        p = lambda t: None if t is None else t.data_ptr()
        bufs = _LearnerBuffers(p(eng.online.flat), p(eng.target.flat), p(eng.grads), p(eng.adam_m), p(eng.adam_v), p(eng.state), p(eng.scalars), p(eng.loss_ring),
                               int(eng.loss_ring.numel()), p(eng.online.wt), p(eng.target.wt), p(eng.online.eff), p(eng.target.eff), p(eng.noise_joint), p(getattr(eng, "rms_sq", None)))
        self.learner = C.c_void_p()
        ok(lib.a0_learner_create_on(C.addressof(desc), C.addressof(bufs), C.addressof(self.learner)), "a0_learner_create_on")
        assert int(lib.a0_learner_param_floats(self.learner)) == L.n_params_padded
        if lc.algo.name == "c51":
            atoms = (C.c_float * L.T)(*[float(x) for x in eng.atoms.cpu().tolist()])
            ok(lib.a0_learner_set_support(self.learner, atoms), "a0_learner_set_support")
        hook = eng.grad_hook
        if hook is not None:      # hook_ok: dist.RcclGradAllReduce — its communicator moves into the handle (an inactive hook, A0_DP_DRYRUN=1, exchanges nothing there either)
            ok(lib.a0_learner_set_exchange(self.learner, C.c_longlong(int(hook.comm) if hook.active else 0)), "a0_learner_set_exchange")
        if lc.noisy_net:
            rng = tr.learner.rng
            ok(lib.a0_learner_set_rng(self.learner, rng.STREAM_NOISE, C.c_ulonglong(rng.offsets.get(rng.STREAM_NOISE, 0))), "a0_learner_set_rng")
        # ---- replay over the ReplayDataset's ring
        self.flat = self.prio and not rp.use_sumtree      # replay.sumtree=false: the reference's flat priority vector (a0_rbuf_desc.prioritize == 2; its `tree` argument is the vector)
        rd = _RbufDesc(int(rp.size), int(rp.obs_bytes), self.B, 2 if self.flat else int(self.prio), float(rc.alpha), float(rc.eps), float(rc.beta0), int(cfg.trainer.total_steps), int(cfg.seed) + 104729)
        self.rbuf = C.c_void_p()
        ok(lib.a0_rbuf_create_on(C.addressof(rd), p(rp.frames), p(rp.act), p(rp.rew), p(rp.done), p(rp.priority) if self.flat else p(rp._tree) if self.prio else None, p(rp._pstate), C.addressof(self.rbuf)),
           "a0_rbuf_create_on")
        # ---- actor (its own env state: the Python Actor's stays where the constructor left it)
        ad = _ActorDesc(self.E, self.T, int(cfg.action_dim), int(bool(lc.dueling_head)), int(lc.n_step_q), float(lc.discount), int(cfg.seed), int(tr.rank),
                        {"stream": 0, "block": 1, "chase": 2}[cfg.env_task], int(lc.reset_noise_freq))
        self.actor = C.c_void_p()
        ok(lib.a0_actor_create(C.addressof(ad), C.addressof(self.actor)), "a0_actor_create")
        # every workspace now, so that no call of the loop allocates; on the launch schedule the actor gets its OWN copy of the network (launch.py:34-36,58-62) and rolls
        # out into the Trainer's stage ring on the Trainer's actor stream
        self.lp = bool(tr.use_lp)
        ok(lib.a0_actor_bind(self.actor, self.learner, int(self.lp)), "a0_actor_bind")
        self.stage = None
        if self.lp:
            sg = tr.stage
            sd = _RbufDesc(int(sg.size), int(sg.obs_bytes), min(self.B, int(sg.size)), 0, float(rc.alpha), float(rc.eps), float(rc.beta0), int(cfg.trainer.total_steps), int(cfg.seed) + 7)
            self._stage_pstate = ops.zeros(1)
            self.stage = C.c_void_p()
            ok(lib.a0_rbuf_create_on(C.addressof(sd), p(sg.frames), p(sg.act), p(sg.rew), p(sg.done), None, p(self._stage_pstate), C.addressof(self.stage)), "a0_rbuf_create_on (stage)")
            self._lp_pending = None
        lp = C.c_void_p()
        ok(lib.a0_learner_loss_buffer(self.learner, C.addressof(lp)), "a0_learner_loss_buffer")
        self.loss_ptr = lp                                # the learner's own per-sample losses: update_priority reads them in place
        self._qs, self._rs, self._nret = (C.c_float * self.T)(), (C.c_float * (self.T * self.E))(), C.c_int()
        self._pending_rollout = False
        self.fqf = lc.algo.name == "fqf"
        self._floss = ops.empty(self.B) if self.fqf else None
        self.frames_ptr = p(rp.frames)
        self.row_bytes = int(rp.row_bytes)

    # ------------------------------------------------------------------ pieces of one iteration
    def _rollout(self, st):
        tr, lib = self.tr, self.lib
        self.ok(lib.a0_actor_rollout(self.actor, self.learner, self.rbuf, C.c_float(tr.epsilon_fn(tr.frame_count)), st), "a0_actor_rollout")

    def _commit(self, st):
        tr, rp = self.tr, self.tr.replay
        n = self.T * self.E
        self.ok(self.lib.a0_rbuf_commit(self.rbuf, n, st), "a0_rbuf_commit")
        tr.frame_count += n
        rp.written += n                                   # the ReplayDataset's counters follow (its buffers ARE the handle's)
        rp.top = min(rp.top + n, rp.size)
        if self.prio:
            beta = C.c_double()
            self.ok(self.lib.a0_rbuf_info(self.rbuf, None, None, C.addressof(beta)), "a0_rbuf_info")
            rp.beta = beta.value

    # the three links of one update (trainer.py:81-104) — kept apart so that a parity harness can stand between them (tests/test_gpu_trace.py walks this very path
    # against the CPU oracle link by link)
    def _sample(self, i: int, n: int, st):
        """The i-th batch of a block of n: sampled indices, ring slots, metadata and importance weights in the handle's device buffers (a0_batch)."""
        if self.prio:
            b = _Batch()
            self.ok(self.lib.a0_rbuf_sample(self.rbuf, C.addressof(b), st), "a0_rbuf_sample")
            return b
        if i % 32 == 0:                                   # uniform replay: the block's batches do not depend on its updates — 32 of them per sampling launch
            self._blk = (_Batch * 32)()
            self.ok(self.lib.a0_rbuf_sample_block(self.rbuf, min(32, n - i), self._blk, st), "a0_rbuf_sample_block")
        return self._blk[i % 32]

    def _update(self, b, st):
        self.ok(self.lib.a0_learner_update(self.learner, self.frames_ptr, b.slot, C.c_longlong(self.row_bytes), b.act, b.rew, b.done, b.weights, None, st), "a0_learner_update")

    def _priority(self, st):
        self.ok(self.lib.a0_rbuf_update_priority(self.rbuf, self.loss_ptr, self.tr.learner.engine.state.data_ptr(), st), "a0_rbuf_update_priority")

    def _block(self, st) -> int:
        tr, lib, ok = self.tr, self.lib, self.ok
        cfg = tr.cfg
        if int(lib.a0_rbuf_len(self.rbuf)) <= cfg.trainer.training_start_steps:
            return 0
        ln = tr.learner
        n = int(cfg.learner.learner_steps)
        if self.fqf and tr._floss_means.numel() < n:
            tr._loss_means, tr._floss_means = tr.ops.zeros(n), tr.ops.zeros(n)
        with tr.ops.range("update_block"):
            for i in range(n):
                b = self._sample(i, n, st)
                self._update(b, st)
                if self.prio:
                    self._priority(st)
                if self.fqf:                              # the `fraction_loss` statistic (trainer.py:99-101): batch mean of the update's fraction losses
                    ok(lib.a0_learner_get_frac_loss(self.learner, self._floss.data_ptr(), st), "a0_learner_get_frac_loss")
                    tr.ops.mean_rows(self._floss, 1, self.B, tr._floss_means[i:i + 1])
        tr._ring0 = ln.updates_issued                     # the Adam launch wrote the block's batch-mean losses to ring slots ring0 .. ring0 + n - 1
        ln.updates_issued += n
        if self.prio and not self.flat:
            tr.replay._top_stale = True                   # the handle defers the tree's top levels to its next sample; ReplayDataset.tree brings them up to date for other readers
        return n

    # ------------------------------------------------------------------ Trainer.run_iteration
    def run_iteration(self, prefetch: bool = False):
        tr, lib, ok = self.tr, self.lib, self.ok
        st = torch.cuda.current_stream().cuda_stream
        tic = time.time()
        if not self._pending_rollout:
            self._rollout(st)
        self._pending_rollout = False
        self._commit(st)
        n_upd = self._block(st)
        blk = tr._block_stats_async(n_upd, self.fqf and n_upd > 0)      # loss ring (+ fraction-loss means) -> page-locked buffers, stream-ordered behind the block
        ok(lib.a0_actor_collect_begin(self.actor, st), "a0_actor_collect_begin")
        if prefetch:
            self._rollout(st)                             # the next iteration's rollout before the host waits for this one's statistics
            self._pending_rollout = True
        ok(lib.a0_actor_collect_end(self.actor, self._qs, self._rs, self.T * self.E, C.addressof(self._nret)), "a0_actor_collect_end")
        tr._block_stats_finish(blk)
        tr.Qs.extend(np.ctypeslib.as_array(self._qs).tolist())
        tr.Rs.extend(np.ctypeslib.as_array(self._rs)[: self._nret.value].tolist())
        result = tr._result()
        if not prefetch:
            torch.cuda.synchronize()
        result.update(fps=tr.num_transitions / (time.time() - tic))
        return result

    # ------------------------------------------------------------------ the launch schedule (Trainer.run_iteration_lp; launch.py:30-63)
    def _snapshot_lp(self):
        """First half of ``actor.futures.sample(eps, state_dict)`` (launch.py:34-36,58-62): epsilon from the frame count as it is now, and the weight snapshot on the
        learner's stream — behind every update enqueued so far, ahead of the next.  Returns (epsilon, event the rollout has to wait for)."""
        tr = self.tr
        eps = tr.epsilon_fn(tr.frame_count)
        cur = torch.cuda.current_stream()
        self.ok(self.lib.a0_actor_snapshot(self.actor, self.learner, cur.cuda_stream), "a0_actor_snapshot")
        ev = torch.cuda.Event()
        ev.record(cur)
        return eps, ev

    def _rollout_lp(self, eps, ev):
        """Second half: the rollout into the free half of the stage on the actor stream, behind the snapshot, without waiting.  Returns its first stage row."""
        tr, lib, ok = self.tr, self.lib, self.ok
        ast = tr.actor_stream
        ast.wait_event(ev)
        start = int(lib.a0_rbuf_write_cursor(self.stage))
        ok(lib.a0_actor_rollout(self.actor, self.learner, self.stage, C.c_float(eps), ast.cuda_stream), "a0_actor_rollout")
        ok(lib.a0_rbuf_commit(self.stage, self.T * self.E, ast.cuda_stream), "a0_rbuf_commit (stage)")
        tr.stage.written += self.T * self.E
        return start

    def run_iteration_lp(self):
        """One pass of launch.py:44-63: collect the finished rollout, issue the next one with the current weights, run the update block on the collected transitions
        while that rollout is in flight.  The host enqueues the update block BEFORE the next rollout's 240 launches (the snapshot, which both depend on, first): the
        learner stream — the critical path — never waits for the host to finish issuing the actor's work; stream order and dependencies are those of
        Trainer.run_iteration_lp, so every number is too (tests/test_gpu_trainer.py::test_native_loop_equals_the_python_classes[launch-*])."""
        tr, lib, ok = self.tr, self.lib, self.ok
        st = torch.cuda.current_stream().cuda_stream
        tic = time.time()
        if self._lp_pending is None:
            self._lp_pending = self._rollout_lp(*self._snapshot_lp())          # launch.py:32-37 primes the pipeline before the loop
        # wait for the rollout in flight and take its statistics (a0_actor_collect synchronises the ACTOR stream)
        ok(lib.a0_actor_collect(self.actor, self._qs, self._rs, self.T * self.E, C.addressof(self._nret), tr.actor_stream.cuda_stream), "a0_actor_collect")
        start = self._lp_pending
        qs, rs = np.ctypeslib.as_array(self._qs).tolist(), np.ctypeslib.as_array(self._rs)[: self._nret.value].tolist()
        eps, ev = self._snapshot_lp()                    # the next rollout acts with the weights and the epsilon of NOW
        # Trainer.step: extend (the finished rollout's rows from the stage into the ring), then the update block
        n = self.T * self.E
        ok(lib.a0_rbuf_extend_from(self.rbuf, self.stage, start, n, st), "a0_rbuf_extend_from")
        rp = tr.replay
        tr.frame_count += n
        rp.written += n
        rp.top = min(rp.top + n, rp.size)
        if self.prio:
            beta = C.c_double()
            ok(lib.a0_rbuf_info(self.rbuf, None, None, C.addressof(beta)), "a0_rbuf_info")
            rp.beta = beta.value
        tr.Qs.extend(qs)
        tr.Rs.extend(rs)
        n_upd = self._block(st)
        blk = tr._block_stats_async(n_upd, self.fqf and n_upd > 0)
        self._lp_pending = self._rollout_lp(eps, ev)     # ... and runs beside the block
        torch.cuda.current_stream().synchronize()        # the update block (and the statistics' copies behind it); the next rollout keeps running on the actor stream
        tr._block_stats_finish(blk)
        result = tr._result()
        result.update(fps=tr.num_transitions / (time.time() - tic))
        return result

    def drain_lp(self):
        """The rollout still in flight when the run ends: waited for and dropped, like Trainer.final does for the Python actor's."""
        if self.lp and self._lp_pending is not None:
            self.ok(self.lib.a0_actor_collect(self.actor, self._qs, self._rs, self.T * self.E, C.addressof(self._nret), self.tr.actor_stream.cuda_stream), "a0_actor_collect")
            self._lp_pending = None

    def detach_exchange(self):
        """The handle stops exchanging gradients (before the communicator's owner destroys it)."""
        if self.learner:
            self.ok(self.lib.a0_learner_set_exchange(self.learner, C.c_longlong(0)), "a0_learner_set_exchange")

    def drain(self):
        """A rollout issued ahead and never consumed: book it (Trainer.final)."""
        if self._pending_rollout:
            st = torch.cuda.current_stream().cuda_stream
            self._pending_rollout = False
            self._commit(st)
            self.ok(self.lib.a0_actor_collect(self.actor, self._qs, self._rs, self.T * self.E, C.addressof(self._nret), st), "a0_actor_collect")
            self.tr.Qs.extend(np.ctypeslib.as_array(self._qs).tolist())
            self.tr.Rs.extend(np.ctypeslib.as_array(self._rs)[: self._nret.value].tolist())

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 — interpreter shutdown
            pass

    def close(self):
        if getattr(self, "actor", None) is None and getattr(self, "rbuf", None) is None and getattr(self, "learner", None) is None:
            return
        torch.cuda.synchronize()
        for h, fn in ((self.actor, self.lib.a0_actor_destroy), (self.rbuf, self.lib.a0_rbuf_destroy), (getattr(self, "stage", None), self.lib.a0_rbuf_destroy),
                      (self.learner, self.lib.a0_learner_destroy)):
            if h is not None and h.value:
                fn(h)
        self.actor = self.rbuf = self.learner = self.stage = None
