"""``Trainer``: the reference's orchestration loop, unchanged in shape, with every tensor resident in HBM.

Mirrors /root/reference agent0/deepq/trainer.py:19-189 — ``Trainer(cfg, use_lp=False)`` with ``run``, ``step``,
``test``, ``logging``, ``final``, the epsilon schedule (46-50, eps(0) = 1 + min_eps as in the reference) and the result
keys (111-118): frames, fraction_loss, loss, return_train, return_train_max, qmax, fps.
Per iteration: ``actors[1].sample(eps)`` rolls ``sample_steps`` env steps and writes the transitions straight into the
replay ring; ``step`` commits them and runs ``learner_steps`` updates (sample -> importance weights -> train ->
priority update), all enqueued on one HIP stream with a single device->host read at the end for the loss means.
wandb / tensorboard are used only if importable and enabled (neither ships in this image).
"""
from __future__ import annotations

import logging
import os
import time

import numpy as np
import torch

from . import agent as agents
from .config import ExpConfig, ReplayEnum, to_dict
from .replay import ReplayDataset, StageRing
from agent0_amd.common.atari_wrappers import make_atari
from agent0_amd.common.utils import set_random_seed


def _git_sha() -> str:
    from .main import _git_sha as sha
    return sha()


PROGRESS_COLUMNS = ("frames", "fraction_loss", "loss", "return_train", "return_train_max", "qmax", "fps")      # trainer.py:111-118 + fps (180-181)


def epsilon_schedule(cfg: ExpConfig):
    def fn(step):
        if step > cfg.trainer.exploration_steps:
            return cfg.actor.min_eps
        return (1.0 - step / cfg.trainer.exploration_steps) + cfg.actor.min_eps
    return fn


class Trainer:
    def __init__(self, cfg: ExpConfig, use_lp: bool = False, ops=None, rank: int = 0, primary: bool = True):
        """``primary=False`` (data-parallel replicas other than rank 0, launch.py here): no run directory, msg.log, tensorboard / wandb, final
        checkpoint or final test — the replicas are identical, so rank 0 reports for all of them."""
        self.cfg, self.use_lp, self.rank, self.primary = cfg, use_lp, rank, primary
        set_random_seed(cfg.seed)
        if cfg.device.value != "cuda":
            raise RuntimeError("agent0_amd runs on MI355X only: set device=cuda (no CPU fallback)")
        if ops is None:
            from agent0_amd.ops import HipOps
            ops = HipOps()
        self.ops = ops
        if ops.trace_enabled() and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            ops.trace_rank(rank)                          # roctx ranges carry the rank ("r3:exchange") in a data-parallel job
        dummy_env = make_atari(cfg.env_id, 1, ops=ops)
        self.obs_shape = tuple(dummy_env.observation_space.shape[1:])
        self.act_dim = int(dummy_env.action_space[0].n)
        dummy_env.close()
        if not cfg.obs_shape or len(tuple(cfg.obs_shape)) != 3:
            cfg.obs_shape = self.obs_shape
        if not cfg.action_dim:
            cfg.action_dim = self.act_dim
        try:
            learner_cls = getattr(agents, f"{cfg.learner.algo.name.upper()}Learner")
        except AttributeError:
            raise NotImplementedError(f"No such learner for {cfg.learner.algo.name}")
        self.learner = learner_cls(cfg, ops=ops)
        self.replay = ReplayDataset(cfg, ops=ops)
        if not use_lp:
            # the reference builds a test actor [0] and a train actor [1] sharing the learner's model (trainer.py:41-44)
            self.actors = [None, agents.Actor(cfg, self.learner.model, replay=self.replay, ops=ops, rank=rank)]
        else:
            # launch.py semantics on one device: the train actor owns a COPY of the network, refreshed when a rollout is issued
            # (launch.py:34-36,58-62), rolls out on its own HIP stream into a stage ring while the learner runs its update block on the
            # current stream.  ``overlap=False`` issues the same work on one stream (used by the tests to show the overlap is race-free).
            self.stage = StageRing(ops, 2 * cfg.actor.sample_steps * cfg.actor.num_envs, self.replay.obs_bytes)
            self.actors = [None, agents.Actor(cfg, None, replay=self.stage, ops=ops, rank=rank)]
            # high priority: a rollout kernel that becomes ready goes ahead of the update block's queued kernels.  No difference with the device env (9.43 - 9.49 ms
            # either way); with a host env, whose workers wait for every step's actions, 23.5 -> 21.7 ms per iteration (profiles/r04_experiments.md).  A0_ACTOR_STREAM_PRIO=0: default priority
            self.actor_stream = torch.cuda.Stream(priority=int(os.environ.get("A0_ACTOR_STREAM_PRIO", "-1")))
            self.overlap = True
            self._pending = None
        self.epsilon_fn = epsilon_schedule(cfg)
        self.writer = None
        self._wandb = None
        if cfg.wandb and primary:
            try:
                import wandb
                wandb.init(project=cfg.name, config=to_dict(cfg))
                self._wandb = wandb
            except ImportError:
                pass
        if cfg.tb and primary:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(cfg.logdir)
            except ImportError:
                pass
        self.logger = logging.getLogger("agent0")
        # the reference gets level INFO and a console handler from Hydra's job-logging defaults; without Hydra they are set here, so that
        # msg.log and the console carry the per-iteration lines (trainer.py:158-169)
        if self.logger.getEffectiveLevel() > logging.INFO:
            self.logger.setLevel(logging.INFO)
        if not logging.getLogger().handlers and not any(isinstance(h, logging.StreamHandler) and not isinstance(h, logging.FileHandler) for h in self.logger.handlers):
            console = logging.StreamHandler()
            console.setFormatter(logging.Formatter("[%(asctime)s][%(name)s][%(levelname)s] - %(message)s"))
            self.logger.addHandler(console)
        if primary:
            try:
                os.makedirs(cfg.logdir, exist_ok=True)
                self.logger.addHandler(logging.FileHandler(os.path.join(cfg.logdir, "msg.log")))
            except OSError:
                pass
        self.num_transitions = cfg.actor.sample_steps * cfg.actor.num_envs
        self.Ls, self.Rs, self.RTs, self.Qs, self.FLs = [], [], [], [], []
        self.frame_count = 0
        L = int(cfg.learner.learner_steps)
        self._loss_means = ops.zeros(max(L, 1))
        self._floss_means = ops.zeros(max(L, 1))

    # ------------------------------------------------------------------ checkpoint / resume
    def save_checkpoint(self, path: str):
        """Weights under the reference's state_dict keys (loadable by the reference's DeepQNet) + optimizer and counters."""
        eng = self.learner.engine
        blob = {"model": {k: v.cpu() for k, v in self.learner.model.state_dict().items()},
                "model_target": {k: v.cpu() for k, v in self.learner.model_target.state_dict().items()},
                "adam_m": eng.adam_m.cpu(), "adam_v": eng.adam_v.cpu(), "state": eng.state.cpu(), "frame_count": self.frame_count,
                "algo": self.cfg.learner.algo.name, "obs_shape": tuple(self.cfg.obs_shape), "action_dim": int(self.cfg.action_dim),
                # what agent0/summary.py:39-58 reads from a run: the test returns ("ITRs"), the frame count, and — there from params.json —
                # game / algo / commit; all in the checkpoint here (agent0_amd/summary.py)
                "ITRs": [float(x) for x in self.RTs], "game": str(self.cfg.env_id), "name": str(self.cfg.name), "sha": _git_sha()}
        if hasattr(eng, "rms_sq"):
            blob["rms_sq"] = eng.rms_sq.cpu()
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(blob, path)
        return path

    def load_checkpoint(self, path: str, weights_only: bool = False):
        blob = torch.load(path, map_location="cpu", weights_only=True)      # tensors, dicts, tuples, strings, ints only
        if blob.get("algo") not in (None, self.cfg.learner.algo.name) or int(blob.get("action_dim", self.cfg.action_dim)) != int(self.cfg.action_dim):
            raise ValueError(f"checkpoint {path} was written for {blob.get('algo')} / {blob.get('action_dim')} actions")
        self.learner.model.load_state_dict(blob["model"])
        self.learner.model_target.load_state_dict(blob.get("model_target", blob["model"]))
        if not weights_only and "adam_m" in blob:
            eng = self.learner.engine
            eng.adam_m.copy_(blob["adam_m"]); eng.adam_v.copy_(blob["adam_v"]); eng.state.copy_(blob["state"])
            if "rms_sq" in blob and hasattr(eng, "rms_sq"):
                eng.rms_sq.copy_(blob["rms_sq"])
            self.frame_count = int(blob.get("frame_count", 0))
            self._updates_done = None
            self.learner.updates_issued = int(eng.state[6])      # the device's count of loss-ring entries travels with the state block

    # ------------------------------------------------------------------ trainer.py:74-119
    def step(self, transitions, returns, qmax):
        self.Qs.extend(qmax)
        self.Rs.extend(returns)
        self._step_device(transitions)
        return self._result()

    def _step_device(self, transitions, defer: bool = False):
        """trainer.py:76-110 without the host-side statistics: commit the rollout, run the update block.  Everything here is enqueued on the
        stream; nothing waits for the device except the one read of the block's loss means at the end."""
        cfg = self.cfg
        self.replay.extend(transitions)
        self.frame_count += self.num_transitions
        n_upd = 0
        has_frac = False
        self._ring0 = None
        if len(self.replay) > cfg.trainer.training_start_steps and self._pipeline_ok():
            with self.ops.range("update_block"):
                n_upd = self._update_block_pipelined()
        elif len(self.replay) > cfg.trainer.training_start_steps:
            rp = self.replay
            if self._loss_means.numel() < cfg.learner.learner_steps:       # learner_steps changed after construction
                self._loss_means = self.ops.zeros(cfg.learner.learner_steps)
                self._floss_means = self.ops.zeros(cfg.learner.learner_steps)
            ring0 = self._loss_ring_start()
            for i in range(cfg.learner.learner_steps):
                b = rp.sample()
                q_loss, f_loss = self.learner.train_batch(rp.frames, b.slot, rp.row_bytes, b.act, b.rew, b.done, b.weights)
                if cfg.replay.policy == ReplayEnum.prioritize:
                    rp.update_priority(b.idx, q_loss, state=self.learner.engine.state)
                B = cfg.learner.batch_size
                if ring0 is None:
                    self.ops.mean_rows(q_loss, 1, B, self._loss_means[i:i + 1])      # one launch; read back once per update block
                if f_loss is not None:
                    self.ops.mean_rows(f_loss, 1, B, self._floss_means[i:i + 1])
                    has_frac = True
                n_upd += 1
        if defer:
            return self._block_stats_async(n_upd, has_frac)
        if n_upd:
            self.Ls.extend(self._block_loss_means(n_upd))                   # the one device->host read of the update block
            # the device's update count (NaN-skipped steps do not count) for the next block's pipelining decision: read here, where the host has just waited
            # for the block anyway, so that the next block can be enqueued behind the rollout without a stop
            self._updates_done = int(self.learner.engine.state[1]) if self._pipeline_candidate() else None
            if has_frac:
                self.FLs.extend(self._floss_means[:n_upd].cpu().tolist())
        return None

    # ------------------------------------------------------------------ statistics read-back that does not stop the stream (run_iteration(prefetch=True))
    def _block_stats_async(self, n_upd: int, has_frac: bool):
        """Enqueue the copies of the block's loss means into page-locked host buffers (stream-ordered behind the block, ahead of whatever is enqueued next);
        ``_block_stats_finish`` reads them once the caller has waited for an event recorded after this call."""
        if not n_upd:
            return (0, None, None, None)
        host = self.__dict__.setdefault("_stat_host", {})
        ring0 = getattr(self, "_ring0", None)
        src = self.learner.engine.loss_ring if ring0 is not None else self._loss_means
        key = ("ring" if ring0 is not None else "means", src.numel())
        if key not in host:
            host[key] = torch.empty(src.numel(), dtype=src.dtype).pin_memory()
        host[key].copy_(src, non_blocking=True)
        fl = None
        if has_frac:
            fk = ("fmeans", self._floss_means.numel())
            if fk not in host:
                host[fk] = torch.empty(self._floss_means.numel(), dtype=self._floss_means.dtype).pin_memory()
            host[fk].copy_(self._floss_means, non_blocking=True)
            fl = host[fk]
        self._updates_done = None
        return (n_upd, ring0, host[key], fl)

    def _block_stats_finish(self, handle):
        n_upd, ring0, buf, fl = handle
        if not n_upd:
            return
        if ring0 is None:
            self.Ls.extend(buf[:n_upd].tolist())
        else:
            self.Ls.extend(float(buf[(ring0 + i) % buf.numel()]) for i in range(n_upd))
        if fl is not None:
            self.FLs.extend(fl[:n_upd].tolist())

    # ------------------------------------------------------------------ the update block with the target network's passes one update ahead
    def _loss_ring_start(self):
        """The per-update loss means come out of the Adam launch (DeviceLearner.loss_ring, slot = update count % ring length) when the fused optimizer tail runs and
        the block fits the ring; returns the first slot of the block about to be issued, or None (then a0_mean_rows per update fills ``_loss_means``)."""
        ln = self.learner
        eng = ln.engine
        ring = getattr(eng, "loss_ring", None)
        ok = (ring is not None and eng.online.fused and hasattr(ln, "updates_issued") and self.cfg.learner.learner_steps <= ring.numel()
              and "train_batch" not in vars(ln) and os.environ.get("A0_LOSS_RING", "1") != "0")
        self._ring0 = ln.updates_issued if ok else None
        return self._ring0

    def _block_loss_means(self, n_upd: int):
        if getattr(self, "_ring0", None) is None:
            return self._loss_means[:n_upd].cpu().tolist()
        ring = self.learner.engine.loss_ring.cpu()
        return [float(ring[(self._ring0 + i) % ring.numel()]) for i in range(n_upd)]

    def _pipeline_candidate(self) -> bool:
        cfg, ln = self.cfg, self.learner
        # OFF by default: measured slower than the strictly serial block on MI355X (profiles/r04_experiments.md) — kept as a tested option
        return (os.environ.get("A0_PIPELINE_TARGET", "0") == "1" and cfg.replay.policy != ReplayEnum.prioritize and cfg.learner.learner_steps >= 2
                and bool(getattr(ln.engine, "target_stage_supported", False)) and ln.use_graph)

    def _pipeline_ok(self) -> bool:
        """Uniform replay (the next batch does not depend on this update's priorities), a learner whose target pass can run as a stage of its own, the
        hot-loop entry points not wrapped by a test harness, and no target sync inside this block: Adam's fused target copy at the end of update k would race
        with the target pass of batch k + 1 — such a block (one in target_update_freq / learner_steps) runs strictly in order."""
        cfg, ln = self.cfg, self.learner
        if not self._pipeline_candidate():
            return False
        if "sample" in vars(self.replay) or "train_batch" in vars(ln) or "update_priority" in vars(self.replay):      # instance-level wrappers (tests/test_gpu_trace.py)
            return False
        if getattr(self, "_updates_done", None) is None:
            self._updates_done = int(ln.engine.state[1])          # one small read per block; NaN-skipped steps do not count, so the device's number is the one to use
        c0, L, f = self._updates_done, int(cfg.learner.learner_steps), int(cfg.learner.target_update_freq)
        return (c0 + L) // f == c0 // f                           # no multiple of f in (c0, c0 + L]: no sync in this block, however many steps a NaN skips

    def _update_block_pipelined(self) -> int:
        """trainer.py:83-104's loop with the same work in the same per-update order, except that batch k + 1 is drawn and pushed through the TARGET network
        (encoder + fc1 GEMM, ~70 us for dqn at B = 512) on a second stream while update k runs: those kernels fill the CUs that update k's latency-bound
        launches (512-row GEMMs, head / loss, reductions, Adam) leave idle.  Two sets of batch and target-stage buffers alternate; update k + 1 waits for its
        stage through an event, the stage of batch k + 2 waits for update k (the last reader of the buffers it overwrites).  Numbers are unchanged."""
        cfg, rp, ln = self.cfg, self.replay, self.learner
        L, B = int(cfg.learner.learner_steps), cfg.learner.batch_size
        if self._loss_means.numel() < L:
            self._loss_means = self.ops.zeros(L)
        if getattr(self, "_tstream", None) is None:
            prio = os.environ.get("A0_TSTREAM_PRIO")
            self._tstream = torch.cuda.Stream() if prio is None else torch.cuda.Stream(priority=int(prio))
            self._tev = [torch.cuda.Event(), torch.cuda.Event()]
            self._mev = torch.cuda.Event()
        self.pipelined_blocks = getattr(self, "pipelined_blocks", 0) + 1
        cur = torch.cuda.current_stream()
        ring0 = self._loss_ring_start()
        batches = [None, None]
        batches[0] = rp.sample(buf=0)
        ln.target_stage_batch(rp.frames, batches[0].slot, rp.row_bytes, 0)          # the first batch's stage has nothing to hide behind: same stream
        for i in range(L):
            p = i & 1
            if i + 1 < L:
                q = p ^ 1
                self._mev.record(cur)                     # update i - 1 (last reader of parity q's buffers) is complete on the main stream here
                self._tstream.wait_event(self._mev)
                with torch.cuda.stream(self._tstream):
                    batches[q] = rp.sample(buf=q)
                    ln.target_stage_batch(rp.frames, batches[q].slot, rp.row_bytes, q)
                    self._tev[q].record(self._tstream)
            if i > 0:
                cur.wait_event(self._tev[p])
            b = batches[p]
            q_loss, _ = ln.train_batch(rp.frames, b.slot, rp.row_bytes, b.act, b.rew, b.done, b.weights, tstage=p)
            if ring0 is None:
                self.ops.mean_rows(q_loss, 1, B, self._loss_means[i:i + 1])
        return L

    def _result(self):
        """The result dict of trainer.py:111-118."""
        return dict(
            frames=self.frame_count,
            fraction_loss=np.mean(self.FLs[-20:]) if len(self.FLs) > 0 else None,
            loss=np.mean(self.Ls[-20:]) if len(self.Ls) > 0 else None,
            return_train=np.mean(self.Rs[-20:]) if len(self.Rs) > 0 else None,
            return_train_max=np.max(self.Rs) if len(self.Rs) > 0 else None,
            qmax=np.mean(self.Qs[-100:]) if len(self.Qs) > 0 else None,
        )

    # ------------------------------------------------------------------ trainer.py:121-156
    def test(self):
        cfg = self.cfg
        if self.actors[0] is None:
            self.actors[0] = agents.Actor(cfg, self.learner.model, replay=None, ops=self.ops, rank=self.rank + 1000)
        if self.use_lp:
            torch.cuda.synchronize()                # the test actor shares the learner's weights: no update may be in flight
        rs = []
        video = []
        self.logger.info("Testing ... ")
        self.actors[0].reset()
        guard = 0
        while len(rs) < cfg.trainer.test_episodes and guard < 200:
            images, returns, _ = self.actors[0].sample(cfg.actor.test_eps, test=True)
            rs.extend(returns)
            if len(video) < 3600:                   # trainer.py:131-132: the newest frame of the first four envs, per step
                video.extend(images)
            guard += 1
        self.RTs.extend(rs)
        # trainer.py:134-135: (n, t, c, h, w) uint8 with the grey channel repeated three times
        self.last_test_video = np.repeat(np.stack(video, axis=1), 3, axis=2) if video else None
        if rs:
            if self.writer is not None:
                self.writer.add_scalar("return_test", np.mean(rs), self.frame_count)
                self.writer.add_scalar("return_test_max", np.max(self.RTs), self.frame_count)
                if self.last_test_video is not None and hasattr(self.writer, "add_video"):
                    self.writer.add_video("test_video", self.last_test_video, self.frame_count, fps=60)
            if self._wandb is not None:
                self._wandb.log({"return_test": np.mean(rs), "frame": self.frame_count})
                self._wandb.log({"return_test_max": np.max(self.RTs), "frame": self.frame_count})
                if self.last_test_video is not None:
                    self._wandb.log({"test_video": self._wandb.Video(self.last_test_video, fps=60, format="mp4"), "frame": self.frame_count})
            self.logger.info(f"TEST ---> Frames: {self.frame_count} | Return Avg: {np.mean(rs):.2f} Max: {np.max(rs)}")
        return rs

    def logging(self, result):
        if not self.primary:
            return
        self._progress_row(result)
        msg = ""
        for k, v in result.items():
            if v is None:
                continue
            if self.writer is not None:
                self.writer.add_scalar(k, v, self.frame_count)
            if self._wandb is not None:
                self._wandb.log({k: v, "frame": self.frame_count})
            if k in ["frames", "loss", "qmax", "fps"] or "return" in k:
                msg += f"{k}: {v:.2f} | "
        self.logger.info(msg)

    def _progress_row(self, result):
        """One CSV row per logged iteration under the run directory (``progress.csv``, columns = the result keys of trainer.py:111-118 plus
        fps): the machine-readable twin of msg.log, aggregated across runs by agent0_amd/summary.py."""
        import csv

        try:
            path = os.path.join(self.cfg.logdir, "progress.csv")
            new = not os.path.exists(path)
            with open(path, "a", newline="") as f:
                w = csv.writer(f)
                if new:
                    w.writerow(PROGRESS_COLUMNS)
                w.writerow(["" if result.get(k) is None else result[k] for k in PROGRESS_COLUMNS])
        except OSError:
            pass

    def _issue_rollout(self):
        """``actor.futures.sample(eps, state_dict)`` (launch.py:34-36,58-62): snapshot the weights, start the rollout, do not wait."""
        actor = self.actors[1]
        eps = self.epsilon_fn(self.frame_count)
        cur = torch.cuda.current_stream()
        st = self.actor_stream if self.overlap else cur
        # snapshot on the learner's stream — ordered after every update enqueued so far and before the next one — then let the
        # actor stream start once the copy has landed
        actor.model._dev.copy_from(self.learner.model._dev)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            pending = actor.sample_async(eps)
        self.stage.written += self.num_transitions  # the next rollout goes to the other half of the stage
        return pending

    def run_iteration_lp(self):
        """One pass of the loop body of launch.py:44-63: collect the finished rollout, issue the next one with the current weights,
        then run the update block on the collected transitions while that rollout is in flight."""
        tic = time.time()
        if self._pending is None:
            self._pending = self._issue_rollout()   # launch.py:32-37 primes the pipeline before the loop
        transitions, returns, qmax = self.actors[1].sample_finish(self._pending)
        actor = self.actors[1]
        if self.overlap and (hasattr(actor.envs, "step_send") or hasattr(actor.envs, "pools")):
            # a HOST env's rollout occupies this thread until its last step (it waits for the worker processes), so the update block — which needs nothing from the
            # thread once enqueued — goes first and the rollout then runs on the actor stream beside it: launch.py's actor processes stepping their emulators while
            # the learner trains (launch.py:44-63).  Same dependencies as the order below (weights snapshot before the block's first Adam, epsilon from the frame
            # count before the commit, the rollout into the other half of the stage), hence the same numbers (tests/test_gpu_trainer.py).
            eps = self.epsilon_fn(self.frame_count)
            cur = torch.cuda.current_stream()
            actor.model._dev.copy_from(self.learner.model._dev)
            snap = torch.cuda.Event()
            snap.record(cur)
            self.Qs.extend(qmax)
            self.Rs.extend(returns)
            stats = self._step_device(transitions, defer=True)
            self.actor_stream.wait_event(snap)
            with torch.cuda.stream(self.actor_stream):
                self._pending = actor.sample_async(eps)
            self.stage.written += self.num_transitions
            torch.cuda.synchronize()
            self._block_stats_finish(stats)
            result = self._result()
            result.update(fps=self.num_transitions / (time.time() - tic))
            return result
        self._pending = self._issue_rollout()
        result = self.step(transitions, returns, qmax)
        torch.cuda.synchronize()
        result.update(fps=self.num_transitions / (time.time() - tic))
        return result

    def _native_loop(self):
        """The library's own handles over this Trainer's buffers (deepq/native_loop.py) when the configuration is one they cover, nothing has wrapped the hot-loop
        methods and no gradient hook is installed; decided at the first iteration, and final from then on (the handles own the actor's and the sampler's state)."""
        nl = getattr(self, "_nl", None)
        if nl is False:
            return None
        from . import native_loop
        ok_now = native_loop.hook_ok(self.learner.engine.grad_hook) and not any(native_loop._wrapped(o) for o in (self, self.replay, self.learner, self.actors[1]))
        if nl is None:
            why = native_loop.eligible(self)
            if why is not None or not ok_now or getattr(self, "_prefetched", None) is not None or self.replay.written != 0 or self.actors[1].steps != 0 or getattr(self, "_pending", None) is not None:
                self._nl = False
                self.native_loop_reason = why or "hot-loop methods wrapped, a gradient hook, or a run already under way"
                return None
            try:
                nl = self._nl = native_loop.NativeLoop(self)
            except RuntimeError as e:            # a create call refused the configuration: NativeLoop has destroyed what it had created; the Python classes run the loop
                self._nl = False
                self.native_loop_reason = f"handle creation failed: {e}"
                if self.primary:
                    self.logger.info("host loop: Python classes (%s)", self.native_loop_reason)
                return None
            if self.primary:
                self.logger.info("host loop: library handles over this Trainer's buffers (agent0_amd/deepq/native_loop.py); A0_NATIVE_LOOP=0 keeps the Python classes in charge")
        elif not ok_now:
            raise RuntimeError("Trainer: a gradient hook or a method wrapper was installed after the native loop had taken over the run")
        return nl

    def run_iteration(self, prefetch: bool = False):
        """One pass of the loop body of trainer.py:176-182; returns the result dict including fps.

        ``prefetch=True`` (``run()`` for every iteration but the last, bench.py's ``main`` schedule): the NEXT iteration's rollout is enqueued before the host
        waits for this iteration's statistics, which travel through page-locked buffers behind an event — same kernels in the same stream order with the same
        arguments (the next epsilon depends on the frame count only), so every number is the one the unpipelined loop produces
        (tests/test_gpu_trainer.py::test_prefetched_rollouts_change_no_number), but the GPU does not idle while Python collects statistics, logs and builds the
        next launch.  A rollout issued ahead is consumed by the next call (whatever its ``prefetch``), or booked into the replay by ``final()``."""
        nl = self._native_loop()
        if nl is not None:
            with self.ops.range("iteration"):             # roctx (A0_ROCTX=1): the handles open `rollout`, `sample`, `update`, `exchange` inside it
                return nl.run_iteration_lp() if self.use_lp else nl.run_iteration(prefetch)
        if self.use_lp:
            with self.ops.range("iteration"):
                return self.run_iteration_lp()
        with self.ops.range("iteration"):
            return self._run_iteration_py(prefetch)

    def _run_iteration_py(self, prefetch: bool):
        tic = time.time()
        # Same work in the same stream order as ``step(*actor.sample(eps))`` — the update block's kernels are ordered behind the rollout's — but
        # the host does not stop between them: the rollout's statistics (episode returns, per-step max-Q: one small read-back) are collected
        # after the update block has been enqueued instead of before, so the GPU never waits for Python at the rollout / update boundary.
        actor = self.actors[1]
        pending, self._prefetched = getattr(self, "_prefetched", None), None
        if pending is None:
            pending = actor.sample_async(self.epsilon_fn(self.frame_count))
        if not prefetch:
            self._step_device(actor.block_of(pending))
            _, returns, qmax = actor.sample_finish(pending)
        else:
            blk = self._step_device(actor.block_of(pending), defer=True)
            st = actor.stats_async(pending)                                     # ahead of the next rollout, which overwrites the statistics buffers
            self._prefetched = actor.sample_async(self.epsilon_fn(self.frame_count))
            returns, qmax = actor.stats_finish(st)
            self._block_stats_finish(blk)
        self.Qs.extend(qmax)
        self.Rs.extend(returns)
        result = self._result()
        if not prefetch:
            torch.cuda.synchronize()
        result.update(fps=self.num_transitions / (time.time() - tic))
        return result

    def run(self):
        cfg = self.cfg
        if cfg.mode.name in ("finetune", "play"):
            if not cfg.checkpoint:
                raise ValueError(f"mode={cfg.mode.name} needs checkpoint=<path>")
            self.load_checkpoint(cfg.checkpoint, weights_only=(cfg.mode.name == "play"))
        if cfg.mode.name == "play":
            return self.final(save=False)
        remaining = max(cfg.trainer.total_steps - self.frame_count, 0)
        trainer_steps = remaining // self.num_transitions + 1
        ahead = os.environ.get("A0_PREFETCH_ROLLOUT", "1") != "0"
        for i in range(trainer_steps):
            self.logging(self.run_iteration(prefetch=ahead and i + 1 < trainer_steps))
        self.final()

    def final(self, save: bool = True):
        if self.use_lp and self._pending is not None:
            self.actors[1].sample_finish(self._pending)      # drain the rollout still in flight
            self._pending = None
        if getattr(self, "_nl", None):
            self._nl.drain_lp() if self.use_lp else self._nl.drain()
        if getattr(self, "_prefetched", None) is not None:   # a rollout issued ahead by run_iteration(prefetch=True) and never consumed: book it, so that replay and counters agree with the device
            pending, self._prefetched = self._prefetched, None
            transitions, returns, qmax = self.actors[1].sample_finish(pending)
            self.replay.extend(transitions)
            self.frame_count += self.num_transitions
            self.Qs.extend(qmax)
            self.Rs.extend(returns)
        try:
            if self.primary:
                self.test()
        finally:
            if save and self.primary:        # after the final test, so that its returns ("ITRs") are part of the run's record — and even if it failed
                try:
                    self.save_checkpoint(os.path.join(self.cfg.logdir, "final.pth"))
                except OSError:
                    pass
        if getattr(self, "_nl", None):
            self._nl.close()
            self._nl = False
        for actor in self.actors:
            if actor is not None:
                actor.close()
