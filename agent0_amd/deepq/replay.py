"""``ReplayDataset``: the reference's replay API over an HBM-resident ring buffer.

Mirrors /root/reference agent0/deepq/replay.py:14-59 — ``extend``, ``__getitem__``, ``__len__``, ``update_priority`` and
the attributes ``priority``, ``top``, ``beta``, ``max_p`` — but transitions never leave the GPU: frames are stored
uncompressed as the actor packs them (st || st_next, agent.py:78-81; 56 448 B each at 84x84, so the default 1 M-entry
buffer is 56.4 GB of the MI355X's 288 GB), and batches are produced by index kernels instead of a DataLoader
(trainer.py:63-72).

Sampling modes (``sample``):
  uniform                     the reference's RandomSampler semantics — a fresh pseudo-random permutation of range(top)
                              per epoch, batches of B, last batch never returned (utils.py:51-56) — as a Feistel bijection
  prioritize, sumtree=False   reference-faithful: uniform sampling, priorities only feed importance weights, new
                              priorities are written to the TAIL of the vector, sum over the whole capacity (Q1,Q2,Q7)
  prioritize, sumtree=True    proportional stratified sampling from the fp32 sum-tree (contract: oracle/sumtree.c),
                              importance weights (top * p/total)^-beta / max
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import os

import torch

from agent0_amd.common.utils import DeviceRng, LinearSchedule
from .config import ExpConfig, ReplayEnum


@dataclass
class TransitionBlock:
    """What ``Actor.sample`` hands to ``ReplayDataset.extend``: transitions already written into ring slots
    [start, start+count) by the actor (zero-copy), or staged in ``staged`` buffers when the actor has no replay bound."""
    count: int
    start: int = -1
    staged: Optional[dict] = None
    source: Optional["StageRing"] = None      # rows [start, start+count) of a StageRing (asynchronous actors, launch mode)

    def __len__(self):
        return self.count


@dataclass
class Batch:
    idx: torch.Tensor        # int64 [B] logical indices (what update_priority takes)
    slot: torch.Tensor       # int32 [B] ring slots
    act: torch.Tensor
    rew: torch.Tensor
    done: torch.Tensor
    prio: torch.Tensor
    weights: torch.Tensor


class StageRing:
    """A small ring in the replay's row format that an asynchronous actor writes while the learner samples the main ring.

    The reference's remote actors return their transitions and the trainer ``extend``s the replay between update blocks
    (launch.py:58-63), so a rollout in flight never touches entries the learner can sample.  Here the in-flight rollout lands in this
    stage (two rollouts deep, so the next one can start while the previous one is being committed) and ``ReplayDataset.extend`` moves
    the rows into the main ring with device-to-device copies."""

    def __init__(self, ops, rows: int, obs_bytes: int):
        self.size, self.obs_bytes, self.row_bytes = int(rows), obs_bytes, 2 * obs_bytes
        self.frames = torch.empty(self.size * self.row_bytes, dtype=torch.uint8, device=ops.device)
        self.act, self.rew, self.done = ops.zeros(self.size, dtype=torch.int32), ops.zeros(self.size), ops.zeros(self.size)
        self.written = 0

    def write_cursor(self) -> int:
        return self.written % self.size


class ReplayDataset:
    def __init__(self, cfg: ExpConfig, ops=None, rng: Optional[DeviceRng] = None):
        if ops is None:
            from agent0_amd.ops import HipOps
            ops = HipOps()
        self.cfg, self.ops = cfg, ops
        self.size = int(cfg.replay.size)
        C, H, W = (int(v) for v in cfg.obs_shape)
        self.obs_bytes = C * H * W
        self.row_bytes = 2 * self.obs_bytes
        if self.obs_bytes % 16:
            raise ValueError("observation byte size must be a multiple of 16")
        self.frames = torch.empty(self.size * self.row_bytes, dtype=torch.uint8, device=ops.device)
        self.act = ops.zeros(self.size, dtype=torch.int32)
        self.rew = ops.zeros(self.size)
        self.done = ops.zeros(self.size)
        self.priority = torch.ones(self.size, device=ops.device)
        self.top = 0
        self.written = 0                      # transitions ever written (ring cursor = written % size)
        self.prioritize = cfg.replay.policy == ReplayEnum.prioritize
        self.use_sumtree = self.prioritize and bool(cfg.replay.sumtree)
        self.rng = rng or DeviceRng(ops, cfg.seed + 104729)
        B = int(cfg.learner.batch_size)
        self.B = B
        self._idx = ops.zeros(B, dtype=torch.int64)
        self._slot = ops.zeros(B, dtype=torch.int32)
        self._act, self._rew, self._done = ops.zeros(B, dtype=torch.int32), ops.zeros(B), ops.zeros(B)
        self._prio, self._w = ops.zeros(B), torch.ones(B, device=ops.device)
        self._idx_out = ops.zeros(B, dtype=torch.int64)
        self._xi = ops.zeros(B)
        self._psum, self._scratch = ops.zeros(1), ops.zeros(256)
        self._pstate = torch.ones(1, device=ops.device)        # max_p on the device
        self._epoch = None
        if self.prioritize:
            self.beta_schedule = LinearSchedule(cfg.replay.beta0, 1.0, cfg.trainer.total_steps)
            self.beta = cfg.replay.beta0
        if self.use_sumtree:
            self.cap2 = 1
            while self.cap2 < self.size:
                self.cap2 <<= 1
            self._tree = ops.zeros(2 * self.cap2)
            # update_priority leaves the levels with < 2048 nodes to the next batch's launch (a0_sumtree_set_from_loss(defer_top) / a0_sumtree_sample_batch(rebuild_top)):
            # `tree` — what every other reader goes through — brings them up to date first
            self._top_stale = False
            self._defer_top = os.environ.get("A0_SUMTREE_DEFER_TOP", "1") != "0"
            self._new_idx = None
            self._val = ops.zeros(max(B, 1))

    # ------------------------------------------------------------------ reference attributes
    @property
    def max_p(self) -> float:
        return float(self._pstate[0])

    def __len__(self):
        return self.top

    @property
    def head(self) -> int:
        return self.written % self.size if self.written > self.size else 0

    def write_cursor(self) -> int:
        return self.written % self.size

    # ------------------------------------------------------------------ insert
    def _stage_tuples(self, transitions) -> TransitionBlock:
        """The reference's transition list (agent.py:78-81: ``(compress(concatenate((st, st_next))), at, rt, dt)`` per env and step) -> staged
        device buffers.  The frame blob may be the raw ``st || st_next`` bytes (bytes / bytearray / memoryview / uint8 array of row_bytes
        elements) or, when its length differs from row_bytes, an lz4 block as the reference stores it (needs the ``lz4`` module)."""
        n = len(transitions)
        rows = np.empty((n, self.row_bytes), dtype=np.uint8)
        act, rew, done = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float32), np.empty(n, dtype=np.float32)
        for i, tr in enumerate(transitions):
            if not isinstance(tr, (tuple, list)) or len(tr) != 4:
                raise TypeError(f"ReplayDataset.extend: item {i} is not a (frames, action, reward, done) tuple")
            blob, at, rt, dt = tr
            if isinstance(blob, np.ndarray):
                flat = np.ascontiguousarray(blob, dtype=np.uint8).reshape(-1)
            else:
                raw = bytes(blob)
                if len(raw) != self.row_bytes:
                    try:
                        from lz4.block import decompress
                    except ImportError as e:
                        raise ValueError(f"ReplayDataset.extend: item {i} holds {len(raw)} bytes, a raw row is {self.row_bytes}; compressed rows need the lz4 module") from e
                    raw = decompress(raw)
                flat = np.frombuffer(raw, dtype=np.uint8)
            if flat.size != self.row_bytes:
                raise ValueError(f"ReplayDataset.extend: item {i} has {flat.size} frame bytes, expected {self.row_bytes} (st || st_next)")
            rows[i] = flat
            act[i], rew[i], done[i] = int(at), float(rt), float(bool(dt))      # the reference's float64 n-step sums are rounded to fp32 once
        dev = self.ops.device
        t = torch.from_numpy(rows).to(dev)
        return TransitionBlock(n, staged={"obs": t[:, : self.obs_bytes].contiguous(), "obs_next": t[:, self.obs_bytes:].contiguous(),
                                          "act": torch.from_numpy(act).to(dev), "rew": torch.from_numpy(rew).to(dev), "done": torch.from_numpy(done).to(dev)})

    def extend(self, transitions):
        """replay.py:45-53.  ``transitions``: the TransitionBlock the device Actor returns (rows already in HBM; the hot path), or the
        reference's own list of ``(frames, at, rt, dt)`` tuples — what its Actor.sample / TrainerNode hand over (agent.py:78-81,
        launch.py:49-62) — which is staged to the device and inserted with a0_replay_insert."""
        if isinstance(transitions, (list, tuple)):
            if len(transitions) == 0:
                return
            transitions = self._stage_tuples(transitions)
        if not isinstance(transitions, TransitionBlock):
            raise TypeError("ReplayDataset.extend takes the TransitionBlock returned by Actor.sample or the reference's list of (frames, action, reward, done) tuples")
        n = transitions.count
        if transitions.source is not None:
            src, done_rows = transitions.source, 0
            assert src.row_bytes == self.row_bytes and transitions.start + n <= src.size and n <= self.size
            dst_f, src_f = self.frames.view(self.size, self.row_bytes), src.frames.view(src.size, src.row_bytes)
            while done_rows < n:                      # at most two pieces: the main ring may wrap
                c = self.write_cursor()
                k = min(n - done_rows, self.size - c)
                a = transitions.start + done_rows
                dst_f[c:c + k].copy_(src_f[a:a + k])
                self.act[c:c + k].copy_(src.act[a:a + k]); self.rew[c:c + k].copy_(src.rew[a:a + k]); self.done[c:c + k].copy_(src.done[a:a + k])
                self.written += k
                done_rows += k
        elif transitions.staged is not None:
            s = transitions.staged
            done_rows = 0
            while done_rows < n:                      # staged blocks may be larger than the ring
                k = min(n - done_rows, self.size)
                sl = slice(done_rows, done_rows + k)
                self.ops.replay_insert(self.frames, self.size, self.obs_bytes, self.write_cursor(), k, s["obs"][sl].reshape(-1), s["obs_next"][sl].reshape(-1),
                                       s["act"][sl].contiguous(), s["rew"][sl].contiguous(), s["done"][sl].contiguous(), self.act, self.rew, self.done)
                self.written += k
                done_rows += k
        else:
            assert transitions.start == self.write_cursor(), "actor wrote to a stale ring position"
            self.written += n
        self.top = min(self.top + n, self.size)
        if self.prioritize:
            if self.use_sumtree:
                val = (self._pstate[0:1].double() ** self.cfg.replay.alpha).float()
                k = min(n, self.size)                     # one launch: the new leaves are a ring range
                self.ops.sumtree_set_range(self._tree, self.cap2, (self.written - k) % self.size, k, self.size, val)
                self._top_stale = False          # the range kernel recomputes every level above 2048 nodes from that level
            else:
                self.ops.priority_tail(self.priority, self.size, min(n, self.size), self._pstate, float(self.cfg.replay.alpha))
            self.beta = self.beta_schedule(n)

    # ------------------------------------------------------------------ host-side item access (API parity, slow path)
    def __getitem__(self, idx: int):
        idx = idx % self.top
        slot = (self.head + idx) % self.size
        row = self.frames[slot * self.row_bytes:(slot + 1) * self.row_bytes].cpu().numpy()
        pr = self.tree[self.cap2 + slot] if self.use_sumtree else self.priority[idx]
        return row, int(self.act[slot]), float(self.rew[slot]), bool(self.done[slot] != 0), pr.cpu(), idx

    # ------------------------------------------------------------------ priorities
    _tree = None
    _top_stale = False

    @property
    def tree(self) -> torch.Tensor:
        """The sum-tree [2 * cap2] with every level up to date."""
        if self._top_stale:
            self.ops.sumtree_top_rebuild(self._tree, self.cap2)
            self._top_stale = False
        return self._tree

    @tree.setter
    def tree(self, t: torch.Tensor):
        self._tree, self._top_stale = t, False

    def update_priority(self, ids: torch.Tensor, priorities: torch.Tensor, state: Optional[torch.Tensor] = None):
        """replay.py:55-59: priority[ids] = (loss + eps)^alpha; max_p = max(max_p, max loss).  ``state``: the learner's status words — when
        the update these losses come from was skipped on a NaN (agent.py:152-158 returns None and trainer.py:103 then skips the call), the
        device-side guard leaves priorities, sum-tree and max_p untouched on both paths."""
        rc = self.cfg.replay
        B = ids.numel()
        ids = ids.to(self.ops.device, torch.int64).contiguous()
        pr = priorities.to(self.ops.device, torch.float32).contiguous()
        if self.use_sumtree and B <= 1024 and self.ops.sumtree_set_from_loss_ok(self.cap2):
            # priorities formed inside the per-subtree kernel: two launches (subtrees, top) instead of three
            self.ops.sumtree_set_from_loss(self._tree, self.cap2, ids, pr, B, float(rc.eps), float(rc.alpha), self._pstate, state, defer_top=self._defer_top)
            self._top_stale = self._defer_top
        elif self.use_sumtree:
            if self._val.numel() < B:
                self._val = self.ops.zeros(B)
            self.ops.priority_from_loss(pr, B, float(rc.eps), float(rc.alpha), self._val, self._pstate, state)
            for o in range(0, B, 1024):       # one workgroup's worth of leaves per call, in batch order: a later duplicate still wins
                k = min(1024, B - o)
                self.ops.sumtree_set(self.tree, self.cap2, ids[o:o + k], self._val[o:o + k], k, state)
        else:
            self.ops.priority_update(self.priority, ids, pr, B, float(rc.eps), float(rc.alpha), self._pstate, state)

    # ------------------------------------------------------------------ sampling
    def _next_uniform(self, B: int, draw: bool = True):
        """DataLoader(shuffle=True) + DataPrefetcher semantics (trainer.py:63-72, utils.py:31-56).  Returns (start, n_perm, seed) of the
        batch; ``draw`` also materialises the permutation elements in ``_idx``."""
        ep = self._epoch
        if ep is None or ep["pos"] + 1 >= ep["nb"]:
            top = self.top
            ep = {"top": top, "nb": (top + B - 1) // B, "pos": 0, "seed": self.rng.next_seed32(self.rng.STREAM_PERM)}
            if ep["nb"] < 2:
                raise RuntimeError("replay holds fewer than two batches: the reference's fetcher cannot return one either")
            self._epoch = ep
        start = ep["pos"] * B
        ep["pos"] += 1
        if draw:
            self.ops.perm_batch(start, B, ep["top"], ep["seed"], self._idx)
        return start, ep["top"], ep["seed"]

    def sample_gathered(self, out_rows: torch.Tensor) -> Batch:
        """``sample()`` plus a dense copy of the sampled rows into ``out_rows`` [B * row_bytes] in ONE launch — the device counterpart of a
        DataLoader batch (trainer.py:63-72, replay.py:32-37).  The learner does not need the copy (conv1 reads ring rows through the slot
        index); this is the API for callers that want the batch materialised, and what bench.py times as 'replay sample GB/s'."""
        B = self.B
        if self.use_sumtree:
            self.rng.uniform(self.rng.STREAM_SUMTREE, self._xi, B)
            self.ops.replay_sample_gather(1, 0, 0, 0, self.tree, self.cap2, self._xi, self.size, 0, self.size, self.frames, self.row_bytes, self.act, self.rew, self.done,
                                          None, B, out_rows, self._idx_out, self._slot, self._act, self._rew, self._done, self._prio)
            self.ops.is_weights(self._prio, B, self.tree[1:2], self.top, float(self.beta), self._w)
        else:
            ep = self._epoch
            if ep is None or ep["pos"] + 1 >= ep["nb"]:
                self._next_uniform(B)           # opens a new epoch (and draws its first batch into _idx, unused here)
                ep = self._epoch
                ep["pos"] = 0
            start = ep["pos"] * B
            ep["pos"] += 1
            self.ops.replay_sample_gather(0, start, ep["top"], ep["seed"], None, 1, None, self.top, self.head, self.size, self.frames, self.row_bytes, self.act, self.rew,
                                          self.done, self.priority if self.prioritize else None, B, out_rows, self._idx_out, self._slot, self._act, self._rew, self._done,
                                          self._prio)
            if self.prioritize:
                self.ops.sum_f32(self.priority, self.size, self._scratch, self._psum)
                self.ops.is_weights(self._prio, B, self._psum, self.top, float(self.beta), self._w)
        return Batch(self._idx_out, self._slot, self._act, self._rew, self._done, self._prio, self._w)

    def sample(self, B: Optional[int] = None, buf: int = 0) -> Batch:
        """``buf=1``: the batch goes to a second set of device buffers (uniform replay only) — the Trainer's pipelined update block draws batch k + 1 and runs
        the target network on it while update k, which still reads batch k's slots, is in flight."""
        B = B or self.B
        assert B == self.B
        if buf:
            if self.prioritize:
                raise ValueError("a second batch buffer exists for uniform replay only (prioritized sampling depends on the previous update's priorities)")
            if getattr(self, "_alt", None) is None:
                ops = self.ops
                self._alt = (ops.zeros(B, dtype=torch.int64), ops.zeros(B, dtype=torch.int32), ops.zeros(B, dtype=torch.int32), ops.zeros(B), ops.zeros(B), ops.zeros(B))
            io, sl, ac, rw, dn, pr = self._alt
            start, n_perm, seed = self._next_uniform(B, draw=False)
            self.ops.replay_sample_slots(start, n_perm, seed, self.top, self.head, self.size, self.act, self.rew, self.done, None, B, io, sl, ac, rw, dn, pr)
            return Batch(io, sl, ac, rw, dn, pr, self._w)
        if self.use_sumtree and B <= 1024:
            # stratified draws, descent, slot + metadata and importance weights in one launch
            rng = self.rng
            stale, self._top_stale = self._top_stale, False
            self.ops.sumtree_sample_batch(rng.seed, rng.STREAM_SUMTREE, rng.reserve(rng.STREAM_SUMTREE, B), self._tree, self.cap2, B, self.top, self.size, float(self.beta),
                                          self.act, self.rew, self.done, self._idx, self._slot, self._act, self._rew, self._done, self._prio, self._w, rebuild_top=stale)
            idx = self._idx
        elif self.use_sumtree:
            self.rng.uniform(self.rng.STREAM_SUMTREE, self._xi, B)
            self.ops.sumtree_sample(self.tree, self.cap2, self._xi, B, self._idx, self._prio)
            # sum-tree leaves are addressed by ring slot: logical index == slot, no deque shift (head = 0)
            self.ops.replay_lookup(self._idx, B, self.size, 0, self.size, self._slot, self.act, self.rew, self.done, None,
                                   self._act, self._rew, self._done, None, self._idx_out)
            self.ops.is_weights(self._prio, B, self.tree[1:2], self.top, float(self.beta), self._w)
            idx = self._idx
        else:
            start, n_perm, seed = self._next_uniform(B, draw=False)
            # permutation element -> slot + metadata in one launch
            self.ops.replay_sample_slots(start, n_perm, seed, self.top, self.head, self.size, self.act, self.rew, self.done,
                                         self.priority if self.prioritize else None, B, self._idx_out, self._slot, self._act, self._rew, self._done, self._prio)
            if self.prioritize:
                self.ops.sum_f32(self.priority, self.size, self._scratch, self._psum)
                self.ops.is_weights(self._prio, B, self._psum, self.top, float(self.beta), self._w)
            idx = self._idx_out
        return Batch(idx, self._slot, self._act, self._rew, self._done, self._prio, self._w)
