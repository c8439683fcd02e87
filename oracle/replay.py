"""Oracle: reference-faithful replay semantics + importance weights (numpy).

Restates /root/reference agent0/deepq/replay.py:14-59 (ReplayDataset) and the
importance-weight block of agent0/deepq/trainer.py:91-96, *including* the
quirks the reference has (SURVEY.md quirk ledger Q1, Q2, Q6, Q7):
  * ``priority.roll()`` result discarded, new priorities written to the TAIL
    ``priority[-n:]`` rather than the slots just filled (replay.py:51-52);
  * ``priority.sum()`` spans the whole capacity vector (trainer.py:92);
  * priority = per-sample loss (trainer.py:104).
Storage is a ring: ``deque(maxlen=size)`` index i == ring slot (head + i) % size.
"""
from __future__ import annotations

import numpy as np
import torch

from .schedules import LinearSchedule


class ReferenceReplay:
    def __init__(self, size: int, prioritize: bool, beta0=0.4, alpha=0.5, eps=0.01, total_steps=int(1e7)):
        self.size, self.prioritize = size, prioritize
        self.alpha, self.eps = alpha, eps
        self.slots = [None] * size  # ring storage
        self.head = 0  # ring slot of deque index 0
        self.count = 0  # live entries (== len(deque))
        self.priority = np.ones(size, dtype=np.float32)
        self.top = 0
        if prioritize:
            self.beta_schedule = LinearSchedule(beta0, 1.0, total_steps)
            self.beta = beta0
            self.max_p = 1.0

    def __len__(self):
        return self.top

    def slot_of(self, idx: int) -> int:
        return (self.head + idx) % self.size

    def extend(self, transitions):
        for t in transitions:
            if self.count < self.size:
                self.slots[(self.head + self.count) % self.size] = t
                self.count += 1
            else:  # deque(maxlen) drops the oldest: overwrite head, advance
                self.slots[self.head] = t
                self.head = (self.head + 1) % self.size
        n = len(transitions)
        self.top = min(self.top + n, self.size)
        if self.prioritize:
            self.priority[-n:] = np.float32(self.max_p ** self.alpha)
            self.beta = self.beta_schedule(n)

    def __getitem__(self, idx: int):
        idx = idx % self.top
        payload, at, rt, dt = self.slots[self.slot_of(idx)]
        return payload, at, rt, dt, self.priority[idx], idx

    def update_priority(self, ids, losses):
        losses = np.asarray(losses, dtype=np.float32)
        # torch's fp32 pow (it lowers alpha == 0.5 to a correctly rounded sqrt; numpy's powf differs by an ulp)
        self.priority[np.asarray(ids)] = torch.from_numpy(losses + np.float32(self.eps)).pow(self.alpha).numpy()
        self.max_p = max(float(losses.max()), self.max_p)


def is_weights(priorities: np.ndarray, priority_sum: float, top: int, beta: float) -> np.ndarray:
    """trainer.py:91-94: w = (top * p/sum)^(-beta); w /= (max w + 1e-8).  fp32."""
    p = np.asarray(priorities, dtype=np.float32)
    probs = torch.from_numpy(p) / priority_sum
    w = (top * probs).pow(-beta)
    return (w / w.max().add(1e-8)).numpy()


class UniformPermutationSampler:
    """DataLoader(shuffle=True) + DataPrefetcher semantics (trainer.py:63-72,
    utils.py:31-56): batches of B from a permutation of range(top) frozen at
    fetcher creation; the final batch is never returned — because preload()
    raises before next() hands out the batch it already holds, the last FULL
    batch is lost too when no partial batch follows it."""

    def __init__(self, top: int, batch: int, perm: np.ndarray):
        assert perm.shape[0] == top
        nb = (top + batch - 1) // batch  # DataLoader drop_last=False
        self.batches = [perm[i * batch:(i + 1) * batch] for i in range(nb)]
        self.pos = 0  # next_data = batches[0] preloaded at construction

    def next(self):
        # data = self.next_data; self.preload() -> raises StopIteration when exhausted
        if self.pos + 1 >= len(self.batches):
            raise StopIteration
        out = self.batches[self.pos]
        self.pos += 1
        return out
