"""Oracle: schedules on the hot path.

LinearSchedule restates /root/reference agent0/common/utils.py:12-28 (note: a
call returns the value *before* the increment).  epsilon restates the lambda at
agent0/deepq/trainer.py:46-50 (so eps(0) = 1 + min_eps, quirk Q15).
"""
from __future__ import annotations


class LinearSchedule:
    def __init__(self, start, end=None, steps=None):
        if end is None:
            end, steps = start, 1
        self.inc = (end - start) / float(steps)
        self.current = start
        self.end = end
        self.rising = end > start

    def __call__(self, steps=1):
        val = self.current
        nxt = self.current + self.inc * steps
        self.current = min(nxt, self.end) if self.rising else max(nxt, self.end)
        return val


def epsilon(step, exploration_steps=int(1e6), min_eps=0.01):
    if step > exploration_steps:
        return min_eps
    return (1.0 - step / exploration_steps) + min_eps
