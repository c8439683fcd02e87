"""Data parallelism over the GPUs of one node: per-rank env/replay shards, one RCCL all-reduce per update.

No reference counterpart (the reference has no torch.distributed / NCCL call anywhere; its only inter-process traffic
is Launchpad's gRPC, launch.py:166-176).  Design: every rank owns its vectorized envs, its replay ring
and sum-tree and samples locally; the flat fp32 gradient buffer (6.7-14 MB) is SUM-reduced once per update over xGMI
with ``torch.distributed`` (backend "nccl" == RCCL on ROCm), together with the NaN-skip flag so replicas stay in
lock-step.  Because the reference reduces the loss by SUM (agent.py:154), a SUM all-reduce with Adam eps = 1e-2/(W*B)
is exactly the reference step on the W*B global batch.
"""
from __future__ import annotations

import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend: str | None = None):
    import torch.distributed as dist

    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradAllReduce:
    """``DeviceLearner.grad_hook``: SUM the flat gradient buffer and MAX the NaN flag across ranks."""

    def __init__(self, n_grad: int, group=None):
        import torch.distributed as dist

        self.dist, self.n, self.group = dist, n_grad, group
        self.world = dist.get_world_size(group)

    def __call__(self, grads: torch.Tensor, state: torch.Tensor):
        if self.world == 1:
            return
        self.dist.all_reduce(grads[: self.n], op=self.dist.ReduceOp.SUM, group=self.group)
        self.dist.all_reduce(state[0:1], op=self.dist.ReduceOp.MAX, group=self.group)
