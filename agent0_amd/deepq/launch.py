"""``python -m agent0.deepq.launch [key=value ...]`` — multi-process entry point.

The reference (agent0/deepq/launch.py:25-205) starts ``num_actors`` Launchpad CourierNodes plus one TrainerNode that
ships the whole state_dict to an actor on every sample RPC and receives pickled lz4 transitions back (launch.py:30-97).
dm-launchpad is not available (its wheel is a missing blob in the reference checkout) and that transport is the
bottleneck this build removes: here one process per GPU runs actor + replay shard + learner on its own device and the
replicas exchange only gradients (agent0_amd/deepq/dist.py).  What the reference gets from Launchpad — rollouts running WHILE the
learner updates, with the weights as they were when the rollout was issued (launch.py:34-36,44-63) — is kept: the actor owns a
device-to-device snapshot of the network and rolls out on a second HIP stream into a stage ring (Trainer(use_lp=True)).  ``num_actors`` maps to the number of ranks, capped by the
GPUs present.  When started without a torch.distributed environment this module re-launches itself under
``torch.distributed.run`` — as a CHILD process and before anything touches the GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

from .dist import env_world, init_process_group, make_grad_hook


class TrainerNode:
    """One data-parallel replica: reference TrainerNode + its ActorNodes collapsed onto one GPU."""

    def __init__(self, cfg, rank: int, world: int):
        from .trainer import Trainer

        cfg.seed = cfg.seed + 1000003 * rank          # per-rank env / replay / exploration streams
        self.trainer = Trainer(cfg, use_lp=True, rank=rank, primary=(rank == 0))      # asynchronous actor on its own stream, weight snapshots
        eng = self.trainer.learner.engine
        if world > 1:
            eng.grad_hook = make_grad_hook(self.trainer.ops, eng.L.n_adam)
            eng.adam_eps = 1e-2 / (world * cfg.learner.batch_size)       # SUM-reduced gradients == one step on the global batch
            # identical initial replicas: broadcast rank 0's parameters
            import torch.distributed as dist
            dist.broadcast(eng.online.flat, src=0)
            eng.online.refresh_wt()
            eng.sync_target(force=True)

    def run(self):
        self.trainer.run()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if "WORLD_SIZE" not in os.environ:
        import torch
        from .config import parse_overrides

        n_gpu = torch.cuda.device_count()             # does not initialise the GPU
        world = max(1, min(parse_overrides(argv).num_actors, n_gpu))
        if world > 1:
            from .dist import free_port
            from .main import _fresh_subdir

            # one rendezvous port per job (two jobs on one box must not collide) and ONE run directory for all ranks
            env = dict(os.environ, A0_RUN_SUBDIR=_fresh_subdir(parse_overrides(argv)))
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                   "--master-port", os.environ.get("MASTER_PORT") or str(free_port()), "-m", "agent0_amd.deepq.launch", *argv]
            raise SystemExit(subprocess.call(cmd, env=env))
    from .main import _fresh_subdir, build_config
    from .config import parse_overrides

    rank, local_rank, world = init_process_group()
    subdir = os.environ.get("A0_RUN_SUBDIR")
    if subdir is None and world > 1:          # started under torch.distributed.run directly: rank 0 names the run directory
        import torch.distributed as dist

        box = [_fresh_subdir(parse_overrides(argv)) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        subdir = box[0]
    cfg = build_config(argv, subdir)
    TrainerNode(cfg, rank, world).run()


if __name__ == "__main__":
    main()
