"""PCIe-inclusive throughput of the host-environment front-end (env_pool.HostEnvPool) on BASELINE configs[1]'s shapes: 256 envs x 80 steps
per rollout, Breakout dqn, observations stepped on the host by worker processes, uploaded through the page-locked ring.  Prints one JSON
line: actor-only env-frames/s, the full iteration (rollout + 20 updates of batch 512) and the bytes that cross PCIe per step.
usage: python tools/bench_host_env.py [workers] [iterations] [device_frame_stack 1|0] [groups] [main|launch]      (groups >= 2: env_pool.HostEnvGroups — the CPU steps one group
while the GPU infers the other)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from agent0_amd.common.env_pool import HostEnvGroups, HostEnvPool, HostSynthSlice
from agent0_amd.deepq import agent as agents
from agent0_amd.deepq.config import parse_overrides
from agent0_amd.deepq.trainer import Trainer

def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    newest = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    groups = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    cfg = parse_overrides(["env_id=Breakout", "actor.num_envs=256", "replay.size=100000", "learner.batch_size=512", "wandb=false", "tb=false",
                           f"logdir={os.path.join(ROOT, 'gpurun_out', 'bench_logs')}"])
    cfg.obs_shape, cfg.action_dim = (4, 84, 84), 4
    launch = len(sys.argv) > 5 and sys.argv[5] == "launch"          # the launch.py schedule: the update block of rollout k beside rollout k + 1 (Trainer(use_lp=True))
    tr = Trainer(cfg, use_lp=launch)
    pool = (HostEnvGroups(HostSynthSlice(cfg.seed), 256, groups=groups, num_workers=workers, ops=tr.ops, newest_frame=newest) if groups > 1 else
            HostEnvPool(HostSynthSlice(cfg.seed), 256, num_workers=workers, ops=tr.ops, newest_frame=newest))
    tr.actors[1].close()
    tr.actors[1] = agents.Actor(cfg, None if launch else tr.learner.model, replay=tr.stage if launch else tr.replay, ops=tr.ops, rank=0, envs=pool)
    start = cfg.trainer.training_start_steps
    cfg.trainer.training_start_steps = 1 << 62
    tr.run_iteration()
    torch.cuda.synchronize()
    waited = lambda: sum(p.wait_s for p in getattr(pool, "pools", [pool]))
    t0, whole0, w0 = time.time(), pool.full_uploads, waited()
    for _ in range(iters):
        tr.run_iteration()
    torch.cuda.synchronize()
    t_act = (time.time() - t0) / iters
    wait_us = 1e6 * (waited() - w0) / (iters * cfg.actor.sample_steps)      # host time per step spent waiting for the workers: the env's own cost as the step path sees it
    whole = (pool.full_uploads - whole0) / (iters * cfg.actor.sample_steps)
    per_step = pool.pcie_bytes_per_step + whole * (pool.obs_bytes if groups == 1 else pool.pools[0].obs_bytes)
    cfg.trainer.training_start_steps = 1000
    for _ in range(3):
        tr.run_iteration()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        tr.run_iteration()
    torch.cuda.synchronize()
    t_full = (time.time() - t0) / iters
    n = cfg.actor.sample_steps * cfg.actor.num_envs
    print(json.dumps({"front_end": "HostEnvPool + HostSynthSlice (host synthetic env, numpy)", "workers": workers, "groups": groups, "schedule": "launch" if launch else "main", "host_cores": os.cpu_count(),
                      "device_frame_stack": pool.newest_frame, "whole_stack_uploads_per_step": round(whole, 3),
                      "actor_only_env_frames_per_sec": round(n / t_act, 1), "actor_only_ms_per_rollout": round(1e3 * t_act, 2),
                      "actor_only_us_per_step": round(1e6 * t_act / cfg.actor.sample_steps, 1), "of_which_waiting_for_the_workers_us": round(wait_us, 1),
                      "library_calls": bool(getattr(pool, "pools", [pool])[0].library_calls), "rollout": "host order" if os.environ.get("A0_HOST_ROLLOUT", "1") != "0" and groups == 1 else "step by step",
                      "iteration_env_frames_per_sec": round(n / t_full, 1), "iteration_ms": round(1e3 * t_full, 2),
                      "pcie_bytes_per_step": round(per_step), "pcie_GBps_at_actor_rate": round(per_step * cfg.actor.sample_steps / t_act / 1e9, 2)}))
    pool.close()


if __name__ == "__main__":      # worker processes are spawned: they re-import this file and must not run the benchmark themselves
    main()
