"""Diagnostics: wall-time split of one Trainer iteration (actor rollout vs learner updates), host-only vs with GPU sync."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agent0_amd.deepq.config import parse_overrides
from agent0_amd.deepq.trainer import Trainer

algo = sys.argv[1] if len(sys.argv) > 1 else "dqn"
cfg = parse_overrides([f"learner.algo={algo}", "actor.num_envs=256", "replay.size=100000", "wandb=false", "tb=false", "logdir=gpurun_out/diag_logs"] + sys.argv[2:])
cfg.obs_shape = (4, 84, 84); cfg.action_dim = 4
cfg.trainer.training_start_steps = 20000
tr = Trainer(cfg)
for _ in range(4):
    tr.run_iteration()
torch.cuda.synchronize()
rp = tr.replay
for use_graph in (False, True, True):
    tr.learner.use_graph = use_graph
    ta = tl = th_a = th_l = 0.0
    n = 5
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.time()
        data, rs, qs = tr.actors[1].sample(0.1)
        t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
        rp.extend(data)
        for i in range(cfg.learner.learner_steps):
            b = rp.sample()
            tr.learner.train_batch(rp.frames, b.slot, rp.row_bytes, b.act, b.rew, b.done, b.weights)
        t3 = time.time(); torch.cuda.synchronize(); t4 = time.time()
        th_a += t1 - t0; ta += t2 - t0; th_l += t3 - t2; tl += t4 - t2
    print(f"graph={use_graph}: actor rollout {1e3*ta/n:.2f} ms (host issue {1e3*th_a/n:.2f}); {cfg.learner.learner_steps} updates {1e3*tl/n:.2f} ms (host issue {1e3*th_l/n:.2f})")
