#!/usr/bin/env python3
"""Headline benchmark: learner FPS of the deepq actor->replay->learner loop (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" is one Trainer iteration exactly as the reference times it (agent0/deepq/trainer.py:176-181):
``sample_steps`` (80) vectorized env steps on ``num_envs`` (256) envs — Q-network forward, epsilon-greedy, env step,
n-step, replay insert — followed by ``learner_steps`` (20) updates of batch 512 drawn from a FULL 1 M-transition HBM
replay (BASELINE.json configs[1]: Breakout dqn, 256 envs, 1 M replay, batch 512).  value = agent steps (transitions
written to replay) per second over all ranks == the reference's "FPS" (README.md:21-30, pre-frameskip); x4 for emulator
frames.  Inputs are synthetic (device-resident env of the right shape, random-init network); nothing is skipped inside
the timed region (forward, loss, backward, Adam, target sync, replay insert/sample, priority bookkeeping all run).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--algo", default="dqn")
    ap.add_argument("--num-envs", type=int, default=256)
    ap.add_argument("--replay-size", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--learner-steps", type=int, default=20)
    ap.add_argument("--sample-steps", type=int, default=80)
    ap.add_argument("--env", default="Breakout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ratio320", action="store_true", help="skip the extra learner_steps=320 measurement")
    ap.add_argument("--no-other-entry", action="store_true", help="skip the extra measurement of the other entry point's schedule")
    ap.add_argument("--entry", choices=("main", "launch"), default="main",
                    help="main: agent0.deepq.main schedule (rollout, then update block, strictly alternating); launch: agent0.deepq.launch schedule "
                         "(next rollout with a weight snapshot on a second stream while the update block runs)")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1 without a gradient exchange: every rank trains its own network on its own game (rank r plays the r-th of the 8 README games; "
                         "BASELINE configs[4] read as 'one game per GPU') — the ranks share only the barriers and the max-over-ranks clock")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the ranks as a child torch.distributed.run even for --gpus 1 (what --gpus N > 1 does when no launcher environment is present)")
    ap.add_argument("overrides", nargs="*", help="extra key=value config overrides")
    return ap.parse_args()


def self_launch(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start ``torch.distributed.run`` as a CHILD process (before anything in this process
    touches the GPU; never an exec), one rank per GPU, relay rank 0's JSON line and the child's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--self-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 or line is not None else 1


def host_cores():
    """Cores this process may actually use: the scheduler affinity mask, further limited by a cgroup CPU quota when one is set (a GPU box hands
    the container of one GPU a share of the host's cores; ``os.cpu_count()`` reports the whole machine)."""
    n_aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    return {"os_cpu_count": os.cpu_count(), "sched_affinity": n_aff, "cgroup_quota": quota, "used": min(n_aff, quota) if quota else n_aff}


def cpu_baseline(args, cfg):
    """SURVEY.md §8(d) "CPU baseline beside it": the CPU oracle (a port of the reference's algorithm, pinned by fixtures generated from the
    reference) timed on this box's host cores, on a bounded sample of the same workload, outside the timed region.  torch's intra-op threads =
    the cores this process may use (``host_cores``), not ``os.cpu_count()``.  Items:
      (i)   one ``learner.train`` at B = 512 per algorithm (dqn, c51, qr at the bench's action count; iqn, fqf at Asterix's A = 9 as in BASELINE configs[3], [4]);
      (ii)  the actor's Q-forward + argmax at E = 16 and E = 256;
      (iii) the reference's replay (oracle.replay.ReferenceReplay): extend, uniform permutation sample of B = 512 rows, importance weights — bytes/s of 56 448-byte rows;
      (iv)  one whole Trainer iteration at BASELINE configs[0]'s sizes (16 envs x 80 steps + 20 updates of B = 512, synthetic env on the CPU).
      (v)   round 6: one whole Trainer iteration at the BENCH workload's sizes (E envs x sample_steps + learner_steps updates of B), measured end to end.
    ``value`` = (v) for scalar / distributional heads; for the quantile networks (an iteration would take minutes) the iteration composed from (i) and (ii): sample_steps x
    act(E) + learner_steps x update(algo), which leaves out env stepping and replay.  lz4 and the data-loader processes the reference also pays are never included."""
    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import recipe
    from oracle import learner as olearner, nets, replay as oreplay
    from oracle.losses import Hyper
    from oracle.trainer import OracleTrainer

    cores = host_cores()
    keep_threads = torch.get_num_threads()
    torch.set_num_threads(cores["used"])
    B, E = args.batch, args.num_envs
    algo0 = cfg.learner.algo.name

    def spec_of(algo):
        A = 9 if algo in ("iqn", "fqf") and algo != algo0 else cfg.action_dim
        duel = cfg.learner.dueling_head if algo == algo0 else False
        return recipe.NetSpec(algo, A, dueling=duel, noisy=False, num_atoms=51 if algo == "c51" else 200)

    def time_update(algo, budget_s):
        spec = spec_of(algo)
        sd = recipe.make_state_dict(spec, 1)
        hp = Hyper(double_q=cfg.learner.double_q, n_step=cfg.learner.n_step_q) if algo == algo0 else Hyper()
        ora = olearner.OracleLearner(spec, sd, sd, hp, batch_size=B)
        frames = recipe.make_frames(B, 3, spec.obs_shape)
        a, r, d, w = recipe.make_transitions(B, spec.action_dim, 4)
        rand = None
        if algo == "iqn":
            g = recipe.gen(5)
            rand = [g.random((B, n, 1), dtype=np.float32) for n in (hp.K, hp.N_dash, hp.N)]
        run = lambda: ora.train(frames.reshape(B, -1), a, r, d.astype(np.float32), w, np.arange(B), rand=rand)
        heavy = algo in ("iqn", "fqf")           # ~0.5 TFLOP per update: one untimed-warm-up-free run is the whole budget
        if not heavy:
            run()
        n, t0 = 0, time.time()
        while n < (1 if heavy else 3) or (not heavy and time.time() - t0 < budget_s):
            run()
            n += 1
        return {"ms": round(1e3 * (time.time() - t0) / n, 1), "runs": n, "action_dim": spec.action_dim, "warm": not heavy}, ora, spec

    items = {}
    upd = {}
    ora0 = spec0 = None
    for algo in dict.fromkeys([algo0, "dqn", "c51", "qr", "iqn", "fqf"]):
        if algo == "mdqn" and algo != algo0:
            continue
        upd[algo], ora, spec = time_update(algo, 4.0 if algo == algo0 else 2.0)
        if algo == algo0:
            ora0, spec0 = ora, spec
    items["i_learner_train_B%d_ms" % B] = upd
    act = {}
    iqn_taus = None
    for e in dict.fromkeys([16, E]):
        obs = torch.from_numpy(recipe.make_frames(e, 5, spec0.obs_shape)[:, :4].copy())
        with torch.no_grad():
            f = lambda: nets.qval(ora0.po, spec0, nets.normalize(obs)).argmax(-1)
            f()
            n, t0 = 0, time.time()
            while n < 5 or time.time() - t0 < 2.0:
                f()
                n += 1
        act[e] = (time.time() - t0) / n
    items["ii_actor_forward_ms"] = {f"E{e}": round(1e3 * t, 3) for e, t in act.items()}
    # (iii) the reference's replay: deque of (blob, a, r, d) tuples + flat priority vector
    row = 2 * int(np.prod(spec0.obs_shape))
    rp = oreplay.ReferenceReplay(100_000, True)
    blobs = recipe.make_frames(1280, 7, spec0.obs_shape)
    chunk = [(blobs[i], 0, 0.0, False) for i in range(1280)]
    n, t0 = 0, time.time()
    while n < 8:
        rp.extend([(b.copy(), a_, r_, d_) for (b, a_, r_, d_) in chunk])       # the copy stands for the reference's per-transition lz4 + bytes object
        n += 1
    t_ext = (time.time() - t0) / n
    g = recipe.gen(9)
    n, t0 = 0, time.time()
    while n < 20:
        idx = g.permutation(rp.top)[:B]
        got = [rp[int(i)] for i in idx]
        batch = np.stack([x[0].reshape(-1) for x in got])                       # collate: B rows -> one [B, 56448] array (trainer.py:63-72)
        prio = np.array([x[4] for x in got], dtype=np.float32)
        oreplay.is_weights(prio, float(torch.from_numpy(rp.priority).sum().item()), rp.top, rp.beta)
        n += 1
    t_smp = (time.time() - t0) / n
    items["iii_reference_replay_GBps"] = {"extend": round(1280 * row / t_ext / 1e9, 3), "sample_B%d_plus_is_weights" % B: round(batch.shape[0] * row / t_smp / 1e9, 3),
                                          "row_bytes": row, "note": "uncompressed rows; the reference additionally lz4-compresses on extend and decompresses in 2 DataLoader workers"}
    # (iv) one whole Trainer iteration at BASELINE configs[0]'s sizes
    spec_c0 = recipe.NetSpec("dqn", 4)
    ot = OracleTrainer(spec_c0, recipe.make_state_dict(spec_c0, 1), num_envs=16, sample_steps=80, batch_size=512, replay_size=100_000, learner_steps=20,
                       training_start_steps=0)
    t0 = time.time()
    ot.iteration()
    t_c0 = time.time() - t0
    items["iv_config0_iteration"] = {"ms": round(1e3 * t_c0, 1), "env_frames_per_sec": round(16 * 80 / t_c0, 1),
                                     "what": "oracle Trainer iteration: 80 steps x 16 synthetic envs (actor forward, eps-greedy, n-step, extend) + 20 updates of B=512, "
                                             "100 000-slot replay, first iteration (training_start_steps lowered to 0)"}
    t_upd, t_act = upd[algo0]["ms"] * 1e-3, act[E]
    t_iter = args.sample_steps * t_act + args.learner_steps * t_upd
    composed = round(args.sample_steps * E / t_iter, 1)
    items["composed_from_i_and_ii"] = {"env_frames_per_sec": composed,
                                       "what": f"{upd[algo0]['runs']} oracle {algo0} updates at B={B} ({upd[algo0]['ms']:.1f} ms each) + oracle actor forwards at E={E} ({t_act*1e3:.2f} ms each), "
                                               f"extrapolated to one iteration of {args.sample_steps} actor steps + {args.learner_steps} updates: env stepping, n-step bookkeeping and replay are NOT "
                                               "included (round 5's `value`)"}
    # (v) round 6: ONE whole iteration of the bench workload itself, measured end to end on the oracle — actor forwards, epsilon-greedy, the synthetic env on the CPU, n-step
    # bookkeeping, replay extend, sampling and `learner_steps` updates of batch B (a 100 000-slot replay: the rows of one rollout are 1.2 GB) — for the scalar / distributional heads
    # (a quantile network's iteration would take minutes: its `value` stays the composed figure)
    value, sample = composed, items["composed_from_i_and_ii"]["what"]
    if algo0 in ("dqn", "mdqn", "c51", "qr") and not cfg.learner.noisy_net:
        ot = OracleTrainer(spec0, recipe.make_state_dict(spec0, 1), num_envs=E, sample_steps=args.sample_steps, batch_size=B, replay_size=100_000, learner_steps=args.learner_steps,
                           training_start_steps=0, policy=cfg.replay.policy.name, n_step=cfg.learner.n_step_q, double_q=cfg.learner.double_q)
        t0 = time.time()
        ot.iteration()
        t_it = time.time() - t0
        value = round(args.sample_steps * E / t_it, 1)
        items["v_bench_workload_iteration"] = {"ms": round(1e3 * t_it, 1), "env_frames_per_sec": value}
        sample = (f"one whole oracle Trainer iteration at the bench workload's sizes, measured end to end ({t_it:.1f} s): {args.sample_steps} steps x {E} synthetic envs on the CPU (actor forward, "
                  f"eps-greedy, n-step, replay extend) + {args.learner_steps} {algo0} updates of B={B}, 100 000-slot replay, first iteration; torch threads = {cores['used']}.  "
                  f"(Composed from the update and forward timings alone, without env stepping and replay: {composed} env-frames/s, items.composed_from_i_and_ii)")
        del ot
    torch.set_num_threads(keep_threads)
    return {"value": value, "unit": "env-frames/sec", "cores": cores["used"], "kind": "port", "host_cores": cores, "sample": sample, "items": items}


def rank_clock(dist, dt: float, dt_local: float, rank: int, world: int, steps: int, device):
    """The contract's clock under data parallelism: MAX over the ranks of the barrier-to-barrier time, plus every rank's own time for its K steps up to its local
    synchronize (a straggler shows in the one JSON line).  Two collectives on ``device`` ("cuda" under RCCL; "cpu" in the gloo test of this function)."""
    import torch

    t = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    loc = torch.zeros(world, device=device, dtype=torch.float64)
    loc[rank] = dt_local
    dist.all_reduce(loc, op=dist.ReduceOp.SUM)
    ms = [round(1e3 * float(x) / steps, 3) for x in loc.tolist()]
    per_rank = {"min": min(ms), "max": max(ms), "ranks": ms,
                "note": "each rank's own time for the K steps up to its local synchronize, before the closing barrier; ms_per_step is the max over ranks of the barrier-to-barrier time"}
    return float(t[0]), per_rank


def exchange_report(hook, world: int):
    """The "rccl" object of a data-parallel line: what the gradient exchange itself reports (hook.report(): RCCL's ncclCommCount / ncclCommUserRank for the communicator
    the gradients travel on and an all-reduce of ones through it), checked against the launcher's WORLD_SIZE.  Collective."""
    if hook is None or not hasattr(hook, "report"):
        return None
    r = hook.report()
    r["matches_world_size"] = bool(r["nranks"] == world and abs(r["allreduce_of_ones"] - world) < 1e-3 and abs(r["allreduce_of_ones_last"] - world) < 1e-3)
    return r


def throughput_fields(world: int, per_iter: int, learner_steps: int, steps: int, warmup: int, dt: float):
    """value = units ALL ranks processed / the max-over-ranks time (weak scaling: per-rank work is fixed)."""
    value = world * per_iter * steps / dt
    return {"value": round(value, 1), "unit": "env-frames/sec", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * dt / steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "emulator_frames_per_sec_x4": round(4 * value, 1),
            "updates_per_sec": round(world * learner_steps * steps / dt, 2)}


# ------------------------------------------------------------------------------------------------ roofline
PEAK_BF16, PEAK_FP32 = 2500.0, 157.3          # TFLOP/s dense MFMA, MI355X_MICROARCH.md
ENC_F1, ENC_F2, ENC_F3 = 2.0 * 400 * 32 * 256, 2.0 * 81 * 64 * 512, 2.0 * 49 * 64 * 576       # conv1 / conv2 / conv3 FLOP per observation (model.py:93-105)
# candidate kernel families of the probe (agent0_amd/ops.py PROBE_TAGS), tried in this order; the one with the largest share of the iteration is the roofline kernel
PROBE_FAMILIES = ("actor_step_enc", "encoder_fused", "encoder_dgrad_fused", "dense_fwd")


def issued_per_mac(family: str, products: int) -> float:
    """bf16 products the family's kernels issue on the matrix pipe per algorithmic fp32 multiply-add: every fp32 operand is an exact sum of three bf16 terms, so a
    product of two costs `products` (9, or 6 with the cross terms below 2^-24 of the product left out: a0_x9_products) bf16 MFMA products; conv1 multiplies bytes
    (exact in bf16) by three weight terms."""
    if family in ("encoder_fused", "actor_step_enc"):
        return (3 * ENC_F1 + products * (ENC_F2 + ENC_F3)) / (ENC_F1 + ENC_F2 + ENC_F3)
    return float(products)


def family_kernel(family: str, algo: str) -> str:
    scalar = algo in ("dqn", "mdqn")
    return {"actor_step_enc": ("a0_actor_step_enc2_kernel" if scalar else "a0_actor_dist_step_enc_kernel") +
                              " (per env: head + action + env step + replay row of step t, then conv1 + conv2 + conv3 of the env's new observation; one workgroup per env)",
            "encoder_fused": "a0_encoder_fused_multi_kernel / a0_encoder_fused_kernel (conv1 + conv2 + conv3 per observation: the update's forward passes in one launch of 256 looping "
                             "workgroups, and the rollout's first encoder)",
            "encoder_dgrad_fused": "a0_encoder_dgrad_fused_x9_kernel (conv3 + conv2 data gradients per observation)",
            "dense_fwd": "a0_igemm_x9_kernel / a0_igemm_x9_group_kernel, the dense layers' forward GEMMs (fc1 512 x 3136 over the pass's rows, the heads)"}.get(family, family)


def algorithmic_flop_per_iteration(cfg) -> dict:
    """forward FLOP per observation from the layer shapes (model.py:93-105,110-123,203-257); an update = the forward passes of the learner
    (online on s, target on s', + online on s' with double-Q / the third pass of mdqn) + the backward pass (2 x forward - conv1's data gradient).  Quantile networks
    count their rows per observation (iqn: N online rows differentiated, N' target rows, K selection rows; fqf: F differentiated, F target, F selection, F - 1 for the
    fraction loss)."""
    lc, A = cfg.learner, cfg.action_dim
    algo = lc.algo.name
    enc, conv1, fc1 = ENC_F1 + ENC_F2 + ENC_F3, ENC_F1, 2.0 * 3136 * 512
    duel = 1 if lc.dueling_head else 0
    if algo in ("dqn", "mdqn", "c51", "qr"):
        atoms = {"dqn": 1, "mdqn": 1, "c51": lc.c51.num_atoms, "qr": lc.qr.num_atoms}[algo]
        fwd = enc + fc1 + 2.0 * 512 * (A + duel) * atoms
        actor = fwd
        update = (3 if (lc.double_q or algo == "mdqn") else 2) * fwd + 2 * fwd - conv1
    else:
        row = 2.0 * lc.iqn.num_cosines * 3136 + fc1 + 2.0 * 512 * (A + duel)
        if algo == "iqn":
            actor = enc + lc.iqn.K * row
            rows = 3 * lc.iqn.N + lc.iqn.N_dash + lc.iqn.K
        else:
            actor = enc + lc.iqn.F * row + 2.0 * 3136 * lc.iqn.F
            rows = 3 * lc.iqn.F + lc.iqn.F + lc.iqn.F + (lc.iqn.F - 1)
        update = (3 if lc.double_q else 2) * enc + 2 * enc - conv1 + rows * row
    n_obs = cfg.actor.sample_steps * cfg.actor.num_envs
    total = n_obs * actor + lc.learner_steps * lc.batch_size * update
    return {"actor_mflop_per_obs": round(actor / 1e6, 2), "update_mflop_per_sample": round(update / 1e6, 1), "total": total}


def newest_profile(suffix: str):
    try:
        name = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(suffix))[-1]
        return name, json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None, None


def roofline_object(probes, cfg, args, products: int, iter_s: float):
    """The bench line's `roofline`: the probed kernel family with the largest measured time per iteration, priced on the pipe it issues on — bf16 MFMA products
    issued / 2.5 PFLOP/s (a fraction that cannot exceed 1) — with the fp32-equivalent rate (every MAC counted once against the fp32 MFMA peak, a bound this family
    can exceed) as a secondary field, the other candidates beside it, and the whole iteration's algorithmic rate."""
    algo = cfg.learner.algo.name
    flop_it = algorithmic_flop_per_iteration(cfg)
    tfl = flop_it["total"] / iter_s / 1e12
    iteration = {"algorithmic_flop": flop_it["total"], "actor_mflop_per_obs": flop_it["actor_mflop_per_obs"], "update_mflop_per_sample": flop_it["update_mflop_per_sample"],
                 "tflops": round(tfl, 1), "frac_fp32_basis": round(tfl / PEAK_FP32, 3), "kernel_floor_ms": None, "kernel_sum_ms": None}
    bname, budget = newest_profile("_budget.json")
    if budget and algo in budget:
        iteration.update(kernel_floor_ms=budget[algo].get("kernel_floor_ms"), kernel_sum_ms=budget[algo].get("kernel_sum_ms"),
                         kernel_floor_note=f"NOT measured in this run: per-kernel floors (MFMA issue of the products actually issued / 6.3 TB/s HBM / 2.5 us launch floor) and the "
                                           f"rocprofv3 kernel-time sum of the same command, profiles/{bname} (tools/budget.py)")
    if not probes:
        return {"bound": "mfma", "achieved": None, "peak": PEAK_BF16, "unit": "TFLOP/s", "frac": None, "traffic": None, "iteration": iteration, "note": "probe switched off (A0_PROBE=none)"}
    cands = []
    for pr in probes:
        per = issued_per_mac(pr["kernel"], products)
        eq = pr["flop"] / (pr["ms"] * 1e-3) / 1e12
        cands.append({"family": pr["kernel"], "kernel": family_kernel(pr["kernel"], algo), "launches_per_iteration": round(pr["launches"] / args.steps, 1),
                      "avg_us": round(1e3 * pr["ms"] / pr["launches"], 2), "ms_per_iteration": round(pr["ms"] / args.steps, 3), "share_of_iteration": round(pr["ms"] / args.steps / (iter_s * 1e3), 3),
                      "algorithmic_gflop_per_launch": round(pr["flop"] / pr["launches"] / 1e9, 3), "bf16_products_per_mac": round(per, 3),
                      "achieved": round(eq * per, 1), "frac": round(eq * per / PEAK_BF16, 4), "fp32_equivalent_tflops": round(eq, 2)})
    cands.sort(key=lambda c: -c["ms_per_iteration"])
    top = cands[0]
    tname, tjson = newest_profile("_pmc_traffic.json")
    traffic, tnote = None, "not measured for this kernel (traffic is null)"
    try:
        if top["family"] == "actor_step_enc":
            c = tjson["calibration"]["stepenc"]
            traffic = round(2 * c["FETCH_SIZE_bytes_raw"] + c["WRITE_SIZE_bytes_raw"])
            tnote = (f"NOT measured in this run: HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (separate rocprofv3 --pmc passes, tools/refresh_profiles.sh), profiles/{tname}; "
                     f"algorithmic {c['algorithmic_read_bytes'] + c['algorithmic_write_bytes']} B (observations + fc1 slabs in, new stacks + replay rows + conv features out)")
        elif top["family"] == "encoder_fused":
            pm = tjson["per_launch"]
            traffic = round(pm["1024"]["hbm_bytes"])
            tnote = f"NOT measured in this run: the learner launch's (1024 observations) 2 x FETCH_SIZE + WRITE_SIZE, profiles/{tname}; algorithmic minimum {pm['1024']['algorithmic_bytes_min']} B"
        elif top["family"] == "dense_fwd":
            traffic = round(tjson["dense_fwd"][algo]["dense_fwd_gemm"]["hbm_bytes"])
            tnote = f"NOT measured in this run: per-launch mean of the family's 2 x FETCH_SIZE + WRITE_SIZE (tools/pmc_quantile.sh), profiles/{tname}"
    except Exception:
        pass
    return {"bound": "mfma", "achieved": top["achieved"], "peak": PEAK_BF16, "unit": "TFLOP/s", "frac": top["frac"], "traffic": traffic, "traffic_note": tnote,
            "kernel": top["kernel"], "family": top["family"], "launches_per_iteration": top["launches_per_iteration"], "avg_us": top["avg_us"],
            "share_of_iteration": top["share_of_iteration"], "algorithmic_gflop_per_launch": top["algorithmic_gflop_per_launch"], "bf16_products_per_mac": top["bf16_products_per_mac"],
            "basis": f"`achieved` = bf16 MFMA FLOPs the kernel ISSUES (algorithmic FLOPs x {top['bf16_products_per_mac']} bf16 products per multiply-add: fp32 operands as exact sums of three bf16 "
                     f"terms, {products} cross products per fp32 x fp32 product (a0_x9_products), 3 per byte x fp32 product in conv1) / the kernel's average duration; `peak` = dense bf16 MFMA, "
                     "the pipe it issues on, so `frac` cannot exceed 1",
            "fp32_equivalent": {"achieved": top["fp32_equivalent_tflops"], "peak": PEAK_FP32, "unit": "TFLOP/s", "frac": round(top["fp32_equivalent_tflops"] / PEAK_FP32, 4),
                                "note": "every multiply-add counted once against the fp32 MFMA peak — the bound of an fp32-input kernel, which this family may exceed because it issues "
                                        "exact bf16-term products on the faster pipe; secondary, not a utilisation"},
            "chosen_by": "largest measured time per iteration among the probed families (`candidates`, sorted); agrees with the top a0_* row of the rocprofv3 kernel statistics of the "
                         "same command under profiles/",
            "measured": "HIP events on the launch stream over a repeat of the timed iterations with hipGraph replay off; every probed launch carries its event pair "
                        "(hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps, what rocprofv3's kernel trace reports)"
                        + ("; on the launch schedule the probe runs without the overlapped rollout stream" if args.entry == "launch" else ""),
            "peak_source": "MI355X_MICROARCH.md: dense bf16 MFMA 2.5 PFLOP/s, fp32 MFMA 157.3 TFLOP/s",
            "candidates": cands, "iteration": iteration}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.self_launch):
        sys.exit(self_launch(args))
    import torch

    from agent0_amd.deepq.dist import dp_forced, init_process_group, make_grad_hook

    rank, local_rank, world = init_process_group()
    dp = world > 1 or dp_forced()            # A0_DP_FORCE=1: the data-parallel path with a one-rank RCCL group (rehearsal on a one-GPU box)
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    from agent0_amd.deepq.config import parse_overrides
    from agent0_amd.deepq.trainer import Trainer
    from agent0_amd.common.atari_wrappers import ACTION_DIMS

    if args.replicas and world > 1:          # q-head shapes differ between games (4 / 6 / 9 / 18 actions): nothing to reduce across them
        args.env = ("Asterix", "BeamRider", "Breakout", "Enduro", "MsPacman", "Qbert", "Seaquest", "SpaceInvaders")[rank % 8]
    cfg = parse_overrides([f"env_id={args.env}", f"learner.algo={args.algo}", f"actor.num_envs={args.num_envs}", f"replay.size={args.replay_size}",
                           f"learner.batch_size={args.batch}", f"learner.learner_steps={args.learner_steps}", f"actor.sample_steps={args.sample_steps}",
                           "wandb=false", "tb=false", f"logdir={os.path.join(ROOT, 'gpurun_out', 'bench_logs')}", *args.overrides])
    cfg.obs_shape = (4, 84, 84)
    cfg.action_dim = ACTION_DIMS.get(cfg.env_id, 18)
    cfg.seed = cfg.seed + 1000003 * rank
    tr = Trainer(cfg, use_lp=(args.entry == "launch"), rank=rank)
    eng = tr.learner.engine
    if dp and not args.replicas:
        import torch.distributed as dist
        eng.grad_hook = make_grad_hook(tr.ops, eng.L.n_adam)
        eng.adam_eps = 1e-2 / (world * cfg.learner.batch_size)
        dist.broadcast(eng.online.flat, src=0)
        eng.online.refresh_wt()
        eng.sync_target(force=True)

    # ---- untimed: fill the replay ring to capacity with real rollouts (no learning), then warm up full iterations
    per_iter = cfg.actor.sample_steps * cfg.actor.num_envs
    start_steps = cfg.trainer.training_start_steps
    cfg.trainer.training_start_steps = 1 << 62
    t_fill = time.time()
    while len(tr.replay) < cfg.replay.size:
        tr.run_iteration()
    torch.cuda.synchronize()
    t_fill = time.time() - t_fill
    cfg.trainer.training_start_steps = min(start_steps, cfg.replay.size - 1)
    # the `main` schedule issues iteration i + 1's rollout before it waits for iteration i's statistics (Trainer.run does the same): same stream order, same
    # numbers; the timed region then holds K update blocks and K rollouts (2 .. K + 1; rollout 1 was issued by the last warm-up step and is complete at the barrier)
    ahead = os.environ.get("A0_PREFETCH_ROLLOUT", "1") != "0"           # (ignored by the `launch` schedule, whose rollouts are in flight during the update block anyway)
    for _ in range(args.warmup):
        tr.run_iteration(prefetch=ahead)

    def barrier():
        torch.cuda.synchronize()
        if dp:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.time()
    last = None
    for _ in range(args.steps):
        last = tr.run_iteration(prefetch=ahead)
    torch.cuda.synchronize()
    dt_local = time.time() - t0          # this rank's own K steps, before it waits for the others: diagnoses a straggler from the one JSON line
    barrier()
    dt = time.time() - t0
    per_rank = None
    if dp:
        import torch.distributed as dist
        dt, per_rank = rank_clock(dist, dt, dt_local, rank, world, args.steps, "cuda")
    # ---- roofline of the dominant kernel: the same iterations once more per candidate kernel family with the hipGraphs switched off, every launch of
    # the family carrying a HIP event pair on its stream (events cannot be read out of a replayed graph; single-kernel families take the dispatch's own begin / end
    # timestamps through hipExtLaunchKernelGGL — what rocprofv3's kernel trace reports).  The family with the largest share of the iteration is THE roofline kernel; the
    # others are listed beside it.  Not part of `value`.
    probes = []
    pick = os.environ.get("A0_PROBE", "auto")
    families = [] if pick == "none" else list(PROBE_FAMILIES) if pick == "auto" else [pick]
    if families:           # every rank repeats the iterations (they contain the gradient all-reduce); rank 0 records
        tr.learner.use_graph = False
        tr.actors[1].use_graph = False
        if args.entry == "launch":
            tr.overlap = False               # kernel timing without a second stream competing for the CUs
        tr.run_iteration()                   # consumes the rollout the timed loop issued ahead: every probed iteration below launches its own 80 steps
        torch.cuda.synchronize()
        for fam in families:
            if rank == 0:
                tr.ops.probe_begin(fam, 64 + args.steps * (4 * cfg.actor.sample_steps + 16 * cfg.learner.learner_steps))
            for _ in range(args.steps):
                tr.run_iteration()
            torch.cuda.synchronize()
            if rank == 0:
                pr = tr.ops.probe_end()
                if pr["launches"] and pr["ms"] > 0:
                    probes.append(pr)
    # ---- metric 2 of BASELINE.json: replay sample GB/s = B * 56 448 B / t(sample + gather); the update itself never gathers
    # (conv1 reads ring rows through the slot index), so the gather kernel is timed on its own here
    replay_gbps = None
    if rank == 0:
        rp = tr.replay
        out_rows = torch.empty(rp.B * rp.row_bytes, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            rp.sample_gathered(out_rows)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_rep = 100
        torch.cuda.synchronize(); e0.record()
        for _ in range(n_rep):
            rp.sample_gathered(out_rows)          # index generation + metadata + 28.9 MB row gather, one launch
        e1.record(); torch.cuda.synchronize()
        replay_gbps = rp.B * rp.row_bytes * n_rep / (e0.elapsed_time(e1) * 1e-3) / 1e9
    # ---- the same metric at the reference's update:data ratio for 256 envs (learner_steps = 320 instead of 20)
    ratio320 = None
    if not dp and not args.no_ratio320:
        tr.learner.use_graph = True
        tr.actors[1].use_graph = True
        if args.entry == "launch":
            tr.overlap = True
        keep_L = cfg.learner.learner_steps
        cfg.learner.learner_steps = 320
        tr.run_iteration(prefetch=ahead)
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(3):
            tr.run_iteration(prefetch=ahead)
        torch.cuda.synchronize()
        d1 = (time.time() - t1) / 3
        cfg.learner.learner_steps = keep_L
        ratio320 = {"learner_steps": 320, "value": round(per_iter / d1, 1), "unit": "env-frames/sec", "ms_per_step": round(1e3 * d1, 2),
                    "updates_per_sec": round(320 / d1, 1)}
    # ---- the other entry point's schedule on the same workload (second trainer, own full replay), so that one bench line carries both
    other = None
    if not dp and not args.no_other_entry:
        import copy
        cfg2 = copy.deepcopy(cfg)
        cfg2.trainer.training_start_steps = 1 << 62
        tr2 = Trainer(cfg2, use_lp=(args.entry != "launch"), rank=rank)
        while len(tr2.replay) < cfg2.replay.size:
            tr2.run_iteration()
        cfg2.trainer.training_start_steps = min(start_steps, cfg2.replay.size - 1)
        for _ in range(max(args.warmup, 3)):          # eager runs, then graph capture
            tr2.run_iteration(prefetch=ahead)
        torch.cuda.synchronize()
        t2 = time.time()
        for _ in range(args.steps):
            tr2.run_iteration(prefetch=ahead)
        torch.cuda.synchronize()
        d2 = (time.time() - t2) / args.steps
        other = {"entry": "agent0.deepq." + ("main" if args.entry == "launch" else "launch"), "value": round(per_iter / d2, 1), "unit": "env-frames/sec",
                 "ms_per_step": round(1e3 * d2, 3), "steps": args.steps}
        del tr2
    # the CPU baseline is a property of the box, not of the rank count: rank 0 times it at every N, while the other ranks wait in the barrier below (so that every rank
    # leaves the process group together)
    cpu_base = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(args, cfg)
        if world > 1:
            cpu_base["note"] = f"timed on rank 0's share of the host cores while the other {world - 1} ranks wait in a barrier; compare with the N = 1 line's"
    if dp:
        barrier()
    exchange = None
    rccl = exchange_report(eng.grad_hook, world) if dp and not args.replicas else None          # collective: before the communicator is closed
    if eng.grad_hook is not None:
        exchange = type(eng.grad_hook).__name__ + (" (captured in the update's hipGraph)" if getattr(eng.grad_hook, "in_graph", False) else " (eager, between three graphs)")
        if getattr(tr, "_nl", None):      # A0_NATIVE_LOOP_DP=1: the learner handle issues the same two all-reduces itself
            exchange = type(eng.grad_hook).__name__ + "'s communicator in the learner handle (a0_learner_set_exchange: eager launches from native code)"
            tr._nl.detach_exchange()
        if hasattr(eng.grad_hook, "close"):
            eng.grad_hook.close()
    if rank != 0:
        if dp:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    cu, hbm, arch = tr.ops.device_info()
    out = {
        "metric": "env-frames/sec (learner FPS: transitions collected and saved to replay per second with the learner running, pre-frameskip)",
        **throughput_fields(world, per_iter, cfg.learner.learner_steps, args.steps, args.warmup, dt),
        "dtype": "f32", "data": "synthetic",
        "arithmetic": {"x9_products": tr.ops.x9_products(),
                       "what": "fp32 results on the bf16 matrix pipe: every fp32 operand is an exact sum of three bf16 terms and the kernels accumulate, in fp32, six of the nine cross products "
                               "of two such sums (the three left out are each below 2^-24 of the product; A0_X9_PRODUCTS=9 / a0_x9_products(9) forms all nine); conv1 multiplies bytes by three "
                               "weight terms, all formed",
                       "record": "profiles/r06_x6_accuracy.txt (tools/check_x6_accuracy.hip): against fp64, six- and nine-product dot products differ by < 1 % in rms error at every "
                                 "reduction length of the path; both are 0.9 - 1.4 x the sequential fp32 fmaf chain's rms error (the excess is the matrix instruction's accumulate rounding, "
                                 "present in both forms); every GPU parity test runs on the six-product default at unchanged tolerances"},
        "config": {"workload": f"{cfg.env_id} {cfg.learner.algo.name}, {cfg.actor.num_envs} vectorized envs x {cfg.actor.sample_steps} steps + "
                               f"{cfg.learner.learner_steps} updates of batch {cfg.learner.batch_size} per iteration, {cfg.replay.size}-transition HBM replay "
                               f"(full), obs 4x84x84 u8, per-rank shards, " + ("independent replicas: one game per rank, no gradient exchange" if args.replicas else
                               "RCCL grad all-reduce" + ("" if world > 1 else " (one-rank group: rehearsal)" if dp else " (inactive at 1 GPU)")),
                   "learner_steps": cfg.learner.learner_steps, "num_envs": cfg.actor.num_envs, "batch_size": cfg.learner.batch_size,
                   "replay_size": cfg.replay.size, "parallelism": (f"replicas{world}" if args.replicas else f"dp{world}"), "entry": f"agent0.deepq.{args.entry}", "rollout_prefetch": bool(ahead and args.entry == "main"),
                   "host_loop": ("library handles over the Python classes' buffers (deepq/native_loop.py: eager launches from native code)" if getattr(tr, "_nl", None)
                                 else "Python classes + hipGraphs"),
                   "gradient_exchange": exchange},
        "per_rank_ms_per_step": per_rank, "gradient_exchange": exchange, "rccl": rccl,
        "device": arch, "replay_fill_s": round(t_fill, 2),
        "at_reference_update_ratio": ratio320, "other_entry": other,
        "replay_sample_GBps": None if replay_gbps is None else round(replay_gbps, 1), "last_loss": None if last is None or last.get("loss") is None else float(last["loss"]),
    }
    out["roofline"] = roofline_object(probes, cfg, args, tr.ops.x9_products(), dt / args.steps)
    out["cpu_baseline"] = cpu_base
    print(json.dumps(out))
    if dp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
