"""Per-kernel floor table of one bench configuration: for every kernel of an iteration — launches, measured average duration (rocprofv3
kernel stats of the bench command), algorithmic FLOPs / bytes, the bound that applies to ITS pipe, the floor that bound gives, and the gap.

    python tools/budget.py <kernel_stats.csv> <config: dqn|c51|qr|mdqn> [ms_per_step] [--json profiles/r06_budget.json] >> profiles/r06_budget.md

A0_BUDGET_PRODUCTS=9 prices the strict nine-product mode (a0_x9_products); the default is the shipped six.  --json merges {config: {kernel_sum_ms, kernel_floor_ms,
ms_per_iteration}} into the named file (what bench.py's roofline.iteration cites).

Bounds (MI355X_MICROARCH.md constants table):
  mfma16   v_mfma_f32_16x16x32_bf16 issues every 16 cycles per SIMD            -> 16 384 bf16 FLOP / 16 cyc / SIMD = 2.5 PFLOP/s over 1024 SIMDs at 2.4 GHz
  mfma32   v_mfma_f32_32x32x16_bf16 issues every 32 cycles per SIMD            -> the same rate
           an exact fp32 product costs 9 bf16 products (3 x 3 terms; 3 where one operand is bytes): floor = issued bf16 FLOP / 2.5 PF
  hbm      achievable HBM3E bandwidth 6.3 TB/s (8 TB/s nominal)
  valu     one wave-instruction per 2 cycles per SIMD at >= 2 waves per SIMD (4 cycles for a lone wave)
  launch   a kernel that does almost nothing still takes ~2.5 us from start to end (ramp-up, first loads, drain): the floor of every tiny kernel
The floor is the LARGER of the pipe bound and the launch floor.
"""
import csv
import json
import os
import re
import sys

PF = 2.5e15
HBM = 6.3e12
LAUNCH = 2.5
GHZ = 2.4e9
SIMDS = 1024


def mfma_us(flop_issued):
    return flop_issued / PF * 1e6


def hbm_us(nbytes):
    return nbytes / HBM * 1e6


B, E, OBS, FEAT = 512, 256, 4 * 84 * 84, 3136
X9 = int(os.environ.get("A0_BUDGET_PRODUCTS", "6"))          # cross products per fp32 x fp32 multiply (a0_x9_products): 6 shipped, 9 strict
ENC_ISSUED = 3 * 2 * 400 * 32 * 256 + X9 * (2 * 81 * 64 * 512 + 2 * 49 * 64 * 576)          # bf16 FLOP per observation as issued without tile padding
# MFMAs per observation with the tiles the kernel uses (16-row blocks: 400 -> 25, 81 -> 6, 49 -> 4 blocks): (25*2*8*3 + 6*4*16*X9 + 4*4*18*X9) * 16 cycles over 4 SIMDs
ENC_TILE_CYC = (25 * 2 * 8 * 3 + 6 * 4 * 16 * X9 + 4 * 4 * 18 * X9) * 16 / 4
DGRAD_FLOP = 12.5e6


def gemm(M, N, K, n=1):
    return n * 2.0 * M * N * K


def model(cfg):
    """name fragment -> (what, bound name, floor us per launch, note).  Shapes of BASELINE configs[1] / [2] and the qr / mdqn variants at A = 4."""
    A = 4
    npass = {"dqn": 2, "c51": 3, "qr": 2, "mdqn": 3}[cfg]
    Npad = {"dqn": 32, "mdqn": 32, "c51": 256, "qr": 800}[cfg]
    n_par = {"dqn": 1_686_180, "mdqn": 1_686_180, "c51": 3_551_902, "qr": 2_094_528}[cfg]
    m = {
        "a0_encoder_fused_kernel": ("actor encoder, 256 obs (1 per CU)", "mfma16 issue", max(ENC_TILE_CYC / GHZ * 1e6, 0), f"{ENC_ISSUED / 1e6:.1f} MFLOP bf16 issued per obs; one obs per CU, so the launch lasts as long as ONE observation: "
                                    f"{ENC_TILE_CYC:.0f} cyc of MFMA issue per SIMD with the kernel's tiles (unpadded: {ENC_ISSUED / 4096 / 4:.0f}); matrix pipe busy 55 % of a wave's lifetime (r04_pmc_encoder.txt): LDS fragment reads + the bf16 term split of act1 / act2 + three layer barriers on 8 waves"),
        "a0_encoder_fused_multi_kernel": (f"learner encoder, {npass} x 512 obs", "mfma16 issue", npass * 512 / 256 * ENC_TILE_CYC / GHZ * 1e6, "the same body looping over 2 (3) x 2 observations per CU"),
        "a0_encoder_dgrad_fused_x9_kernel": ("conv3 + conv2 data gradients, 512 obs", "mfma16 issue", mfma_us(512 * DGRAD_FLOP * X9), f"12.5 MFLOP fp32 per obs x {X9} products; one workgroup per CU (123 KB LDS)"),
        "a0_conv23_wgrad_fused_kernel": ("conv2 + conv3 weight gradients", "mfma32 issue", mfma_us(512 * 8.92e6 * X9), f"8.92 MFLOP per obs x {X9}"),
        "a0_conv1_wgrad_fused_kernel": ("conv1 weight gradient", "mfma issue (x3)", mfma_us(512 * 6.55e6 * 3), "6.55 MFLOP per obs, bytes x three terms of d1"),
        "a0_igemm_x9_kernel<OpMatKC, OpMatKC, EpiSlab, 2, 2, 1, 1, 2>": ("actor fc1 256 x 512 x 3136 (dqn / mdqn: 8 split-K slabs) or actor head GEMM (c51 / qr)", "mfma32 issue",
                                                                       max(mfma_us(gemm(E, 512, FEAT) * X9), LAUNCH) if cfg in ("dqn", "mdqn") else max(mfma_us(gemm(E, Npad, 512) * X9), LAUNCH),
                                                                       "256 workgroups x 12 k tiles: the k loop is 12 x 576 cyc = 2.9 us of the 13; prologue (first tiles from MALL), slab epilogue and ramp are the rest. "
                                                                       "Unsplit (bias + ReLU in the epilogue, no slabs) would be 32 workgroups x 98 k tiles = 23.5 us: split-K wins"),
        "a0_igemm_x9_kernel<OpMatKC, OpMatKC, EpiSlab, 4, 1, 1, 2, 2>": ("actor fc1 (c51 / qr: a0_dense_fwd, 16 slabs)", "mfma32 issue", mfma_us(gemm(E, 512, FEAT) * X9), "as above"),
        "a0_actor_qhead_env_kernel": ("actor tail + env step + replay row, 256 envs", "hbm", hbm_us(E * (OBS + 3 * OBS) + 8 * E * 512 * 4), "28 KB read + 85 KB written per env, + the fc1 slabs; wave 0's serial tail (slab sums, head, Philox, n-step) is the critical path"),
        "a0_actor_step_enc": ("actor tail + env step + replay row, then the new observation's encoder (main schedule, scalar heads: steps 1..T-1)", "mfma16 issue + hbm",
                                     ENC_TILE_CYC / GHZ * 1e6 + hbm_us(E * (OBS + 3 * OBS) + 8 * E * 512 * 4), "the encoder's issue floor plus the tail's traffic (a0_actor_step_enc2_kernel runs conv1's channels 0..2 beside the tail; timing-only builds: 28.7 us without stores, tail and catch-up); "
                                     "saves the boundary between a0_actor_qhead_env_kernel and a0_encoder_fused_kernel (~4 us per step)"),
        "a0_actor_dist_step_enc_kernel": ("distributional actor tail + env step + replay row, then the new observation's encoder (main schedule, c51 / qr: steps 1..T-1)", "mfma16 issue + hbm",
                                          ENC_TILE_CYC / GHZ * 1e6 + hbm_us(E * (OBS + 3 * OBS) + 8 * E * Npad * 4), "as a0_actor_step_enc_kernel, head slabs instead of fc1 slabs"),
        "a0_actor_dist_tail_env_kernel": ("distributional actor tail + env step + replay row", "hbm", hbm_us(E * (OBS + 3 * OBS) + 8 * E * Npad * 4), "as above, head slabs instead of fc1 slabs"),
        "a0_reduce_bias_act_kernel": ("fc1 slab sum + bias + ReLU (dist actors)", "launch", LAUNCH, "16 x 0.5 MB of slabs: 1.3 us of traffic under a launch floor"),
        "a0_igemm_x9_group_kernel": (f"{npass} grouped fc1 GEMMs 512 x 512 x 3136 (and, c51 / qr, the grouped head GEMMs)", "mfma32 issue", mfma_us(gemm(B, 512, FEAT, npass) * X9), "fc1 group; the head group is smaller"),
        "a0_igemm_x9_trio_kernel": ("fc1 data gradient + fc1 weight gradient + the head's weight gradient, one launch (scalar heads)", "mfma32 issue", mfma_us((gemm(B, 512, FEAT, 2) + gemm(B, Npad, 512)) * X9),
                                    "the pair below with the head's 8 tiles as a third problem in grid row 0"),
        "a0_igemm_x9_pair_kernel<OpMatKC, OpMatXC, EpiMaskMat, OpMatXC, OpMatXC, EpiWgradSlab, 2, 2": ("the head's data gradient + weight gradient side by side (c51 / qr)", "mfma32 issue",
                                    max(LAUNCH, mfma_us(gemm(B, 512, Npad, 2) * X9)), "64 + 32 (c51) / 104 (qr) tiles of 64 x 64 in one round; lasts as long as the longer k loop"),
        "a0_igemm_x9_pair_kernel": ("fc1 data gradient + weight gradient, one launch", "mfma32 issue", mfma_us(gemm(B, 512, FEAT, 2) * X9), f"2 x 1.64 GFLOP x {X9}; 784 workgroups on 512 slots"),
        "a0_igemm_x9_kernel<OpMatKC, OpMatXC, EpiMaskMat": ("head / fc1 data gradient (unpaired launches)", "mfma32 issue", mfma_us(gemm(B, 512, max(Npad, 512)) * X9), ""),
        "a0_igemm_x9_kernel<OpMatXC, OpMatXC, EpiWgradSlab": ("fc1 weight gradient (unpaired launches: probe pass)", "mfma32 issue", mfma_us(gemm(B, 512, FEAT) * X9), ""),
        "a0_igemm_kernel<OpMatXC, OpMatXC, EpiWgradSlab": ("head weight gradient", "launch", max(LAUNCH, mfma_us(gemm(B, Npad, 512) * 16)), "fp32 MFMA chain"),
        "a0_dqn_head_loss_slabs_kernel": ("fc1 slab sums + heads + loss + head gradient + dh", "hbm / L2", max(LAUNCH, hbm_us(npass * 4 * B * 512 * 4 + 2 * B * 512 * 4)), "reads 2-3 x 4 slabs x 1 MB, writes h and dh"),
        "a0_c51_head_loss_slabs_kernel": ("C51 from head slabs to loss + head gradient", "launch / latency", LAUNCH + 2.0, "three dependent L2 round trips + the 51-step projection scans"),
        "a0_qr_head_loss_slabs_kernel": ("QR from head slabs to quantile Huber loss + head gradient", "valu", max(LAUNCH, 512 * 200 * 200 * 8 / 64 * 2 / SIMDS / GHZ * 1e6 * 256 / 200) + 2.0,
                                         "20.5 M pairs x 8 vector instructions; lanes 200 of 256 busy; + staging of 3 x 4 KB per sample"),
        "a0_adam_sync_kernel": ("Adam + target copy + loss statistic", "hbm", hbm_us(n_par * 4 * 7), "28 B per parameter (p, g, m, v read; p, m, v written)"),
        "a0_reduce_segments_kernel": ("slab reductions of the weight gradients", "hbm / L2", max(LAUNCH, hbm_us(8.4e6 + 3e6)), "conv1's 256 slabs x 32 KB + conv2 / conv3 / head slabs"),
        "a0_reduce_bias_act_multi_kernel": ("fc1 slab sums of the update's passes", "launch", LAUNCH, ""),
        "a0_conv_wt_kernel": ("refresh of the fused kernels' weight copies", "launch", LAUNCH, "0.5 MB"),
        "a0_sample_gather_kernel": ("replay sample + gather (bench metric 2 only)", "hbm", hbm_us(2 * B * 2 * OBS), "not part of an iteration"),
        "a0_sample_slots_multi_kernel": ("20 uniform batches' slots + metadata", "launch", LAUNCH, ""),
        "a0_noisy_multi_v4_kernel<false>": ("NoisyNet compose, both networks", "hbm", hbm_us(2 * (512 * FEAT + Npad * 512) * 12), "12 B per weight; actor resets compose one network"),
        "a0_noisy_multi_v4_kernel<true>": ("NoisyNet sigma gradients", "hbm", hbm_us((512 * FEAT + Npad * 512) * 8), ""),
        "a0_rng_normal_kernel": ("NoisyNet noise draws (~10 K normals)", "launch", LAUNCH, "not an HBM pass: 40 KB; pure launch cost"),
        "a0_sumtree_batch_kernel": ("sum-tree: top rebuild + stratified descent + batch metadata", "latency", LAUNCH + 5.0, "20 levels, 11 in LDS + 5 dependent L2 round trips"),
        "a0_sumtree_set_sub_kernel": ("sum-tree priority update (subtrees)", "latency", LAUNCH + 3.0, ""),
        "a0_sumtree_set_range_kernel": ("sum-tree: a rollout's 20 480 new leaves", "latency", LAUNCH + 6.0, "one workgroup, level-synchronous"),
        "a0_mean_rows_kernel": ("per-step mean max-Q", "launch", LAUNCH, ""),
    }
    return m


# launches per iteration (80 actor steps, 20 updates; NoisyNet: a reset every 4 actor steps and one per update)
def per_iteration(cfg, name):
    noisy = cfg == "c51"
    if name.startswith(("a0_actor_step_enc", "a0_actor_dist_step_enc_kernel")):      # a0_actor_step_enc_kernel / a0_actor_step_enc2_kernel (the default: conv1 beside the tail)
        return 79
    if name.startswith(("a0_encoder_fused_kernel", "a0_actor_qhead_env_kernel", "a0_actor_dist_tail_env_kernel")):
        return 1                     # a rollout's first encoder and last tail; the 79 steps between run a0_actor_step_enc_kernel / a0_actor_dist_step_enc_kernel
    if name.startswith(("a0_encoder_fused_kernel", "a0_actor_qhead_env_kernel", "a0_actor_dist_tail_env_kernel", "a0_reduce_bias_act_kernel")):
        return 80
    if name.startswith("a0_igemm_x9_kernel<OpMatKC, OpMatKC, EpiSlab"):
        return 80
    if name.startswith(("a0_rng_normal_kernel", "a0_noisy_multi_v4_kernel<false>")):
        return 40 if noisy else 0
    if name.startswith("a0_igemm_x9_group_kernel"):
        return 20 if cfg in ("dqn", "mdqn") else 40
    if name.startswith(("a0_mean_rows_kernel", "a0_sumtree_set_range_kernel", "a0_sample_slots_multi_kernel")):
        return 1
    if name.startswith(("a0_sample_gather_kernel", "a0_igemm_x9_kernel<OpMatXC, OpMatXC, EpiWgradSlab", "a0_igemm_x9_kernel<OpMatKC, OpMatXC, EpiMaskMat",
                        "a0_igemm_kernel<OpMatXC, OpMatXC, EpiWgradSlab")):
        return 0                     # bench metric 2 / probe pass only (the unpaired gradient GEMMs: pair and trio launches are off while the probe brackets kernels)
    return 20


def main():
    argv = list(sys.argv[1:])
    jpath = None
    if "--json" in argv:
        i = argv.index("--json")
        jpath = argv[i + 1]
        del argv[i:i + 2]
    path, cfg = argv[0], argv[1]
    ms = float(argv[2]) if len(argv) > 2 else None
    rows = list(csv.DictReader(open(path)))
    md = model(cfg)
    tot = tot_floor = 0.0
    out = []
    for r in rows:
        name, calls, avg = r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e3
        if name.startswith("void "):
            name = name[5:]
        if name.startswith("a0_igemm_x9"):          # the trailing template argument is the product count (round 6): the model rows are keyed without it
            name = re.sub(r", [69](, (true|false))?>", ">", name, count=1)      # (and, for the plain GEMM, whether the launch runs without staging masks)
        per_iter = per_iteration(cfg, name)
        hit = next((k for k in md if name.startswith(k)), None)
        if hit is None or per_iter == 0:
            if hit is None and name.startswith("a0_") and calls > 20:
                print(f"(no model row for {name.split('(')[0]}: {calls} calls x {avg:.1f} us)", file=sys.stderr)
            continue
        what, bound, floor, note = md[hit]
        floor = max(floor, LAUNCH)
        t = per_iter * avg
        tot += t
        tot_floor += per_iter * floor
        out.append((t, f"| `{name.split('(')[0][:62]}` | {what} | {per_iter:.0f} | {avg:.1f} | {bound} | {floor:.1f} | {avg - floor:+.1f} | {per_iter * (avg - floor) / 1e3:.2f} | {note} |"))
    print(f"\n### {cfg}" + (f" — {ms} ms per iteration measured" if ms else "") + f": kernels sum to {tot / 1e3:.2f} ms, their floors to {tot_floor / 1e3:.2f} ms\n")
    print("| kernel | what | launches / iteration | measured µs | bound | floor µs | gap µs | gap ms / iteration | what the gap is |")
    print("|---|---|---|---|---|---|---|---|---|")
    for _, line in sorted(out, key=lambda x: -x[0]):
        print(line)
    if jpath:
        try:
            acc = json.load(open(jpath))
        except (OSError, ValueError):
            acc = {}
        acc[cfg] = {"kernel_sum_ms": round(tot / 1e3, 3), "kernel_floor_ms": round(tot_floor / 1e3, 3), "ms_per_iteration": ms, "products": X9, "source": os.path.basename(path)}
        json.dump(acc, open(jpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
