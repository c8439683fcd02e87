"""Builds the tracked evidence under profiles/ from the scratch output of tools/refresh_profiles.sh + tools/bench_configs.sh
(gpurun_out/<round>/): bench lines, rocprofv3 kernel statistics, per-grid durations of the fused kernels, PMC traffic, the other
BASELINE configurations, and the summary table.   usage: python tools/make_profiles.py [round_tag]"""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
SRC = os.path.join(ROOT, "gpurun_out", tag)
DST = os.path.join(ROOT, "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


bench = last_json(os.path.join(SRC, "bench.json"))
launch = last_json(os.path.join(SRC, "bench_launch.json"))
json.dump(bench, open(os.path.join(DST, f"{tag}_bench_dqn.json"), "w"), indent=1)
json.dump(launch, open(os.path.join(DST, f"{tag}_bench_dqn_launch_entry.json"), "w"), indent=1)
shutil.copy(os.path.join(SRC, "kernel_stats.csv"), os.path.join(DST, f"{tag}_bench_dqn_kernel_stats.csv"))
by_grid = json.load(open(os.path.join(SRC, "fused_by_grid.json")))
json.dump(by_grid, open(os.path.join(DST, f"{tag}_fused_kernels_by_grid.json"), "w"), indent=1)

# ---- PMC traffic: counter unit KB; FETCH_SIZE x2 on gfx950 (64 B counted per 128-B request on wide coalesced reads), WRITE_SIZE x1
pm = json.load(open(os.path.join(SRC, "pmc_summary.json")))
def kb(key, c):
    v = pm.get(f"{key}:{c}")
    return None if v is None else v["mean"] * 1024.0
E256, OBS = 256, 4 * 84 * 84
cal_r, cal_w = kb("envcommit:1792", "FETCH_SIZE"), kb("envcommit:1792", "WRITE_SIZE")     # grid (7, 256) x 256 threads
for k in list(pm):
    if k.startswith("envcommit:"):
        g = k.split(":")[1]
        cal_r, cal_w = kb(f"envcommit:{g}", "FETCH_SIZE"), kb(f"envcommit:{g}", "WRITE_SIZE")
# calibration kernels that still run (round 4): a0_sample_gather_kernel copies B = 512 rows of 56 448 B (28.90 MB in, 28.90 MB out, bench.py's replay-sample
# measurement); a0_actor_qhead_env_kernel reads 256 observations (7.23 MB) and writes the new stack + the replay row (21.68 MB) per actor step
calib = {}
for name, rd, wr, what in (("gather", 512 * 2 * OBS, 512 * 2 * OBS, "a0_sample_gather_kernel: 512 rows x 56 448 B copied"),
                           ("qenv", E256 * OBS + 8 * E256 * 512 * 4, E256 * 3 * OBS, "a0_actor_qhead_env_kernel: 256 observations + the fc1 GEMM's eight split-K slabs (4.19 MB) read, "
                                    "new stacks + replay rows written"),
                           ("stepenc", E256 * OBS + 8 * E256 * 512 * 4, E256 * 3 * OBS + E256 * 3136 * 4, "a0_actor_step_enc2_kernel (round 5): the same tail traffic, then the encoder of the new observation "
                                       "(read back by the workgroup that wrote it: an L2 hit, not counted as algorithmic) + 3.2 MB of conv features written")):
    for k in list(pm):
        if k.startswith(name + ":") and k.endswith(":FETCH_SIZE"):
            g = k.split(":")[1]
            r_, w_ = kb(f"{name}:{g}", "FETCH_SIZE"), kb(f"{name}:{g}", "WRITE_SIZE")
            if r_ and w_:
                calib[name] = {"kernel": what, "algorithmic_read_bytes": rd, "algorithmic_write_bytes": wr, "FETCH_SIZE_bytes_raw": r_, "WRITE_SIZE_bytes_raw": w_,
                               "read_correction_implied": round(rd / r_, 3), "write_correction_implied": round(wr / w_, 3), "launches": pm[k]["n"]}
traffic = {
    "kernels": "a0_encoder_fused_kernel<7,3,2,84,true[,true]> (enc: one observation per workgroup for the actor's 256, looping for the learner's 512), a0_encoder_dgrad_fused_x9_kernel (dgrad); "
               "a0_env_step_commit_kernel as the calibration kernel",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
              "--no-ratio320 --no-other-entry --replay-size 100000 (tools/refresh_profiles.sh)",
    "corrections": "counter unit KB; FETCH_SIZE x2 on gfx950 (64 B counted per 128-B request on wide coalesced reads), WRITE_SIZE x1.  Calibrated IN THIS RUN on the kernels of "
                   "known traffic listed under `calibration` (" + "; ".join(f"{k}: read x{v['read_correction_implied']}, write x{v['write_correction_implied']}" for k, v in calib.items()) + ")"
                   + (f"; and on a0_env_step_commit_kernel ({E256 * OBS / 1e6:.2f} MB in, {E256 * 3 * OBS / 1e6:.2f} MB out per launch): FETCH_SIZE x2 = {2 * cal_r / 1e6:.2f} MB, WRITE_SIZE = {cal_w / 1e6:.2f} MB"
                      if cal_r and cal_w else " (a0_env_step_commit_kernel, rounds 1-3's calibration kernel, no longer runs in the bench: the actor's tail performs the env step)"),
    "calibration": calib,
    "per_launch": {},
}
for n_obs, grid in ((256, 256 * 512), (512, "loop"), (1024, "multi")):       # "loop": the looping instantiation (one pass of 512); "multi": round 4, the update's two passes of 512 in one launch
    r, w = kb(f"enc:{grid}", "FETCH_SIZE"), kb(f"enc:{grid}", "WRITE_SIZE")
    if r is None or w is None:
        continue
    traffic["per_launch"][str(n_obs)] = {
        "observations": n_obs, "hbm_read_bytes": 2 * r, "hbm_write_bytes": w, "hbm_bytes": 2 * r + w,
        "algorithmic_bytes_min": n_obs * (OBS + 49 * 64 * 4) + 470016,
        "note": "actor launches: observations in (7.2 MB), conv features out (3.2 MB, exact), bf16-term weights (0.47 MB, fetched once per XCD L2)" if n_obs == 256 else
                ("learner launches, average of the online pass (also stores act1/act2 for the backward pass: +36.8 MB) and the target pass (features only)" if n_obs == 512 else
                 "learner launches since round 4: the target pass on s' (features only) and the online pass on s (also stores act1/act2 for the backward pass: +36.8 MB) in ONE launch")}
r, w = kb(f"dgrad:{256 * 512}", "FETCH_SIZE"), kb(f"dgrad:{256 * 512}", "WRITE_SIZE")        # 256 looping workgroups for 512 observations
if r is not None and w is not None:
    traffic["dgrad_per_launch_512"] = {"hbm_read_bytes": 2 * r, "hbm_write_bytes": w,
                                       "algorithmic_bytes": {"read": 512 * (49 * 64 + 81 * 64 + 400 * 32) * 4, "write": 512 * (81 * 64 + 400 * 32) * 4}}
# ---- the quantile configurations' dense forward GEMMs and the short-reduction kernel (tools/pmc_quantile.sh)
qp = os.path.join(SRC, "pmc_quantile.json")
if os.path.exists(qp):
    traffic["dense_fwd"] = json.load(open(qp))
json.dump(traffic, open(os.path.join(DST, f"{tag}_pmc_traffic.json"), "w"), indent=1)

# ---- the other BASELINE configurations
names = {"cfg2_c51_rainbow": "configs[2] Breakout c51 double+dueling+noisy, n_step=3, prioritized (sum-tree)", "cfg3_iqn_asterix": "configs[3] Asterix iqn (iqr)",
         "cfg4_fqf_asterix": "configs[4] Asterix fqf, one rank", "cfg_qr": "Breakout qr", "cfg_mdqn": "Breakout mdqn"}
other = {"note": "BASELINE.json configs[2..4] (+ qr, mdqn) at full size on one MI355X, python bench.py --steps 4 --warmup 2 --algo ... (tools/bench_configs.sh); these are "
                 "parity-test configurations, recorded for completeness — the headline line is " + f"{tag}_bench_dqn.json"}
for f, name in names.items():
    path = os.path.join(SRC, f + ".json")
    if os.path.exists(path):
        try:
            d = last_json(path)
            other[name] = {"env_frames_per_sec": d["value"], "ms_per_iteration": d["ms_per_step"], "updates_per_sec": d["updates_per_sec"], "workload": d["config"]["workload"],
                           "last_loss": d["last_loss"]}
        except Exception as e:      # noqa: BLE001
            other[name] = {"error": str(e)}
json.dump(other, open(os.path.join(DST, f"{tag}_other_configs.json"), "w"), indent=1)

# ---- BASELINE configs[2..4] with their own kernel tables and roofline (tools/prof_configs.sh)
cfg_lines = [f"# BASELINE configs[2..4] (+ qr, mdqn) at full size ({tag}, MI355X gfx950, 1 GPU): bench line with the roofline of that configuration's dominant kernel + rocprofv3 kernel statistics", "",
             "`R=" + tag + " bash tools/prof_configs.sh`: per configuration `python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 --algo ...` once plain (the JSON line) "
             "and once under `rocprofv3 --kernel-trace --stats` (the table; it covers the untimed replay fill, warm-up, the timed and the probe iterations).", ""]
for short, title in (("c51", "configs[2] Breakout c51 double-Q + dueling + NoisyNet, n_step = 3, prioritized sum-tree replay"), ("iqn", "configs[3] Asterix iqn (iqr)"),
                     ("fqf", "configs[4] Asterix fqf, one rank"), ("qr", "Breakout qr (quantile regression, 200 quantiles: named in north_star)"), ("mdqn", "Breakout mdqn (Munchausen dqn)")):
    bj, ks = os.path.join(SRC, f"{short}_bench.json"), os.path.join(SRC, f"{short}_kernel_stats.csv")
    if not (os.path.exists(bj) and os.path.exists(ks)):
        continue
    d = last_json(bj)
    json.dump(d, open(os.path.join(DST, f"{tag}_{short}_bench.json"), "w"), indent=1)
    shutil.copy(ks, os.path.join(DST, f"{tag}_{short}_kernel_stats.csv"))
    r = d.get("roofline") or {}
    cfg_lines += [f"## {title}", "",
                  f"* {d['value']:.0f} env-frames/s, {d['ms_per_step']} ms per iteration, {d['updates_per_sec']} updates/s (`{tag}_{short}_bench.json`).",
                  f"* roofline of the kernel family with the largest measured share ({r.get('share_of_iteration')} of the iteration: {r.get('launches_per_iteration')} launches x {r.get('avg_us')} us): "
                  f"{r.get('achieved')} {r.get('unit')} of bf16 MFMA products issued = **{r.get('frac')}** of the {r.get('peak')} dense bf16 peak ({r.get('bf16_products_per_mac')} products per multiply-add); "
                  f"fp32-equivalent {(r.get('fp32_equivalent') or {}).get('achieved')} TFLOP/s.  {str(r.get('kernel'))[:160]}...",
                  "* other probed families: " + "; ".join(f"{c['family']} {c['launches_per_iteration']} x {c['avg_us']} us = {c['ms_per_iteration']} ms, frac {c['frac']}" for c in (r.get('candidates') or [])[1:]) + ".",
                  f"* whole iteration: {(r.get('iteration') or {}).get('tflops')} TFLOP/s algorithmic = {(r.get('iteration') or {}).get('frac_fp32_basis')} of the fp32 MFMA peak.", "",
                  "| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    rows_c = list(csv.DictReader(open(ks)))
    for rr in rows_c[:14]:
        nm = rr["Name"].split("(")[0].replace("void ", "")
        cfg_lines.append(f"| `{nm}` | {rr['Calls']} | {float(rr['TotalDurationNs']) / 1e6:.2f} | {float(rr['AverageNs']) / 1e3:.1f} | {float(rr['Percentage']):.2f} |")
    cfg_lines.append("")
if len(cfg_lines) > 4:
    open(os.path.join(DST, f"{tag}_configs_rocprof_summary.md"), "w").write("\n".join(cfg_lines))

# ---- data-parallel rehearsal (one-rank RCCL group, A0_DP_FORCE=1) and the host-environment front-end, when their runs are there
for src, dst in (("bench_dpforce.json", f"{tag}_bench_dqn_dp_rehearsal.json"), ("host_env.json", f"{tag}_host_env_front_end.json"),
                 ("host_env_2groups.json", f"{tag}_host_env_front_end_2groups.json")):
    path = os.path.join(SRC, src)
    if os.path.exists(path):
        try:
            json.dump(last_json(path), open(os.path.join(DST, dst), "w"), indent=1)
        except Exception as e:      # noqa: BLE001
            print("skipped", src, e)

for src, dst in (("suite8.json", f"{tag}_fqf_suite8.json"), ("pmc_encoder.txt", f"{tag}_pmc_encoder.txt"), ("qr_quantile_huber_ubench.json", f"{tag}_qr_quantile_huber_ubench.json"),
                 ("c_host_loop.json", f"{tag}_c_host_loop.json"), ("learning.json", f"{tag}_learning.json")):
    if os.path.exists(os.path.join(SRC, src)):
        shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))
if os.path.exists(os.path.join(ROOT, "gpurun_out", f"{tag}_learning.json")):
    shutil.copy(os.path.join(ROOT, "gpurun_out", f"{tag}_learning.json"), os.path.join(DST, f"{tag}_learning.json"))

# ---- summary table
rows = list(csv.DictReader(open(os.path.join(SRC, "kernel_stats.csv"))))
tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
tot_calls = sum(int(r["Calls"]) for r in rows)
lines = [f"# rocprofv3 --kernel-trace --stats  --  python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry   ({tag}, MI355X gfx950, 1 GPU)", "",
         "Commands (on the GPU box, `tools/refresh_profiles.sh`; this file is generated from their output by `tools/make_profiles.py`): `cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && "
         "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/<round>/prof -- python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry`;",
         "HBM traffic from two further runs, `rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` and `rocprofv3 --pmc WRITE_SIZE --kernel-trace ...` (counters never combined with other trace domains).",
         "The run covers the untimed replay fill (49 actor-only iterations), 2 warm-up, 8 timed iterations of BASELINE configs[1] and the 8 probe iterations (hipGraph replay off).",
         f"Files: `{tag}_bench_dqn_kernel_stats.csv` (full table), `{tag}_bench_dqn.json` (bench line of the un-profiled default run), `{tag}_bench_dqn_launch_entry.json` (`--entry launch`), "
         f"`{tag}_fused_kernels_by_grid.json`, `{tag}_pmc_traffic.json`, `{tag}_other_configs.json`.", "",
         "| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
for r in rows[:26]:
    nm = r["Name"].split("(")[0].replace("void ", "")
    lines.append(f"| `{nm}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
lines += ["", f"Total GPU kernel time: {tot_ns / 1e6:.1f} ms over {tot_calls} launches.", ""]
enc = {k.split("grid=")[1]: v for k, v in by_grid.items() if "a0_encoder_fused_kernel" in k}
multi = [v for k, v in by_grid.items() if "a0_encoder_fused_multi_kernel" in k]
stepk = [(k, v) for k, v in by_grid.items() if "a0_actor_step_enc" in k or "a0_actor_dist_step_enc" in k]
dg = [v for k, v in by_grid.items() if "dgrad" in k]
FLOP = 15.47e6
roof = bench.get("roofline") or {}
PROD = roof.get("bf16_products_per_mac")
lines += [f"## Dominant kernel by measured time: {str(roof.get('kernel'))[:60]}", "",
          f"* bench.py's in-run probe (HIP events carried by the launch: the dispatch's own begin / end timestamps): {roof.get('avg_us')} us average, {roof.get('launches_per_iteration')} launches per iteration = "
          f"{roof.get('share_of_iteration')} of the iteration; {roof.get('algorithmic_gflop_per_launch')} GFLOP algorithmic per launch x {PROD} bf16 products per multiply-add = {roof.get('achieved')} TFLOP/s issued = "
          f"**{roof.get('frac')} of the 2.5 PFLOP/s dense bf16 MFMA peak** (`roofline.frac`); fp32-equivalent {(roof.get('fp32_equivalent') or {}).get('achieved')} TFLOP/s "
          f"({(roof.get('fp32_equivalent') or {}).get('frac')} of the fp32 MFMA peak, secondary)."]
for k, v in stepk:
    lines += [f"* kernel trace of the same command: `{k.split('|')[0]}` {v['avg_us']:.2f} us average over {v['launches']} launches (min {v['min_us']:.2f}, max {v['max_us']:.2f}): the probe's figure and the trace's agree "
              f"to {abs(v['avg_us'] - (roof.get('avg_us') or 0)) / v['avg_us'] * 100:.1f} %."]
st = traffic.get("calibration", {}).get("stepenc")
if st:
    lines += [f"* HBM traffic per launch (PMC, separate passes): read {2 * st['FETCH_SIZE_bytes_raw'] / 1e6:.1f} MB (FETCH_SIZE x 2), write {st['WRITE_SIZE_bytes_raw'] / 1e6:.1f} MB; algorithmic "
              f"{st['algorithmic_read_bytes'] / 1e6:.1f} / {st['algorithmic_write_bytes'] / 1e6:.1f} MB.  At ~28 us that is ~1.5 TB/s: matrix-pipe / issue bound, not HBM bound."]
lines += ["* other probed families of the same run (`roofline.candidates`): " + "; ".join(f"{c['family']} {c['launches_per_iteration']} x {c['avg_us']} us = {c['ms_per_iteration']} ms per iteration, "
                                                                                           f"{c['achieved']} TFLOP/s issued = {c['frac']}" for c in (roof.get('candidates') or [])[1:]) + ".",
          f"* whole iteration (`roofline.iteration`): {(roof.get('iteration') or {}).get('algorithmic_flop', 0) / 1e12:.3f} TFLOP algorithmic in {bench['ms_per_step']} ms = {(roof.get('iteration') or {}).get('tflops')} TFLOP/s = "
          f"{(roof.get('iteration') or {}).get('frac_fp32_basis')} of the fp32 MFMA peak; kernels sum to {(roof.get('iteration') or {}).get('kernel_sum_ms')} ms, their floors to {(roof.get('iteration') or {}).get('kernel_floor_ms')} ms.", ""]
lines += ["## `a0_encoder_fused_multi_kernel` / `a0_encoder_fused_kernel` (conv1 + conv2 + conv3 per observation: the learner's forward passes in one launch of 256 looping workgroups; the rollout's first encoder)", ""]
if multi:
    m = multi[0]
    lines += [f"* kernel trace: learner launch (2 x 512 observations) {m['avg_us']:.2f} us over {m['launches']} launches = {1024 * FLOP / m['avg_us'] / 1e6:.1f} TFLOP/s algorithmic"
              + (f" x {PROD} = {1024 * FLOP * PROD / m['avg_us'] / 1e6:.0f} TFLOP/s issued = {1024 * FLOP * PROD / m['avg_us'] / 1e6 / 2500:.3f} of the bf16 peak" if PROD else "") + "."]
if "131072" in enc:
    a = enc["131072"]
    lines += [f"* kernel trace: 256 observations (a rollout's first step) {a['avg_us']:.2f} us over {a['launches']} launches."]
pl = traffic["per_launch"]
if "1024" in pl:
    lines += [f"* HBM traffic per learner launch (PMC): {pl['1024']['hbm_bytes'] / 1e6:.1f} MB (read {pl['1024']['hbm_read_bytes'] / 1e6:.1f}, write {pl['1024']['hbm_write_bytes'] / 1e6:.1f}; algorithmic minimum "
              f"{pl['1024']['algorithmic_bytes_min'] / 1e6:.1f} MB + 36.8 MB of act1 / act2 kept for the backward pass)."]
lines += ["* all three layers run on `v_mfma_f32_16x16x32_bf16` with fp32 operands as exact sums of three bf16 terms (conv1: bytes x three weight terms, 3 MFMAs per 32 k; conv2 / conv3: six of the nine cross "
          "products of three activation terms x three weight terms by default — `a0_x9_products`, nine in the strict mode), fp32 accumulation.", ""]
if dg:
    d = dg[0]
    lines += ["## `a0_encoder_dgrad_fused_x9_kernel` (conv3 + conv2 data gradients per observation, split operands on the bf16 pipe)", "",
              f"* {d['avg_us']:.1f} us average over {d['launches']} launches of 512 observations = {512 * 12.5e6 / d['avg_us'] / 1e6:.1f} TFLOP/s algorithmic (12.5 MFLOP per observation)"
              + (f"; HBM read {traffic['dgrad_per_launch_512']['hbm_read_bytes'] / 1e6:.1f} MB / write {traffic['dgrad_per_launch_512']['hbm_write_bytes'] / 1e6:.1f} MB per launch "
                 f"(algorithmic {traffic['dgrad_per_launch_512']['algorithmic_bytes']['read'] / 1e6:.1f} / {traffic['dgrad_per_launch_512']['algorithmic_bytes']['write'] / 1e6:.1f} MB)." if "dgrad_per_launch_512" in traffic else "."), ""]
r320 = bench.get("at_reference_update_ratio") or {}
oe = bench.get("other_entry") or {}
cb = bench.get("cpu_baseline") or {}
lines += ["## Bench lines of this build", "",
          f"* `agent0.deepq.main` schedule: {bench['value']:.0f} env-frames/s, {bench['ms_per_step']} ms per iteration (80 x 256 env steps + 20 updates of batch 512), {bench['updates_per_sec']} updates/s; "
          f"at the reference's update:data ratio (learner_steps=320) {r320.get('value')} env-frames/s and {r320.get('updates_per_sec')} updates/s; replay sample+gather {bench.get('replay_sample_GBps')} GB/s; "
          f"CPU port {cb.get('value')} env-frames/s on {cb.get('cores')} host threads.",
          f"* `agent0.deepq.launch` schedule (rollout with a weight snapshot on a second stream while the update block runs): {oe.get('value')} env-frames/s in the same run ({oe.get('ms_per_step')} ms per "
          f"iteration); standalone `--entry launch` run: {launch['value']:.0f}.", ""]
open(os.path.join(DST, f"{tag}_bench_dqn_rocprof_summary.md"), "w").write("\n".join(lines))
# ---- the per-kernel floor tables (tools/budget.py) of the headline configuration and of configs[2], qr, mdqn
import subprocess
parts = ["# Per-kernel floor table (" + tag + ")\n", open(os.path.join(ROOT, "tools", "budget.py")).read().split('"""')[1].split("Bounds")[0].strip().splitlines()[0] + "\n",
         "Bounds: MFMA issue (v_mfma_f32_16x16x32_bf16 every 16 cycles, v_mfma_f32_32x32x16_bf16 every 32 cycles per SIMD = 2.5 PFLOP/s; an exact fp32 product costs 9 bf16 products — 6 as shipped, a0_x9_products — 3 where "
         "one operand is bytes), achievable HBM 6.3 TB/s, one vector wave-instruction per 2 cycles per SIMD, and a launch floor of 2.5 us (ramp-up, first loads, drain); the floor of a kernel is the "
         "larger of its pipe bound and the launch floor.  `gap ms / iteration` = launches x (measured - floor).  Measured = rocprofv3 kernel statistics of the bench command.\n"]
for cfgname, stats, bj in (("dqn", "kernel_stats.csv", "bench.json"), ("c51", "c51_kernel_stats.csv", "c51_bench.json"), ("qr", "qr_kernel_stats.csv", "qr_bench.json"),
                           ("mdqn", "mdqn_kernel_stats.csv", "mdqn_bench.json")):
    if os.path.exists(os.path.join(SRC, stats)):
        ms = last_json(os.path.join(SRC, bj))["ms_per_step"] if os.path.exists(os.path.join(SRC, bj)) else ""
        parts.append(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "budget.py"), os.path.join(SRC, stats), cfgname, str(ms), "--json", os.path.join(DST, f"{tag}_budget.json")],
                                    capture_output=True, text=True).stdout)
open(os.path.join(DST, f"{tag}_budget.md"), "w").write("\n".join(parts))
print("profiles/ refreshed from", SRC)
