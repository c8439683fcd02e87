# PMC passes over the quantile configurations (BASELINE configs[3] iqn, configs[4] fqf): HBM traffic (FETCH_SIZE, WRITE_SIZE) and matrix-pipe activity of their
# dominant kernel family — the dense FORWARD GEMMs a0_igemm_x9_kernel<OpMatKC, OpMatKC, Epi...> that bench.py's probe tags dense_fwd — and of a0_short_k_fwd_kernel.
# One rocprofv3 run per counter group, --kernel-trace only.  Output: gpurun_out/$R/pmc_quantile.json (tools/make_profiles.py puts it into profiles/<R>_pmc_traffic.json)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; R=${R:-r04}
mkdir -p gpurun_out/$R
for algo in ${ALGOS:-iqn fqf}; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    A0_PROBE=none rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/$R/pmcq_${algo}_g$i -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ratio320 --no-other-entry --replay-size 100000 --algo $algo --env Asterix > gpurun_out/$R/pmcq_${algo}_g$i.log 2>&1
  done
done
R=$R python3 - <<'PY'
import csv, glob, collections, json, os, re
R = os.environ["R"]
out = {"source": "rocprofv3 --pmc <group> --kernel-trace (one pass per group) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ratio320 --no-other-entry "
                 "--replay-size 100000 --algo iqn|fqf --env Asterix (tools/pmc_quantile.sh); FETCH_SIZE / WRITE_SIZE in KB, FETCH x2 on gfx950 (see `calibration`)",
       "note": "per-launch means over every launch of the kernel family in the run (actor's 8 192-row and learner's 16 384 / 32 768-row layers, heads); hbm_bytes = 2 * FETCH + WRITE"}
for algo in ("iqn", "fqf"):
    fam = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{R}/pmcq_{algo}_g*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            key = None
            if "a0_igemm_x9_kernel<OpMatKC, OpMatKC, Epi" in k or "a0_igemm_x9_kernel<OpMatKC, OpPlanesKC, Epi" in k:      # (the actor's fc1 reads W as term planes since round 6)
                key = "dense_fwd_gemm"
            elif "a0_short_k_fwd_kernel" in k:
                key = "short_k_fwd"
            if key:
                fam[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for key, d in fam.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        e = {"launches": len(next(iter(d.values()))), "counters_mean": {c: round(v, 1) for c, v in m.items()}}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["hbm_read_bytes"] = 2 * 1024.0 * m["FETCH_SIZE"]; e["hbm_write_bytes"] = 1024.0 * m["WRITE_SIZE"]; e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
            # MFMA-busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md, PMC units): busy share of the kernel's SIMD-cycles
            e["matrix_pipe_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128.0), 4)
            e["matrix_pipe_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)"
        res[key] = e
    out[algo] = res
json.dump(out, open(f"gpurun_out/{R}/pmc_quantile.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -rf gpurun_out/$R/pmcq_*_g?
