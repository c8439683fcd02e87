# PMC passes over the fused encoder (tools/ubench_encoder.py): one rocprofv3 run per counter group, kernel-trace only.
# usage: bash tools/pmc_encoder.sh [stage_mask]   (stage mask -> A0_FUSED_STAGES, default 7 = all three convolutions)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export A0_FUSED_STAGES=${1:-7}
out=gpurun_out/pmc_enc_$A0_FUSED_STAGES
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -- python3 tools/ubench_encoder.py > $out/g$i.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, collections, sys
for g in sorted(glob.glob(sys.argv[1] + "/g*/")):
    for f in glob.glob(g + "*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "a0_encoder_fused_kernel" in r["Kernel_Name"]:
                acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for gs, d in sorted(acc.items()):
            if gs == "131072":
                print("stages", sys.argv[1][-1], "grid", gs, {k: round(sum(v) / len(v)) for k, v in d.items()})
PY
