# PMC passes over one kernel of one Python script: one rocprofv3 run per counter group, kernel-trace only (never combined with other trace domains).
# usage: bash tools/pmc_script.sh <kernel-name-substring> <script.py> [args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pat=$1; shift
out=gpurun_out/${R:-r03}/pmc_$pat
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -- python3 "$@" > $out/g$i.log 2>&1
done
python3 - $out $pat <<'PY'
import csv, glob, collections, sys, json
tot = {}
for g in sorted(glob.glob(sys.argv[1] + "/g*/")):
    for f in glob.glob(g + "*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if sys.argv[2] in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in sorted(acc.items()):
            tot.setdefault(k, {}).update({c: round(sum(v) / len(v)) for c, v in d.items()})
for k, d in tot.items():
    print(k, json.dumps(d))
json.dump(tot, open(sys.argv[1] + ".json", "w"), indent=1)
PY
rm -rf $out
