cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for i in 1 2 3; do
  A0_NATIVE_LOOP=0 python3 - > gpurun_out/r06/det_$i.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import learning_runs as LR, torch, hashlib
r = LR.run("dqn", {"replay.policy": "prioritize"}, 1_300_000, task="chase")
print("curve", r["curve"], r["loss"], r["qmax"])
PY
  grep curve gpurun_out/r06/det_$i.log
done
