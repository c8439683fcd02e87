# round 6, first GPU call: the whole GPU suite on the six-product default, the accuracy record, and same-box A/B of A0_X9_PRODUCTS=9|6 on configs[1], [3], [4]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/check_x6_accuracy.hip -o /tmp/x6 && /tmp/x6 > gpurun_out/r06/x6_accuracy.txt 2> gpurun_out/r06/x6_accuracy.err; tail -3 gpurun_out/r06/x6_accuracy.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r06/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06/pytest_gpu.log
ab() {  # algo env products
  A0_X9_PRODUCTS=$3 python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps $4 --warmup 2 --algo $1 --env $2 2> gpurun_out/r06/ab_$1_$3.err | tee gpurun_out/r06/ab_$1_$3.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 products=$3', d['ms_per_step'], d['value'], d['last_loss'], r['family'], r['avg_us'], r['frac'], [(c['family'], c['avg_us'], c['ms_per_iteration']) for c in r['candidates']])"
}
for p in 9 6 9 6; do ab dqn Breakout $p 20; done
for p in 9 6; do ab iqn Asterix $p 4; ab fqf Asterix $p 4; done
