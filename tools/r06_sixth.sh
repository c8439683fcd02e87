cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
cat > /tmp/t.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from agent0_amd.ops import HipOps
hip = HipOps()
N, K = 512, 3136
for R in (8192, 16384):
    X = torch.randn(R * K, device="cuda").clamp_min(0); W = torch.randn(N * K, device="cuda") * 0.02; b = torch.zeros(N, device="cuda"); Y = torch.empty(R * N, device="cuda")
    sc = torch.empty(4, device="cuda")
    for _ in range(3): hip.dense_fwd(X, K, W, b, Y, R, N, K, True, sc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): hip.dense_fwd(X, K, W, b, Y, R, N, K, True, sc)
    e1.record(); torch.cuda.synchronize()
    print(f"two_wg={os.environ.get('A0_X9_TWO_WG','0')} rows {R}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  checksum {float(Y.double().sum()):.6e}", flush=True)
PY
for k in 0 1 0 1; do A0_X9_TWO_WG=$k python3 /tmp/t.py 2>&1 | grep rows; done
ab() {
  A0_X9_TWO_WG=$3 python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 --algo $1 --env $2 2> gpurun_out/r06/ab2_$1_$3.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 two_wg=$3', d['ms_per_step'], d['value'], d['last_loss'], r['family'], r['avg_us'], r['frac'])"
}
for p in 0 1 0 1; do ab iqn Asterix $p; done
