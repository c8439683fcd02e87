# round 6: the embedding product's backward in the data-gradient GEMM's epilogue (a0_dense_dgrad_hadamard) — parity tests and same-box A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout -k 10 1000 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_engine.py tests/test_gpu_trainer.py tests/test_gpu_trace.py -m gpu -q -x --timeout 600 -k "embedding_product or iqn or fqf or quant or full_size or handle" > gpurun_out/r06/pytest_had.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06/pytest_had.log
ab() {  # algo env off steps
  if [ $3 = 1 ]; then export A0_NO_DGRAD_HADAMARD=1; else unset A0_NO_DGRAD_HADAMARD; fi
  python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps $4 --warmup 2 --algo $1 --env $2 2> gpurun_out/r06/abh_$1_$3.err | tee gpurun_out/r06/abh_$1_$3.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 separate=$3', d['ms_per_step'], d['value'], d['last_loss'], r['family'], r['avg_us'], r['frac'])"
}
for p in 1 0 1 0; do ab iqn Asterix $p 4; done
for p in 1 0 1 0; do ab fqf Asterix $p 4; done
