# round 6 evidence, part B: BASELINE configs[2..4] (+ qr, mdqn) with their kernel tables, the roctx marker trace, the six-product accuracy record, the plain-C host
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/check_x6_accuracy.hip -o /tmp/x6 2> /dev/null && /tmp/x6 > gpurun_out/r06/x6_accuracy.txt 2> gpurun_out/r06/x6_accuracy.err; tail -4 gpurun_out/r06/x6_accuracy.txt
R=r06 bash tools/prof_configs.sh 2>&1 | tail -6
A0_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d gpurun_out/r06/roctx -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ratio320 --no-other-entry --replay-size 100000 > gpurun_out/r06/roctx.log 2>&1; echo "roctx rc=$?"
ls gpurun_out/r06/roctx/*/ | head -20
for f in gpurun_out/r06/roctx/*/*marker_api_stats.csv gpurun_out/r06/roctx/*/*marker*stats*.csv; do [ -f "$f" ] && cp "$f" gpurun_out/r06/roctx_marker_stats.csv && head -12 "$f"; done
f=$(ls gpurun_out/r06/roctx/*/*marker_api_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && head -400 "$f" > gpurun_out/r06/roctx_marker_trace_head.csv
rm -rf gpurun_out/r06/roctx
