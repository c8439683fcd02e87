# kernel-trace gap analysis of the Python host (hipGraphs): where the GPU idles between kernels in the timed iterations
# (the plain C host under rocprofv3 --kernel-trace did not finish within seven minutes on this pool: not profiled)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; R=${R:-r04}
mkdir -p gpurun_out/$R
CFG=${CFG:-1}
if [ $CFG = 2 ]; then EXTRA="--algo c51 learner.double_q=true learner.dueling_head=true learner.noisy_net=true learner.n_step_q=3 replay.policy=prioritize"; else EXTRA=""; fi
A0_PROBE=none timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$R/gap_py -- python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 10 --warmup 3 --replay-size 100000 $EXTRA > gpurun_out/$R/gap_py.log 2>&1 || exit 1
echo "== Python host, config $CFG"; python3 tools/gap_analysis.py $(ls gpurun_out/$R/gap_py/*/*kernel_trace.csv | head -1) ${SKIP:-0.62}
rm -rf gpurun_out/$R/gap_py
