# timing-only builds of the split-operand GEMM (tools/build_experiment.sh ... x9_limiter_timing.patch) on the SMALL layers of the dqn iteration: the actor's fc1 GEMM + tail at
# 256 / 512 rows (tools/ubench_actor_tail.py) and the whole B = 512 update (tools/ubench_update.py), default build first and last
cd $GRAFT_REPO_ROOT
run() { if [ "$1" = default ]; then python3 "${@:2}" 2>/dev/null | tail -${N:-2}; else python3 tools/with_lib.py tools/variants/libagent0_hip_$1.so "${@:2}" 2>/dev/null | tail -${N:-2}; fi; }
for v in default "$@" default; do echo "== $v"; N=2 run $v tools/ubench_actor_tail.py; N=1 run $v tools/ubench_update.py; done
