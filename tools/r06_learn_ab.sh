cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for p in 9 6; do
  A0_X9_PRODUCTS=$p A0_NATIVE_LOOP=0 python3 tests/learning_runs.py dqn_prio > gpurun_out/r06/learn_prio_$p.log 2>&1; grep "curve" gpurun_out/r06/learn_prio_$p.log | sed "s/^/products=$p /"
done
A0_X9_PRODUCTS=6 A0_NATIVE_LOOP=0 python3 tests/learning_runs.py dqn > gpurun_out/r06/learn_dqn_6.log 2>&1; grep "curve" gpurun_out/r06/learn_dqn_6.log | sed "s/^/dqn products=6 /"
