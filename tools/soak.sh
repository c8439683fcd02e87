# Soak runs of the agent0.deepq.main entry point on one MI355X: five algorithm families, 256 envs, a 300 k ring that wraps several times,
# target syncs, prioritized sum-tree, NoisyNet, n-step.  Prints the last per-iteration line of each run and the count of NaN / error lines.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/soak
for spec in "dqn 10000000 learner.double_q=true" \
            "c51 6000000 learner.double_q=true learner.dueling_head=true learner.noisy_net=true learner.n_step_q=3 replay.policy=prioritize" \
            "iqn 1500000" "fqf 1500000" "qr 6000000 learner.n_step_q=3"; do
  set -- $spec; a=$1; n=$2; shift 2
  timeout -k 10 400 python -m agent0.deepq.main env_id=Asterix learner.algo=$a actor.num_envs=256 replay.size=300000 trainer.total_steps=$n \
      trainer.training_start_steps=50000 env_task=${TASK:-stream} wandb=false tb=false logdir=/tmp/soak/$a "$@" > /tmp/soak_$a.log 2>&1
  echo "== $a rc=$? lines=$(wc -l < /tmp/soak_$a.log)"
  grep -c "host loop: library handles" /tmp/soak_$a.log | sed 's/^/native-loop lines: /'
  grep "frames:" /tmp/soak_$a.log | tail -1 | cut -c40-230
  echo "nan/error lines: $(grep -ci 'nan\|error\|Traceback' /tmp/soak_$a.log)"
done
true
