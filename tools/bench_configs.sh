# BASELINE.json configs[2..4] at full size on one GPU (parity-test cases, not the headline bench line): throughput for the record.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${R:-r03}
python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 --algo c51 learner.double_q=true learner.dueling_head=true learner.noisy_net=true learner.n_step_q=3 replay.policy=prioritize > gpurun_out/${R:-r03}/cfg2_c51_rainbow.json 2> gpurun_out/${R:-r03}/cfg2.err
python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 --algo iqn --env Asterix > gpurun_out/${R:-r03}/cfg3_iqn_asterix.json 2> gpurun_out/${R:-r03}/cfg3.err
python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 --algo fqf --env Asterix > gpurun_out/${R:-r03}/cfg4_fqf_asterix.json 2> gpurun_out/${R:-r03}/cfg4.err
python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 --algo qr > gpurun_out/${R:-r03}/cfg_qr.json 2> gpurun_out/${R:-r03}/cfgqr.err
python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 4 --warmup 2 --algo mdqn > gpurun_out/${R:-r03}/cfg_mdqn.json 2> gpurun_out/${R:-r03}/cfgmdqn.err
for f in cfg2_c51_rainbow cfg3_iqn_asterix cfg4_fqf_asterix cfg_qr cfg_mdqn; do python3 -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/${R:-r03}/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['updates_per_sec'], d['last_loss'])
except Exception as e:
    print('$f FAILED', e)
"; done

