# timing-only builds of the fused encoder (tools/build_experiment.sh ... encoder_timing_experiments.patch) against the default build: tools/ubench_encoder_fwd.py, one call
cd $GRAFT_REPO_ROOT
python3 tools/ubench_encoder_fwd.py 2>/dev/null | tail -1
for v in "$@"; do python3 tools/with_lib.py tools/variants/libagent0_hip_$v.so tools/ubench_encoder_fwd.py 2>/dev/null | tail -1; done
python3 tools/ubench_encoder_fwd.py 2>/dev/null | tail -1
