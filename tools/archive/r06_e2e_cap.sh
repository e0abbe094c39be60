#!/usr/bin/env bash
# the binary on un-called configs[2]: the lanes on text batches with a cap on a lane's EM steps (the rest handed to the wavefront kernel)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r06_e2e
for v in "base" "c4w1:NGSLD_REPLAY_LANES_FROM=0 NGSLD_TEST_LANE_ITER_CAP=4 NGSLD_LANE_WAVES=1" "c6w1:NGSLD_REPLAY_LANES_FROM=0 NGSLD_TEST_LANE_ITER_CAP=6 NGSLD_LANE_WAVES=1" "c4w3:NGSLD_REPLAY_LANES_FROM=0 NGSLD_TEST_LANE_ITER_CAP=4 NGSLD_LANE_WAVES=3" "c6w3:NGSLD_REPLAY_LANES_FROM=0 NGSLD_TEST_LANE_ITER_CAP=6 NGSLD_LANE_WAVES=3" "c8w3:NGSLD_REPLAY_LANES_FROM=0 NGSLD_TEST_LANE_ITER_CAP=8 NGSLD_LANE_WAVES=3" "c5w2:NGSLD_REPLAY_LANES_FROM=0 NGSLD_TEST_LANE_ITER_CAP=5 NGSLD_LANE_WAVES=2"; do
  name=${v%%:*}; envs=""; [ "$v" != "$name" ] && envs=${v#*:}
  env $envs E2E_ONLY=${E2E_ONLY:-mono20} timeout 500 python tools/e2e_uncalled.py > gpurun_out/r06_e2e/e2e_$name.json 2>gpurun_out/r06_e2e/err_$name.txt
  python -c "
import json; d=json.load(open('gpurun_out/r06_e2e/e2e_$name.json')); print('$name', '$envs', {k:(v['seconds']) for k,v in d['runs'].items()})" | tee -a gpurun_out/r06_e2e/cap_ab.txt
done
