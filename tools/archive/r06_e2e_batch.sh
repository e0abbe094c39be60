#!/usr/bin/env bash
# the binary on un-called configs[2] (20 % monomorphic sites): text batch size x which replay kernel takes the batches
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r06_e2e
for v in "b19" "b21:NGSLD_TEXT_BATCH_PAIRS=2097152" "b21L:NGSLD_TEXT_BATCH_PAIRS=2097152 NGSLD_REPLAY_LANES_FROM=1048576" "b22:NGSLD_TEXT_BATCH_PAIRS=4194304" "b22L:NGSLD_TEXT_BATCH_PAIRS=4194304 NGSLD_REPLAY_LANES_FROM=1048576" "b23L:NGSLD_TEXT_BATCH_PAIRS=8388608 NGSLD_REPLAY_LANES_FROM=1048576"; do
  name=${v%%:*}; envs=""; [ "$v" != "$name" ] && envs=${v#*:}
  env $envs E2E_ONLY=${E2E_ONLY:-mono20} timeout 500 python tools/e2e_uncalled.py > gpurun_out/r06_e2e/e2e_$name.json 2>gpurun_out/r06_e2e/err_$name.txt
  python -c "
import json; d=json.load(open('gpurun_out/r06_e2e/e2e_$name.json')); print('$name', '$envs', {k:(v['seconds']) for k,v in d['runs'].items()})" | tee -a gpurun_out/r06_e2e/batch_ab.txt
done
