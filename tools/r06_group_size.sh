#!/usr/bin/env bash
# the binary on un-called configs[2] (20 % monomorphic sites): pairs per group of text batches (engine_run.hip, run_grouped)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r06_e2e
for g in ${SIZES:-4194304 8388608 16777216 33554432 67108864}; do
  NGSLD_TEST_TEXT_GROUP_PAIRS=$g E2E_ONLY=${E2E_ONLY:-mono20} timeout 500 python tools/e2e_uncalled.py > gpurun_out/r06_e2e/e2e_g$g.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r06_e2e/e2e_g$g.json')); print('group $g', {k:(v['seconds']) for k,v in d['runs'].items()})" | tee -a gpurun_out/r06_e2e/group_size_ab.txt
done
grep -v "chunk\|^\[trace\] batch" gpurun_out/r05_e2e/trace_mono20.txt | grep "trace" | tail -20
