#!/usr/bin/env bash
# Same-box counter comparison of bench variants: tools/pmc_compare.sh "<env assignments> -- <bench args>" ...
# (env part may be empty: " -- --sites 30000 --ind 1000").  Output: gpurun_out/pmc_compare.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${PMC_OUT:-pmc_compare.txt}
mkdir -p $(dirname $OUT); : > $OUT
for V in "$@"; do
  ENVS=${V%% -- *}; ARGS=${V#* -- }
  echo "=== env [$ENVS] bench.py $ARGS" >> $OUT
  env $ENVS python $R/bench.py --no-cpu --no-sink --no-e2e --no-traffic --steps 2 --warmup 1 $ARGS 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('pairs/s %.4g  ms %.2f  iters %.3f  frac %.3f  valu %.3f  kernel %s' % (d['value'], d['roofline']['kernel_ms_per_launch'], d['config']['mean_executed_em_iterations'], d['roofline']['frac'], d['roofline']['fp64_valu']['frac'], d['roofline']['kernel']), d['config']['pairs_per_step'])" >> $OUT
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" ${PMC_EXTRA:+"$PMC_EXTRA"}; do
    env $ENVS bash $R/tools/pmc_once.sh "$SET" --no-sink --no-e2e $ARGS >> $OUT 2>&1
  done
done
cat $OUT
