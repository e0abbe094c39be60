#!/usr/bin/env bash
# round 4: how the drop-in binary's loop depends on the runtime's hardware queues (GPU_MAX_HW_QUEUES, default 4):
#   tools/r04_hwq_ab.sh            the resident run of configs[2] (one context: three busy streams)
#   tools/r04_hwq_ab.sh streamed   the same matrix in row slabs (two alternating contexts) and as three parts on one device
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
if [ "${1:-}" = streamed ]; then
  O=gpurun_out/r04/hw_queues_streamed_ab.txt; : > $O
  for q in 4 8 16 4 8 16; do
    for how in "NGSLD_TEST_SLAB_SITES=20000" "NGSLD_E2E_DEVICES=0,0,0"; do
      E2E_NO_TRACE=1 E2E_ENV="GPU_MAX_HW_QUEUES=$q $how" bash tools/e2e_breakdown.sh > /dev/null 2>&1
      echo "#### GPU_MAX_HW_QUEUES=$q $how" >> $O
      grep -E "real|total" gpurun_out/r04/e2e_breakdown.txt >> $O
    done
  done
  cat $O; exit 0
fi
O=gpurun_out/r04/hw_queues_ab.txt; : > $O
for q in 4 2 8 1 4 8; do
  E2E_NO_TRACE=1 E2E_ENV="GPU_MAX_HW_QUEUES=$q" bash tools/e2e_breakdown.sh > /dev/null 2>&1
  echo "#### GPU_MAX_HW_QUEUES=$q" >> $O
  grep -E "create|pair kernels|total" gpurun_out/r04/e2e_breakdown.txt >> $O
done
cat $O
