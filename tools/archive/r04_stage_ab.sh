#!/usr/bin/env bash
# round 4: staging-buffer size of the matrix upload (NGSLD_TEST_STAGE_BYTES), phases of the drop-in binary on configs[2]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
: > gpurun_out/r04/e2e_stage_ab.txt
for sb in 268435456 67108864 33554432 268435456 67108864 33554432; do
  E2E_NO_TRACE=1 E2E_ENV="NGSLD_TRACE=1 NGSLD_TEST_STAGE_BYTES=$sb" bash tools/e2e_breakdown.sh > /dev/null 2>&1
  echo "#### NGSLD_TEST_STAGE_BYTES=$sb" >> gpurun_out/r04/e2e_stage_ab.txt
  grep -E "set_geno|upload|create|free|real|pair kernels" gpurun_out/r04/e2e_breakdown.txt | grep -v "chunk [1-9]" >> gpurun_out/r04/e2e_stage_ab.txt
done
cat gpurun_out/r04/e2e_stage_ab.txt
