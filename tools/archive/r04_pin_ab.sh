#!/usr/bin/env bash
# round 4: pinned buffers as registered huge-page memory (default) against hipHostMalloc (NGSLD_PIN_REGISTER=0):
# phases of the drop-in binary on configs[2], the bench's host-resident leg, and the record-route tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
O=gpurun_out/r04/pin_ab.txt; : > $O
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_text.py tests/test_gpu_replay.py tests/test_gpu_golden.py -q -x 2>&1 | grep -E "passed|failed" >> $O
for v in 1 0 1 0; do  # (1 = registered huge-page memory, opt-in; 0 = hipHostMalloc, the default)
  E2E_NO_TRACE=1 E2E_ENV="NGSLD_PIN_REGISTER=$v" bash tools/e2e_breakdown.sh > /dev/null 2>&1
  echo "#### NGSLD_PIN_REGISTER=$v" >> $O
  grep -E "upload|create|free|real|pair kernels|total" gpurun_out/r04/e2e_breakdown.txt >> $O
done
for v in 1 0; do
  echo "#### bench NGSLD_PIN_REGISTER=$v" >> $O
  NGSLD_PIN_REGISTER=$v python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value %.4e host_resident %.4e ratio %.4f kernel_ms %.2f checksum %d' % (d['value'], d['value_host_resident'], d['value_host_resident']/d['value'], d['roofline']['kernel_ms_per_launch'], d['config']['rank_records'][0]['records_checksum_u64']))" >> $O
done
cat $O
