#!/usr/bin/env bash
# Round 4, same-box A/B of the launch tails: the last pairs of every launch cut into short runs (build_runs).
# bench lines (run_device: one launch) and the host-resident leg (ngsld_run: 12 launches, direct writes) per setting.
set -x
out=gpurun_out/r04/tail_ab.txt
mkdir -p gpurun_out/r04
: > $out
B="python bench.py --steps 4 --warmup 1 --no-e2e --no-traffic --no-cpu"
for v in "0 0" "2 262144" "1 262144" "2 131072" "2 524288" "4 262144" "1 524288"; do
  set -- $v
  echo "== NGSLD_TEST_TAIL_LEN=$1 NGSLD_TEST_TAIL_PAIRS=$2" >> $out
  NGSLD_TEST_TAIL_LEN=$1 NGSLD_TEST_TAIL_PAIRS=$2 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.4e ms_per_step %.2f kernel_ms %.2f host_resident %.4e ratio %.4f checksum %d replayed %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['value_host_resident'], d['value_host_resident']/d['value'], d['config']['rank_records'][0]['records_checksum_u64'], d['config']['pairs_replayed_exact_order_rank0_last_step']))" >> $out
done
echo "== default, copy mode (NGSLD_TEST_RUN_DIRECT=0)" >> $out
NGSLD_TEST_RUN_DIRECT=0 python tools/sink_probe.py 2>&1 | grep -v amdgpu.ids | grep "pass [12]" >> $out
echo "== default" >> $out
python tools/sink_probe.py 2>&1 | grep -v amdgpu.ids | grep "pass [12]" >> $out
cat $out
python -m pytest tests/test_gpu_prep.py tests/test_gpu_replay.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_text.py tests/test_gpu_stream.py tests/test_gpu_launch_chunks.py tests/test_gpu_run_kernel.py -x -q -m gpu > gpurun_out/r04/pytest_b2.txt 2>&1
tail -5 gpurun_out/r04/pytest_b2.txt
