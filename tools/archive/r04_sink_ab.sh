#!/usr/bin/env bash
# Round 4, same-box A/B of how record batches reach the host (ngsld_run without text output): one / two compute streams,
# tapered batches, D2H copies or direct writes into the pinned host buffers.  Each line: tools/sink_probe.py on configs[2].
set -x
out=gpurun_out/r04/sink_ab.txt
mkdir -p gpurun_out/r04
: > $out
for v in "1 0 0" "1 1 0" "2 1 0" "2 0 0" "2 0 1" "1 0 1"; do
  set -- $v
  echo "== NGSLD_TEST_RUN_STREAMS=$1 NGSLD_TEST_RUN_TAPER=$2 NGSLD_TEST_RUN_DIRECT=$3" >> $out
  NGSLD_TEST_RUN_STREAMS=$1 NGSLD_TEST_RUN_TAPER=$2 NGSLD_TEST_RUN_DIRECT=$3 python tools/sink_probe.py 2>&1 | grep -v amdgpu.ids >> $out
done
NGSLD_TRACE=1 NGSLD_TEST_RUN_STREAMS=2 NGSLD_TEST_RUN_TAPER=1 python tools/sink_probe.py 2>&1 | grep -v amdgpu.ids | head -80 > gpurun_out/r04/sink_trace_2s_taper.txt
NGSLD_TRACE=1 NGSLD_TEST_RUN_STREAMS=2 NGSLD_TEST_RUN_DIRECT=1 python tools/sink_probe.py 2>&1 | grep -v amdgpu.ids | head -80 > gpurun_out/r04/sink_trace_2s_direct.txt
python bench.py --steps 5 --warmup 2 --no-e2e --no-traffic --no-cpu > gpurun_out/r04/bench_b1.json 2> gpurun_out/r04/bench_b1.err
python -m pytest tests/test_gpu_replay.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_text.py -x -q -m gpu > gpurun_out/r04/pytest_b1.txt 2>&1
tail -5 gpurun_out/r04/pytest_b1.txt
NGSLD_TEST_RUN_DIRECT=1 python -m pytest tests/test_gpu_replay.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r04/pytest_b1_direct.txt 2>&1
tail -5 gpurun_out/r04/pytest_b1_direct.txt
cat $out
