#!/bin/bash
# round 5, second measurement: uncalled input after the memoised store build; the new suite entries; per-shard times; e2e on uncalled input
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_b; mkdir -p $O
python -m pytest tests/test_gpu_zz_throughput.py -q -k uncalled -s 2>&1 | tail -8 > $O/floors.txt
python -m pytest tests/test_gpu_every_pair.py -q -k "whole_table" -s 2>&1 | tail -8 > $O/whole_table.txt
python tools/shard_times.py c3 c4 > $O/shard_times.json 2> $O/shard_times.err
python tools/e2e_uncalled.py > $O/e2e_uncalled.json 2> $O/e2e_uncalled.err
tail -5 $O/*.txt $O/*.err
