#!/usr/bin/env bash
# Round 4 -- the final tree once more (after the last engine / CLI changes): GPU suite, smoke, the driver's bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04/${LATE_DIR:-late}; mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_late.txt 2>&1
grep -E "passed|failed|parity:" $O/pytest_gpu_late.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r04_late.json 2> $O/bench_r04_late.err
python - $O/bench_r04_late.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('value %.4e ms_per_step %.2f kernel %.2f host_resident %.4e ratio %.4f frac %.4f e2e %.3f checksum %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['value_host_resident'], d['value_host_resident'] / d['value'], d['roofline']['frac'], d['e2e_file_to_tsv_s']['seconds'], d['config']['rank_records'][0]['records_checksum_u64']))
PY
