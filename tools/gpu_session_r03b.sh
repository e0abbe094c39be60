set -x
mkdir -p gpurun_out/r03
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r03/pytest_gpu_b.txt 2>&1
tail -6 gpurun_out/r03/pytest_gpu_b.txt
python bench.py --no-cpu --no-e2e --no-sink > gpurun_out/r03/bench_refactor.json 2> gpurun_out/r03/bench_refactor.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_refactor.json').read().strip().splitlines()[-1])
print('refactor:', d['value'], d['roofline']['kernel_ms_per_launch'], d['config']['rank_records'][0]['records_checksum_u64'], 'baseline checksum 2191640741019688465')
PY
NINDS="512 513 576 577 640 641 704 768 832 833 896 1024" tools/sweep_variants.sh "default=" > gpurun_out/r03/sweep_dispatch.txt 2>&1
cat gpurun_out/r03/sweep_dispatch.txt
