set -x
mkdir -p gpurun_out/r03
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03/pytest_gpu_d.txt 2>&1
tail -8 gpurun_out/r03/pytest_gpu_d.txt
timeout 600 python bench.py --no-cpu --no-e2e --no-sink > gpurun_out/r03/bench_ghost.json 2> gpurun_out/r03/bench_ghost.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_ghost.json').read().strip().splitlines()[-1])
print('ghost:', d['value'], d['roofline']['kernel_ms_per_launch'], d['config']['rank_records'][0]['records_checksum_u64'], d['roofline']['traffic'], d['roofline']['traffic_per_pair'], d['roofline']['traffic_detail'])
PY
NINDS="500 577 640 1000 1153 1200 1280 2000 2305 2560 4000 4609 5120" timeout 1500 tools/sweep_variants.sh "default=" "slots10=NGSLD_SLOTS10=1" > gpurun_out/r03/sweep_ghost.txt 2>&1
cat gpurun_out/r03/sweep_ghost.txt
