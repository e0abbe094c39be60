set -x
mkdir -p gpurun_out/r03
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r03/pytest_gpu_c.txt 2>&1
tail -12 gpurun_out/r03/pytest_gpu_c.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r03/prof_prep -o prep -- python bench.py --no-cpu --no-e2e --no-sink --steps 2 --warmup 1 > gpurun_out/r03/bench_prep.json 2> gpurun_out/r03/bench_prep.err
find gpurun_out/r03/prof_prep -name "*kernel_stats*" | head
f=$(find gpurun_out/r03/prof_prep -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-200
rocprofv3 --kernel-trace --stats -d gpurun_out/r03/prof_prep_c4 -o prep -- python bench.py --no-cpu --no-e2e --no-sink --config c4 --sites 60000 --steps 1 --warmup 0 > gpurun_out/r03/bench_prep_c4.json 2> gpurun_out/r03/bench_prep_c4.err
f=$(find gpurun_out/r03/prof_prep_c4 -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_prep.json').read().strip().splitlines()[-1])
print('prep:', d['value'], d['roofline']['kernel_ms_per_launch'], d['config']['rank_records'][0], d['config']['one_off_prep_ms_rank0'], d['config']['pairs_replayed_exact_order_rank0_last_step'])
PY
