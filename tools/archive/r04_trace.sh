set -x
mkdir -p gpurun_out/r04
python tools/r04_sink_trace.py > gpurun_out/r04/alt_default.txt 2>&1
NGSLD_REPLAY=0 python tools/r04_sink_trace.py > gpurun_out/r04/alt_noreplay.txt 2>&1
NGSLD_TEST_BATCH_PAIRS=16777216 python tools/r04_sink_trace.py > gpurun_out/r04/alt_batch24.txt 2>&1
NGSLD_TEST_TAIL_LEN=0 python tools/r04_sink_trace.py > gpurun_out/r04/alt_notail.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/trace_sink -o sink -- python $GRAFT_REPO_ROOT/tools/r04_sink_trace.py 100000 500 2 > $GRAFT_REPO_ROOT/gpurun_out/r04/trace_sink.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04/trace_sink/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = open('gpurun_out/r04/trace_sink_summary.txt', 'w')
prev_end = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'][:60]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    out.write(f"{name:60s} dur {(e - s) / 1e3:10.1f} us  gap_before {gap:10.1f} us\n")
    prev_end = e
out.close()
PY
rm -rf gpurun_out/r04/trace_sink
tail -70 gpurun_out/r04/trace_sink_summary.txt
cat gpurun_out/r04/alt_*.txt | grep -v amdgpu
