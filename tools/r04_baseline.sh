set -x
mkdir -p gpurun_out/r04
python tools/sink_probe.py > gpurun_out/r04/sink_probe_base.txt 2>&1
NGSLD_TRACE=1 python tools/sink_probe.py > gpurun_out/r04/sink_probe_trace.txt 2>&1
python bench.py --steps 5 --warmup 2 --no-e2e --no-traffic --cpu-seconds 4 > gpurun_out/r04/bench_base.json 2> gpurun_out/r04/bench_base.err
tail -c 1500 gpurun_out/r04/bench_base.json
