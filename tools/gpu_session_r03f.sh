set -x
mkdir -p gpurun_out/r03
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03/pytest_gpu_f.txt 2>&1
tail -6 gpurun_out/r03/pytest_gpu_f.txt
timeout 700 python tools/fuzz_soak.py 10060 10900 > gpurun_out/r03/fuzz_soak_r03_newshapes.txt 2>&1
tail -3 gpurun_out/r03/fuzz_soak_r03_newshapes.txt
timeout 500 python tools/fuzz_soak.py 400 3400 > gpurun_out/r03/fuzz_soak_r03.txt 2>&1
tail -3 gpurun_out/r03/fuzz_soak_r03.txt
timeout 900 bash tools/bench_configs.sh gpurun_out/r03/configs_r03_mid.jsonl > gpurun_out/r03/configs_r03_mid.txt 2>&1
cat gpurun_out/r03/configs_r03_mid.txt
