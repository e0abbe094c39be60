mkdir -p gpurun_out/r04/late
timeout 2400 python tools/fuzz_soak.py 20000 40000 > gpurun_out/r04/late/fuzz_soak_r04_long.txt 2>&1; tail -1 gpurun_out/r04/late/fuzz_soak_r04_long.txt
timeout 900 python tools/text_soak.py 1000 2500 > gpurun_out/r04/late/text_soak_r04.txt 2>&1; tail -1 gpurun_out/r04/late/text_soak_r04.txt
