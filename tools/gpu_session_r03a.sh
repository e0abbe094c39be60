set -x
mkdir -p gpurun_out/r03
rocminfo | grep -c "gfx950" > gpurun_out/r03/ngpu.txt
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r03/pytest_gpu_a.txt 2>&1
tail -5 gpurun_out/r03/pytest_gpu_a.txt
python bench.py > gpurun_out/r03/bench_base.json 2> gpurun_out/r03/bench_base.err
tail -c 600 gpurun_out/r03/bench_base.json
tools/sweep_variants.sh "default=" "ab=NGSLD_PAIR_KERNEL=ab" "run10=NGSLD_RUN_SLOTS=10" > gpurun_out/r03/sweep_513_1024.txt 2>&1
cat gpurun_out/r03/sweep_513_1024.txt
ROUNDS=3 BENCH_ARGS="--no-cpu --no-sink --no-e2e" tools/ab.sh "default=" "nosetprio=NGSLD_LIB=$PWD/ngsld_amd/ab/libngsld_nosetprio.so" > gpurun_out/r03/ab_setprio.txt 2>&1
cat gpurun_out/r03/ab_setprio.txt
