set -x
mkdir -p gpurun_out/r03
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03/pytest_gpu_i.txt 2>&1
tail -6 gpurun_out/r03/pytest_gpu_i.txt
A=$PWD/ngsld_amd/ab
NINDS="1000 1025 1100 1152 1153 1280 2000 2304 2305 2560 4608 5120" timeout 1500 tools/sweep_variants.sh "r02=NGSLD_LIB=$A/libngsld_r02.so" "now=" > gpurun_out/r03/sweep_bycount.txt 2>&1
cat gpurun_out/r03/sweep_bycount.txt
