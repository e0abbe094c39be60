set -x
mkdir -p gpurun_out/r03
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03/pytest_gpu_e.txt 2>&1
tail -8 gpurun_out/r03/pytest_gpu_e.txt
NINDS="641 704 768 832 896" timeout 900 tools/sweep_variants.sh "default=" "multi=NGSLD_PAIR_KERNEL=multi" "ab=NGSLD_PAIR_KERNEL=ab" > gpurun_out/r03/sweep_ab_vs_multi_ghost.txt 2>&1
cat gpurun_out/r03/sweep_ab_vs_multi_ghost.txt
