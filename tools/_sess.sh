mkdir -p gpurun_out/r03/late3
O=gpurun_out/r03/late3
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_final.txt 2>&1
grep -E "passed|failed|parity:|rror" $O/pytest_gpu_final.txt | tail -8
