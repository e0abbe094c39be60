#!/bin/bash
# One GPU call for the shared-reciprocal quotients of the replay kernels (ld_replay_lkl.hip: div_operand_plain):
#   1. the GPU suite on the new library (tests/test_gpu_replay_lkl.py holds device against host replay bit for bit);
#   2. same-box A/B of bench.py on un-called input against the library of the tree before (ngsld_amd/ab/libngsld_6c35aec.so);
#   3. the new test on the OLD library (were it to fail under 1: the kernel or the test?);
#   4. the driver's line.
# Output under gpurun_out/shared_rcp/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/shared_rcp; mkdir -p $O
export OLD_LIB=$PWD/ngsld_amd/ab/libngsld_6c35aec.so
( time python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=8 ) > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
tools/r05_ab_value.sh "--mono-frac 0.2" "--sfs" > $O/ab_value.txt 2>&1
cat $O/ab_value.txt
NGSLD_LIB=$OLD_LIB python -m pytest tests/test_gpu_replay_lkl.py -q -p no:cacheprovider -k plain_divisions > $O/new_test_on_old_lib.txt 2>&1
tail -3 $O/new_test_on_old_lib.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json
