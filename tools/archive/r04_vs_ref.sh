#!/usr/bin/env bash
# round 4: the binary against the reference's program -- the new test, then a soak
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_vs_ref_program.py -q -x 2>&1 | grep -E "passed|failed|error|Error|differ|hip |ref " | head -40 > gpurun_out/r04/vs_ref_program_test.txt
cat gpurun_out/r04/vs_ref_program_test.txt
timeout 1500 python tools/cli_soak.py 100 700 > gpurun_out/r04/cli_soak.txt 2>&1; tail -12 gpurun_out/r04/cli_soak.txt
