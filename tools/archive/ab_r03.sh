#!/usr/bin/env bash
# Round 4, same-box A/B: round 3's library (ngsld_amd/ab/libngsld_r03.so, built from commit d2e3e5e) against this round's on the
# BASELINE shapes with and without --ignore_miss_data, and a few cohort sizes in between -- kernel rates (the kernels did not
# change this round: the flag list in write_pair, the odd-site test and the launch tails are what could have moved them).
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_round4_vs_r03.txt
: > $out
for cfg in "c1:--config c1 --steps 20 --warmup 5" "c2:" "c2m:--ignore-miss" "c3:--config c3 --sites 12000" "c3m:--config c3 --sites 12000 --ignore-miss" "c4:--config c4 --sites 40000" "c4m:--config c4 --sites 40000 --ignore-miss" "n64:--sites 60000 --ind 64" "n250:--sites 60000 --ind 250" "n640:--sites 60000 --ind 640" "n768:--sites 50000 --ind 768" "n1500:--config c3 --sites 10000 --ind 1500" "n3000:--config c4 --sites 30000 --ind 3000" "hard:--hard-calls"; do
  name=${cfg%%:*}; args=${cfg#*:}
  echo "=== $name ($args)" >> $out
  ROUNDS=2 BENCH_ARGS="--no-cpu --no-sink --no-e2e --no-traffic $args" timeout 900 tools/ab.sh "r03=NGSLD_LIB=$PWD/ngsld_amd/ab/libngsld_r03.so" "r04=NGSLD_X=0" >> $out 2>&1
done
cat $out
