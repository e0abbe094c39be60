set -x
mkdir -p gpurun_out/r03
A=$PWD/ngsld_amd/ab
for cfg in "c1:--config c1" "c2:" "c2m:--ignore-miss" "c3:--config c3 --sites 12000" "c3m:--config c3 --sites 12000 --ignore-miss" "c4:--config c4 --sites 40000" "c4m:--config c4 --sites 40000 --ignore-miss"; do
  name=${cfg%%:*}; args=${cfg#*:}
  echo "=== $name ($args)"
  ROUNDS=2 BENCH_ARGS="--no-cpu --no-sink --no-e2e --no-traffic $args" timeout 900 tools/ab.sh "r02=NGSLD_LIB=$A/libngsld_r02.so" "ghost=NGSLD_LIB=$A/libngsld_prev6702.so" "now="
done > gpurun_out/r03/ab_round3_vs_r02.txt 2>&1
cat gpurun_out/r03/ab_round3_vs_r02.txt
