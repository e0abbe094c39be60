#!/usr/bin/env bash
# Same-box A/B: from how many individuals per lane on does masking converged groups off pay?  default = 5; variants 3, 4,
# never (-DNGSLD_MASK_DONE=0).
A=$PWD/ngsld_amd/ab
for shape in "--ind 24" "--ind 32" "--ind 40" "--ind 48" "--ind 72"; do
  echo "== $shape"
  BENCH_ARGS="--no-cpu --no-sink --no-e2e --config c1 --sites 12000 --steps 5 --warmup 2 $shape" ROUNDS=2 tools/ab.sh "from5=X=1" "from4=NGSLD_LIB=$A/libngsld_mask4.so" "from3=NGSLD_LIB=$A/libngsld_mask3.so" "never=NGSLD_LIB=$A/libngsld_nomask.so"
done
