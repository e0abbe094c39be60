mkdir -p gpurun_out/r03
( NINDS="2049 2304 4097 4608" timeout 900 bash tools/sweep_variants.sh "now=" "abm=NGSLD_PAIR_KERNEL=abm" ) > gpurun_out/r03/sweep_abm3.txt 2>&1
