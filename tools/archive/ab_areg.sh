#!/usr/bin/env bash
# Same-box: the a/b kernel (row vector in registers where it fits, AbRegs in ld_pair_ab.hip) against two wavefronts per pair
# around the upper end of its range.   tools/ab_areg.sh > gpurun_out/sweep_ab_range.txt
NINDS="${NINDS:-641 704 768 832 833 896 897 960}" tools/sweep_variants.sh "default=" "multi=NGSLD_PAIR_KERNEL=multi" "ab=NGSLD_PAIR_KERNEL=ab"
