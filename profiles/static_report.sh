#!/usr/bin/env bash
# Static evidence for DESIGN.md §4: resource usage of the shipped pair kernels and the instruction mix of the
# headline kernel's EM loop (hipcc cross-compiles; no GPU needed).  Usage: profiles/static_report.sh > profiles/r01/static_report.txt
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cat > $T/k.hip <<EOT
#include "$R/ngsld_amd/csrc/ld_device.h"
template __global__ void ngsld::pair_ld_run_kernel<8,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_run_kernel<8,true>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_pf_kernel<8,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_group_kernel<8,3,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_group_kernel<16,7,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_group_kernel<32,7,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<8,2,false,true>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<8,4,false,true>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_stream_kernel<false>(ngsld::PairArgs);
EOT
cd $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c k.hip -o k.o -save-temps \
  -Rpass-analysis=kernel-resource-usage 2> res.txt || true
echo "== kernel resources (hipcc -Rpass-analysis=kernel-resource-usage, gfx950) =="
grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" res.txt | sed 's/remark:[^:]*:[0-9]*:[0-9]*: *//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' \
  | sed 's/_ZN5ngsld//; s/EvNS_8PairArgsE//'
S=k-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN5ngsld18pair_ld_run_kernelILi8ELb0EEEvNS_8PairArgsE:/,/s_endpgm/' $S > pf.s
L=$(grep -n "Inner Loop Header: Depth=2" pf.s | tail -1 | cut -d: -f1)
# the hot path of one EM iteration = the loop header block up to its first branch (the shared-reciprocal step and its
# reduction); what follows in the loop is the single-reciprocal redo path and the convergence bookkeeping
E=$(awk -v s=$L 'NR>s && /s_cbranch/ {print NR; exit}' pf.s)
X=$(awk -v s=$L 'NR>s && /^\.LBB0_[0-9]+:.*Depth=1$/ {print NR; exit}' pf.s)
echo
echo "== pair_ld_run_kernel<8,false>: instruction mix of the hot block of one EM iteration (ISA lines $L..$E) =="
sed -n "${L},${E}p" pf.s | grep -v "^\s*;" | grep -v "^\." | awk '{print $1}' | sort | uniq -c | sort -rn
echo
echo "== same kernel: whole EM loop incl. the redo path and the convergence test (ISA lines $L..$X) =="
sed -n "${L},${X}p" pf.s | grep -v "^\s*;" | grep -v "^\." | awk '{print $1}' | sort | uniq -c | sort -rn
rm -rf $T
