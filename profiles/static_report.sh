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
template __global__ void ngsld::pair_ld_run_kernel<9,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_run_kernel<10,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_run_kernel<10,true>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_group_kernel<8,3,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_group_kernel<16,7,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_group_kernel<32,7,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_stream_kernel<false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_bres_kernel<12,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_bres_kernel<20,false>(ngsld::PairArgs);
EOT
# the multi-wavefront kernels are built with the scheduler ld_pair_wn.hip is built with (csrc/Makefile: FLAGS_ld_pair_wn)
cat > $T/kw.hip <<EOT
#include "$R/ngsld_amd/csrc/ld_device.h"
template __global__ void ngsld::pair_ld_kernel<6,4,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<6,4,true>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<7,4,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<8,2,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<8,2,true>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<8,4,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<8,4,true>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<8,8,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<10,4,false>(ngsld::PairArgs);
template __global__ void ngsld::pair_ld_kernel<10,8,false>(ngsld::PairArgs);
EOT
# the a/b-form kernels live in their translation unit (ld_pair_ab.hip): compiled as it is built, a few shapes picked from its report
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c $R/ngsld_amd/csrc/ld_pair_ab.hip -o $T/ab.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 || true ) | grep -A9 -E "Function Name: .*(pair_ld_abm_kernelILi(9|12|13|15)ELi(2|4|8)ELb0|pair_ld_ab_kernelILi(12|15)ELb0)" \
  | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" > $T/res_ab.txt || true
cd $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c k.hip -o k.o -save-temps \
  -Rpass-analysis=kernel-resource-usage 2> res.txt || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-sched-strategy=iterative-ilp -c kw.hip -o kw.o \
  -Rpass-analysis=kernel-resource-usage 2>> res.txt || true
echo "== kernel resources (hipcc -Rpass-analysis=kernel-resource-usage, gfx950) =="
cat res.txt res_ab.txt | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/^[^ ]*:[0-9]*:[0-9]*: remark: *//; s/remark:[^:]*:[0-9]*:[0-9]*: *//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' \
  | sed 's/_ZN5ngsld//; s/EvNS_8PairArgsE//'
S=k-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN5ngsld18pair_ld_run_kernelILi8ELb0EEEvNS_8PairArgsE:/,/s_endpgm/' $S > pf.s
echo
echo "== pair_ld_run_kernel<8,false>: instruction mix of the hot block of one EM iteration =="
echo "   (the basic block with the most f64 FMAs: the shared-reciprocal step and its reduction, up to the sanity branch;"
echo "    the convergence test that follows is another ~12 VALU instructions on its common path)"
python3 - pf.s <<'PY'
import re, sys
from collections import Counter
blocks, cur = [], []
for l in open(sys.argv[1]):
    if re.match(r'^\.LBB', l) or 's_cbranch' in l or 's_branch' in l:
        if cur:
            blocks.append(cur)
        cur = []
    else:
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.'):
            cur.append(t.split()[0])
if cur:
    blocks.append(cur)
hot = max(blocks, key=lambda b: sum(1 for x in b if x.startswith('v_fma')))
for k, v in sorted(Counter(hot).items(), key=lambda kv: -kv[1]):
    print(f"{v:7d} {k}")
print(f"{len(hot):7d} instructions in the block")
PY
rm -rf $T
