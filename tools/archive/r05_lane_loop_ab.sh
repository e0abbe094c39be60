#!/bin/bash
# One GPU call for the lane-per-pair replay's individual loop without its register copies (ld_replay_lkl.hip):
#   candidates  ngsld_amd/ab/libngsld_v2.so  one register set, the next individual fetched into it between the step's two phases
#               ngsld_amd/ab/libngsld_v1.so  two register sets used in turn (142 registers: three wavefronts to a SIMD)
#   baseline    ngsld_amd/ab/libngsld_rcp.so the tree before (shared reciprocal, staging set copied into place)
#   (CANDIDATES=v3: the shared reciprocal without its per-individual branch -- the step taken again with plain divisions if a lane asks)
#   (ROUNDS=1 NO_SUITE=1: the comparison alone, once)
# Same-box bench.py --mono-frac 0.2 of the three, two rounds; the faster candidate, if it beats the baseline by 1 %, becomes
# ngsld_amd/libngsld.so ON THE BOX and the GPU suite runs on it.  Output under gpurun_out/lane_loop/ (chosen.txt names it).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lane_loop; mkdir -p $O
AB=$PWD/ngsld_amd/ab
one() {  # ms per step of bench.py --mono-frac 0.2 on library $1
  NGSLD_LIB=$1 python bench.py --mono-frac 0.2 --steps 3 --warmup 1 --no-cpu --no-e2e --no-traffic --no-sink --no-unfiltered --no-other-configs 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['config'].get('replay_rank0_last_step',{}); print('%.2f %.5g %s %s %s' % (d['ms_per_step'], d['value'], r.get('pairs_on_device'), r.get('pairs_on_host'), d['config']['rank_records'][0]['records_checksum_u64']))"
}
: > $O/ab.txt
for r in $(seq 1 ${ROUNDS:-2}); do for v in rcp ${CANDIDATES:-v2 v1}; do echo "round $r $v $(one $AB/libngsld_$v.so)" | tee -a $O/ab.txt; done; done
best=$(python - <<PY
import collections
ms=collections.defaultdict(list); chk=collections.defaultdict(set)
for l in open("$O/ab.txt"):
    p=l.split()
    if len(p)>=8: ms[p[2]].append(float(p[3])); chk[p[2]].add(p[7])
m={k:min(v) for k,v in ms.items()}
ok=[k for k in m if k != "rcp" and chk[k]==chk["rcp"] and m[k] < 0.99*m["rcp"]]
print(min(ok, key=lambda k:m[k]) if ok else "none")
PY
)
echo "chosen: $best" | tee $O/chosen.txt
if [ "$best" != none ] && [ -z "$NO_SUITE" ]; then
  cp $AB/libngsld_$best.so ngsld_amd/libngsld.so
  ( time python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=8 ) > $O/pytest_gpu.txt 2>&1
  grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/pytest_gpu.txt | tail -6
fi
