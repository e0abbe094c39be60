#!/bin/bash
# One GPU call for the TILED sort key of the lane-per-pair replay (ld_replay_lkl.hip: replay_keys_kernel, NGSLD_REPLAY_TILE):
# same library, same box, bench.py --mono-frac 0.2 at 100,000 x 500 under several tilings ("0,0" = the order before: rarer site,
# then the other), two rounds; --sfs once.  If the library's default beats "0,0" by 1 % with the same records checksum, the GPU
# suite runs on it.  Output under gpurun_out/tile/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/tile; mkdir -p $O
one() {  # $1 = tiling ("" = the library's default), $2.. = bench args
  local t=$1; shift
  env ${t:+NGSLD_REPLAY_TILE=$t} python bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-e2e --no-traffic --no-sink --no-unfiltered --no-other-configs 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['config'].get('replay_rank0_last_step',{}); print('%.2f %.5g %s %s %s' % (d['ms_per_step'], d['value'], r.get('pairs_on_device'), r.get('pairs_on_host'), d['config']['rank_records'][0]['records_checksum_u64']))"
}
: > $O/ab.txt
for r in 1 2; do for t in 0,0 default 6,3 5,4 4,3; do
  tt=$t; [ $t = default ] && tt=""
  echo "mono20 round $r tile $t $(one "$tt" --mono-frac 0.2)" | tee -a $O/ab.txt
done; done
for t in 0,0 default; do
  tt=$t; [ $t = default ] && tt=""
  echo "sfs round 1 tile $t $(one "$tt" --sfs)" | tee -a $O/ab.txt
done
ok=$(python - <<PY
import collections
ms=collections.defaultdict(list); chk=collections.defaultdict(set)
for l in open("$O/ab.txt"):
    p=l.split()
    if len(p)>=10 and p[0]=="mono20": ms[p[4]].append(float(p[5])); chk[p[4]].add(p[9])
m={k:min(v) for k,v in ms.items()}
print("yes" if "default" in m and "0,0" in m and chk["default"]==chk["0,0"] and m["default"] < 0.99*m["0,0"] else "no")
PY
)
echo "default beats 0,0: $ok" | tee $O/chosen.txt
if [ "$ok" = yes ]; then
  ( time python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=8 ) > $O/pytest_gpu.txt 2>&1
  grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/pytest_gpu.txt | tail -6
fi
