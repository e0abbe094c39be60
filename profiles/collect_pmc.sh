#!/usr/bin/env bash
# Collect the rocprofv3 evidence bench.py's roofline numbers are checked against (run on the GPU box):
#   1) --kernel-trace --stats            -> average duration of pair_ld_kernel
#   2) --pmc passes, one counter set each (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined
#      with trace domains other than the kernel trace, per the pool's rules)
# Usage: profiles/collect_pmc.sh <tag> [bench args...]; summaries land in gpurun_out/prof_<tag>/.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=/tmp/prof_$TAG; OUT=$R/gpurun_out/prof_$TAG
mkdir -p $O $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python $R/bench.py --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered "$@" > $OUT/stats.log 2>&1
cp $O/stats/run_kernel_stats.csv $OUT/kernel_stats.csv
grep -m1 '^{' $OUT/stats.log > $OUT/bench_under_rocprof.json
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE32_INSTS SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT"; do
  N=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $O/pmc_$N -o run -- python $R/bench.py --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --steps 1 --warmup 0 "$@" > $OUT/pmc_$N.log 2>&1
  python - "$O/pmc_$N" "$OUT/pmc_$N.csv" <<'PY'
import csv, glob, sys
src, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(src + "/*counter_collection.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "ngsld::" in r.get("Kernel_Name", ""):     # pair kernel + prep kernel (the byte-count calibration)
                rows.append(r)
keys = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
        "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]
with open(dst, "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=keys, extrasaction="ignore")
    w.writeheader()
    for r in rows:
        w.writerow(r)
print(dst, len(rows), "rows")
PY
done
grep ngsld $OUT/kernel_stats.csv | cut -c1-200
python $R/profiles/summarise.py $OUT
