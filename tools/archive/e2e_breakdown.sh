#!/usr/bin/env bash
# Where the drop-in binary's wall time goes on configs[2] (100,000 x 500, 100 kb, --extend_out, text to /dev/null):
# NGSLD_TIMING=1 phases of three runs + the kernel trace of one (rocprofv3).  -> gpurun_out/r04/e2e_breakdown.txt
set -e
mkdir -p gpurun_out/r04
out=$PWD/gpurun_out/r04/e2e_breakdown.txt
D=/dev/shm/e2e_$$; mkdir -p $D
python - $D <<'PY'
import sys, os
sys.path.insert(0, ".")
import torch
from ngsld_amd import synth
d = sys.argv[1]
synth.make_gl_torch(100000, 500, 3, torch.device("cuda", 0)).cpu().numpy().tofile(os.path.join(d, "in.glf"))
chrs, pos = synth.make_positions(100000, 3)
synth.write_pos(os.path.join(d, "in.pos"), chrs, pos)
PY
CMD="ngsld_amd/bin/ngsLD --geno $D/in.glf --n_ind 500 --n_sites 100000 --pos $D/in.pos --max_kb_dist 100 --extend_out --n_threads 16 --verbose 0 --out /dev/null"
case " ${E2E_ENV:-} " in *NGSLD_E2E_DEVICES=*) DEVS=$(echo "${E2E_ENV}" | tr " " "\n" | grep NGSLD_E2E_DEVICES= | cut -d= -f2); CMD="$CMD --devices $DEVS";; esac  # (E2E_ENV="NGSLD_E2E_DEVICES=0,0,0": several parts in one process)
: > $out
for i in 1 2 3; do
  echo "== run $i ${E2E_ENV:-}" >> $out
  /usr/bin/env bash -c "time env NGSLD_TIMING=1 ${E2E_ENV:-} $CMD" >> $out 2>&1
done
if [ -z "${E2E_NO_TRACE:-}" ]; then
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/e2e_trace -o t -- env ${E2E_ENV:-} $R/$CMD > /dev/null 2>&1 || true
cd $R
python - >> $out <<'PY'
import csv, glob, collections
k = glob.glob('/tmp/e2e_trace/**/*kernel_trace.csv', recursive=True)
m = glob.glob('/tmp/e2e_trace/**/*memory_copy_trace.csv', recursive=True)
rows = list(csv.DictReader(open(k[0]))) if k else []
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r['Kernel_Name'].split('(')[0][-60:]
    agg[n][0] += 1; agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
print(f"== kernel trace: first kernel start to last kernel end {(t1 - t0) / 1e6:.1f} ms")
for n, (c, ms) in sorted(agg.items(), key=lambda x: -x[1][1])[:12]:
    print(f"{ms:10.2f} ms  {c:6d} x  {n}")
# union of busy time of the pair kernels vs. everything
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
busy = 0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"device busy with some kernel: {busy / 1e6:.1f} ms of {(t1 - t0) / 1e6:.1f}")
if m:
    cr = list(csv.DictReader(open(m[0])))
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in cr:
        d = r.get('Direction', '?'); agg[d][0] += 1; agg[d][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    for d, (c, ms, _) in agg.items(): print(f"copies {d}: {c} x, {ms:.1f} ms in flight")
PY
rm -rf /tmp/e2e_trace
fi
rm -rf $D
cat $out
