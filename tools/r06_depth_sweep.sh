#!/usr/bin/env bash
# "a stress variant at depth 2 should be reported once" -- the headline workload at other read depths, final tree.
#   tools/r06_depth_sweep.sh   -> gpurun_out/r06_depth/sweep.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_depth; mkdir -p $O
cd $R
B="--steps 3 --warmup 1 --no-cpu --no-sink --no-e2e --no-traffic --no-unfiltered --no-other-configs"
for d in 1 2 5 10 20; do
  python bench.py --depth $d $B > $O/d$d.json 2>$O/err.txt
  python - $O/d$d.json $d <<'PY' | tee -a $O/sweep.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d["config"]; r=d["roofline"]; ri=c.get("replay_rank0_last_step") or {}
print(f"depth {sys.argv[2]:>2s}: {d['value']:.4e} pairs/s  {d['ms_per_step']:.1f} ms a step  mean executed EM steps {c['mean_executed_em_iterations']}  kernel {r['kernel']} {r['kernel_ms_per_launch']:.1f} ms  fp64 VALU frac {r['fp64_valu']['frac']:.3f}  flagged {ri.get('pairs_flagged')} (device {ri.get('pairs_on_device')}, host {ri.get('pairs_on_host')})")
PY
done
