#!/usr/bin/env bash
# Round 5 -- the final tree: smoke, the driver's own bench line, rocprofv3 kernel stats + counter passes of the default bench,
# the other BASELINE configurations, the un-called inputs at full size, the binary end to end.  (The GPU suite runs by itself.)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_final; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r05_final.json 2> $O/bench_r05_final.err
python - $O/bench_r05_final.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
u = d['unfiltered_input']
print('value %.4e ms_per_step %.2f kernel %.2f host_resident %.4e ratio %.4f frac %.4f e2e %.3f checksum %d cpu_baseline %.4g (%s, %d threads)' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['value_host_resident'], d['value_host_resident'] / d['value'], d['roofline']['frac'], d['e2e_file_to_tsv_s']['seconds'], d['config']['rank_records'][0]['records_checksum_u64'], d['cpu_baseline']['value'], d['cpu_baseline']['kind'], d['cpu_baseline']['cores']))
for k in ('mono_frac_0.2', 'sfs'):
    print('unfiltered_input', k, '%.4g pairs/s, replay off %.4g, first pass %.3f s,' % (u[k]['value'], u[k]['value_replay_off'], u[k]['first_pass_s']), u[k]['replay'])
PY
bash profiles/collect_pmc.sh r05_last --steps 3 --warmup 1 > $O/collect_pmc.log 2>&1
cp -r gpurun_out/prof_r05_last $O/last
bash tools/bench_configs.sh $O/configs_r05.jsonl | tee $O/configs_r05.txt
COMMON="--no-cpu --no-e2e --no-traffic"
for a in "--mono-frac 0.2" "--sfs" "--mono-frac 0.2 --depth 30" "--hard-calls" "--hard-calls --mono-frac 0.2"; do
  timeout 900 python bench.py $a --steps 3 --warmup 1 $COMMON 2>/dev/null | tail -1 >> $O/uncalled_r05.jsonl
done
python - $O/uncalled_r05.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["config"]["workload"], "|", f'{d["value"]:.4g} pairs/s | host-resident {d["value_host_resident"]:.4g} |', f'{d["ms_per_step"]:.1f} ms | kernel {d["roofline"]["kernel_ms_per_launch"]:.1f} ms |',
          d["config"]["replay_rank0_last_step"], "| first pass", d["config"]["first_pass_s_rank0"], "| replay off", d["config"]["replay_off"])
PY
python tools/e2e_uncalled.py > $O/e2e_uncalled.json 2> /dev/null
python -c "
import json; d=json.load(open('$O/e2e_uncalled.json'))
for k,v in d['runs'].items(): print('binary end to end,', k, v['seconds'])"
