#!/bin/bash
# the driver's N = 2 launch (both ranks on one device over gloo) on an un-called matrix against the 1-rank run: pair counts, executed-iteration
# totals and record checksums of the ranks must add up to the 1-rank run's (every rank builds the exact store of ITS slab)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A="--sites 12000 --mono-frac 0.2 --steps 1 --warmup 1 --no-cpu --no-e2e --no-traffic --no-sink"
python bench.py --gpus 1 --sites 24000 --mono-frac 0.2 --steps 1 --warmup 1 --no-cpu --no-e2e --no-traffic --no-sink 2>/dev/null | tail -1 > gpurun_out/ranks1.json
NGSLD_BENCH_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 $A 2>/dev/null | tail -1 > gpurun_out/ranks2.json
python - <<'PY'
import json
a=json.loads(open('gpurun_out/ranks1.json').read()); b=json.loads(open('gpurun_out/ranks2.json').read())
ra=a['config']['rank_records']; rb=b['config']['rank_records']
print('1 rank :', [(r['pairs'], r['executed_iterations'], r['records_checksum_u64']) for r in ra], a['config']['replay_rank0_last_step'])
print('2 ranks:', [(r['pairs'], r['executed_iterations'], r['records_checksum_u64']) for r in rb], b['config']['replay_rank0_last_step'])
print('pairs add up', sum(r['pairs'] for r in rb)==ra[0]['pairs'], 'iterations add up', sum(r['executed_iterations'] for r in rb)==ra[0]['executed_iterations'],
      'checksums add up', sum(r['records_checksum_u64'] for r in rb)%(1<<64)==ra[0]['records_checksum_u64'])
PY
