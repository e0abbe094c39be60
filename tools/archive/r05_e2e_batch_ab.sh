#!/bin/bash
# un-called configs[2] through the binary: text batch size A/B (a batch of 2^22 rows takes the lane-per-pair replay kernel)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for b in 524288 2097152 4194304; do
  for lf in default; do
    echo "== NGSLD_TEXT_BATCH_PAIRS=$b"
    NGSLD_TEXT_BATCH_PAIRS=$b E2E_ONLY=mono20,sfs python tools/e2e_uncalled.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v['seconds'] for k,v in d['runs'].items()})"
  done
done
echo "== 2097152 with lanes from 2^21"
NGSLD_REPLAY_LANES_FROM=2097152 NGSLD_TEXT_BATCH_PAIRS=2097152 E2E_ONLY=mono20,sfs python tools/e2e_uncalled.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v['seconds'] for k,v in d['runs'].items()})"
