#!/bin/bash
# un-called configs[2] through the binary: the lane-per-pair replay kernel on the text batches too (NGSLD_REPLAY_LANES_FROM=0), with the
# hand-back cap per lane, against the product (wavefront per pair on launches below 2^22 records at up to 512 individuals)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v['seconds'] for k,v in d['runs'].items()})"; }
echo "== product"; E2E_ONLY=mono20,sfs python tools/e2e_uncalled.py 2>/dev/null | show
for cap in 6 12 24; do for w in 1 2 4; do
  echo "== lanes on every batch, cap $cap, $w wavefronts per SIMD"
  NGSLD_REPLAY_LANES_FROM=0 NGSLD_TEST_LANE_ITER_CAP=$cap NGSLD_LANE_WAVES=$w E2E_ONLY=mono20,sfs python tools/e2e_uncalled.py 2>/dev/null | show
done; done
