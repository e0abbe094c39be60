#!/bin/bash
# un-called input of LARGE cohorts through the binary (text batches): the lane-per-pair replay kernel on every batch (product from 513
# individuals on) against the wavefront-per-pair kernel (NGSLD_REPLAY_LANES_FROM=4194304: as for 500 individuals), cap / wavefront A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $1"; env $1 E2E_MONO=0.2 python tools/e2e_stream.py ${SITES:-120000} ${IND:-2000} 48 16 2>&1 | grep "resident\|streamed"; }
run "NGSLD_REPLAY_LANES_FROM=4194304"
run "NGSLD_X=product"
run "NGSLD_TEST_LANE_ITER_CAP=6"
run "NGSLD_TEST_LANE_ITER_CAP=24"
run "NGSLD_LANE_WAVES=2"
run "NGSLD_LANE_WAVES=4"
