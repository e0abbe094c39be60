#!/usr/bin/env bash
# Host hardening run (CPU box): the host-side tests and the CLI's error paths on the AddressSanitizer + UBSan build of the
# host code (make -C ngsld_amd/csrc asan).  Device code is not instrumented (no GPU sanitizers on this pool).
#   tests/run_asan.sh            -> exit code 0 = no sanitizer report
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -j8 -C "$R/ngsld_amd/csrc" all asan
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export NGSLD_LIB="$R/ngsld_amd/libngsld_asan.so"
cd "$R"
LD_PRELOAD="$ASAN $UBSAN" python -m pytest tests/test_host_io.py tests/test_stream_host.py tests/test_replay_host.py tests/test_abi.py \
  -q -m "not gpu" -p no:cacheprovider "$@"
# the CLI's argument / input error paths (no device is reached)
T=$(mktemp -d); trap 'rm -rf "$T"' EXIT
python - "$T" <<'PY'
import sys, numpy as np
d = sys.argv[1]
np.random.default_rng(1).random((40, 6, 3)).tofile(d + "/in.glf")
open(d + "/in.pos", "w").write("".join(f"chr1\t{10 * (k + 1)}\n" for k in range(40)))
open(d + "/bad.pos", "w").write("".join(f"chr1\t{10 * (40 - k)}\n" for k in range(40)))
PY
B="$R/ngsld_amd/bin/ngsLD_asan"
run() { set +e; "$B" "$@" > "$T/out" 2> "$T/err"; rc=$?; set -e; if grep -q "Sanitizer\|runtime error" "$T/err"; then cat "$T/err"; exit 1; fi; echo "rc=$rc  $*" | cut -c1-150; }
run --geno "$T/in.glf" --n_ind 6 --n_sites 40                                   # no --pos with a distance limit
run --geno "$T/in.glf" --n_ind 6 --n_sites 41 --pos "$T/in.pos"                 # size check
run --geno "$T/in.glf" --n_ind 6 --pos "$T/in.pos"                              # n_sites missing
run --geno "$T/in.glf" --n_ind 6 --n_sites 40 --pos "$T/in.pos" --min_maf 2     # range check
run --geno "$T/in.glf" --n_ind 6 --n_sites 40 --pos "$T/in.pos" --rnd_sample 0  # range check
run --geno "$T/in.glf" --n_ind 6 --n_sites 40 --pos "$T/in.pos" --devices 0,x   # malformed list
run --geno "$T/in.glf" --n_ind 6 --n_sites 40 --pos "$T/in.pos"                 # reaches ngsld_create: no device here
run --geno "$T/in.glf" --n_ind 6 --n_sites 40 --pos "$T/bad.pos" --devices 0,0  # multi path: positions not increasing
echo "asan run clean"
