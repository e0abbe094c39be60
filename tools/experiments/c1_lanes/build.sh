#!/usr/bin/env bash
# builds the experiment's own shared object beside its source (not the library's Makefile, not __graft_entry__.build())
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared -o libc1lanes.so c1_lanes.hip
