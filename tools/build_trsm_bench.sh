#!/bin/bash
# Builds tools/bin/trsm_bench$SUFFIX with bgp_llchol.hip recompiled with any extra flags ("$@").
set -e
cd "$(dirname "$0")/../bayes-skopt_amd/csrc"
make -s
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
$HIPCC $FLAGS "$@" -c bgp_llchol.hip -o /tmp/bgp_llchol_tb.o
$HIPCC $FLAGS -I. -c ../../tools/trsm_bench.hip -o /tmp/trsm_bench.o
mkdir -p ../../tools/bin
$HIPCC --offload-arch=gfx950 /tmp/trsm_bench.o /tmp/bgp_llchol_tb.o \
  bgp_api.o bgp_kbuild.o bgp_chol.o bgp_warp.o bgp_post.o bgp_bench.o -o ../../tools/bin/trsm_bench${SUFFIX}
