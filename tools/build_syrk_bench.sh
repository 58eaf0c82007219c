#!/bin/bash
# Builds tools/bin/syrk_bench$SUFFIX with bgp_chol.hip recompiled under -DPF_TRACE and any extra flags ("$@").
set -e
cd "$(dirname "$0")/../bayes-skopt_amd/csrc"
make -s
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
$HIPCC $FLAGS -DPF_TRACE "$@" -c bgp_chol.hip -o /tmp/bgp_chol_sb.o
$HIPCC $FLAGS -I. -c ../../tools/syrk_bench.hip -o /tmp/syrk_bench.o
mkdir -p ../../tools/bin
$HIPCC --offload-arch=gfx950 /tmp/syrk_bench.o /tmp/bgp_chol_sb.o \
  bgp_api.o bgp_kbuild.o bgp_llchol.o bgp_warp.o bgp_post.o bgp_bench.o -o ../../tools/bin/syrk_bench${SUFFIX}
