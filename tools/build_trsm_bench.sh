#!/bin/bash
# Builds tools/bin/trsm_bench$SUFFIX: round 1's trsm8_kernel / left-looking update (tools/legacy/bgp_llchol.hip, extra flags "$@").
set -e
cd "$(dirname "$0")/../bayes-skopt_amd/csrc"
make -s
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
$HIPCC $FLAGS -I. -I../../tools/legacy "$@" -c ../../tools/legacy/bgp_llchol.hip -o /tmp/bgp_llchol_tb.o
$HIPCC $FLAGS -I. -c ../../tools/trsm_bench.hip -o /tmp/trsm_bench.o
mkdir -p ../../tools/bin
$HIPCC --offload-arch=gfx950 /tmp/trsm_bench.o /tmp/bgp_llchol_tb.o \
  bgp_api.o bgp_kbuild.o bgp_chol.o bgp_syrk4.o bgp_warp.o bgp_post.o bgp_bench.o bgp_comm.o bgp_gram.o bgp_ps.o bgp_mcmc.o -ldl -o ../../tools/bin/trsm_bench${SUFFIX}
