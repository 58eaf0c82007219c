#!/bin/bash
# Builds tools/bin/syrk_bench$SUFFIX: round 1's syrk2_kernel (tools/legacy/legacy_kernels.hip, any extra flags "$@").
set -e
cd "$(dirname "$0")/../bayes-skopt_amd/csrc"
make -s
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
$HIPCC $FLAGS -I. -I../../tools/legacy "$@" -c ../../tools/legacy/legacy_kernels.hip -o /tmp/bgp_legacy_sb.o
$HIPCC $FLAGS -I. -c ../../tools/syrk_bench.hip -o /tmp/syrk_bench.o
mkdir -p ../../tools/bin
$HIPCC --offload-arch=gfx950 /tmp/syrk_bench.o /tmp/bgp_legacy_sb.o \
  bgp_api.o bgp_kbuild.o bgp_chol.o bgp_syrk4.o bgp_warp.o bgp_post.o bgp_bench.o bgp_comm.o bgp_gram.o bgp_ps.o bgp_mcmc.o -ldl -o ../../tools/bin/syrk_bench${SUFFIX}
