#!/bin/bash
# Builds tools/bin/syrk4_bench: bgp_syrk4.hip under -DS4_BENCH (ablation / trace instantiations) and round 1's
# syrk2_kernel from tools/legacy/ (the bit-identical A/B reference), linked with the product objects.
set -e
cd "$(dirname "$0")/../bayes-skopt_amd/csrc"
make -s
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
$HIPCC $FLAGS -I. -I../../tools/legacy -c ../../tools/legacy/legacy_kernels.hip -o /tmp/bgp_legacy_sb.o
$HIPCC $FLAGS -DS4_BENCH "$@" -c bgp_syrk4.hip -o /tmp/bgp_syrk4_sb.o
$HIPCC $FLAGS -I. -c ../../tools/syrk4_bench.hip -o /tmp/syrk4_bench.o
mkdir -p ../../tools/bin
$HIPCC --offload-arch=gfx950 /tmp/syrk4_bench.o /tmp/bgp_legacy_sb.o /tmp/bgp_syrk4_sb.o \
  bgp_api.o bgp_kbuild.o bgp_chol.o bgp_warp.o bgp_post.o bgp_bench.o bgp_comm.o bgp_gram.o bgp_ps.o bgp_mcmc.o -ldl -o ../../tools/bin/syrk4_bench${SUFFIX}
$HIPCC $FLAGS -I. -c ../../tools/trsm4_bench.hip -o /tmp/trsm4_bench.o
$HIPCC --offload-arch=gfx950 /tmp/trsm4_bench.o /tmp/bgp_syrk4_sb.o \
  bgp_api.o bgp_kbuild.o bgp_chol.o bgp_warp.o bgp_post.o bgp_bench.o bgp_comm.o bgp_gram.o bgp_ps.o bgp_mcmc.o -ldl -o ../../tools/bin/trsm4_bench${SUFFIX}
