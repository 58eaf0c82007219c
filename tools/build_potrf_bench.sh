#!/bin/bash
# Builds tools/bin/potrf_bench: the library objects with bgp_chol.hip recompiled under -DPF_TRACE.
set -e
cd "$(dirname "$0")/../bayes-skopt_amd/csrc"
make -s
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
$HIPCC $FLAGS -DPF_TRACE -c bgp_chol.hip -o /tmp/bgp_chol_trace.o
$HIPCC $FLAGS -I. -c ../../tools/potrf_bench.hip -o /tmp/potrf_bench.o
mkdir -p ../../tools/bin
$HIPCC --offload-arch=gfx950 /tmp/potrf_bench.o /tmp/bgp_chol_trace.o \
  bgp_api.o bgp_kbuild.o bgp_syrk4.o bgp_warp.o bgp_post.o bgp_bench.o bgp_comm.o bgp_gram.o bgp_ps.o bgp_mcmc.o -ldl -o ../../tools/bin/potrf_bench
