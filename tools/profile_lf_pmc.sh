#!/bin/bash
# Counters of the launch-free factorisation's kernel (ps_kernel), one shape per process so that the dispatches of a pass are
# all of one shape: kernel table, FETCH_SIZE, WRITE_SIZE and the SQ / GRBM set in SEPARATE passes (MI355X_MICROARCH.md).
# usage (repo root, GPU box): bash tools/profile_lf_pmc.sh r04_LF
set -u
TAG=${1:-rXX_LF}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SUM=$OUT/${TAG}_pmc.txt
: > $SUM
# second argument "pairs": the shapes the chain pairs take (bgp_pair_auto_rule) instead of the single-chain ones
if [ "${2:-}" = "pairs" ]; then
  SHAPES=("lml 4096 32 1 20" "lml 2048 16 1 20" "lml 2048 16 2 20" "lml 1024 8 8 20")
else
  SHAPES=("lml 1024 8 32 20" "lml 2048 16 16 20" "lml 4096 32 1 20" "cov 1000 8 10000 6")
fi
for shape in "${SHAPES[@]}"; do
  name=$(echo $shape | tr ' ' '_')
  echo "===== $shape" >> $SUM
  python3 $ROOT/tools/lf_shape.py $shape >> $SUM 2>/dev/null
  rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_${name}_kt -- python3 $ROOT/tools/lf_shape.py $shape > /dev/null 2>&1
  python3 $ROOT/tools/rocprof_summary.py $(find $OUT -name "${TAG}_${name}_kt_results.db" | head -1) | grep -E "kernel|ps_kernel|kbuild|gemm4|tri_matmul" >> $SUM
  dbs=""
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    cn=$(echo $c | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $c -d $OUT -o ${TAG}_${name}_${cn} -- python3 $ROOT/tools/lf_shape.py $shape > /dev/null 2>&1
    db=$(find $OUT -name "${TAG}_${name}_${cn}_results.db" | head -1)
    [ -n "$db" ] && dbs="$dbs $db"
  done
  python3 $ROOT/tools/rocprof_pmc_summary.py /tmp/pmc_$$.txt $dbs | grep -E "counter|ps_kernel" >> $SUM
done
rm -f $OUT/${TAG}_*_results.db
cat $SUM
