#!/bin/bash
# rocprofv3 evidence for BASELINE config D (n = 4096, d = 32): kernel table + FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC
# passes of tools/config_d.py (that configuration ONLY), summaries under gpurun_out/prof/ -> copy to profiles/.
# usage (from the repo root on the box): bash tools/profile_config_d.sh r03_D
set -u
TAG=${1:-rXX_D}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BGP_STREAMS=1
python3 $ROOT/tools/config_d.py 6 > $OUT/${TAG}_line.json 2> $OUT/${TAG}_line.err
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_bench -- python3 $ROOT/tools/config_d.py 6 > $OUT/${TAG}_trace.log 2>&1
python3 $ROOT/tools/rocprof_summary.py $(find $OUT -name "${TAG}_bench_results.db" | head -1) $OUT/${TAG}_kernel_stats.txt > /dev/null
python3 $ROOT/tools/rocprof_by_grid.py $(find $OUT -name "${TAG}_bench_results.db" | head -1) > $OUT/${TAG}_by_grid.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT -o ${TAG}_$c -- python3 $ROOT/tools/config_d.py 2 > $OUT/${TAG}_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $OUT -o ${TAG}_mfma -- python3 $ROOT/tools/config_d.py 2 > $OUT/${TAG}_mfma.log 2>&1
python3 $ROOT/tools/rocprof_pmc_summary.py $OUT/${TAG}_pmc.txt $(find $OUT -name "${TAG}_FETCH_SIZE_results.db" | head -1) $(find $OUT -name "${TAG}_WRITE_SIZE_results.db" | head -1) $(find $OUT -name "${TAG}_mfma_results.db" | head -1) > /dev/null
python3 $ROOT/tools/make_pmc_traffic.py $OUT/${TAG}_pmc_traffic.json $(find $OUT -name "${TAG}_FETCH_SIZE_results.db" | head -1) $(find $OUT -name "${TAG}_WRITE_SIZE_results.db" | head -1) $(find $OUT -name "${TAG}_mfma_results.db" | head -1) syrk4_kernel "tools/profile_config_d.sh: tools/config_d.py (BASELINE config D only: n=4096, d=32, 8 matrices per batch)" > /dev/null
rm -f $OUT/${TAG}_*_results.db
cat $OUT/${TAG}_line.json; cat $OUT/${TAG}_kernel_stats.txt | head -20
