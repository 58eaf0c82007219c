#!/bin/bash
# Collects the per-round profile set on the GPU box: bench line, rocprofv3 kernel stats, PMC passes.
# usage (from the repo root on the box): bash tools/profile_round.sh r01_v6
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# per-kernel numbers: every launch on one stream (kernel alone on the GPU), like bench.py's instrumented pass
export BGP_STREAMS=1
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_bench -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-extras > $OUT/${TAG}_trace.log 2>&1
python3 $ROOT/tools/rocprof_summary.py $(find $OUT -name "${TAG}_bench_results.db" | head -1) $OUT/${TAG}_bench_kernel_stats.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT -o ${TAG}_$c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-extras > $OUT/${TAG}_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $OUT -o ${TAG}_mfma -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-extras > $OUT/${TAG}_mfma.log 2>&1
python3 $ROOT/tools/rocprof_pmc_summary.py $OUT/${TAG}_pmc.txt $(find $OUT -name "${TAG}_FETCH_SIZE_results.db" | head -1) $(find $OUT -name "${TAG}_WRITE_SIZE_results.db" | head -1) $(find $OUT -name "${TAG}_mfma_results.db" | head -1) > /dev/null
python3 $ROOT/tools/make_pmc_traffic.py $OUT/${TAG}_pmc_traffic.json $(find $OUT -name "${TAG}_FETCH_SIZE_results.db" | head -1) $(find $OUT -name "${TAG}_WRITE_SIZE_results.db" | head -1) $(find $OUT -name "${TAG}_mfma_results.db" | head -1) syrk4_kernel > /dev/null
python3 $ROOT/tools/rocprof_by_grid.py $(find $OUT -name "${TAG}_bench_results.db" | head -1) > $OUT/${TAG}_bench_by_grid.txt
rm -f $OUT/${TAG}_*_results.db  # (tens of MB each: only the summaries travel back)
ls -la $OUT | tail -20
cat $OUT/${TAG}_bench.json | cut -c1-400
