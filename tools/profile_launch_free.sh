#!/bin/bash
# rocprofv3 kernel table of the launch-free factorisation next to the launch schedule (tools/launch_free_only.py): per call
# the launch-free path is ONE ps_chain_kernel + ONE ps_tile_kernel behind the Gram build; the launch schedule ~3 launches per
# block column.  usage (from the repo root on the box): bash tools/profile_launch_free.sh r03_LF
set -u
TAG=${1:-rXX_LF}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/launch_free_only.py 20 > $OUT/${TAG}_line.json 2> $OUT/${TAG}_line.err
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_bench -- python3 $ROOT/tools/launch_free_only.py 20 > $OUT/${TAG}_trace.log 2>&1
python3 $ROOT/tools/rocprof_summary.py $(find $OUT -name "${TAG}_bench_results.db" | head -1) $OUT/${TAG}_kernel_stats.txt > /dev/null
rm -f $OUT/${TAG}_*_results.db
cat $OUT/${TAG}_line.json; cat $OUT/${TAG}_kernel_stats.txt | head -20
