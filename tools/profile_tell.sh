#!/bin/bash
# rocprofv3 kernel tables of config-E tells (tools/pvrs_tell_probe.py: PVRS; tools/ei128_probe.py: EI over 128 hyper-posterior
# draws), summaries under gpurun_out/prof/ -> copy to profiles/.  usage (repo root on the box): bash tools/profile_tell.sh r03_E
set -u
TAG=${1:-rXX_E}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in pvrs_tell ei128; do
  rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_$v -- python3 $ROOT/tools/${v}_probe.py > $OUT/${TAG}_${v}_trace.log 2>&1
  python3 $ROOT/tools/rocprof_summary.py $(find $OUT -name "${TAG}_${v}_results.db" | head -1) $OUT/${TAG}_${v}_kernel_stats.txt > /dev/null
  grep -E "^tell|posterior\(|lml\(" $OUT/${TAG}_${v}_trace.log | head -8
  head -14 $OUT/${TAG}_${v}_kernel_stats.txt
done
rm -f $OUT/${TAG}_*_results.db
