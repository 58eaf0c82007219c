#!/bin/bash
# Kernel timeline of the config C hot path with the product's walker-group streams: which kernels run side by side.
# usage (repo root, GPU box): bash tools/profile_overlap.sh TAG [ENV=VAL ...]
TAG=${1:-ov}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace -d $OUT -o ${TAG} -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-extras > $OUT/${TAG}.log 2>&1
DB=$(find $OUT -name "${TAG}_results.db" | head -1)
python3 $ROOT/tools/rocprof_overlap.py $DB 1500 90 > $OUT/${TAG}_overlap.txt
python3 $ROOT/tools/rocprof_gaps.py $DB 1500 > $OUT/${TAG}_gaps.txt
rm -f $(find $OUT -name "${TAG}_results.db")
head -12 $OUT/${TAG}_overlap.txt; cat $OUT/${TAG}_gaps.txt
