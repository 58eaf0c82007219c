set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BGP_DIST_FORCE=1 BGP_DIST_BACKEND=rccl RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29731 HSA_ENABLE_IPC_MODE_LEGACY=0
SUM=$OUT/r06_sharded_resident_timeline.txt
: > $SUM
for shape in "2048 16 40 30" "1024 8 64 100"; do
  for mode in "" "sharded"; do
    echo "===== resident_probe.py $shape $mode" >> $SUM
    python3 $ROOT/tools/resident_probe.py $shape $mode 2>/dev/null | tail -1 >> $SUM
    name=$(echo "$shape $mode" | tr ' ' '_')
    rocprofv3 --kernel-trace -d $OUT -o r06_tl_$name -- python3 $ROOT/tools/resident_probe.py $shape $mode > /dev/null 2>&1
    python3 $ROOT/tools/rocprof_timeline.py $(find $OUT -name "r06_tl_${name}_results.db" | head -1) 200 >> $SUM
  done
done
rm -f $OUT/r06_tl_*_results.db
cat $SUM
