cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
for cfg in "1024 8 64 100" "128 2 100 100"; do
  tag=$(echo $cfg | tr ' ' '_')
  python3 $R/tools/resident_probe.py $cfg
  rocprofv3 --kernel-trace -d $R/gpurun_out/r05/rt -o rt_$tag -- python3 $R/tools/resident_probe.py $cfg > /dev/null 2>&1
  python3 $R/tools/rocprof_timeline.py $(find $R/gpurun_out/r05/rt -name "rt_${tag}_results.db" | head -1) 40
done
rm -rf $R/gpurun_out/r05/rt
