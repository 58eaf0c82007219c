# per-launch-size kernel durations of config C's hot path with the Gram generation inside the trailing update on / off (same box)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 1 0; do
  export BGP_SYRK_GEN=$v BGP_STREAMS=1
  rocprofv3 --kernel-trace -d gpurun_out/gen_$v -o gen_$v -- python3 bench.py --no-extras --steps 10 --warmup 2 > gpurun_out/gen_$v.json 2> gpurun_out/gen_$v.err
  python3 tools/rocprof_by_grid.py $(find gpurun_out/gen_$v -name "*_results.db" | head -1) > gpurun_out/gen_${v}_by_grid.txt
  find gpurun_out/gen_$v -name "*.db" -delete
done
