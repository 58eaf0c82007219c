# per-launch-size kernel durations of config C's hot path, the library in tree against the one under $1 (a saved libbgp.so), same box
REF=$1
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp bayes-skopt_amd/lib/libbgp.so /tmp/libbgp_new.so
for v in new ref; do
  if [ $v = ref ]; then cp $REF bayes-skopt_amd/lib/libbgp.so; fi
  rocprofv3 --kernel-trace -d gpurun_out/ab_$v -o ab_$v -- python3 bench.py --no-extras --steps 10 --warmup 2 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python3 tools/rocprof_by_grid.py $(find gpurun_out/ab_$v -name "*_results.db" | head -1) syrk4 > gpurun_out/ab_${v}_by_grid.txt
  find gpurun_out/ab_$v -name "*.db" -delete
done
cp /tmp/libbgp_new.so bayes-skopt_amd/lib/libbgp.so
paste gpurun_out/ab_new_by_grid.txt gpurun_out/ab_ref_by_grid.txt | awk '{print $3, $4, "new", $8, "ref", $19, "ratio", $8/$19}'
