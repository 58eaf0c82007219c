# config C hot path alone, N alternations of the library in tree against the one under $1 (a saved libbgp.so): ms per step (three passes) and kernel split
REF=$1
for rep in 1 2 3; do
  for v in new ref; do
    if [ $v = ref ]; then cp bayes-skopt_amd/lib/libbgp.so /tmp/libbgp_new.so; cp $REF bayes-skopt_amd/lib/libbgp.so; fi
    echo -n "$v: "; python bench.py --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([round(t,3) for t in d['timed_passes_ms_per_step']], {k: round(v,4) for k,v in d['kernel_ms_per_half_step'].items()}, round(d['roofline']['frac'],4))"
    if [ $v = ref ]; then cp /tmp/libbgp_new.so bayes-skopt_amd/lib/libbgp.so; fi
  done
done
