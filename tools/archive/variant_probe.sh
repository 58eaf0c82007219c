#!/bin/bash
# Config C headline (bench.py --no-extras) under a list of environment variants: "NAME=VAL,NAME=VAL" per argument
# ("-" = defaults).  Usage: tools/variant_probe.sh - BGP_STAGGER=2 BGP_STREAMS=4,BGP_STAGGER=1
for v in "$@"; do
  envs=""
  if [ "$v" != "-" ]; then envs=$(echo "$v" | tr ',' ' '); fi
  for rep in 1 2; do
    line=$(env $envs python bench.py --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1)
    echo "$v rep$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("evals/s %.0f ms/step %.3f  dev_ms/half(1 stream) %.3f" % (d["value"], d["ms_per_step"], d["device_ms_per_half_step"]))')"
  done
done
