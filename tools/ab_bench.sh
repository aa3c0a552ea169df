#!/bin/bash
# A/B two builds of the library on ONE box, interleaved: tools/ab_bench.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    ODEHIP_LIB=$L python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-model 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],3))"
  done
done
