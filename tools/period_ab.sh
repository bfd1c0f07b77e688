#!/bin/bash
# Same-box A/B of the period bench under two settings of one environment variable, alternating.
# usage: tools/period_ab.sh VAR valueA valueB [pairs] [extra bench args]
VAR=$1; A=$2; B=$3; PAIRS=${4:-2}; shift 4
for i in $(seq 1 $PAIRS); do
  for m in $A $B; do
    env $VAR=$m timeout 400 python bench.py --no-a3 --no-cpu --steps ${STEPS:-20} --warmup ${WARMUP:-5} "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('$VAR=$m', 'ms_per_period', round(d['ms_per_step'], 2), 'eval', k.get('k_eval_ranks'))"
  done
done
