#!/bin/bash
# Sample the GPU's clock / power while the period bench runs under a setting (is one setting drawing the clocks down?)
# usage: tools/clock_watch.sh VAR value
VAR=$1; VAL=$2
env $VAR=$VAL timeout 400 python bench.py --no-a3 --no-cpu --steps 40 --warmup 2 > /tmp/cw_$VAL.json 2>/dev/null &
PID=$!
sleep 25
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.7
done
wait $PID
python -c "
import json
d = json.loads(open('/tmp/cw_$VAL.json').read().strip().splitlines()[-1])
print('$VAR=$VAL ms_per_period', round(d['ms_per_step'], 2))"
