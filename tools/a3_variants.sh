#!/bin/bash
# a3 end-to-end (bench.py --workload bare) under variants of the preparation stream
ulimit -c 0
run() { env "$@" timeout 200 python bench.py --workload bare --no-cpu --users 10000000 --items 1000000 --bare-batch 262144 --bare-triples 4194304 --item-zipf ${ZIPF:-0} --steps ${STEPS:-6} --warmup ${WARM:-1} 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s Mtri/s %8.1f kernel %.3f e2e %.3f' % (sys.argv[1], r['value']/1e6, r['roofline']['frac'], r['roofline']['end_to_end_frac']), {k:v['avg_us'] for k,v in r['kernels'].items()})" "$*"; }
for v in "$@"; do run $v; done
