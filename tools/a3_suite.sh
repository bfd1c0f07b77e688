#!/bin/bash
# The a3 (bare fused embed+loss+SGD) step at table scale, four configurations; one JSON line each into gpurun_out/<tag>_a3_*.json
TAG=${1:-a3}
mkdir -p gpurun_out
B="python bench.py --workload bare --bare-batch 262144 --steps 3 --warmup 1"
$B --users 10000000 --items 1000000 --d 32 --item-zipf 0 > gpurun_out/${TAG}_a3_d32_z0.json 2> gpurun_out/${TAG}_a3.err
$B --users 10000000 --items 1000000 --d 32 --item-zipf 1 > gpurun_out/${TAG}_a3_d32_z1.json 2>> gpurun_out/${TAG}_a3.err
$B --users 10000000 --items 1000000 --d 64 --item-zipf 1 > gpurun_out/${TAG}_a3_d64_z1.json 2>> gpurun_out/${TAG}_a3.err
$B --users 50000000 --items 5000000 --d 128 --bare-dtype f16 --item-zipf 0 > gpurun_out/${TAG}_a3_d128h.json 2>> gpurun_out/${TAG}_a3.err
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_a3_*.json")):
    try:
        r = json.load(open(f))
        print(f.split("_a3_")[1][:-5], "kernel frac %.3f  e2e frac %.3f  %.2f Gtriples/s " % (r["roofline"]["frac"], r["roofline"]["end_to_end_frac"], r["value"] / 1e9),
              {k: v["avg_us"] for k, v in r["kernels"].items()})
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/${TAG}_a3.err
