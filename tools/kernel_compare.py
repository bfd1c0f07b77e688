import sys, json
a = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("ms", a["ms_per_step"], b["ms_per_step"])
for k in sorted(a.get("kernels", {})):
    ka, kb = a["kernels"][k], b.get("kernels", {}).get(k)
    if kb: print("%-28s n=%5d  %8.2f us  %8.2f us   total %7.2f -> %7.2f ms" % (k, ka["launches"], ka["avg_us"], kb["avg_us"], ka["total_ms"], kb["total_ms"]))
