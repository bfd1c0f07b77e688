"""bench.a3_object's legs in one process, in a chosen order (is a leg slower after another one?)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
order = sys.argv[1].split(",")
sys.argv = ["bench.py", "--no-cpu"]
a = bench.parse()
dev = torch.device("cuda:0")
keep = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("PRE_STREAMS", "0")))]      # (shifts the pool stream the legs get)
cfg = {c[0]: c for c in bench.A3_CONFIGS}
for tag in order:
    _, users, items, d, dt, zipf = cfg[tag]
    b = copy.copy(a)
    b.users, b.items, b.d, b.bare_dtype, b.item_zipf = users, items, d, dt, zipf
    b.bare_batch, b.bare_triples, b.steps, b.warmup = 262144, 1 << 22, 16, 4
    r = bench.bench_bare(b, dev)
    print(tag, "Mtri/s %.1f kernel %.3f e2e %.3f" % (r["value"] / 1e6, r["roofline"]["frac"], r["roofline"]["end_to_end_frac"]), flush=True)
