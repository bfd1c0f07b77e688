#!/bin/bash
# fabric bytes of the index preparation's kernels (FETCH_SIZE / WRITE_SIZE in separate passes, KB units)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_prep_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_prep_$c -o t -- python tools/prep_probe.py --reps 2 "$@" > /tmp/o_$c.txt 2>&1
done
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_prep_%s/**/*counter_collection.csv"%c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==c and "k_prep" in r["Kernel_Name"]:
                import re
                k=re.search(r"k_prep_\w+", r["Kernel_Name"]).group(0)
                e=acc[k][c]; e[0]+=1; e[1]+=float(r["Counter_Value"])
tot=0
for k,v in sorted(acc.items()):
    f=v["FETCH_SIZE"]; w=v["WRITE_SIZE"]
    fb=f[1]/max(f[0],1)*1024; wb=w[1]/max(w[0],1)*1024
    print("%-30s launches %3d fetch(raw) %7.1f MB write %7.1f MB" % (k, f[0], fb/1e6, wb/1e6))
PY
rm -rf gpurun_out/pmc_prep_FETCH_SIZE gpurun_out/pmc_prep_WRITE_SIZE
