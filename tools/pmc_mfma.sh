#!/bin/bash
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pmcout
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcout -- python3 $GRAFT_REPO_ROOT/tools/bench_steps.py > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
f=$(find /tmp/pmcout -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"k_transfer\w+(<[^>]*>)?", r["Kernel_Name"])
    if m:
        a = acc[(m.group(0), r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (kn, k), (c, v) in sorted(acc.items()):
    print("%-40s %-30s n %5d avg %.4g" % (kn, k, c, v / c))
PY
