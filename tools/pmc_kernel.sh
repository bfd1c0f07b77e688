#!/bin/bash
# rocprofv3 --pmc passes for one python script; prints per-kernel averages of each counter.
#   tools/pmc_kernel.sh <kernel-name-substring> <script> [env...]
K=$1; shift; SCRIPT=$1; shift
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT"; do
  rm -rf /tmp/pmcout
  env "$@" rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcout -- python3 $GRAFT_REPO_ROOT/$SCRIPT > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
  f=$(find /tmp/pmcout -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no counters for: $grp"; tail -3 /tmp/pmc.log; continue; }
  python3 - "$f" "$K" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        a = acc[((re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"]) or re.search(r".{1,40}", r["Kernel_Name"])).group(0), r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (kn, k), (c, v) in sorted(acc.items()):
    print("%-46s %-24s launches %5d  avg %.4g" % (kn, k, c, v / c))
PY
done
