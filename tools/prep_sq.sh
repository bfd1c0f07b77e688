#!/bin/bash
# where the wavefronts of the index preparation's kernels spend their cycles (SQ counters, one pass): parked on s_waitcnt / barriers
# (WAIT_ANY), issue stalls (WAIT_INST_ANY, of which LDS), issuing (ACTIVE_INST_ANY), LDS bank-conflict cycles of all LDS cycles
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_prep_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/pmc_prep_sq -o t -- python3 tools/prep_probe.py --reps 2 "$@" > /tmp/o_sq.txt 2>&1
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob("gpurun_out/pmc_prep_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_prep" in r["Kernel_Name"]:
            k=re.search(r"k_prep_\w+", r["Kernel_Name"]).group(0)
            e=acc[k][r["Counter_Name"]]; e[0]+=1; e[1]+=float(r["Counter_Value"])
for k,v in sorted(acc.items()):
    g=lambda n: v[n][1]/max(v[n][0],1)
    wc=g("SQ_WAVE_CYCLES") or 1
    print("%-16s waves %7.0f  wave-cycles %10.0f  parked %4.1f%%  issue-stall %4.1f%% (LDS %4.1f%%)  issuing %4.1f%%  LDS conflict %4.1f%% of LDS cycles" % (
        k, g("SQ_WAVES"), wc, 100*g("SQ_WAIT_ANY")/wc, 100*g("SQ_WAIT_INST_ANY")/wc, 100*g("SQ_WAIT_INST_LDS")/wc, 100*g("SQ_ACTIVE_INST_ANY")/wc,
        100*g("SQ_LDS_BANK_CONFLICT")/max(g("SQ_LDS_IDX_ACTIVE"),1)))
PY
rm -rf gpurun_out/pmc_prep_sq
