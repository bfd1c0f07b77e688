#!/bin/bash
# where the wavefronts of a command's kernels spend their cycles (SQ counters, one pass): parked on s_waitcnt / barriers (WAIT_ANY), issue
# stalls (WAIT_INST_ANY, of which LDS), issuing (ACTIVE_INST_ANY), LDS bank-conflict share.   usage: tools/sq_kernels.sh <kernel-name regex> python3 <script> [args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PAT=$1; shift
rm -rf gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/pmc_sq -o t -- "$@" > /tmp/o_sq.txt 2>&1
PAT="$PAT" python3 - <<PY
import csv,glob,collections,re,os
pat=re.compile(os.environ["PAT"])
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat.search(r["Kernel_Name"]):
            k=re.sub(r"\(anonymous namespace\)::|void |\(Sml\w+\)", "", r["Kernel_Name"])[:40]
            e=acc[k][r["Counter_Name"]]; e[0]+=1; e[1]+=float(r["Counter_Value"])
for k,v in sorted(acc.items()):
    g=lambda n: v[n][1]/max(v[n][0],1)
    wc=g("SQ_WAVE_CYCLES") or 1
    print("%-40s waves %6.0f  quad-cycles/wave %7.0f  parked %4.1f%%  issue-stall %4.1f%% (LDS %4.1f%%)  issuing %4.1f%%  LDS conflict %4.1f%%" % (
        k, g("SQ_WAVES"), wc/max(g("SQ_WAVES"),1), 100*g("SQ_WAIT_ANY")/wc, 100*g("SQ_WAIT_INST_ANY")/wc, 100*g("SQ_WAIT_INST_LDS")/wc, 100*g("SQ_ACTIVE_INST_ANY")/wc,
        100*g("SQ_LDS_BANK_CONFLICT")/max(g("SQ_LDS_IDX_ACTIVE"),1)))
PY
rm -rf gpurun_out/pmc_sq
