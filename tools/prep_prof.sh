#!/bin/bash
# per-kernel times of the index preparation alone (tools/prep_probe.py under rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_prep
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_prep -o t -- python tools/prep_probe.py --reps 3 "$@" > /tmp/o.txt 2>&1
tail -2 /tmp/o.txt
python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/prof_prep/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f))):
        if "k_prep" in r["Name"] or "fillBuffer" in r["Name"]:
            print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
