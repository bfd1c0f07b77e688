#!/bin/bash
# where the device idles during a driver stage: kernel trace of tools/time_driver_stage.py, gaps on the busiest queue
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_drv
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_drv -o t -- python tools/time_driver_stage.py 3 > /tmp/drv.txt 2>&1
tail -3 /tmp/drv.txt
python - <<PY
import csv, glob, collections
rows=[]
for f in glob.glob("gpurun_out/prof_drv/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
byq=collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
q=max(byq, key=lambda k: len(byq[k]))
ev=sorted(byq[q])
t_end=ev[-1][1]
# last stage only: last 0.128 s
ev=[e for e in ev if e[0] > t_end - 128e6]
busy=sum(e[1]-e[0] for e in ev)
gaps=[(ev[i+1][0]-ev[i][1], ev[i][2][:40], ev[i+1][2][:40]) for i in range(len(ev)-1)]
big=[g for g in gaps if g[0] > 30000]
print("kernels %d busy %.1f ms  span %.1f ms  gaps>30us: %d totalling %.1f ms; all gaps %.1f ms" % (len(ev), busy/1e6, (ev[-1][1]-ev[0][0])/1e6, len(big), sum(g[0] for g in big)/1e6, sum(g[0] for g in gaps)/1e6))
for g in sorted(big, reverse=True)[:14]: print("  %.0f us after %s before %s" % (g[0]/1e3, g[1], g[2]))
PY
rm -rf gpurun_out/prof_drv
