#!/bin/bash
# where the training queue idles during a bench period: kernel trace of `bench.py --no-cpu --no-a3 --no-roofline`, gaps on the busiest queue
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gap
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gap -o t -- python3 bench.py --no-cpu --no-a3 --no-roofline --steps 2 --warmup 1 > /tmp/gap.txt 2>&1
tail -1 /tmp/gap.txt | cut -c1-200
python3 - <<PY
import csv, glob, collections
rows=[]
for f in glob.glob("gpurun_out/prof_gap/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
byq=collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q in sorted(byq, key=lambda k: -len(byq[k]))[:3]:
    ev=sorted(byq[q])
    t_end=ev[-1][1]
    ev=[e for e in ev if e[0] > t_end - 100e6]          # the last ~100 ms: inside the last timed period
    if len(ev) < 2: continue
    busy=sum(e[1]-e[0] for e in ev)
    gaps=[(ev[i+1][0]-ev[i][1], ev[i][2], ev[i+1][2]) for i in range(len(ev)-1)]
    print("queue %s: kernels %d busy %.2f ms span %.2f ms; all gaps %.2f ms" % (q, len(ev), busy/1e6, (ev[-1][1]-ev[0][0])/1e6, sum(g[0] for g in gaps)/1e6))
    import re
    short=lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n)[:34]
    agg=collections.defaultdict(lambda:[0,0])
    for g,a,b in gaps:
        k=(short(a), short(b)); agg[k][0]+=1; agg[k][1]+=g
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:16]:
        print("   %5d gaps, %8.1f us total, %6.2f us avg: after %-34s before %s" % (v[0], v[1]/1e3, v[1]/1e3/v[0], k[0], k[1]))
    dur=collections.defaultdict(lambda:[0,0])
    for s,e,n in ev: dur[short(n)][0]+=1; dur[short(n)][1]+=e-s
    for k,v in sorted(dur.items(), key=lambda kv:-kv[1][1])[:10]:
        print("   busy %8.1f us in %5d x %s" % (v[1]/1e3, v[0], k))
PY
rm -rf gpurun_out/prof_gap
