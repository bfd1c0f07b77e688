"""Does teacher forcing (tests/test_host_logic.py: run_g12) change the GPU run at all?  Prints how many per-batch losses differ
between the forced and the free-running G12 / G15 sequences (0 = the forced state never reached the engine)."""
import os, sys, pathlib, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
os.environ.setdefault("LOCAL_RANK", "0")
import test_host_logic as H
for name in ("g12_midsize", "g15_fulldepth"):
    res = []
    for tf in (True, False):
        meta, log, flat, z = H.run_g12(pathlib.Path(tempfile.mkdtemp()), teacher_forced=tf, name=name)
        res.append(np.array([a[2] for a in flat]))
    tags = z["batch_tag"]
    d = res[0] != res[1]
    print(name, "losses that differ forced vs free:", int(d.sum()), "of", d.size, "| per stage:", [int(d[tags[:, 0] == s].sum()) for s in range(int(tags[:, 0].max()) + 1)])
