"""The index lists of one of the A/B cases (tests/test_hip_parity.py: _prep_ab_triples), prepared `reps` times and read back
(sml_index_lists_read) each time: every duplicated row once, its slots in occurrence order, the unique marks -- against numpy.
usage: python tools/prep_stress.py <case> <reps>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SML_PREP", "hand")
import numpy as np, torch
import test_hip_parity as tp
from sml_amd.engine import HipEngine
case = sys.argv[1]; reps = int(sys.argv[2])
U, I, B, tri = tp._prep_ab_triples(case)
want = tp._expected_lists(tri, B)
DEV = torch.device("cuda:0")
eng = HipEngine(DEV, 32, B)
t = torch.from_numpy(tri).to(DEV)
bad = 0
for rep in range(reps):
    L = eng.index_lists(eng.bare_prepare(t, B, U, I))
    n = tri.shape[0]
    for b, (wu, wi, wuniq) in enumerate(want):
        Bb = min(B, n - b * B)
        if not np.array_equal(L["uniq"][3 * b * B:3 * b * B + 3 * Bb], wuniq): print(rep, b, "uniq"); bad += 1
        for tab, runs, off, cnt, vals, wr in ((0, L["runs_u"], L["off_u"], L["cnt_u"], L["val_u"], wu), (1, L["runs_i"], L["off_i"], L["cnt_i"], L["val_i"], wi)):
            k = int(cnt[b]); rec = runs[off[b]:off[b] + k]
            if k != len(wr): print(rep, b, tab, "count", k, len(wr)); bad += 1; continue
            for row, pos, ln, _, s0, s1, s2, s3 in rec.tolist():
                w = wr.get(row)
                if w is None or ln != w.size or not np.array_equal(vals[pos:pos + ln], w):
                    print(rep, b, tab, "row", row, pos, ln, None if w is None else w.size); bad += 1; break
print("bad", bad)
