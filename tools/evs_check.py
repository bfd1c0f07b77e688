#!/usr/bin/env python3
"""Read back the sliced evaluation's prepared entries and check them against the test rows (layout debugging)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sml_amd.engine import HipEngine
dev = torch.device("cuda", 0)
d, U, I, n, neg = 32, 700, int(os.environ.get("I", "123000")), int(os.environ.get("N", "1031")), int(os.environ.get("NEG", "999"))
torch.manual_seed(0)
eng = HipEngine(dev, d, 1024)
rows = torch.cat([torch.randint(0, U, (n, 1)), torch.randint(0, I, (n, 1 + neg))], 1).to(dev)
entries, seg_off = eng._sliced_rows(rows, I)
torch.cuda.synchronize()
E = entries.cpu().numpy().view(np.uint32); S = seg_off.cpu().numpy(); R = rows.cpu().numpy()
ns = (I + 1023) // 1024; n_mb = (n + 63) // 64
assert S[0] == 0 and (np.diff(S) >= 0).all(), "segment table"
bad = 0
for s in range(ns):
    for mb in range(n_mb):
        a, b = S[s * n_mb + mb], S[s * n_mb + mb + 1]
        e = E[a:b]
        row = (e >> 25).astype(np.int64); item = ((e & 0x1ff80) // 128).astype(np.int64); pad = (e & 1).astype(bool)
        assert a % 2 == 0 and (b - a) % 2 == 0, ("odd segment", s, mb, a, b)
        assert (np.diff(row) >= 0).all(), ("rows not ascending", s, mb)
        assert (row[0::2] == row[1::2]).all(), ("pair spans rows", s, mb)
        for rr in range(64):
            r = mb * 64 + rr
            if r >= n: continue
            want = np.sort(R[r, 2:][(R[r, 2:] >> 10) == s] & 1023)
            got = np.sort(item[(row == rr) & ~pad])
            if not (len(want) == len(got) and (want == got).all()):
                bad += 1
                if bad < 5: print("mismatch", s, mb, rr, want[:8], got[:8], len(want), len(got))
print("total entries", S[-1], "candidates", n * neg, "bad units", bad)
