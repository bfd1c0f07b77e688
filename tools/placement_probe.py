#!/usr/bin/env python3
"""Which CUs does a CU-masked HIP stream use?  Prints, per mask, the number of distinct (XCD, SE, CU) seen and per-XCD counts."""
import ctypes, os, sys, collections
import numpy as np
import torch  # loads the HIP runtime
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "placement_probe.so"))
lib.probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
torch.zeros(1, device="cuda")


def run(mask_bits, n_blocks=2048, threads=512, spin=2000):
    words = np.zeros(8, dtype=np.uint32)
    for b in mask_bits:
        words[b // 32] |= np.uint32(1 << (b % 32))
    out = np.zeros((n_blocks, 2), dtype=np.uint32)
    rc = lib.probe(words.ctypes.data, 8 if mask_bits is not None and len(mask_bits) else 0, n_blocks, threads, spin, out.ctypes.data)
    assert rc == 0, rc
    xcc, hw = out[:, 0], out[:, 1]
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    keys = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per = collections.Counter(k[0] for k in keys)
    return len(keys), dict(sorted(per.items())), keys


if __name__ == "__main__":
    n, per, all_keys = run([])
    print("no mask:", n, per)
    for name, bits in (("bits 0..63", range(64)), ("bits 0..191", range(192)), ("bits 192..255", range(192, 256)),
                       ("every 4th bit", range(0, 256, 4)), ("bits 0..7", range(8)), ("bits 8..15", range(8, 16))):
        n, per, keys = run(list(bits))
        print(name, "->", n, "CUs", per)
    a = run(list(range(192)))[2]
    b = run(list(range(192, 256)))[2]
    print("overlap of [0,192) and [192,256):", len(a & b))
