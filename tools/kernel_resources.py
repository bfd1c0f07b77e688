#!/usr/bin/env python3
"""Register / spill / scratch / LDS table of every kernel in sml_amd/csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage,
gfx950, device code only, the product's flags).  Exit code 1 if any kernel spills or touches scratch memory (round 6: a patched
local COPY of a by-value argument struct that is indexed dynamically anywhere lives in scratch -- caught here).
usage: python tools/kernel_resources.py [file.hip ...]"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "sml_amd", "csrc")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return [re.sub(r"\(anonymous namespace\)::|^void ", "", o).split("(")[0] for o in out]
    except OSError:
        return names


def report(path):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16",
           "--cuda-device-only", "-c", path, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (?:Function Name: (\S+)|\s*([A-Za-z \[\]/]+): (\d+))", line)
        if not m:
            continue
        if m.group(1):
            cur = {"name": m.group(1)}
            rows.append(cur)
        elif cur is not None:
            cur[m.group(2).strip()] = int(m.group(3))
    for r, n in zip(rows, demangle([r["name"] for r in rows])):
        r["name"] = n
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    files = args or [os.path.join(CSRC, f) for f in ("transfer_net.hip", "mf_kernels.hip")]
    print("%-52s %5s %5s %6s %6s %7s %7s %4s" % ("kernel", "VGPR", "AGPR", "spillV", "spillS", "scratch", "LDS", "occ"))
    bad = 0
    for f in files:
        for r in sorted(report(f), key=lambda r: r["name"]):
            sv = r.get("VGPRs Spill", 0)
            sc = r.get("ScratchSize [bytes/lane]", 0)
            bad += (sv > 0) or (sc > 0)
            print("%-52s %5d %5d %6d %6d %7d %7d %4d" % (r["name"][:52], r.get("VGPRs", 0), r.get("AGPRs", 0), sv, r.get("SGPRs Spill", 0), sc,
                                                       r.get("LDS Size [bytes/block]", 0), r.get("Occupancy [waves/SIMD]", 0)))
    print("kernels with spilled VGPRs or scratch:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
