#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/profile_round.sh to the small summaries kept under profiles/.

    python tools/summarize_prof.py gpurun_out/prof_<tag> <tag>

Writes (into gpurun_out/prof_<tag>/summary/, to be copied to profiles/):
  <tag>_<run>_kernel_stats.csv      per-kernel calls / total / average duration (from *kernel_stats.csv)
  <tag>_pmc_per_launch.json         per kernel: launches and average FETCH_SIZE / WRITE_SIZE per launch
                                    (raw counter units as rocprofv3 reports them: KB)
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    """k_transfer_fwd<32, 1, 4>(...) -> k_transfer_fwd<32,1,4>"""
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace(" ", "")


def kernel_stats(run_dir):
    rows = {}
    for path in glob.glob(os.path.join(run_dir, "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r.get("Name", ""))
                calls = int(float(r.get("Calls", 0)))
                tot = float(r.get("TotalDurationNs", 0.0))
                e = rows.setdefault(k, [0, 0.0])
                e[0] += calls
                e[1] += tot
    return rows


def counters(run_dir, counter):
    acc = {}
    for path in glob.glob(os.path.join(run_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                k = short(r.get("Kernel_Name", ""))
                e = acc.setdefault(k, [0, 0.0])
                e[0] += 1
                e[1] += float(r.get("Counter_Value", 0.0))
    return acc


def all_counters(run_dir):
    """{kernel: {counter: [launches, sum]}} over every counter in the run."""
    acc = {}
    for path in glob.glob(os.path.join(run_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                e = acc.setdefault(short(r.get("Kernel_Name", "")), {}).setdefault(r.get("Counter_Name"), [0, 0.0])
                e[0] += 1
                e[1] += float(r.get("Counter_Value", 0.0))
    return acc


# fp32 MFMA work of one instruction, for the achieved-FLOP cross-check: SQ_INSTS_VALU_MFMA_MOPS_F32 counts
# MFMA "mega-ops" in units of 512 flops as rocprofv3 reports it on gfx950 (v_mfma_f32_16x16x4_f32 = 2*16*16*4 = 2,048 flops = 4 units)
def mfma_summary(run_dir):
    out = {}
    for k, cs in all_counters(run_dir).items():
        if not k.startswith("k_t"):
            continue
        n = max(cs.get("GRBM_GUI_ACTIVE", [1, 0])[0], 1)
        avg = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
        e = {"launches": n, "avg_per_launch": avg}
        busy, act = avg.get("SQ_VALU_MFMA_BUSY_CYCLES"), avg.get("GRBM_GUI_ACTIVE")
        if busy and act:
            # the SQ counter is summed over the chip's 1,024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE is the launch's
            # active cycles per shader engine sample: busy fraction of the matrix pipes = busy / (active * 1024)
            e["mfma_busy_frac_of_all_simds"] = busy / (act * 1024.0)
        out[k] = e
    return out


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(root, "summary")
    os.makedirs(out, exist_ok=True)
    pmc = {}
    for run in sorted(os.listdir(root)):
        d = os.path.join(root, run)
        if not os.path.isdir(d) or run == "summary":
            continue
        if run.endswith("_stats"):
            rows = kernel_stats(d)
            if rows:
                with open(os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, run[:-6])), "w") as f:
                    f.write("kernel,calls,total_ms,avg_us\n")
                    for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
                        f.write('"%s",%d,%.3f,%.2f\n' % (k, c, t / 1e6, t / 1e3 / max(c, 1)))
        if run.endswith("_mfma"):
            m = mfma_summary(d)
            if m:
                with open(os.path.join(out, "%s_pmc_mfma.json" % tag), "w") as f:
                    json.dump(m, f, indent=1, sort_keys=True)
        for suffix, counter in (("_fetch", "FETCH_SIZE"), ("_write", "WRITE_SIZE")):
            if run.endswith(suffix):
                for k, (c, v) in counters(d, counter).items():
                    pmc.setdefault(run[:-len(suffix)], {}).setdefault(k, {})[counter] = {
                        "launches": c, "avg_counter_per_launch": v / max(c, 1)}
    if pmc:
        with open(os.path.join(out, "%s_pmc_per_launch.json" % tag), "w") as f:
            json.dump(pmc, f, indent=1, sort_keys=True)
    print("summaries in", out, ":", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
