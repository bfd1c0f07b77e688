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


# SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles the matrix pipes were busy, SUMMED over the chip's SIMDs (checked on the
# table-sized forward: 8.78 M wave-level v_mfma_f32_16x16x4_f32 x 32 cycles per SIMD each = 281 M, the counter's value);
# SQ_INSTS_VALU_MFMA_MOPS_F32 counts fp32 MFMA work in units of 512 flops (35.1 M x 512 = 18.0 GFLOP = the launch's
# fc1 + fc2 flops); GRBM_GUI_ACTIVE is summed over the 8 XCDs.  MFMA-busy fraction of a kernel = busy cycles / (1,024
# SIMDs x the kernel's duration x the shader clock).  The duration comes from the UN-counted kernel trace of the same
# command (counter passes serialise and slow the launches), the clock is the 2.4 GHz nominal: the fraction is of ALL
# 256 CUs' matrix pipes (the training stream owns 192 of them: divide by 0.75 for the share of what it could use).
N_SIMD, CLOCK_MHZ = 1024, 2400.0


def mfma_summary(run_dir, durations_us):
    out = {}
    for k, cs in all_counters(run_dir).items():
        if not k.startswith(("k_t", "k_mf_fwd")):
            continue
        n = max(cs.get("GRBM_GUI_ACTIVE", [1, 0])[0], 1)
        avg = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
        e = {"launches": n, "avg_per_launch": avg}
        busy, act = avg.get("SQ_VALU_MFMA_BUSY_CYCLES"), avg.get("GRBM_GUI_ACTIVE")
        if busy is not None and act:
            e["mfma_busy_frac_in_the_counter_pass"] = busy / (act / 8.0 * N_SIMD)
        dur = durations_us.get(k)
        if busy is not None and dur:
            e["avg_us_kernel_trace"] = dur
            e["mfma_busy_frac_of_all_simds"] = busy / (N_SIMD * dur * CLOCK_MHZ)
        mops = avg.get("SQ_INSTS_VALU_MFMA_MOPS_F32")
        if mops is not None and dur:
            e["mfma_tflops_by_counter"] = mops * 512.0 / (dur * 1e-6) / 1e12
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
                name = "yelp_period" if run[:-6] == "period" else run[:-6]      # (the name bench.py's rocprof_avg_us looks for)
                with open(os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, name)), "w") as f:
                    f.write("kernel,calls,total_ms,avg_us\n")
                    for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
                        f.write('"%s",%d,%.3f,%.2f\n' % (k, c, t / 1e6, t / 1e3 / max(c, 1)))
        if run.endswith("_mfma"):
            stats = kernel_stats(os.path.join(root, run[:-5] + "_stats"))
            m = mfma_summary(d, {k: t / 1e3 / max(c, 1) for k, (c, t) in stats.items()})
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
    # the index preparation's own rate (bare runs): per prepared epoch, the k_prep_* kernels' time (kernel trace) and
    # fabric bytes (FETCH x 2 + WRITE, KB -> bytes: the gfx950 correction of MI355X_MICROARCH.md)
    prep = {}
    for run in sorted(os.listdir(root)):
        if not (run.startswith("bare_") and run.endswith("_stats")):
            continue
        name = run[:-6]
        rows = {k: v for k, v in kernel_stats(os.path.join(root, run)).items() if k.startswith("k_prep_")}
        epochs = rows.get("k_prep_hist", [0, 0.0])[0]
        if not epochs:
            continue
        e = {"epochs_prepared": epochs, "triples_per_epoch": 1 << 22,
             "us_per_epoch": sum(t for _, t in rows.values()) / 1e3 / epochs,
             "kernels_us_per_epoch": {k: t / 1e3 / epochs for k, (c, t) in sorted(rows.items())}}
        c = pmc.get(name, {})
        by = 0.0
        for k in rows:
            if k in c and "FETCH_SIZE" in c[k] and "WRITE_SIZE" in c[k]:
                by += c[k]["FETCH_SIZE"]["launches"] * (2.0 * c[k]["FETCH_SIZE"]["avg_counter_per_launch"]
                                                        + c[k]["WRITE_SIZE"]["avg_counter_per_launch"]) * 1024.0
        if by:
            e["fabric_bytes_per_epoch"] = by / epochs
            e["fabric_GBps"] = by / epochs / (e["us_per_epoch"] * 1e-6) / 1e9
            e["algorithmic_bytes_per_epoch"] = (1 << 22) * (24 + 3)       # the triples once + a mark per occurrence
        prep[name] = e
    if prep:
        with open(os.path.join(out, "%s_index_prep.json" % tag), "w") as f:
            json.dump(prep, f, indent=1, sort_keys=True)
    print("summaries in", out, ":", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
