#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag>
# Kernel-trace stats and the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, they do not fit one
# pass) for the headline period workload and for the bare a3 step; raw output under gpurun_out/prof_<tag>/,
# summaries (what gets committed) under profiles/ via tools/summarize_prof.py.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
PERIOD="bench.py --no-cpu --no-a3 --steps 4 --warmup 2"      # (the step counter passes 1,024 -- closed-form replay -- in the second period)
BARE="bench.py --workload bare --users 10000000 --items 1000000 --bare-batch 262144 --steps 2 --warmup 1"
run() { # name, rocprof args..., -- cmd
    local name=$1; shift
    timeout 900 rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || echo "rocprofv3 $name failed (see $OUT/$name.log)"
}
run period_stats --kernel-trace --stats --output-format csv -d "$OUT/period_stats" -- python3 $PERIOD
# (counter passes on ONE queue: rocprofv3 --pmc crashes in the launch path when the side-stream evaluations
# share the device with the training stream; the kernels and their traffic are the same)
run period_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/period_fetch" -- python3 $PERIOD --no-roofline --no-overlap
run period_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/period_write" -- python3 $PERIOD --no-roofline --no-overlap
# MFMA utilisation of the transfer kernels (north_star: "rocprof ... MFMA utilisation against MI355X peak"): busy cycles of
# the matrix pipe, fp32 MFMA ops and the launch's active cycles, per kernel, in a counter pass of their own
run period_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/period_mfma" -- python3 $PERIOD --no-roofline --no-overlap
for z in 0 1; do
    run bare_z${z}_stats --kernel-trace --stats --output-format csv -d "$OUT/bare_z${z}_stats" -- python3 $BARE --item-zipf $z
    run bare_z${z}_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/bare_z${z}_fetch" -- python3 $BARE --item-zipf $z
    run bare_z${z}_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/bare_z${z}_write" -- python3 $BARE --item-zipf $z
done
python3 tools/summarize_prof.py "$OUT" "$TAG"
# gpurun merges at most 64 MiB back: the raw traces of seven periods exceed that -- keep the logs and the summaries
if [ -z "${KEEP_RAW:-}" ]; then
    for d in "$OUT"/*/; do
        case "$d" in */summary/) ;; *) rm -rf "$d" ;; esac
    done
fi
