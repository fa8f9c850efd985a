#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh
# then, back in the container:  python tools/summarize_profiles.py r03
# For each of the three configurations the bench line reports (headline = BASELINE configs[1], biglm = configs[3],
# lattice_beam15 = configs[4]): one kernel trace with --stats and two counter passes (FETCH_SIZE, WRITE_SIZE).  Trace and
# counters are separate runs (gpurun refuses --pmc together with the trace domains); the profiled program is python3 itself.
# One channel group (--groups 1): every launch covers the whole batch and nothing overlaps, so the per-kernel average
# durations are the kernels' own (the tracer serialises the two streams of the default two-group run anyway).
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"
O="$R/gpurun_out/prof"
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd "$R"
COMMON="--groups 1 --cpu-sample 0 --no-service-point --no-traffic --no-legs"
declare -A CFG
CFG[headline]=""
CFG[biglm]="--biglm --max-tokens 131072"
CFG[lattice_beam15]="--beam 15 --lattice-beam 8 --lattice-links 25165824 --arena-per-frame 60000 --max-tokens 262144 --determinize"
for name in headline biglm lattice_beam15; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_$name" -- python3 bench.py $COMMON ${CFG[$name]} --steps 2 --warmup 1 > "$O/bench_kt_$name.json" 2> "$O/kt_$name.err"
  # counter passes: kernels enqueued one by one (--no-hip-graph) so that every dispatch is attributed
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch_$name" -- python3 bench.py $COMMON ${CFG[$name]} --steps 1 --warmup 0 --no-hip-graph > "$O/bench_fetch_$name.json" 2> "$O/fetch_$name.err"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write_$name" -- python3 bench.py $COMMON ${CFG[$name]} --steps 1 --warmup 0 --no-hip-graph > "$O/bench_write_$name.json" 2> "$O/write_$name.err"
done
# the DEFAULT (two-group) headline run under the tracer, for the record
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_groups2" -- python3 bench.py --cpu-sample 0 --no-service-point --no-traffic --no-legs --steps 2 --warmup 1 > "$O/bench_kt_groups2.json" 2> "$O/kt_groups2.err"
# reduce the per-dispatch files to per-kernel sums here (the merge-back is limited to 64 MiB)
python3 - "$O" <<'PY'
import collections, csv, glob, json, os, sys
O = sys.argv[1]
for d in sorted(glob.glob(os.path.join(O, "fetch_*")) + glob.glob(os.path.join(O, "write_*"))):
    fs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("wfst::", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    json.dump({k: {"launches": n, "sum_kb": v} for k, (n, v) in agg.items()}, open(d + ".json", "w"), indent=1)
    os.remove(fs[0])
PY
find "$O" -name "*_kernel_trace.csv" -delete
ls "$O"
