#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh
# then, back in the container:  python tools/summarize_profiles.py r01
# Kernel trace + stats and the two PMC passes are separate runs (gpurun refuses --pmc together with
# the trace domains); the profiled program is python3 itself.
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"
O="$R/gpurun_out/prof"
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd "$R"
# One channel group (--groups 1): every launch covers the whole batch and nothing overlaps, so the per-kernel average
# durations are the kernels' own (the tracer serialises the two streams of the default two-group run anyway: a traced
# two-group run measures neither the overlap nor the kernels; it is kept below as kt_groups2 for the record).
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 bench.py --groups 1 --steps 2 --warmup 1 --cpu-sample 0 --no-service-point > "$O/bench_kt.json" 2> "$O/kt.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_groups2" -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-service-point > "$O/bench_kt_groups2.json" 2> "$O/kt_groups2.err"
# counter passes: kernels enqueued one by one (--no-hip-graph) so that every dispatch is attributed
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 bench.py --groups 1 --steps 1 --warmup 0 --cpu-sample 0 --no-service-point --no-hip-graph > "$O/bench_fetch.json" 2> "$O/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 bench.py --groups 1 --steps 1 --warmup 0 --cpu-sample 0 --no-service-point --no-hip-graph > "$O/bench_write.json" 2> "$O/write.err"
# lattice mode (insert_kernel<true>, closure_kernel<true>, lattice_prune_kernel, nbest_kernel)
B=128 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_lattice" -- python3 tools/lattice_fullsize.py > "$O/lattice.log" 2> "$O/lattice.err"
# keep the merge-back small: the per-dispatch traces are not needed, the stats and counter CSVs are
find "$O" -name "*_kernel_trace.csv" -size +8M -delete
find "$O" -name "*_counter_collection.csv" -size +40M -delete
ls -la "$O"/*/* | head -40
tail -1 "$O/bench_kt.json" | cut -c1-300
