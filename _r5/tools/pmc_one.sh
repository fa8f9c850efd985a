#!/bin/bash
# one rocprofv3 --pmc pass:  bash tools/pmc_one.sh "CTR1 CTR2 ..." [bench args]
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; O="$R/gpurun_out/pmc1"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd "$R"
C="$1"; shift
timeout 600 rocprofv3 --pmc $C --output-format csv -d "$O/p" -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-service-point --no-hip-graph "$@" > "$O/p.json" 2> "$O/p.err"
python3 tools/pmc_summary.py "$O/p" | cut -c1-700
find "$O" -name "*_counter_collection.csv" -delete
