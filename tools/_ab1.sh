#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
export WFST_BENCH_BREAKDOWN=1
python3 bench.py --cpu-sample 0 --no-service-point --no-legs --steps 10 --warmup 3 2>&1 | grep -E "host-side|ms_per_step" | sed -e 's/.*"ms_per_step": \([0-9.]*\).*/ms_per_step \1/'
python3 bench.py --cpu-sample 0 --no-service-point --no-legs --lattice-links 8388608 --steps 4 --warmup 1 2>&1 | grep -E "host-side|ms_per_step" | sed -e 's/.*"ms_per_step": \([0-9.]*\).*/ms_per_step \1/'
