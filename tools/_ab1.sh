#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
python3 bench.py --groups 1 --cpu-sample 0 --no-service-point --no-legs --beam 15 --arena-per-frame 60000 --max-tokens 262144 --steps 3 --warmup 2 --debug 64 2>&1 | grep "insert:\|ms_per_step" | sed -e 's/.*"ms_per_step": \([0-9.]*\).*"kernel_ms_per_step": \({[^}]*}\).*/ms_per_step \1 \2/'
python3 bench.py --groups 1 --cpu-sample 0 --no-service-point --no-legs --beam 15 --lattice-beam 8 --lattice-links 25165824 --arena-per-frame 60000 --max-tokens 262144 --steps 3 --warmup 2 --debug 64 2>&1 | grep "insert:\|ms_per_step" | sed -e 's/.*"ms_per_step": \([0-9.]*\).*"kernel_ms_per_step": \({[^}]*}\).*/ms_per_step \1 \2/'
