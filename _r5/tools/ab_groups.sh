#!/bin/bash
# the beam-15 lattice leg (pipelined determinizer) at several channel-group counts, one box
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; mkdir -p gpurun_out/ab
ARGS="--beam 15 --lattice-beam 8 --lattice-links 25165824 --arena-per-frame 60000 --max-tokens 262144 --determinize --pipeline-determinizer --steps 6 --cpu-sample 2 --warmup 2 --no-service-point --no-traffic --no-legs --no-cpu-baseline"
for g in "$@"; do
  timeout 200 python bench.py $ARGS --groups $g --detail-out gpurun_out/ab/groups_$g.json > /dev/null 2> gpurun_out/ab/groups_$g.err || tail -3 gpurun_out/ab/groups_$g.err
  python - $g <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab/groups_%s.json" % sys.argv[1]))
k = d["roofline"]["kernel_ms_per_step"]
print("groups %s  %.2f ms/step  expand %.1f insert %.1f closure %.1f  parity %s" % (sys.argv[1], d["ms_per_step"], k["expand"], k["insert"], k["closure"], d["config"].get("parity", "")[:5]), flush=True)
PY
done
