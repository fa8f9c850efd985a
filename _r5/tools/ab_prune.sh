#!/bin/bash
# A/B on one box: the beam-15 lattice leg (no determinizer) with the raw frames of the back-pruning on several workgroups per
# channel (default) and on one (debug 0x40000), library variant "ab" (built with -DWFST_AB_SWITCHES).
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; mkdir -p gpurun_out/ab
ARGS="--beam 15 --lattice-beam 8 --lattice-links 25165824 --arena-per-frame 60000 --max-tokens 262144 --steps 4 --cpu-sample 2 --warmup 2 --no-service-point --no-traffic --no-legs --no-cpu-baseline"
for rep in 1 2; do
  for dbg in 0 262144; do
    WFST_LIB_VARIANT=ab timeout 150 python bench.py $ARGS --debug $dbg --detail-out gpurun_out/ab/prune_${dbg}_$rep.json > gpurun_out/ab/prune_${dbg}_$rep.line 2> gpurun_out/ab/prune_${dbg}_$rep.err || tail -3 gpurun_out/ab/prune_${dbg}_$rep.err
    python - $dbg $rep <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab/prune_%s_%s.json" % (sys.argv[1], sys.argv[2])))
k = d["roofline"]["kernel_ms_per_step"]
print("AB debug %-7s rep %s  %.2f ms/step  expand %.1f insert %.1f closure %.1f  parity %s | %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], k["expand"], k["insert"], k["closure"], d["config"].get("parity", "")[:5], d["config"].get("lattice_parity", "")[:4]), flush=True)
PY
  done
done
