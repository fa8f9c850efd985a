#!/bin/bash
# A/B on one box: the beam-13 lattice leg with and without the several-workgroup raw pass (debug 0x40000 = off), variant "ab"
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; mkdir -p gpurun_out/ab
ARGS="--lattice-links 25165824 --steps 6 --cpu-sample 2 --warmup 2 --no-service-point --no-traffic --no-legs --no-cpu-baseline"
for rep in 1 2; do
  for dbg in 0 262144; do
    WFST_LIB_VARIANT=ab timeout 150 python bench.py $ARGS --debug $dbg --detail-out gpurun_out/ab/p13_${dbg}_$rep.json > /dev/null 2> gpurun_out/ab/p13_${dbg}_$rep.err || tail -3 gpurun_out/ab/p13_${dbg}_$rep.err
    python - $dbg $rep <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab/p13_%s_%s.json" % (sys.argv[1], sys.argv[2])))
k = d["roofline"]["kernel_ms_per_step"]
print("beam13 debug %-7s rep %s  %.2f ms/step  expand %.1f insert %.1f closure %.1f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], k["expand"], k["insert"], k["closure"]), flush=True)
PY
  done
done
