#!/bin/bash
# the beam-15 lattice leg (pipelined determinizer, or AB_LATTICE_ARGS) under several ENVIRONMENT settings read by the "ab" library
# variant (built with -DWFST_AB_SWITCHES), one box, interleaved:   bash tools/ab_env.sh "" "WFST_PRUNE_RAW_MIN=300000" ...
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; mkdir -p gpurun_out/ab
ARGS="${AB_LATTICE_ARGS:---beam 15 --lattice-beam 8 --lattice-links 25165824 --arena-per-frame 60000 --max-tokens 262144 --determinize --pipeline-determinizer} --steps ${STEPS:-6} --cpu-sample 2 --warmup 2 --no-service-point --no-traffic --no-legs --no-cpu-baseline"
for rep in $(seq 1 ${REPS:-1}); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    env WFST_LIB_VARIANT=ab $v timeout 240 python bench.py $ARGS --detail-out gpurun_out/ab/env_${i}_$rep.json > /dev/null 2> gpurun_out/ab/env_${i}_$rep.err || tail -3 gpurun_out/ab/env_${i}_$rep.err
    python - "$i" "$rep" "$v" <<'PY'
import json, sys
i, rep, v = sys.argv[1:4]
try:
    d = json.load(open("gpurun_out/ab/env_%s_%s.json" % (i, rep)))
    k = d["roofline"]["kernel_ms_per_step"]
    print("AB [%-34s] rep %s  %.2f ms/step  expand %.1f insert %.1f closure %.1f  parity %s" % (v, rep, d["ms_per_step"], k["expand"], k["insert"], k["closure"], str(d["config"].get("parity", ""))[:5]), flush=True)
except Exception as e:
    print("AB [%-34s] rep %s  FAILED %r" % (v, rep, e), flush=True)
PY
  done
done
