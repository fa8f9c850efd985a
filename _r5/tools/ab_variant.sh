#!/bin/bash
# the beam-15 lattice leg (pipelined determinizer, or AB_LATTICE_ARGS) with several LIBRARY VARIANTS (build.py --variant NAME;
# "" = the product library), one box, interleaved:   bash tools/ab_variant.sh base ""
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; mkdir -p gpurun_out/ab
ARGS="${AB_LATTICE_ARGS:---beam 15 --lattice-beam 8 --lattice-links 25165824 --arena-per-frame 60000 --max-tokens 262144 --determinize --pipeline-determinizer} --steps ${STEPS:-6} --cpu-sample 2 --warmup 2 --no-service-point --no-traffic --no-legs --no-cpu-baseline"
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    env WFST_LIB_VARIANT=$v timeout 240 python bench.py $ARGS --detail-out gpurun_out/ab/var_${i}_$rep.json > /dev/null 2> gpurun_out/ab/var_${i}_$rep.err || tail -3 gpurun_out/ab/var_${i}_$rep.err
    python - "$i" "$rep" "$v" <<'PY'
import json, sys
i, rep, v = sys.argv[1:4]
try:
    d = json.load(open("gpurun_out/ab/var_%s_%s.json" % (i, rep)))
    k = d["roofline"]["kernel_ms_per_step"]
    print("AB [variant %-12s] rep %s  %.2f ms/step  expand %.1f insert %.1f closure %.1f  parity %s" % (v or "(product)", rep, d["ms_per_step"], k["expand"], k["insert"], k["closure"], str(d["config"].get("parity", ""))[:5]), flush=True)
except Exception as e:
    print("AB [variant %-12s] rep %s  FAILED %r" % (v, rep, e), flush=True)
PY
  done
done
