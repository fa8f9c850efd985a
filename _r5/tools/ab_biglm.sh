#!/bin/bash
# the biglm leg (BASELINE configs[3]) under several argument sets, one box:   bash tools/ab_biglm.sh "" "--groups 4" ...
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; mkdir -p gpurun_out/ab
ARGS="--biglm --max-tokens 131072 --steps ${STEPS:-10} --cpu-sample 2 --warmup 3 --no-service-point --no-traffic --no-legs --no-cpu-baseline"
for rep in $(seq 1 ${REPS:-1}); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    timeout 240 python bench.py $ARGS $v --detail-out gpurun_out/ab/big_${i}_$rep.json > /dev/null 2> gpurun_out/ab/big_${i}_$rep.err || tail -3 gpurun_out/ab/big_${i}_$rep.err
    python - "$i" "$rep" "$v" <<'PY'
import json, sys
i, rep, v = sys.argv[1:4]
try:
    d = json.load(open("gpurun_out/ab/big_%s_%s.json" % (i, rep)))
    k = d["roofline"]["kernel_ms_per_step"]
    print("AB [%-34s] rep %s  %.2f ms/step  expand %.1f insert %.1f closure %.1f  parity %s groups %s" % (v, rep, d["ms_per_step"], k["expand"], k["insert"], k["closure"], str(d["config"].get("parity", ""))[:5], d["config"].get("channel_groups")), flush=True)
except Exception as e:
    print("AB [%-34s] rep %s  FAILED %r" % (v, rep, e), flush=True)
PY
  done
done
