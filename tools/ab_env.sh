#!/bin/bash
# A/B of bench.py under environment settings on ONE box: each argument is "VAR=val VAR2=val|bench args" (either side may be empty).
#   gpurun -- 'bash tools/ab_env.sh "|" "GPU_MAX_HW_QUEUES=8|--groups 4"'
R="${GRAFT_REPO_ROOT:-$PWD}"
cd "$R"
mkdir -p gpurun_out/abe
COMMON="${AB_COMMON:---steps 10 --warmup 3 --no-service-point --no-legs --cpu-sample 4 --no-cpu-baseline}"
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    e="${v%%|*}"; a="${v#*|}"
    env $e python bench.py $COMMON $a > "gpurun_out/abe/a${i}_$rep.json" 2> "gpurun_out/abe/a${i}_$rep.err" || tail -3 "gpurun_out/abe/a${i}_$rep.err"
    python - "$i" "$rep" "$v" <<'PY'
import json, sys
i, rep, v = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(open("gpurun_out/abe/a%s_%s.json" % (i, rep)).read().strip().splitlines()[-1])
    k = d["roofline"]["kernel_ms_per_step"]
    print("AB [%-44s] rep %s  %.3f ms/step  expand %.2f insert %.2f closure %.2f  %s" % (v, rep, d["ms_per_step"], k["expand"], k["insert"], k["closure"], d["config"].get("parity", "")[:24]), flush=True)
except Exception as e:
    print("AB [%-44s] rep %s  FAILED %r" % (v, rep, e), flush=True)
PY
  done
done
