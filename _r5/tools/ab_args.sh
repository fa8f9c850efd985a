#!/bin/bash
# A/B timing of bench.py argument sets on ONE box (box-to-box spread is ~4 %, more than most kernel changes):
#   gpurun -- 'bash tools/ab_args.sh "" "--debug 65536" ...'     ("" = the defaults)
# Each argument set runs bench.py REPS times (default 2), interleaved (A B C A B C), without the CPU legs; prints ms per step.
R="${GRAFT_REPO_ROOT:-$PWD}"   # (A/B switches of wfst_options.debug need a library built with -DWFST_AB_SWITCHES)
cd "$R"
mkdir -p gpurun_out/ab
COMMON="${AB_COMMON:---steps 10 --warmup 3 --no-service-point --no-legs --cpu-sample 4 --no-cpu-baseline}"
i=0
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    python bench.py $COMMON $v > "gpurun_out/ab/a${i}_$rep.json" 2> "gpurun_out/ab/a${i}_$rep.err" || tail -3 "gpurun_out/ab/a${i}_$rep.err"
    python - "$i" "$rep" "$v" <<'PY'
import json, sys
i, rep, v = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(open("gpurun_out/ab/a%s_%s.json" % (i, rep)).read().strip().splitlines()[-1])
    k = d["roofline"]["kernel_ms_per_step"]
    print("AB [%-40s] rep %s  %.3f ms/step  expand %.2f insert %.2f closure %.2f  %s" % (v, rep, d["ms_per_step"], k["expand"], k["insert"], k["closure"], d["config"].get("parity", "")[:24]), flush=True)
except Exception as e:
    print("AB [%-40s] rep %s  FAILED %r" % (v, rep, e), flush=True)
PY
  done
done
