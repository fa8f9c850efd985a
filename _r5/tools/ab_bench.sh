#!/bin/bash
# A/B timing of library variants on ONE box (box-to-box spread is ~4 %, more than most kernel changes):
#   python asr-decoder_amd/build.py --variant NAME -DWFST_EXP_...=1      (here, once per variant)
#   gpurun -- 'bash tools/ab_bench.sh "" NAME1 NAME2 ...'                 ("" = the product library)
# Each variant runs bench.py twice, interleaved (A B C A B C), without the CPU legs; prints ms per step.
R="${GRAFT_REPO_ROOT:-$PWD}"
cd "$R"
mkdir -p gpurun_out/ab
ARGS="${AB_ARGS:---steps 10 --warmup 3 --no-service-point --no-traffic --no-legs --cpu-sample 4 --no-cpu-baseline --max-tokens 65536}"
for rep in 1 2; do
  for v in "$@"; do
    tag="${v:-product}"
    WFST_LIB_VARIANT="$v" python bench.py $ARGS > "gpurun_out/ab/${tag}_$rep.json" 2> "gpurun_out/ab/${tag}_$rep.err" || tail -3 "gpurun_out/ab/${tag}_$rep.err"
    python - "$tag" "$rep" <<'PY'
import json, sys
tag, rep = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open("gpurun_out/ab/%s_%s.json" % (tag, rep)).read().strip().splitlines()[-1])
    k = d["roofline"]["kernel_ms_per_step"]
    print("AB %-24s rep %s  %.3f ms/step  expand %.2f insert %.2f closure %.2f" % (tag, rep, d["ms_per_step"], k["expand"], k["insert"], k["closure"]), flush=True)
except Exception as e:
    print("AB %-24s rep %s  FAILED %r" % (tag, rep, e), flush=True)
PY
  done
done
