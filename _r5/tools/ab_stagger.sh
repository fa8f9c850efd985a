#!/bin/bash
# A/B on one box: the pipelined beam-15 lattice leg with the channel groups' frame chains staggered by g x N x 100 us
# (debug 0x80000 | N << 20; library variant "ab").  bash tools/ab_stagger.sh 0 10 20 30
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; mkdir -p gpurun_out/ab
ARGS="--beam 15 --lattice-beam 8 --lattice-links 25165824 --arena-per-frame 60000 --max-tokens 262144 --determinize --pipeline-determinizer --steps 6 --cpu-sample 2 --warmup 2 --no-service-point --no-traffic --no-legs --no-cpu-baseline"
for n in "$@"; do
  dbg=$(( n > 0 ? (524288 + n * 1048576) : 0 ))
  WFST_LIB_VARIANT=ab timeout 200 python bench.py $ARGS --debug $dbg --detail-out gpurun_out/ab/stagger_$n.json > /dev/null 2> gpurun_out/ab/stagger_$n.err || tail -3 gpurun_out/ab/stagger_$n.err
  python - $n <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab/stagger_%s.json" % sys.argv[1]))
print("stagger %s x 100 us per group  %.2f ms/step  parity %s" % (sys.argv[1], d["ms_per_step"], d["config"].get("parity", "")[:5]), flush=True)
PY
done
