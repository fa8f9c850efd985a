#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
for g in 3 3; do
python3 bench.py --biglm --max-tokens 131072 --groups $g --cpu-sample 0 --no-service-point --no-legs --steps 6 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms_per_step']
print('groups $g  %.2f ms/step  expand %.2f insert %.2f closure %.2f' % (d['ms_per_step'], k['expand'], k['insert'], k['closure']))"
done
timeout 900 python -m pytest tests/test_gpu_biglm.py -x -q -m gpu 2>&1 | tail -3
