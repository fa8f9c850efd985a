#!/bin/bash
# quick bench: prints ms/step, kernel split, parity
python bench.py --cpu-sample ${CPU_SAMPLE:-4} --cpu-seconds 0.5 --no-service-point --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['kernel_ms_per_step'].items()}, d['config'].get('parity'))"
