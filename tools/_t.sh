#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
timeout 1500 python -m pytest tests/test_gpu_lattice.py tests/test_gpu_running_prune.py tests/test_gpu_fuzz.py tests/test_gpu_biglm.py -x -q -m gpu 2>&1 | tail -3
python3 bench.py --groups 2 --cpu-sample 0 --no-service-point --no-legs --lattice-links 8388608 --steps 3 --warmup 2 --debug 32 2>&1 | grep "per pass\|prune passes" | cut -c1-330
for rep in 1 2; do
python3 bench.py --groups 2 --cpu-sample 2 --no-service-point --no-legs --lattice-links 8388608 --steps 6 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms_per_step']
print('beam13 %.2f ms/step  %.0f f/s expand %.2f insert %.2f closure %.2f %s' % (d['ms_per_step'], d['value'], k['expand'], k['insert'], k['closure'], d['config'].get('parity','')[:40]))"
done
python3 bench.py --groups 2 --cpu-sample 2 --no-service-point --no-legs --beam 15 --lattice-beam 8 --lattice-links 25165824 --arena-per-frame 60000 --max-tokens 262144 --steps 3 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms_per_step']
print('beam15 %.2f ms/step  %.0f f/s expand %.2f insert %.2f closure %.2f %s' % (d['ms_per_step'], d['value'], k['expand'], k['insert'], k['closure'], d['config'].get('parity','')[:40]))"
