for b in 64 128 256 512; do
timeout 800 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --batch $b 2>gpurun_out/err.log | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('batch', d['config']['global_batch'], round(d['value']), round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['roofline']['kernel_ms_per_step'].items()}, round(d['roofline']['frac'],4))"; done
