export WFST_BENCH_BREAKDOWN=1
timeout 500 python bench.py --steps 3 --warmup 1 --cpu-sample 8 2>gpurun_out/err.log | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config']['parity'][:5], {k:round(v,1) for k,v in d['roofline']['kernel_ms_per_step'].items()}, d['roofline']['kernel'], round(d['roofline']['frac'],4))"; grep host-side gpurun_out/err.log
