run() { echo -n "$* : "; env "$@" python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['config']['mean_active_tokens_per_frame']; print(round(d['value']), round(d['ms_per_step'],2), 'tokens/frame', round(t), 'ms per 1k mean tokens', round(d['ms_per_step']/t*1000,2), d['roofline']['kernel_ms_per_step'])"; }
run A=1
run WFST_BENCH_SAME_UTT=0
run WFST_BENCH_SAME_UTT=5
run WFST_BENCH_SAME_UTT=17
run WFST_BENCH_SAME_UTT=64
