run() { echo -n "$* : "; env "$@" python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"; }
run A=1
run A=1
