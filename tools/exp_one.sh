run() { echo -n "$* : "; env "$@" python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"; }
run WFST_LOG2_PARTS=4
run WFST_LOG2_PARTS=5
run WFST_LOG2_PARTS=5 WFST_JOINT_MAX=1280
run WFST_LOG2_PARTS=5 WFST_EXPAND_WGS=1024
