# scratch: channel groups / workgroup counts sweep (bench value only)
run() { echo -n "$* : "; env "$@" python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"; }
run WFST_GROUPS=1
run WFST_GROUPS=2
run WFST_GROUPS=4
run WFST_GROUPS=8
run WFST_GROUPS=2 WFST_EXPAND_WGS=1024 WFST_INSERT_WGS=384
run WFST_GROUPS=4 WFST_EXPAND_WGS=512 WFST_INSERT_WGS=192
run WFST_GROUPS=4 WFST_NO_GRAPH=1
