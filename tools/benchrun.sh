for cfg in "11 768 1024" "12 768 768" "12 1536 768"; do set -- $cfg; export WFST_LOG2_LDS_SLOTS=$1 WFST_JOINT_MAX=$2 WFST_INSERT_WGS=$3;
timeout 500 python bench.py --steps 3 --warmup 1 --cpu-sample 8 2>gpurun_out/err.log | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('lds',os.environ['WFST_LOG2_LDS_SLOTS'], 'joint', os.environ['WFST_JOINT_MAX'], 'wgs', os.environ['WFST_INSERT_WGS'], round(d['value']), round(d['ms_per_step'],2), d['config']['parity'][:5], {k:round(v,1) for k,v in d['roofline']['kernel_ms_per_step'].items()})"; done
