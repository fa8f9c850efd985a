python -m pytest tests/test_gpu_biglm.py -x -q 2>&1 | tail -3
for g in 2 0; do
python bench.py --biglm --steps 5 --warmup 2 --no-service-point --no-legs --cpu-sample 4 --no-cpu-baseline --max-tokens 131072 --groups $g > gpurun_out/biglm_g.json 2> gpurun_out/biglm_g.err
python - $g <<'PY'
import json,sys
d=json.loads(open("gpurun_out/biglm_g.json").read().strip().splitlines()[-1])
print("groups", sys.argv[1], d["value"], d["ms_per_step"], d["config"].get("parity"), d["roofline"]["kernel_ms_per_step"])
PY
done
