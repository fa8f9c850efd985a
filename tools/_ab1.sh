python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_token_gc.py tests/test_gpu_fuzz.py tests/test_gpu_lattice.py -x -q 2>&1 | tail -6
for rep in 1 2; do
AB_ARGS="--steps 10 --warmup 3 --no-service-point --cpu-sample 0" bash tools/ab_bench.sh "" 2>&1 | grep "rep 1"
AB_ARGS="--steps 10 --warmup 3 --no-service-point --cpu-sample 0 --debug 32768" bash tools/ab_bench.sh "" 2>&1 | grep "rep 1" | sed 's/product/oldexp /'
done
AB_ARGS="--steps 10 --warmup 3 --no-service-point --cpu-sample 4" bash tools/ab_bench.sh "" 2>&1 | grep "rep 1"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/ab/product_1.json").read().strip().splitlines()[-1])
print(d["config"].get("parity"), d["config"].get("work_counts_sample"))
PY
