python -m pytest tests/test_gpu_lattice.py tests/test_gpu_running_prune.py tests/test_gpu_fuzz.py tests/test_gpu_determinize.py tests/test_gpu_lifecycle.py -x -q 2>&1 | tail -6
python bench.py --steps 3 --warmup 1 --no-service-point --no-legs --cpu-sample 2 --no-cpu-baseline --lattice-links 6291456 --arena-per-frame 20000 --max-tokens 131072 > gpurun_out/lat13.json 2> gpurun_out/lat13.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/lat13.json").read().strip().splitlines()[-1])
print("lat13", d["value"], d["ms_per_step"], d["config"].get("parity"), d["config"].get("lattice_parity"), d["roofline"]["kernel_ms_per_step"])
PY
