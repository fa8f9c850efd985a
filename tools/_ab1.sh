python -m pytest tests/test_gpu_lattice.py tests/test_gpu_running_prune.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --no-service-point --no-legs --cpu-sample 2 --no-cpu-baseline --lattice-links 6291456 --arena-per-frame 20000 --max-tokens 131072 --debug 32 > gpurun_out/lat13.json 2> gpurun_out/lat13.err; grep "wfst dbg\] p" gpurun_out/lat13.err
python bench.py --steps 2 --warmup 1 --no-service-point --no-legs --cpu-sample 2 --no-cpu-baseline --beam 15 --lattice-beam 8 --lattice-links 25165824 --arena-per-frame 60000 --max-tokens 262144 --debug 32 > gpurun_out/lat15.json 2> gpurun_out/lat15.err; grep "wfst dbg\] p" gpurun_out/lat15.err
python - <<'PY'
import json
for n in ("lat13","lat15"):
    d=json.loads(open("gpurun_out/%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["config"].get("parity"), d["config"].get("lattice_parity"), d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"])
PY
