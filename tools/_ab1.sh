python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
AB_ARGS="--steps 10 --warmup 3 --no-service-point --no-legs --cpu-sample 0" bash tools/ab_bench.sh "" 2>&1 | grep "AB "
