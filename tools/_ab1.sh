export AB_ARGS="--steps 10 --warmup 3 --no-service-point --no-legs --cpu-sample 0"
bash tools/ab_bench.sh "" t512 t384 t320 2>&1 | grep "AB "
