export AB_ARGS="--steps 10 --warmup 3 --no-service-point --no-legs --cpu-sample 0"
bash tools/ab_bench.sh "" s1392 2>&1 | grep "AB "
