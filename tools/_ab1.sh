export AB_ARGS="--steps 10 --warmup 3 --no-service-point --cpu-sample 0"
bash tools/ab_bench.sh "" s1280 s1024 2>&1 | grep "AB "
AB_ARGS="$AB_ARGS --groups 1" bash tools/ab_bench.sh "" s1280 s1024 2>&1 | grep "rep 1" | sed 's/rep 1/g1   /'
