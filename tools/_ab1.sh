export AB_ARGS="--steps 4 --warmup 2 --no-service-point --cpu-sample 0 --workload single --max-active 7000 --min-active 200 --max-tokens 131072"
bash tools/ab_bench.sh "" 2>&1 | grep "rep 1"
AB_ARGS="$AB_ARGS --debug 32768" bash tools/ab_bench.sh "" 2>&1 | grep "rep 1" | sed 's/product/oldexp /'
AB_ARGS="$AB_ARGS --debug 16384" bash tools/ab_bench.sh "" 2>&1 | grep "rep 1" | sed 's/product/noseedt/'
AB_ARGS="$AB_ARGS --debug 49152" bash tools/ab_bench.sh "" 2>&1 | grep "rep 1" | sed 's/product/old+nos/'
