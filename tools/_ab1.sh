for mu in -2.6 -2.7 -2.8; do
python bench.py --steps 2 --warmup 1 --no-service-point --no-legs --cpu-sample 0 --workload single --mu $mu --max-active 7000 --min-active 200 --max-tokens 131072 > gpurun_out/sp_mu.json 2> gpurun_out/sp_mu.err
python - $mu <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sp_mu.json").read().strip().splitlines()[-1])
print("mu",sys.argv[1], "mean active", d["config"]["mean_active_tokens_per_frame"], "expanded", d["config"]["mean_expanded_tokens_per_frame"], "peak", d["config"]["peak_tokens_in_a_frame"], "ms", d["ms_per_step"])
PY
done
