for b in 128 64; do for c in 0 100 25; do for hf in "" "--host-feed"; do
echo "== batch $b chunk $c $hf"; WFST_BENCH_CHUNK=$c python bench.py --batch $b --cpu-sample 4 --cpu-seconds 0.5 --no-service-point --no-legs --no-traffic --no-cpu-baseline --steps 6 --warmup 2 $hf 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],2), 'frames/s', round(d['value']), d['config'].get('parity'))"
done; done; done
