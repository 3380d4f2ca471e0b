set -e
for c in config4 config5; do
for nb in 248 992; do
timeout -k 10 1000 python bench.py --workload $c --mode batch --batch $nb --steps 2 --warmup 1 --no-cpu-baseline --no-resident > gpurun_out/bench_${c}_${nb}.json 2> gpurun_out/bench_${c}_${nb}.err || { tail -5 gpurun_out/bench_${c}_${nb}.err; exit 1; }
python - $c $nb <<'PY'
import json, sys
d=json.loads(open(f'gpurun_out/bench_{sys.argv[1]}_{sys.argv[2]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], round(d['value'],1), round(d['ms_per_step'],1), d['proof_accepted'], d['mode'])
PY
done
done
