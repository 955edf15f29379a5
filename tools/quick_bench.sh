python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep > gpurun_out/qb.json 2>gpurun_out/qb.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/qb.json').read().strip().splitlines()[-1])
for k in ("value","long_run","replayed_batch","index_order","hw_dispatch","queue_packed","full_lds_layout","unrelated_batches"):
    v=d.get(k); print(k, v if not isinstance(v,dict) else round(v["value"]))
PY
tail -3 gpurun_out/qb.err
