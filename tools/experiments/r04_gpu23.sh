for rep in 1 2; do for lanes in 128 256 64; do
LWKZG_REDUCE_LANES=$lanes python bench.py --direct-bits 0 --no-config-legs --no-cpu-baseline --steps 10 > /tmp/l.json 2>/dev/null
python - <<PY
import json
d=json.load(open("bench_detail.json"))
print("lanes=$lanes", round(d["value"]), {k: round(v["avg_ms"],3) for k,v in d["kernels"].items()})
PY
done; done
