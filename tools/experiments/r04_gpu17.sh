for mid in 384 0; do
LWKZG_MID_PROOF_HOST=$mid python bench.py --op blob_proof --batch 256 --no-cpu-baseline > /tmp/l.json 2>/dev/null
python - <<PY
import json
l=json.load(open("/tmp/l.json")); d=json.load(open("bench_detail.json"))
print("mid=$mid", round(l["value"]), round(l["ms_per_step"],2), {k:(v["launches"], round(v["avg_ms"],3)) for k,v in d["kernels"].items()})
PY
done
LWKZG_MID_PROOF_HOST=384 python bench.py --op blob_proof --batch 256 --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('steps 40:', l['value'], l['ms_per_step'])"
