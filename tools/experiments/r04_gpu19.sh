timeout 900 python -m pytest tests/test_gpu_proof_parity.py -m gpu -x -q -k "one_pass or commit_and_prove or pass" 2>&1 | tail -3
for mid in 384 0; do
LWKZG_MID_PROOF_HOST=$mid python bench.py --op commit_prove --batch 256 --no-cpu-baseline > /tmp/l.json 2>/dev/null
python - <<PY
import json
l=json.load(open("/tmp/l.json")); d=json.load(open("bench_detail.json"))
print("mid=$mid", round(l["value"]), round(l["ms_per_step"],2), {k:(v["launches"], round(v["avg_ms"],3)) for k,v in d["kernels"].items() if "challenge" in k or "accumulate" in k or "midstate" in k})
PY
done
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print(l['value'], {k:(round(l['configs'][k]['value']), round(l['configs'][k]['ms_per_step'],2)) for k in ('blob_proof_b256','commit_prove_b256','blob_proof_b256_two_streams')})"; done
