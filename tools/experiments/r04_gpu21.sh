for args in "--steps 10 --warmup 3" "--steps 40 --warmup 10" "--steps 100 --warmup 20"; do
for mid in 384 0; do
LWKZG_MID_PROOF_HOST=$mid python bench.py --op commit_prove --batch 256 --no-cpu-baseline $args 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('commit_prove mid=$mid $args:', round(l['value']), round(l['ms_per_step'],2))"
LWKZG_MID_PROOF_HOST=$mid python bench.py --op blob_proof --batch 256 --no-cpu-baseline $args 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('blob_proof   mid=$mid $args:', round(l['value']), round(l['ms_per_step'],2))"
done; done
