for ch in 4 16 2 8; do
for i in 1 2 3; do LWKZG_MID_PROOF_CHUNKS=$ch python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('chunks $ch', round(l['value']), {k:(round(l['configs'][k]['value']), round(l['configs'][k]['ms_per_step'],2)) for k in ('blob_proof_b256','commit_prove_b256')})"; done
LWKZG_MID_PROOF_CHUNKS=$ch python bench.py --op commit_prove --batch 256 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('  op commit_prove', round(l['value']), round(l['ms_per_step'],2))"
LWKZG_MID_PROOF_CHUNKS=$ch python bench.py --op blob_proof --batch 256 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('  op blob_proof', round(l['value']), round(l['ms_per_step'],2))"
done
