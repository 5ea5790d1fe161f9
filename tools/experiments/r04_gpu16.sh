mkdir -p gpurun_out/r04p
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r04p/line.json 2> gpurun_out/r04p/err.txt
python - <<'PY'
import json
l=json.load(open("gpurun_out/r04p/line.json"))
print(l["value"], {k: (round(l["configs"][k]["value"]), round(l["configs"][k]["ms_per_step"],2)) for k in ("blob_proof_b256","blob_proof_b256_two_streams","commit_prove_b256")})
PY
done
LWKZG_MID_PROOF_HOST=0 python bench.py --no-cpu-baseline > gpurun_out/r04p/line0.json 2> gpurun_out/r04p/err0.txt
python - <<'PY'
import json
l=json.load(open("gpurun_out/r04p/line0.json"))
print("mid=0", l["value"], {k: (round(l["configs"][k]["value"]), round(l["configs"][k]["ms_per_step"],2)) for k in ("blob_proof_b256","blob_proof_b256_two_streams","commit_prove_b256")})
PY
python bench.py --op blob_proof --batch 256 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('op blob_proof 256:', l['value'], l['ms_per_step'])"
LWKZG_MID_PROOF_HOST=0 python bench.py --op blob_proof --batch 256 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('op blob_proof 256 mid=0:', l['value'], l['ms_per_step'])"
python bench.py --op blob_proof --batch 256 --caller-streams 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('two streams:', l['value'], l['ms_per_step'])"
