set -x
mkdir -p gpurun_out/r04c
python -m pytest tests/test_gpu_proof_parity.py -m gpu -x -q -k "small or two_caller" 2>&1 | tail -8 > gpurun_out/r04c/small_tests.txt
tail -4 gpurun_out/r04c/small_tests.txt
for lim in 0 64 256; do
  LWKZG_SMALL_PROOF_HOST=$lim python tools/config_sweep.py --direct-bits default > gpurun_out/r04c/sweep_default_limit$lim.json 2> gpurun_out/r04c/sweep_err_$lim.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r04c/sweep_default_limit$lim.json"))
print("limit $lim", [(r["batch"], round(r["device_resident_ms"],3), round(r["host_abi_ms"],3)) for r in d["blob_proof"]])
PY
done
