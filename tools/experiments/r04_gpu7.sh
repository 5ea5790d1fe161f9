mkdir -p gpurun_out/r04g
timeout 600 python -m pytest tests/test_gpu_setups_unstructured.py tests/test_gpu_parity.py -m gpu -x -q -k "bucket or repairs or adversarial" 2>&1 | tail -3
for cfg in "1 2" "1 1" "0 2"; do
  set -- $cfg
  LWKZG_BUCKET_ASM=$1 LWKZG_SPLIT=$2 python bench.py --direct-bits 0 --no-config-legs --no-cpu-baseline --steps 10 > gpurun_out/r04g/bench_a$1_s$2.json 2> gpurun_out/r04g/err_a$1_s$2.txt
  cp bench_detail.json gpurun_out/r04g/detail_a$1_s$2.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r04g/detail_a$1_s$2.json"))
print("asm=$1 split=$2", round(d["value"]), {k: round(v["avg_ms"],3) for k,v in d["kernels"].items()})
PY
done
