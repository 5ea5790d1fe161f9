mkdir -p gpurun_out/r04f
for cfg in "1 2" "0 2" "1 3" "1 4"; do
  set -- $cfg
  LWKZG_BUCKET_TURNS=$1 LWKZG_SPLIT=$2 python bench.py --direct-bits 0 --no-config-legs --no-cpu-baseline --steps 10 > gpurun_out/r04f/bench_t$1_s$2.json 2> gpurun_out/r04f/err_t$1_s$2.txt
  cp bench_detail.json gpurun_out/r04f/detail_t$1_s$2.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r04f/detail_t$1_s$2.json"))
print("turns=$1 split=$2", round(d["value"]), {k: round(v["avg_ms"],3) for k,v in d["kernels"].items()})
PY
done
