cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_asm5
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "direct_commitments or adversarial or every_commitment" > $O/parity.txt 2>&1; tail -1 $O/parity.txt
run() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs > $O/$n.json 2> $O/$n.err; }
run t128 LWKZG_DIRECT_T128=1
run t256 LWKZG_DIRECT_T128=0
run t128_b LWKZG_DIRECT_T128=1
run t256_b LWKZG_DIRECT_T128=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_asm5/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, "default", round(d["default_engine"]["value"]), "bucket", round(d["bucket_engine"]["value"]), "host", round(d["host_abi"]["value"]))
    except Exception as e:
        print(f, "ERR", e)
PY
