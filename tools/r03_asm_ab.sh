set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_asm
rm -rf $O; mkdir -p $O
cd $R
# parity first: the commitment tests of every engine and width run through the asm kernel (n >= 5 blobs)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "direct or commit or msm or closed_form or adversarial" > $O/parity.txt 2>&1
echo "pytest rc=$?" >> $O/parity.txt
tail -5 $O/parity.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs > $O/bench_asm.json 2> $O/bench_asm_err.txt
LWKZG_DIRECT_ASM=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs > $O/bench_cpp.json 2> $O/bench_cpp_err.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs > $O/bench_asm2.json 2>> $O/bench_asm_err.txt
python - <<'PY'
import json
for f in ("bench_asm","bench_cpp","bench_asm2"):
    try:
        d=json.loads(open("gpurun_out/r03_asm/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"]), d["ms_per_step"], {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, "default", round(d["default_engine"]["value"]), {k:round(v["avg_ms"],3) for k,v in d["default_engine"]["kernels"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/bench_asm_err.txt
