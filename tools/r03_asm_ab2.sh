cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_asm2
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "direct_commitments or adversarial" > $O/parity.txt 2>&1; tail -2 $O/parity.txt
run() { # name, lib
  LWKZG_LIBRARY=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/$1.json 2> $O/$1.err
}
run main ""
for v in sdst block2 serial; do run $v $R/lambdaworks_kzg_amd/lib_$v/liblambdaworks_kzg.so; done
run main2 ""
LWKZG_DIRECT_ASM=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/cpp.json 2> $O/cpp.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_asm2/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
