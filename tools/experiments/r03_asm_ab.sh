# same-box A/B of the hand-scheduled kernel: the shipped library, then every build named in $VARIANTS (tools/r03_asm_variants.sh), then the shipped one again
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_asm_ab
rm -rf $O; mkdir -p $O
cd $R
run() { # name, lib, extra args
  LWKZG_LIBRARY=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs $3 > $O/$1.json 2> $O/$1.err
}
run main ""
for v in $VARIANTS; do run $v $R/lambdaworks_kzg_amd/lib_$v/liblambdaworks_kzg.so; done
run main2 ""
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_asm_ab/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, "default", round(d["default_engine"]["value"]) if "default_engine" in d else "")
    except Exception as e:
        print(f, "ERR", e)
PY
