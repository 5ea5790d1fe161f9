cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_asm4
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "direct_commitments or adversarial" > $O/parity.txt 2>&1; tail -1 $O/parity.txt
run() { # name, env assignments...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/$n.json 2> $O/$n.err
}
run stag16 LWKZG_DIRECT_STAGGER=16
run stag0 LWKZG_DIRECT_STAGGER=0
run stag8 LWKZG_DIRECT_STAGGER=8
run stag16_fill2048 LWKZG_DIRECT_STAGGER=16 LWKZG_DIRECT_FILL=2048
run stag0_fill2048 LWKZG_DIRECT_STAGGER=0 LWKZG_DIRECT_FILL=2048
run stag16_fill4096 LWKZG_DIRECT_STAGGER=16 LWKZG_DIRECT_FILL=4096
run stag16_b LWKZG_DIRECT_STAGGER=16
run stag0_b LWKZG_DIRECT_STAGGER=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_asm4/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
