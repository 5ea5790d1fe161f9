cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_setups_unstructured.py -m gpu -q -x -k "direct or commit or msm or closed_form or adversarial or second_pass or unstructured" 2>&1 | tail -3
run() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}, 'default', round(d['default_engine']['value']), 'host', round(d['host_abi']['value']))"; }
run fold_asm LWKZG_FOLD_ASM=1
run fold_cpp LWKZG_FOLD_ASM=0
run fold_asm_b LWKZG_FOLD_ASM=1
