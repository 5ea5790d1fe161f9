cd $GRAFT_REPO_ROOT
run() { n=$1; shift; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --batch $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', $B, round(d['value']), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items() if 'direct' in k})"; }
SER=$GRAFT_REPO_ROOT/lambdaworks_kzg_amd/lib_serial/liblambdaworks_kzg.so
B=256; run interleaved_1wave LWKZG_DIRECT_FILL=256; run serial_1wave LWKZG_DIRECT_FILL=256 LWKZG_LIBRARY=$SER
B=512; run interleaved_2waves LWKZG_DIRECT_FILL=512; run serial_2waves LWKZG_DIRECT_FILL=512 LWKZG_LIBRARY=$SER
