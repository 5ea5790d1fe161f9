#!/bin/bash
# r06 call 39: a pipelined proof call parses its sub-batches while the host threads still hash (shipped) against parsing behind the digests (lib_x: -DLWK_PARSE_AFTER_DIGESTS)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_proof_parity.py -x -q -m gpu 2>&1 | tail -2
for arm in shipped behind shipped behind shipped behind; do
  if [ $arm = behind ]; then export LWKZG_LIBRARY=$R/lambdaworks_kzg_amd/lib_x/liblambdaworks_kzg.so; else unset LWKZG_LIBRARY; fi
  python bench.py --op blob_proof --batch 256 --steps 60 --warmup 10 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$arm', l['value'], l['ms_per_step'])" | tee -a $O/g39_ab.txt
done
