#!/bin/bash
# r06 call 42: the head of a 513 ... 1024-blob proof call hashed by the host threads, beside the hash kernel of the rest: parity against r05's schedule, then the clock
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_plan.py tests/test_gpu_proof_parity.py -x -q -m gpu 2>&1 | tail -6
for arm in head r05 head r05; do
  if [ $arm = r05 ]; then export LWKZG_EXPERIMENTAL=1 LWKZG_PROOF_HEAD_HOST=0; else unset LWKZG_PROOF_HEAD_HOST; fi
  for b in 1024 768; do
    python bench.py --op blob_proof --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$arm', $b, l['value'], l['ms_per_step'])" | tee -a $O/g42_ab.txt
  done
done
unset LWKZG_PROOF_HEAD_HOST
python bench.py --op blob_proof --batch 1024 --mode ckzg --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('ckzg head', l['value'], l['ms_per_step'])"
