#!/bin/bash
# soaks on the round's final code (after the transform moved to fr28.cuh)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02soak2; mkdir -p $O; cd $R
timeout 900 python tools/soak_ckzg.py --batches 600 --sample 8 > $O/soak_ckzg.json 2> $O/err.txt
timeout 900 python tools/soak.py --batches 600 --direct-bits 16 > $O/soak.json 2>> $O/err.txt
timeout 500 python tools/soak_verify.py 240 2>> $O/err.txt | tail -1 > $O/soak_verify_default.json
LWKZG_DIRECT=16 timeout 500 python tools/soak_verify.py 240 2>> $O/err.txt | tail -1 > $O/soak_verify_direct16.json
LWKZG_TEST_STRESS_SECONDS=90 timeout 900 python -m pytest tests/test_gpu_proof_parity.py -x -q -s -k "everything" > $O/stress90.log 2>&1
echo "rc=$?" >> $O/stress90.log
cat $O/*.json; grep -E "mixed concurrency|passed|failed|rc=" $O/stress90.log
