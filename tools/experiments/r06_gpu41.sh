#!/bin/bash
# r06 call 41: longer soaks on the round's final code: commitments on the 16-bit table, proof + verification through both forms (host batches up to 2300 blobs: the staged path), small batches, the c-kzg front end, two processes sharing the GPU
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1500 python tools/soak.py --batches 900 --direct-bits 16 2> $O/g41_err.txt | tail -1 > $O/g41_soak_long.jsonl
timeout 900 python tools/soak_verify.py 600 2>> $O/g41_err.txt | tail -1 >> $O/g41_soak_long.jsonl
LWKZG_DIRECT=16 timeout 900 python tools/soak_verify.py 400 2>> $O/g41_err.txt | tail -1 >> $O/g41_soak_long.jsonl
timeout 600 python tools/soak_ckzg.py --batches 120 2>> $O/g41_err.txt | tail -1 >> $O/g41_soak_long.jsonl
timeout 900 python tools/soak_small.py --rounds 1200 2>> $O/g41_err.txt | tail -1 >> $O/g41_soak_long.jsonl
timeout 900 bash tools/stress_mirror.sh > $O/g41_stress_mirror.txt 2>&1
cat $O/g41_soak_long.jsonl | cut -c1-300; tail -5 $O/g41_stress_mirror.txt
