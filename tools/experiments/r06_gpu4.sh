#!/bin/bash
# r06 call 4: what clock do latency-shaped launches run at; the 1 -> 8 projection from one-GPU shard timings; the eight-rank / eight-context rehearsals; large verification batches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
./tools/ubench_clock_latency_bin | tee $O/g4_clock_latency.txt
python tools/scaling_projection.py --out $O/scaling_projection.json 2>$O/g4_proj_err.txt | tee $O/g4_projection_table.md; tail -3 $O/g4_proj_err.txt
for n in 8192 12288 16384; do python tools/verify_device_loop.py --n $n --calls 3 --tag "n=$n" 2>>$O/g4_err.txt | tee -a $O/g4_large.jsonl; done
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_dist.py -x -q -m gpu -k "eight or gpus_8 or shards" 2>&1 | tail -5
