#!/bin/bash
# kernel timeline of the pipelined 256-blob proof call on the final code (one caller stream): profiles/r05_proof_b256_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/kt_proof256; rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --output-format csv -d $O -o kt -- python3 bench.py --op blob_proof --batch 256 --steps 6 --warmup 4 --no-cpu-baseline --no-extra-legs > $O/line.json 2> $O/err.txt
python3 tools/timeline.py $O/kt_kernel_trace.csv 44 > $R/gpurun_out/r05/proof_b256_timeline.txt
tail -30 $R/gpurun_out/r05/proof_b256_timeline.txt
