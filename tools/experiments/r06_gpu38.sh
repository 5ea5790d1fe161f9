#!/bin/bash
# r06 call 38: BASELINE configs[2] (256 device-resident blobs per call): kernels and copies of one call on one time axis -- where do the 1.8 ms in front of the second sub-batch's quotient go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt_p256 -o kt -- python3 bench.py --op blob_proof --batch 256 --steps 6 --warmup 4 --no-cpu-baseline --no-extra-legs > $O/g38_line.json 2> $O/g38_err.txt
python3 tools/experiments/trace_timeline.py $O/kt_p256 $O/g38_proof256_timeline.txt 130
rm -rf $O/kt_p256
head -124 $O/g38_proof256_timeline.txt | tail -75
tail -1 $O/g38_line.json | cut -c1-200
